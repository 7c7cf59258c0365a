// wr_plan.hip — on-device batch plans for the fused BPRMF step.
//
// Counterpart of the reference's host-side batching: DataLoader(shuffle=True) slicing the epoch into
// consecutive batches of `batch_size` with a short last batch (src/helpers/BaseRunner.py:188-193,201) and
// BaseModel.Dataset.collate_batch (src/models/BaseModel.py:96-127).  The reference hands each batch to
// autograd, which resolves duplicate rows with dense scatter-adds; here every batch is sorted once so each
// table row has a single owner in the step kernels (wr_bpr.hip):
//   - triplets stably sorted by (batch, user)          -> tu, tp, tn, torig
//   - the 2B (item, source) occurrences of a batch stably sorted by (batch, item) -> oc_item, oc_src
// One device-wide LSD radix sort per table over composite keys (batch << row_bits | row), restricted to the
// bits that are actually used.  The radix sort itself is rocPRIM's (ROCm system library, stable); key
// construction, range checks and unpacking are the kernels below.
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "wr_common.h"

namespace wr {

static inline unsigned bits_for(int64_t n_values) {  // bits needed to represent 0..n_values-1
    unsigned b = 0;
    while ((int64_t(1) << b) < n_values) ++b;
    return b == 0 ? 1 : b;
}

template <typename Idx, typename Key>
__global__ __launch_bounds__(kBlock) void plan_user_keys(const Idx *__restrict__ u, int64_t n, int64_t B, unsigned row_bits,
                                                          int64_t n_users, Key *__restrict__ keys,
                                                          uint32_t *__restrict__ vals, int *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t uu = (int64_t)u[i];
    if (uu < 0 || uu >= n_users) {
        if (err) *err = 1;
        uu = 0;
    }
    keys[i] = ((Key)(i / B) << row_bits) | (Key)uu;
    vals[i] = (uint32_t)i;
}

template <typename Idx, typename Key>
__global__ __launch_bounds__(kBlock) void plan_unpack_user(const Key *__restrict__ keys_sorted,
                                                            const uint32_t *__restrict__ perm, const Idx *__restrict__ p,
                                                            const Idx *__restrict__ nn, int64_t n, int64_t B,
                                                            unsigned user_bits, unsigned item_bits, int64_t n_items,
                                                            int *__restrict__ tu, int *__restrict__ tp, int *__restrict__ tn,
                                                            int *__restrict__ torig, Key *__restrict__ ikeys,
                                                            uint32_t *__restrict__ ivals, int *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / B;
    const int64_t j = i - b * B;
    const int64_t Bb = (b * B + B <= n) ? B : (n - b * B);
    const uint32_t o = perm[i];
    int64_t pi = (int64_t)p[o], ni = (int64_t)nn[o];
    if (pi < 0 || pi >= n_items || ni < 0 || ni >= n_items) {
        if (err) *err = 1;
        pi = (pi < 0 || pi >= n_items) ? 0 : pi;
        ni = (ni < 0 || ni >= n_items) ? 0 : ni;
    }
    tu[i] = (int)(keys_sorted[i] & (((Key)1 << user_bits) - 1));
    tp[i] = (int)pi;
    tn[i] = (int)ni;
    if (torig) torig[i] = (int)o;
    const int64_t base = 2 * b * B;
    const Key hi = (Key)b << item_bits;
    ikeys[base + j] = hi | (Key)pi;
    ivals[base + j] = (uint32_t)(j << 1);
    ikeys[base + Bb + j] = hi | (Key)ni;
    ivals[base + Bb + j] = (uint32_t)((j << 1) | 1);
}

// After the item sort: writes the item row of every sorted occurrence and flags, in tp/tn (bit 31), the
// occurrences whose item row occurs more than once in its batch.  Composite keys carry the batch id, so
// equal neighbouring keys are always in the same batch.
template <typename Key>
__global__ __launch_bounds__(kBlock) void plan_unpack_item(const Key *__restrict__ keys_sorted,
                                                            const uint32_t *__restrict__ src_sorted, int64_t n2, int64_t B,
                                                            unsigned item_bits, int *__restrict__ oc_item,
                                                            int *__restrict__ tp, int *__restrict__ tn) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n2) return;
    const Key k = keys_sorted[i];
    oc_item[i] = (int)(k & (((Key)1 << item_bits) - 1));
    const bool shared = (i > 0 && keys_sorted[i - 1] == k) || (i + 1 < n2 && keys_sorted[i + 1] == k);
    if (shared) {
        const uint32_t s = src_sorted[i];
        const int64_t batch = (int64_t)(k >> item_bits);
        const int64_t t = batch * B + (int64_t)(s >> 1);
        int *dst = (s & 1u) ? tn : tp;
        dst[t] |= (int)0x80000000;  // one writer per (triplet, side)
    }
}

// ----------------------------------------------------------------------------------------------- small batches, one launch
// For batches of at most kSmallBatch triplets (the reference's default batch is 2,048, BaseRunner.py:32) the whole plan of a
// batch is built by ONE workgroup in LDS: bitonic sort of the (user << 32 | position) composites, then of the
// (item << 32 | side << 31 | sorted index) composites — unique keys, so the result is the stable order the other builders
// produce, bit for bit.  One launch instead of the ~20 of two device radix sorts, and nothing for the host to read back
// besides the index-range flag: what the LightGCN step (a plan per optimizer step) and hipGraph capture need.
constexpr int kSmallBatch = 4096;
constexpr int kSmallThreads = 1024;

// Two compare-exchange stages per pass: a thread takes the four keys base + {0, s, 2s, 3s} (bits s and 2s of base clear),
// applies the stage of stride 2s to (0, 2), (1, 3) and the stage of stride s to (0, 1), (2, 3) in registers, and writes them
// back — every key moves through LDS once per TWO stages and the workgroup meets at half as many barriers (the kernel is
// bound by the LDS pipe and its barriers: 144 stages for the 2,048 + 4,096 keys of a default batch).
template <typename K>
__device__ __forceinline__ void bitonic_sort_lds(K *__restrict__ a, int n_pow2) {
    auto cx = [](K &x, K &y, bool up) {
        if ((x > y) == up) { const K t = x; x = y; y = t; }
    };
    for (int size = 2; size <= n_pow2; size <<= 1) {
        int stride = size >> 1;
        for (; stride >= 2; stride >>= 2) {               // strides (stride, stride / 2) together
            const int s1 = stride >> 1;
            for (int q = threadIdx.x; q < n_pow2 / 4; q += kSmallThreads) {
                const int base = ((q & ~(s1 - 1)) << 2) | (q & (s1 - 1));
                const bool up = (base & size) == 0;
                K k0 = a[base], k1 = a[base + s1], k2 = a[base + 2 * s1], k3 = a[base + 3 * s1];
                cx(k0, k2, up); cx(k1, k3, up);
                cx(k0, k1, up); cx(k2, k3, up);
                a[base] = k0; a[base + s1] = k1; a[base + 2 * s1] = k2; a[base + 3 * s1] = k3;
            }
            __syncthreads();
        }
        if (stride == 1) {                                 // an odd number of stages: the last one alone
            for (int t = threadIdx.x; t < n_pow2 / 2; t += kSmallThreads) {
                const int lo = 2 * t;
                const bool up = (lo & size) == 0;
                K x = a[lo], y = a[lo + 1];
                if ((x > y) == up) { a[lo] = y; a[lo + 1] = x; }
            }
            __syncthreads();
        }
    }
}

// K: composite type.  64-bit: (row << 32) | (side << 31) | position.  32-bit, when the row bits, the side bit and the position
// bits of a batch fit 31 bits together (ml-scale tables at the default batch: 13 + 1 + 11): (row << shift) | (side << (shift
// - 1)) | position — same order, half the LDS traffic per compare-exchange stage (the kernel is bound by the LDS pipe:
// 144 stages for 2,048 + 4,096 keys).  shift_u / shift_i: position of the row inside the user / item composites.
template <typename Idx, typename K>
__global__ __launch_bounds__(kSmallThreads) void plan_small_kernel(const Idx *__restrict__ u, const Idx *__restrict__ p,
                                                                    const Idx *__restrict__ nn, int64_t n, int64_t B,
                                                                    int64_t n_users, int64_t n_items, int *__restrict__ tu,
                                                                    int *__restrict__ tp, int *__restrict__ tn,
                                                                    int *__restrict__ torig, int *__restrict__ oc_item,
                                                                    int *__restrict__ oc_src, int *__restrict__ err,
                                                                    int shift_u, int shift_i) {
    extern __shared__ unsigned long long lds_raw[];       // 2 * P composites | sp[P] | sn[P] (ints)
    K *keys = reinterpret_cast<K *>(lds_raw);
    const int64_t b = blockIdx.x, lo = b * B;
    const int Bb = (int)((lo + B <= n) ? B : (n - lo));
    int P = 1;
    while (P < Bb) P <<= 1;
    int *sp = reinterpret_cast<int *>(keys + 2 * P), *sn = sp + P;
    const K pad = ~(K)0;
    const K pos_mask_u = (((K)1) << shift_u) - 1, pos_mask_i = (((K)1) << (shift_i - 1)) - 1;
    for (int i = threadIdx.x; i < P; i += kSmallThreads) {
        K k = pad;
        if (i < Bb) {
            int64_t uu = (int64_t)u[lo + i];
            if (uu < 0 || uu >= n_users) {
                if (err) *err = 1;
                uu = 0;
            }
            k = ((K)uu << shift_u) | (K)(uint32_t)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    bitonic_sort_lds<K>(keys, P);
    for (int t = threadIdx.x; t < Bb; t += kSmallThreads) {
        const K k = keys[t];
        const int o = (int)(uint32_t)(k & pos_mask_u);
        int64_t pi = (int64_t)p[lo + o], ni = (int64_t)nn[lo + o];
        if (pi < 0 || pi >= n_items || ni < 0 || ni >= n_items) {
            if (err) *err = 1;
            pi = (pi < 0 || pi >= n_items) ? 0 : pi;
            ni = (ni < 0 || ni >= n_items) ? 0 : ni;
        }
        tu[lo + t] = (int)(k >> shift_u);
        if (torig) torig[lo + t] = (int)(lo + o);
        sp[t] = (int)pi;
        sn[t] = (int)ni;
    }
    __syncthreads();
    const int P2 = 2 * P;
    for (int i = threadIdx.x; i < P2; i += kSmallThreads) {
        K k = pad;
        if (i < Bb) k = ((K)(uint32_t)sp[i] << shift_i) | (K)(uint32_t)i;
        else if (i >= P && i - P < Bb) k = ((K)(uint32_t)sn[i - P] << shift_i) | (((K)1) << (shift_i - 1)) | (K)(uint32_t)(i - P);
        keys[i] = k;
    }
    __syncthreads();
    bitonic_sort_lds<K>(keys, P2);
    for (int q = threadIdx.x; q < 2 * Bb; q += kSmallThreads) {
        const K k = keys[q];
        const int item = (int)(k >> shift_i);
        const int side = (int)((k >> (shift_i - 1)) & 1), t = (int)(uint32_t)(k & pos_mask_i);
        oc_item[2 * lo + q] = item;
        oc_src[2 * lo + q] = (t << 1) | side;
        const bool shared = (q > 0 && (int)(keys[q - 1] >> shift_i) == item) || (q + 1 < 2 * Bb && (int)(keys[q + 1] >> shift_i) == item);
        if (shared) {
            if (side) sn[t] |= (int)0x80000000; else sp[t] |= (int)0x80000000;   // one writer per (triplet, side)
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Bb; t += kSmallThreads) {
        tp[lo + t] = sp[t];
        tn[lo + t] = sn[t];
    }
}

// Counting form of the same builder, for tables small enough that one counter per row fits LDS beside the batch (ml-scale:
// 6,040 users / 3,706 items at the default batch): count the rows, exclusive scan, scatter the positions behind their row's
// cursor, then put every row's segment in ascending position order (a position counts the smaller ones of its segment —
// segments are a handful long; a batch of one single row costs B / 1,024 passes over it and stays correct).  A dozen
// barriers instead of the 72 of the two bitonic sorts; the same arrays, bit for bit.
constexpr int kCountPerThread = 2 * kSmallBatch / kSmallThreads;

template <typename RowOf>
__device__ __forceinline__ void count_sort_lds(int *__restrict__ cnt, int *__restrict__ cur, int *__restrict__ out,
                                               int *__restrict__ wave_tot, int n_rows, int E, int total, RowOf row_of) {
    for (int j = threadIdx.x; j < n_rows; j += kSmallThreads) cnt[j] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += kSmallThreads) {
        const int r = row_of(e);
        if (r >= 0) atomicAdd(&cnt[r], 1);                       // integer counts: order irrelevant
    }
    __syncthreads();
    const int per = (n_rows + kSmallThreads - 1) / kSmallThreads;
    const int c0 = threadIdx.x * per;
    int local = 0;
    for (int j = 0; j < per; ++j)
        if (c0 + j < n_rows) local += cnt[c0 + j];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wave_tot[w];
    for (int j = 0; j < per; ++j)
        if (c0 + j < n_rows) {
            cur[c0 + j] = run;
            run += cnt[c0 + j];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += kSmallThreads) {
        const int r = row_of(e);
        if (r >= 0) out[atomicAdd(&cur[r], 1)] = e;              // the cursors end at their segments' ends
    }
    __syncthreads();
    int v[kCountPerThread], dst[kCountPerThread];
#pragma unroll
    for (int k = 0; k < kCountPerThread; ++k) {
        const int q = threadIdx.x + k * kSmallThreads;
        dst[k] = -1;
        if (q < total) {
            v[k] = out[q];
            const int r = row_of(v[k]);
            const int m = cnt[r];
            dst[k] = q;
            if (m > 1) {
                const int s0 = cur[r] - m;
                int rank = 0;
                for (int x = 0; x < m; ++x) rank += out[s0 + x] < v[k] ? 1 : 0;
                dst[k] = s0 + rank;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCountPerThread; ++k)
        if (dst[k] >= 0) out[dst[k]] = v[k];
    __syncthreads();
}

template <typename Idx>
__global__ __launch_bounds__(kSmallThreads) void plan_small_count_kernel(const Idx *__restrict__ u, const Idx *__restrict__ p,
                                                                          const Idx *__restrict__ nn, int64_t n, int64_t B,
                                                                          int64_t n_users, int64_t n_items, int *__restrict__ tu,
                                                                          int *__restrict__ tp, int *__restrict__ tn,
                                                                          int *__restrict__ torig, int *__restrict__ oc_item,
                                                                          int *__restrict__ oc_src, int *__restrict__ err,
                                                                          int nk, int P) {
    extern __shared__ int lds_count[];                    // cnt[nk] | cur[nk] | out[2P] | sp[P] | sn[P]
    __shared__ int wave_tot[kSmallThreads / 64];
    int *cnt = lds_count, *cur = cnt + nk, *out = cur + nk, *sp = out + 2 * P, *sn = sp + P;
    int *ur = out + P;                                    // the user side sorts P positions: its rows live in the upper half
    const int64_t b = blockIdx.x, lo = b * B;
    const int Bb = (int)((lo + B <= n) ? B : (n - lo));
    for (int i = threadIdx.x; i < Bb; i += kSmallThreads) {
        int64_t uu = (int64_t)u[lo + i];
        if (uu < 0 || uu >= n_users) {
            if (err) *err = 1;
            uu = 0;
        }
        ur[i] = (int)uu;
    }
    __syncthreads();
    count_sort_lds(cnt, cur, out, wave_tot, (int)n_users, Bb, Bb, [&](int e) { return ur[e]; });
    for (int t = threadIdx.x; t < Bb; t += kSmallThreads) {
        const int o = out[t];
        int64_t pi = (int64_t)p[lo + o], ni = (int64_t)nn[lo + o];
        if (pi < 0 || pi >= n_items || ni < 0 || ni >= n_items) {
            if (err) *err = 1;
            pi = (pi < 0 || pi >= n_items) ? 0 : pi;
            ni = (ni < 0 || ni >= n_items) ? 0 : ni;
        }
        tu[lo + t] = ur[o];
        if (torig) torig[lo + t] = (int)(lo + o);
        sp[t] = (int)pi;
        sn[t] = (int)ni;
    }
    __syncthreads();
    // occurrences: element e = side * P + t, i.e. inside an item's segment the positive side first, each side by t
    auto item_of = [&](int e) { const int t = e & (P - 1); return t < Bb ? ((e >= P ? sn[t] : sp[t]) & 0x7fffffff) : -1; };
    count_sort_lds(cnt, cur, out, wave_tot, (int)n_items, 2 * P, 2 * Bb, item_of);
    for (int q = threadIdx.x; q < 2 * Bb; q += kSmallThreads) {
        const int e = out[q];
        const int side = e >= P ? 1 : 0, t = e & (P - 1);
        const int item = side ? sn[t] : sp[t];           // one reader and writer per (triplet, side): no flag set yet
        oc_item[2 * lo + q] = item;
        oc_src[2 * lo + q] = (t << 1) | side;
        if (cnt[item] > 1) {
            if (side) sn[t] = item | (int)0x80000000; else sp[t] = item | (int)0x80000000;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Bb; t += kSmallThreads) {
        tp[lo + t] = sp[t];
        tn[lo + t] = sn[t];
    }
}

constexpr size_t kCountLdsMax = 128 * 1024;
// WR_PLAN_SMALL=bitonic in the environment keeps the sorting form (A/B runs, and the tests that compare the two forms)
static bool small_plan_force_bitonic() {
    const char *e = getenv("WR_PLAN_SMALL");
    return e && strcmp(e, "bitonic") == 0;
}

template <typename Idx>
static int32_t plan_build_small(const Idx *u, const Idx *p, const Idx *nn, int64_t n, int64_t B, int64_t n_users,
                                int64_t n_items, int32_t *tu, int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item,
                                int32_t *oc_src, int32_t *err_flag, void *stream_) {
    WR_REQUIRE(u && p && nn, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(tu && tp && tn && oc_item && oc_src, WR_E_NULL, "plan output arrays must not be NULL");
    WR_REQUIRE(n > 0 && n < (int64_t(1) << 31) && B > 0 && B <= kSmallBatch, WR_E_RANGE,
               "small plan builder: batch size %lld (1..%d)", (long long)B, kSmallBatch);
    WR_REQUIRE(n_users > 0 && n_users < (int64_t(1) << 31) && n_items > 0 && n_items < (int64_t(1) << 31), WR_E_SHAPE,
               "small plan builder: table sizes out of range");
    const int64_t nb = (n + B - 1) / B;
    int P = 1, pos_bits = 0;
    while (P < (B < n ? B : n)) { P <<= 1; ++pos_bits; }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    {   // small tables: one LDS counter per row (plan_small_count_kernel)
        const int64_t nk = n_users > n_items ? n_users : n_items;
        const size_t lds = (size_t)(2 * nk + 4 * (int64_t)P) * 4;
        if (lds <= kCountLdsMax && !small_plan_force_bitonic()) {
            static bool attr_set = false;
            if (!attr_set) {
                WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_count_kernel<Idx>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCountLdsMax));
                attr_set = true;
            }
            hipLaunchKernelGGL((plan_small_count_kernel<Idx>), dim3((unsigned)nb), dim3(kSmallThreads), lds, stream, u, p, nn, n, B,
                               n_users, n_items, tu, tp, tn, torig, oc_item, oc_src, err_flag, (int)nk, P);
            WR_LAUNCH_CHECK("plan_small_count_kernel");
            return WR_OK;
        }
    }
    // 32-bit composites when rows + side + positions fit 31 bits (the all-ones padding key stays above every real one)
    const bool narrow = (int)bits_for(n_users) + pos_bits <= 31 && (int)bits_for(n_items) + 1 + pos_bits <= 31;
    if (narrow) {
        const size_t lds = (size_t)2 * P * 4 + (size_t)2 * P * 4 + 8;
        hipLaunchKernelGGL((plan_small_kernel<Idx, uint32_t>), dim3((unsigned)nb), dim3(kSmallThreads), lds, stream, u, p, nn, n, B,
                           n_users, n_items, tu, tp, tn, torig, oc_item, oc_src, err_flag, pos_bits, pos_bits + 1);
    } else {
        const size_t lds = (size_t)2 * P * 8 + (size_t)2 * P * 4;
        if (lds > 64 * 1024) {
            WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel<Idx, unsigned long long>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL((plan_small_kernel<Idx, unsigned long long>), dim3((unsigned)nb), dim3(kSmallThreads), lds, stream, u, p,
                           nn, n, B, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src, err_flag, 32, 32);
    }
    WR_LAUNCH_CHECK("plan_small_kernel");
    return WR_OK;
}

constexpr int kHotRun = 32;     // must match wr_bpr.hip
constexpr int kHotPiece = 256;

// The head of a run longer than kHotRun finds the run's end by binary search in its batch's (ascending) key array and
// appends the run, cut into pieces of kHotPiece, to the batch's hot lists.
// mult = 2 for the item occurrences (2B positions per batch), 1 for the user-sorted triplets (B per batch).
__device__ __forceinline__ void hot_run_at(const int *__restrict__ keys, int64_t i, int64_t n, int64_t B, int mult,
                                           int64_t cap_pieces, int64_t cap_runs, int *__restrict__ piece_q,
                                           int *__restrict__ piece_len, int *__restrict__ run_q, int *__restrict__ run_first,
                                           int *__restrict__ run_np, int *__restrict__ counts, int count_stride) {
    const int64_t b = i / (mult * B);
    const int64_t base = mult * b * B;
    const int64_t Bb = (b * B + B <= n) ? B : (n - b * B);
    const int64_t end = base + mult * Bb;
    if (i >= end) return;
    const int r = keys[i];
    if (i > base && keys[i - 1] == r) return;                   // not a run head
    if (i + kHotRun >= end || keys[i + kHotRun] != r) return;    // run of at most kHotRun positions
    int64_t lo = i + kHotRun, hi = end;                           // keys[lo] == r; keys[hi] != r or hi == end
    while (lo + 1 < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] == r) lo = mid; else hi = mid;
    }
    const int len = (int)(hi - i);
    const int np = (len + kHotPiece - 1) / kHotPiece;
    const int first = atomicAdd(&counts[count_stride * b], np);
    const int ridx = atomicAdd(&counts[count_stride * b + 1], 1);
    if (first + np > cap_pieces || ridx >= cap_runs) return;      // cannot happen with wr_bprmf_hot_caps capacities
    const int q0 = (int)(i - base);
    for (int k = 0; k < np; ++k) {
        piece_q[b * cap_pieces + first + k] = q0 + k * kHotPiece;
        piece_len[b * cap_pieces + first + k] = (k + 1 < np) ? kHotPiece : (len - k * kHotPiece);
    }
    run_q[b * cap_runs + ridx] = q0;
    run_first[b * cap_runs + ridx] = first;
    run_np[b * cap_runs + ridx] = np;
}

// One thread per four sorted positions.  Almost no position starts a hot run, so the common case is decided from two
// 16-byte loads (positions i..i+3 and i+kHotRun..i+kHotRun+3): a key that differs from the key kHotRun positions on
// cannot head a run that long.  Groups that may hold one (or that touch a batch's end) take the exact per-position path.
__global__ __launch_bounds__(kBlock) void plan_hot_runs_kernel(const int *__restrict__ keys, int64_t n, int64_t B, int mult,
                                                                int64_t cap_pieces, int64_t cap_runs, int *__restrict__ piece_q,
                                                                int *__restrict__ piece_len, int *__restrict__ run_q,
                                                                int *__restrict__ run_first, int *__restrict__ run_np,
                                                                int *__restrict__ counts, int count_stride) {
    static_assert(kHotRun % 4 == 0, "the far load must stay 16-byte aligned");
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    const int64_t total = mult * n;
    if (i >= total) return;
    if ((reinterpret_cast<uintptr_t>(keys) & 15) == 0 && i + 3 + kHotRun < total) {
        const int4 a = *reinterpret_cast<const int4 *>(keys + i);
        const int4 f = *reinterpret_cast<const int4 *>(keys + i + kHotRun);
        // within a batch keys ascend, so keys[i] == keys[i + kHotRun] is necessary for a run head at i; across a batch
        // boundary the test can only err towards the exact path
        if (a.x != f.x && a.y != f.y && a.z != f.z && a.w != f.w) return;
    }
    for (int j = 0; j < 4; ++j)
        if (i + j < total)
            hot_run_at(keys, i + j, n, B, mult, cap_pieces, cap_runs, piece_q, piece_len, run_q, run_first, run_np, counts,
                       count_stride);
}

struct PlanLayout {
    bool wide;          // 64-bit composite keys
    unsigned user_bits, item_bits, batch_bits;
    int64_t key_bytes;  // per key array (2n keys)
    int64_t val_bytes;  // per value array (2n values)
    size_t sort_temp;   // rocPRIM temporary storage
    int64_t total;
};

template <typename Key>
static hipError_t sort_temp_bytes(int64_t n2, unsigned end_bit, size_t &bytes) {
    Key *k = nullptr;
    uint32_t *v = nullptr;
    bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n2, 0u, end_bit, (hipStream_t)0);
}

static int32_t plan_layout(int64_t n, int64_t B, int64_t n_users, int64_t n_items, PlanLayout &L) {
    WR_REQUIRE(n > 0 && n < (int64_t(1) << 31), WR_E_SHAPE, "n_triplets=%lld out of range (1..2^31)", (long long)n);
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "batch_size=%lld out of range (1..2^29)", (long long)B);
    WR_REQUIRE(n_users > 0 && n_users < (int64_t(1) << 31) && n_items > 0 && n_items < (int64_t(1) << 31), WR_E_SHAPE,
               "table sizes out of range");
    const int64_t nb = (n + B - 1) / B;
    L.user_bits = bits_for(n_users);
    L.item_bits = bits_for(n_items);
    L.batch_bits = bits_for(nb);
    const unsigned need = (L.user_bits > L.item_bits ? L.user_bits : L.item_bits) + L.batch_bits;
    L.wide = need > 32;
    const int64_t n2 = 2 * n;
    L.key_bytes = align_up(n2 * (L.wide ? 8 : 4), 256);
    L.val_bytes = align_up(n2 * 4, 256);
    size_t t = 0;
    hipError_t e = L.wide ? sort_temp_bytes<uint64_t>(n2, L.item_bits + L.batch_bits, t)
                          : sort_temp_bytes<uint32_t>(n2, L.item_bits + L.batch_bits, t);
    if (e != hipSuccess) return fail_hip(e, "rocprim::radix_sort_pairs (size query)");
    size_t t2 = 0;
    e = L.wide ? sort_temp_bytes<uint64_t>(n, L.user_bits + L.batch_bits, t2)
               : sort_temp_bytes<uint32_t>(n, L.user_bits + L.batch_bits, t2);
    if (e != hipSuccess) return fail_hip(e, "rocprim::radix_sort_pairs (size query)");
    L.sort_temp = (size_t)align_up((int64_t)(t > t2 ? t : t2), 256);
    L.total = 2 * L.key_bytes + 2 * L.val_bytes + (int64_t)L.sort_temp;
    return WR_OK;
}

template <typename Idx, typename Key>
static int32_t plan_build_impl(const Idx *u, const Idx *p, const Idx *nn, int64_t n, int64_t B, int64_t n_users,
                               int64_t n_items, int32_t *tu, int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item,
                               int32_t *oc_src, int32_t *err_flag, void *workspace, const PlanLayout &L,
                               hipStream_t stream) {
    char *ws = reinterpret_cast<char *>(workspace);
    Key *keyA = reinterpret_cast<Key *>(ws);
    Key *keyB = reinterpret_cast<Key *>(ws + L.key_bytes);
    uint32_t *valA = reinterpret_cast<uint32_t *>(ws + 2 * L.key_bytes);
    uint32_t *valB = reinterpret_cast<uint32_t *>(ws + 2 * L.key_bytes + L.val_bytes);
    void *temp = ws + 2 * L.key_bytes + 2 * L.val_bytes;
    size_t temp_bytes = L.sort_temp;
    const int64_t n2 = 2 * n;
    const unsigned g1 = (unsigned)((n + kBlock - 1) / kBlock), g2 = (unsigned)((n2 + kBlock - 1) / kBlock);

    hipLaunchKernelGGL((plan_user_keys<Idx, Key>), dim3(g1), dim3(kBlock), 0, stream, u, n, B, L.user_bits, n_users, keyA,
                       valA, err_flag);
    WR_LAUNCH_CHECK("plan_user_keys");
    WR_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keyA, keyB, valA, valB, (size_t)n, 0u, L.user_bits + L.batch_bits,
                                     stream));
    // keyB/valB hold the user-sorted order; item keys/values go to keyA/valA (2n entries)
    hipLaunchKernelGGL((plan_unpack_user<Idx, Key>), dim3(g1), dim3(kBlock), 0, stream, keyB, valB, p, nn, n, B, L.user_bits,
                       L.item_bits, n_items, tu, tp, tn, torig, keyA, valA, err_flag);
    WR_LAUNCH_CHECK("plan_unpack_user");
    temp_bytes = L.sort_temp;
    WR_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keyA, keyB, valA, reinterpret_cast<uint32_t *>(oc_src), (size_t)n2,
                                     0u, L.item_bits + L.batch_bits, stream));
    hipLaunchKernelGGL((plan_unpack_item<Key>), dim3(g2), dim3(kBlock), 0, stream, keyB,
                       reinterpret_cast<const uint32_t *>(oc_src), n2, B, L.item_bits, oc_item, tp, tn);
    WR_LAUNCH_CHECK("plan_unpack_item");
    return WR_OK;
}

template <typename Idx>
static int32_t plan_build(const Idx *u, const Idx *p, const Idx *nn, int64_t n, int64_t B, int64_t n_users,
                          int64_t n_items, int32_t *tu, int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item,
                          int32_t *oc_src, int32_t *err_flag, void *workspace, int64_t workspace_bytes, void *stream_) {
    WR_REQUIRE(u && p && nn, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(tu && tp && tn && oc_item && oc_src, WR_E_NULL, "plan output arrays must not be NULL");
    PlanLayout L;
    int32_t rc = plan_layout(n, B, n_users, n_items, L);
    if (rc != WR_OK) return rc;
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= L.total, WR_E_WORKSPACE,
               "plan workspace %lld B < %lld B", (long long)workspace_bytes, (long long)L.total);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (L.wide)
        return plan_build_impl<Idx, uint64_t>(u, p, nn, n, B, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src, err_flag,
                                              workspace, L, stream);
    return plan_build_impl<Idx, uint32_t>(u, p, nn, n, B, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src, err_flag,
                                          workspace, L, stream);
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_bprmf_plan_hot_runs(const int32_t *keys, int32_t kind, int64_t n_triplets, int64_t batch_size, int32_t *piece_q,
                               int32_t *piece_len, int32_t *run_q, int32_t *run_first, int32_t *run_np, int32_t *counts,
                               void *stream) {
    WR_REQUIRE(keys && piece_q && piece_len && run_q && run_first && run_np && counts, WR_E_NULL, "hot-run arrays: NULL");
    WR_REQUIRE(n_triplets > 0 && n_triplets < (int64_t(1) << 30) && batch_size > 0, WR_E_SHAPE, "hot runs: bad sizes");
    WR_REQUIRE(kind == 0 || kind == 1, WR_E_RANGE, "hot runs: kind must be 0 (item occurrences) or 1 (user positions)");
    int64_t cp = 0, cr = 0;
    wr_bprmf_hot_caps(batch_size, kind, &cp, &cr);
    const int mult = kind == 0 ? 2 : 1;
    const int64_t total = mult * n_triplets;
    hipLaunchKernelGGL(plan_hot_runs_kernel, dim3((unsigned)((total + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), keys, n_triplets, batch_size, mult, cp, cr, piece_q, piece_len,
                       run_q, run_first, run_np, counts + (kind == 0 ? 0 : 2), 4);
    WR_LAUNCH_CHECK("plan_hot_runs_kernel");
    return WR_OK;
}

int64_t wr_bprmf_plan_small_max_batch(void) { return kSmallBatch; }

int32_t wr_bprmf_plan_build_small_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                      int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                      int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                      void *stream) {
    return plan_build_small<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                                     err_flag, stream);
}

int32_t wr_bprmf_plan_build_small_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                      int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                      int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                      void *stream) {
    return plan_build_small<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                                     err_flag, stream);
}

int64_t wr_bprmf_plan_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items) {
    PlanLayout L;
    const int32_t rc = plan_layout(n_triplets, batch_size, n_users, n_items, L);
    if (rc != WR_OK) return rc < 0 ? rc : -(int64_t)rc;
    return L.total;
}

int32_t wr_bprmf_plan_build_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                               err_flag, workspace, workspace_bytes, stream);
}

int32_t wr_bprmf_plan_build_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                               err_flag, workspace, workspace_bytes, stream);
}

}  // extern "C"
