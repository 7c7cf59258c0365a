// wr_plan_fast.hip — hand-written batch-plan builder: bucket scatter + per-bucket LDS bitonic sort.
//
// Produces EXACTLY the arrays of the generic builder (wr_plan.hip, radix sort over composite keys): triplets of each
// batch stably sorted by user, item occurrences sorted by (item, positive-before-negative, sorted triplet index), bit
// 31 of tp/tn flagging item rows with several occurrences.  It replaces the reference's host-side batching
// (src/helpers/BaseRunner.py:188-193, src/models/BaseModel.py:96-127) like the generic one; tests compare the two
// bit for bit.
//
// Why: the generic radix sort moves every (key,value) pair 4x through HBM at ~2 TB/s; here every pair is written
// once into a (batch, row-range) bucket and sorted inside LDS.
//   F1 user scatter : (user<<32 | original index) appended to bucket (batch, user >> shift_u)   [atomic slot counter]
//   F2 user sort    : one workgroup per bucket: LDS bitonic sort, writes tu/tp/tn/torig at the bucket's prefix, and
//                     appends the two item occurrences (item<<32 | side<<31 | sorted index) to the item buckets
//   F3 item sort    : one workgroup per bucket: LDS bitonic sort, writes oc_item/oc_src, flags shared rows in tp/tn
// Buckets have a fixed capacity (2x the mean + 64); if any bucket overflows (skewed ids) flags[1] is set and the
// caller must rebuild with the generic builder — never a wrong plan.  Ties are impossible (composites are unique), so
// the unstable bucket placement does not leak into the result.
#include "wr_common.h"

namespace wr {

constexpr int kBuckets = 256;    // row-range buckets per batch
constexpr int kMaxCap = 4096;    // largest bucket an LDS sort handles here (32 KiB of 8-byte composites)

struct FastLayout {
    unsigned user_bits, item_bits, shift_u, shift_i;
    int cap_u, cap_i;         // bucket capacities (entries)
    int64_t nb;
    int64_t cnt_bytes, ubuf_bytes, ibuf_bytes, total;
};

static inline unsigned bits_for_rows(int64_t n_values) {
    unsigned b = 0;
    while ((int64_t(1) << b) < n_values) ++b;
    return b == 0 ? 1 : b;
}

static bool fast_layout(int64_t n, int64_t B, int64_t n_users, int64_t n_items, FastLayout &L) {
    if (n <= 0 || n >= (int64_t(1) << 31) || B <= 0 || B > (int64_t(1) << 24)) return false;
    if (n_users <= 0 || n_users >= (int64_t(1) << 31) || n_items <= 0 || n_items >= (int64_t(1) << 31)) return false;
    L.nb = (n + B - 1) / B;
    L.user_bits = bits_for_rows(n_users);
    L.item_bits = bits_for_rows(n_items);
    L.shift_u = L.user_bits > 8 ? L.user_bits - 8 : 0;
    L.shift_i = L.item_bits > 8 ? L.item_bits - 8 : 0;
    const int64_t cu = 2 * ((B + kBuckets - 1) / kBuckets) + 64;
    const int64_t ci = 2 * ((2 * B + kBuckets - 1) / kBuckets) + 64;
    if (cu > kMaxCap || ci > kMaxCap) return false;
    L.cap_u = (int)cu;
    L.cap_i = (int)ci;
    L.cnt_bytes = align_up(2 * L.nb * kBuckets * 4, 256);
    L.ubuf_bytes = align_up(L.nb * kBuckets * (int64_t)L.cap_u * 8, 256);
    L.ibuf_bytes = align_up(L.nb * kBuckets * (int64_t)L.cap_i * 8, 256);
    L.total = L.cnt_bytes + L.ubuf_bytes + L.ibuf_bytes;
    return true;
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void fast_user_scatter(const Idx *__restrict__ u, int64_t n, int64_t B, int64_t n_users,
                                                             unsigned shift_u, int cap_u, int *__restrict__ cnt_u,
                                                             unsigned long long *__restrict__ ubuf, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t uu = (int64_t)u[i];
    if (uu < 0 || uu >= n_users) {
        flags[0] = 1;
        uu = 0;
    }
    const int64_t b = i / B;
    const int64_t bucket = b * kBuckets + (uu >> shift_u);
    const int slot = atomicAdd(&cnt_u[bucket], 1);
    if (slot < cap_u) ubuf[bucket * cap_u + slot] = ((unsigned long long)uu << 32) | (unsigned long long)(uint32_t)i;
    else flags[1] = 1;
}

// Ascending bitonic sort of n2 (power of two) 64-bit keys in LDS by the whole workgroup.
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *a, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += kBlock) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const int l = i + j;
                const unsigned long long x = a[i], y = a[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    a[i] = y;
                    a[l] = x;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// exclusive prefix of this bucket's count inside its batch (sum of the counts of the buckets before it)
__device__ __forceinline__ int bucket_prefix(const int *__restrict__ cnt_batch, int bucket, int cap, int *scratch) {
    int a = 0;
    for (int j = threadIdx.x; j < bucket; j += kBlock) a += min(cnt_batch[j], cap);
    // integer block sum through LDS (order irrelevant for integers)
    __syncthreads();
    if (threadIdx.x == 0) scratch[0] = 0;
    __syncthreads();
    if (a) atomicAdd(&scratch[0], a);
    __syncthreads();
    return scratch[0];
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void fast_user_sort(const Idx *__restrict__ p, const Idx *__restrict__ nn, int64_t n,
                                                          int64_t B, int64_t n_items, int cap_u, int cap_i, unsigned shift_i,
                                                          const int *__restrict__ cnt_u, const unsigned long long *__restrict__ ubuf,
                                                          int *__restrict__ cnt_i, unsigned long long *__restrict__ ibuf,
                                                          int *__restrict__ tu, int *__restrict__ tp, int *__restrict__ tn,
                                                          int *__restrict__ torig, int *__restrict__ flags) {
    extern __shared__ unsigned long long keys[];
    __shared__ int scratch[1];
    const int64_t b = blockIdx.x / kBuckets;
    const int bucket = blockIdx.x % kBuckets;
    const int count = min(cnt_u[blockIdx.x], cap_u);
    const int prefix = bucket_prefix(cnt_u + b * kBuckets, bucket, cap_u, scratch);
    if (count == 0) return;
    const int n2 = pow2_ceil(count);
    const unsigned long long *src = ubuf + (int64_t)blockIdx.x * cap_u;
    for (int j = threadIdx.x; j < n2; j += kBlock) keys[j] = (j < count) ? src[j] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, n2);
    for (int r = threadIdx.x; r < count; r += kBlock) {
        const unsigned long long kv = keys[r];
        const uint32_t orig = (uint32_t)kv;
        const int tloc = prefix + r;                 // position inside the batch, user-sorted
        const int64_t t = b * B + tloc;
        int64_t pi = (int64_t)p[orig], ni = (int64_t)nn[orig];
        if (pi < 0 || pi >= n_items || ni < 0 || ni >= n_items) {
            flags[0] = 1;
            pi = (pi < 0 || pi >= n_items) ? 0 : pi;
            ni = (ni < 0 || ni >= n_items) ? 0 : ni;
        }
        tu[t] = (int)(kv >> 32);
        tp[t] = (int)pi;
        tn[t] = (int)ni;
        if (torig) torig[t] = (int)orig;
        // item occurrences: composite (item, side, sorted index) — positives before negatives for equal items,
        // then by sorted triplet index: the order a stable sort of [positives | negatives] gives
        {
            const int64_t bk = b * kBuckets + (pi >> shift_i);
            const int slot = atomicAdd(&cnt_i[bk], 1);
            if (slot < cap_i) ibuf[bk * cap_i + slot] = ((unsigned long long)pi << 32) | (unsigned long long)(uint32_t)tloc;
            else flags[1] = 1;
        }
        {
            const int64_t bk = b * kBuckets + (ni >> shift_i);
            const int slot = atomicAdd(&cnt_i[bk], 1);
            if (slot < cap_i)
                ibuf[bk * cap_i + slot] = ((unsigned long long)ni << 32) | 0x80000000ull | (unsigned long long)(uint32_t)tloc;
            else flags[1] = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void fast_item_sort(int64_t n, int64_t B, int cap_i, const int *__restrict__ cnt_i,
                                                          const unsigned long long *__restrict__ ibuf, int *__restrict__ oc_item,
                                                          int *__restrict__ oc_src, int *__restrict__ tp, int *__restrict__ tn) {
    extern __shared__ unsigned long long keys[];
    __shared__ int scratch[1];
    const int64_t b = blockIdx.x / kBuckets;
    const int bucket = blockIdx.x % kBuckets;
    const int count = min(cnt_i[blockIdx.x], cap_i);
    const int prefix = bucket_prefix(cnt_i + b * kBuckets, bucket, cap_i, scratch);
    if (count == 0) return;
    const int n2 = pow2_ceil(count);
    const unsigned long long *src = ibuf + (int64_t)blockIdx.x * cap_i;
    for (int j = threadIdx.x; j < n2; j += kBlock) keys[j] = (j < count) ? src[j] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, n2);
    const int64_t base = 2 * b * B + prefix;
    for (int r = threadIdx.x; r < count; r += kBlock) {
        const unsigned long long kv = keys[r];
        const int item = (int)(kv >> 32);
        const uint32_t lo = (uint32_t)kv;
        const int side = (int)(lo >> 31);
        const int tloc = (int)(lo & 0x7fffffffu);
        oc_item[base + r] = item;
        oc_src[base + r] = (tloc << 1) | side;
        // equal items always share a bucket, so neighbours inside the sorted bucket decide "several occurrences"
        const bool shared = (r > 0 && (int)(keys[r - 1] >> 32) == item) || (r + 1 < count && (int)(keys[r + 1] >> 32) == item);
        if (shared) {
            int *dst = side ? tn : tp;
            dst[b * B + tloc] |= (int)0x80000000;  // one writer per (triplet, side)
        }
    }
}

template <typename Idx>
static int32_t plan_build_fast(const Idx *u, const Idx *p, const Idx *nn, int64_t n, int64_t B, int64_t n_users,
                               int64_t n_items, int32_t *tu, int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item,
                               int32_t *oc_src, int32_t *flags, void *workspace, int64_t workspace_bytes, void *stream_) {
    WR_REQUIRE(u && p && nn, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(tu && tp && tn && oc_item && oc_src && flags, WR_E_NULL, "plan output arrays / flags must not be NULL");
    FastLayout L;
    WR_REQUIRE(fast_layout(n, B, n_users, n_items, L), WR_E_RANGE,
               "fast plan builder not applicable to n=%lld, batch=%lld (use the generic builder)", (long long)n, (long long)B);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= L.total, WR_E_WORKSPACE,
               "fast plan workspace %lld B < %lld B", (long long)workspace_bytes, (long long)L.total);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char *ws = reinterpret_cast<char *>(workspace);
    int *cnt_u = reinterpret_cast<int *>(ws);
    int *cnt_i = cnt_u + L.nb * kBuckets;
    unsigned long long *ubuf = reinterpret_cast<unsigned long long *>(ws + L.cnt_bytes);
    unsigned long long *ibuf = reinterpret_cast<unsigned long long *>(ws + L.cnt_bytes + L.ubuf_bytes);
    WR_HIP(hipMemsetAsync(cnt_u, 0, (size_t)(2 * L.nb * kBuckets * 4), stream));
    const unsigned g1 = (unsigned)((n + kBlock - 1) / kBlock);
    const unsigned gb = (unsigned)(L.nb * kBuckets);
    hipLaunchKernelGGL((fast_user_scatter<Idx>), dim3(g1), dim3(kBlock), 0, stream, u, n, B, n_users, L.shift_u, L.cap_u,
                       cnt_u, ubuf, flags);
    WR_LAUNCH_CHECK("fast_user_scatter");
    const size_t lds_u = (size_t)8 << (31 - __builtin_clz((unsigned)(2 * L.cap_u - 1)));  // pow2_ceil(cap_u) * 8
    const size_t lds_i = (size_t)8 << (31 - __builtin_clz((unsigned)(2 * L.cap_i - 1)));
    hipLaunchKernelGGL((fast_user_sort<Idx>), dim3(gb), dim3(kBlock), lds_u, stream, p, nn, n, B, n_items, L.cap_u, L.cap_i,
                       L.shift_i, cnt_u, ubuf, cnt_i, ibuf, tu, tp, tn, torig, flags);
    WR_LAUNCH_CHECK("fast_user_sort");
    hipLaunchKernelGGL(fast_item_sort, dim3(gb), dim3(kBlock), lds_i, stream, n, B, L.cap_i, cnt_i, ibuf, oc_item, oc_src, tp,
                       tn);
    WR_LAUNCH_CHECK("fast_item_sort");
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_bprmf_plan_fast_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items) {
    FastLayout L;
    if (!fast_layout(n_triplets, batch_size, n_users, n_items, L)) return 0;  // 0: not applicable, use the generic builder
    return L.total;
}

int32_t wr_bprmf_plan_build_fast_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                                    flags, workspace, workspace_bytes, stream);
}

int32_t wr_bprmf_plan_build_fast_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, tu, tp, tn, torig, oc_item, oc_src,
                                    flags, workspace, workspace_bytes, stream);
}

}  // extern "C"
