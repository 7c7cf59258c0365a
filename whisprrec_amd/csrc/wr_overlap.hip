// wr_overlap.hip — plan-time marks of the chained step launch (wr_bprmf_run_sgd_chain, wr_bpr.hip).
//
// Reference semantics kept: BaseRunner.fit's loop (src/helpers/BaseRunner.py:194-200) is a strict sequence of
// batch-synchronous steps.  The overlapped stream runs the item phase of step k beside the user phase of step k+1; that
// is only legal for the user runs of batch k+1 that read no item row the item phase of step k rewrites.  This file finds
// the others ("deferred" runs) while the plan is built — index work only, it never looks at table values:
//   O1 overlap_mark_multi    per batch a bitmap over the item rows: bit set = the row has several occurrences in the
//                            batch (bit 31 of tp / tn), i.e. the item phase of that step rewrites it
//   O2 overlap_mark_deferred a triplet of batch b whose positive or negative row is set in the bitmap of batch b-1 makes
//                            its user's run deferred: bit (position of the run's head) of the batch's head mask
//   O3 overlap_compact       per batch the deferred heads in ascending position order (popcount prefix over the mask;
//                            fixed order -> the loss partials of the deferred launch are summed in a fixed order) + count
// All accesses are range-checked against n_items / the batch: the kernels are safe on a plan whose builder overflowed
// (arrays partly unwritten) — such a plan is rebuilt and marked again.
#include <cstdlib>
#include "wr_common.h"

namespace wr {

__global__ __launch_bounds__(kBlock) void overlap_mark_multi(const int *__restrict__ tp, const int *__restrict__ tn, int64_t n,
                                                              int64_t B, int64_t n_items, int64_t words,
                                                              unsigned *__restrict__ bitmap) {
    // grid: (workgroups per batch, batches) — no 64-bit division per thread
    const int64_t b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t t = b * B + j;
    if (j >= B || t >= n) return;
    const int pr = tp[t], nr = tn[t];
    if (pr < 0) {
        const unsigned r = (unsigned)pr & 0x7fffffffu;
        if ((int64_t)r < n_items) atomicOr(&bitmap[b * words + (r >> 5)], 1u << (r & 31u));
    }
    if (nr < 0) {
        const unsigned r = (unsigned)nr & 0x7fffffffu;
        if ((int64_t)r < n_items) atomicOr(&bitmap[b * words + (r >> 5)], 1u << (r & 31u));
    }
}

constexpr int kMaxWalk = 64;   // a run longer than this is a hot user run (> kHotRun = 32): such batches take the ordinary step

__global__ __launch_bounds__(kBlock) void overlap_mark_deferred(const int *__restrict__ tu, const int *__restrict__ tp,
                                                                 const int *__restrict__ tn, int64_t n, int64_t B,
                                                                 int64_t n_items, int64_t words, int64_t dwords,
                                                                 const unsigned *__restrict__ prev_bitmap,
                                                                 const unsigned *__restrict__ bitmap,
                                                                 unsigned *__restrict__ tdef) {
    const int64_t b = blockIdx.y;
    const int64_t j0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t t = b * B + j0;
    if (j0 >= B || t >= n) return;
    const unsigned *bm = b == 0 ? prev_bitmap : bitmap + (b - 1) * words;
    if (bm == nullptr) return;
    const unsigned p = (unsigned)tp[t] & 0x7fffffffu, q = (unsigned)tn[t] & 0x7fffffffu;
    bool hit = false;
    if ((int64_t)p < n_items) hit = (bm[p >> 5] >> (p & 31u)) & 1u;
    if (!hit && (int64_t)q < n_items) hit = (bm[q >> 5] >> (q & 31u)) & 1u;
    if (!hit) return;
    const int64_t base = b * B;
    const int u = tu[t];
    int64_t h = t;
    for (int s = 0; s < kMaxWalk && h > base && tu[h - 1] == u; ++s) --h;   // head of the user's run
    const int64_t j = h - base;
    atomicOr(&tdef[b * dwords + (j >> 5)], 1u << (j & 31));
}

// The same marks with the previous batch's bitmap staged in LDS (item tables of up to ~1.1 M rows: 4 bytes per 32 rows): the
// global version is bound by its two random 4-byte L2 requests per triplet (4.2 M triplets per 64-batch plan: 38-41 us);
// here a workgroup of 1,024 threads copies the bitmap once (coalesced) and looks up kLdsTile triplets in LDS.
constexpr int kLdsThreads = 1024;
constexpr int kLdsTile = 16384;
constexpr size_t kLdsBitmapMax = 144 * 1024;

__global__ __launch_bounds__(kLdsThreads) void overlap_mark_deferred_lds(const int *__restrict__ tu, const int *__restrict__ tp,
                                                                         const int *__restrict__ tn, int64_t n, int64_t B,
                                                                         int64_t n_items, int64_t words, int64_t dwords,
                                                                         const unsigned *__restrict__ prev_bitmap,
                                                                         const unsigned *__restrict__ bitmap,
                                                                         unsigned *__restrict__ tdef) {
    extern __shared__ unsigned lds_bitmap[];
    const int64_t b = blockIdx.y;
    const unsigned *bm = b == 0 ? prev_bitmap : bitmap + (b - 1) * words;
    if (bm == nullptr) return;                                     // workgroup-uniform
    const int64_t base = b * B;
    const int64_t j0 = (int64_t)blockIdx.x * kLdsTile + threadIdx.x;
    constexpr int PER = kLdsTile / kLdsThreads;
    unsigned pk[PER], qk[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                                // the tile's index loads fly while the bitmap is copied
        const int64_t j = j0 + (int64_t)k * kLdsThreads;
        const bool in = j < B && base + j < n;
        pk[k] = in ? (unsigned)tp[base + j] & 0x7fffffffu : 0xffffffffu;
        qk[k] = in ? (unsigned)tn[base + j] & 0x7fffffffu : 0xffffffffu;
    }
    if ((reinterpret_cast<uintptr_t>(bm) & 15) == 0) {             // 16 bytes per load where the batch's bitmap allows it
        const uint4 *bm4 = reinterpret_cast<const uint4 *>(bm);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_bitmap);
        const int64_t w4 = words / 4;
        for (int64_t i = threadIdx.x; i < w4; i += kLdsThreads) l4[i] = bm4[i];
        for (int64_t i = 4 * w4 + threadIdx.x; i < words; i += kLdsThreads) lds_bitmap[i] = bm[i];
    } else {
        for (int64_t i = threadIdx.x; i < words; i += kLdsThreads) lds_bitmap[i] = bm[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const unsigned p = pk[k], q = qk[k];
        bool hit = false;
        if ((int64_t)p < n_items) hit = (lds_bitmap[p >> 5] >> (p & 31u)) & 1u;
        if (!hit && (int64_t)q < n_items) hit = (lds_bitmap[q >> 5] >> (q & 31u)) & 1u;
        if (!hit) continue;
        const int64_t t = base + j0 + (int64_t)k * kLdsThreads;
        const int u = tu[t];
        int64_t h = t;
        for (int s = 0; s < kMaxWalk && h > base && tu[h - 1] == u; ++s) --h;   // head of the user's run
        const int64_t jh = h - base;
        atomicOr(&tdef[b * dwords + (jh >> 5)], 1u << (jh & 31));
    }
}

// one workgroup per batch: positions of the set bits of the batch's head mask, ascending
__global__ __launch_bounds__(kBlock) void overlap_compact(const unsigned *__restrict__ tdef, int64_t dwords, int64_t cap,
                                                           int *__restrict__ def_q, int *__restrict__ def_count) {
    __shared__ int wave_tot[kBlock / 64];
    const int64_t b = blockIdx.x;
    const unsigned *m = tdef + b * dwords;
    const int64_t per = (dwords + kBlock - 1) / kBlock;
    const int64_t w0 = (int64_t)threadIdx.x * per;
    int local = 0;
    for (int64_t j = 0; j < per; ++j)
        if (w0 + j < dwords) local += __popc(m[w0 + j]);
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int at = incl - local;
    int total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) at += wave_tot[w];
        total += wave_tot[w];
    }
    for (int64_t j = 0; j < per; ++j) {
        if (w0 + j >= dwords) break;
        unsigned bits = m[w0 + j];
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            if (at < cap) def_q[b * cap + at] = (int)((w0 + j) * 32 + bit);
            ++at;
        }
    }
    if (threadIdx.x == 0) def_count[b] = total;   // may exceed cap: the host then takes the ordinary step for this plan
}

}  // namespace wr

using namespace wr;

static int32_t overlap_marks(const int32_t *tu, const int32_t *tp, const int32_t *tn, int64_t n_triplets, int64_t batch_size,
                             int64_t n_items, const int32_t *prev_bitmap, int32_t *bitmap, bool bitmap_ready, int32_t *tdef,
                             int32_t *def_q, int64_t def_cap, int32_t *def_count, void *stream_) {
    WR_REQUIRE(tu && tp && tn && bitmap && tdef && def_q && def_count, WR_E_NULL, "overlap marks: NULL argument");
    WR_REQUIRE(n_triplets > 0 && n_triplets < (int64_t(1) << 31) && batch_size > 0 && batch_size <= (int64_t(1) << 24) &&
                   n_items > 0 && n_items < (int64_t(1) << 31) && def_cap > 0,
               WR_E_SHAPE, "overlap marks: bad sizes");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int64_t nb = (n_triplets + batch_size - 1) / batch_size;
    const int64_t words = (n_items + 31) / 32, dwords = (batch_size + 31) / 32;
    WR_HIP(hipMemsetAsync(tdef, 0, (size_t)(nb * dwords * 4), stream));
    WR_REQUIRE(nb <= 65535, WR_E_SHAPE, "overlap marks: %lld batches in one plan (at most 65535)", (long long)nb);
    const dim3 grid((unsigned)((batch_size + kBlock - 1) / kBlock), (unsigned)nb);
    if (!bitmap_ready) {
        WR_HIP(hipMemsetAsync(bitmap, 0, (size_t)(nb * words * 4), stream));
        hipLaunchKernelGGL(overlap_mark_multi, grid, dim3(kBlock), 0, stream, tp, tn, n_triplets, batch_size, n_items,
                           words, reinterpret_cast<unsigned *>(bitmap));
        WR_LAUNCH_CHECK("overlap_mark_multi");
    }
    if ((size_t)words * 4 <= kLdsBitmapMax && batch_size >= kLdsTile && getenv("WR_MARKS_GLOBAL") == nullptr) {
        static bool attr_set = false;
        if (!attr_set) {
            WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(overlap_mark_deferred_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBitmapMax));
            attr_set = true;
        }
        const dim3 grid_lds((unsigned)((batch_size + kLdsTile - 1) / kLdsTile), (unsigned)nb);
        hipLaunchKernelGGL(overlap_mark_deferred_lds, grid_lds, dim3(kLdsThreads), (size_t)words * 4, stream, tu, tp, tn,
                           n_triplets, batch_size, n_items, words, dwords, reinterpret_cast<const unsigned *>(prev_bitmap),
                           reinterpret_cast<const unsigned *>(bitmap), reinterpret_cast<unsigned *>(tdef));
        WR_LAUNCH_CHECK("overlap_mark_deferred_lds");
    } else {
        hipLaunchKernelGGL(overlap_mark_deferred, grid, dim3(kBlock), 0, stream, tu, tp, tn, n_triplets, batch_size, n_items,
                           words, dwords, reinterpret_cast<const unsigned *>(prev_bitmap),
                           reinterpret_cast<const unsigned *>(bitmap), reinterpret_cast<unsigned *>(tdef));
        WR_LAUNCH_CHECK("overlap_mark_deferred");
    }
    hipLaunchKernelGGL(overlap_compact, dim3((unsigned)nb), dim3(kBlock), 0, stream, reinterpret_cast<const unsigned *>(tdef),
                       dwords, def_cap, def_q, def_count);
    WR_LAUNCH_CHECK("overlap_compact");
    return WR_OK;
}

extern "C" {

int32_t wr_bprmf_plan_overlap_marks(const int32_t *tu, const int32_t *tp, const int32_t *tn, int64_t n_triplets,
                                    int64_t batch_size, int64_t n_items, const int32_t *prev_bitmap, int32_t *bitmap,
                                    int32_t *tdef, int32_t *def_q, int64_t def_cap, int32_t *def_count, void *stream) {
    return overlap_marks(tu, tp, tn, n_triplets, batch_size, n_items, prev_bitmap, bitmap, false, tdef, def_q, def_cap,
                         def_count, stream);
}

int32_t wr_bprmf_plan_overlap_deferred(const int32_t *tu, const int32_t *tp, const int32_t *tn, int64_t n_triplets,
                                       int64_t batch_size, int64_t n_items, const int32_t *prev_bitmap, const int32_t *bitmap,
                                       int32_t *tdef, int32_t *def_q, int64_t def_cap, int32_t *def_count, void *stream) {
    return overlap_marks(tu, tp, tn, n_triplets, batch_size, n_items, prev_bitmap, const_cast<int32_t *>(bitmap), true, tdef,
                         def_q, def_cap, def_count, stream);
}

}  // extern "C"
