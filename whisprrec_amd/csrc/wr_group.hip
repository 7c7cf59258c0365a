// wr_group.hip — the fused SGD step WITHOUT a per-batch sort (gfx950 / MI355X).
//
// Reference path restated (paths relative to the reference root): the loop of src/helpers/BaseRunner.py:194-200 —
// zero_grad / BPRMF.predict (src/models/general/BPRMF.py:69-80) / BPRLoss (src/utils/loss.py:37-39) / backward /
// torch.optim.SGD.step — has no per-batch index work at all.  The sorted batch plan of wr_plan*.hip (two bucket scatters +
// two LDS sorts per batch, 36 B of plan per triplet) is what this file takes off the step: a "group plan" leaves the
// triplets where they are and only GROUPS what recurs inside a batch.
//
// Group plan (per batch, built by ONE launch per chunk of batches, index work only):
//   flags   one uint4 per 32 triplets {US, PS, NS, DF}: bit t of US = the user of triplet t occurs several times in the
//           batch; PS / NS = its positive / negative item row does; DF = one of its three rows is rewritten by the tiles
//           of the batch BEFORE (the launch defers such triplets behind those tiles).
//   lists   the shared occurrences only (~6 % of the triplets by user, ~12 % of the occurrences by item at 1M x 1M,
//           B = 65,536), sorted by (row, position): (user, t << 1) and (item, t << 1 | negative), in R segments per batch
//           and side (a segment = a range of hashed rows), with their lengths.
// How: a workgroup owns (batch, side, range of 2^18 hashed rows: the rows whose hashed id has the range's low bits, so that
// the popular — low — ids of a popularity-sorted id space are dealt round the ranges) and keeps two bitmaps of that range in LDS: a first scan
// of the batch's ids sets "seen" bits and, where the bit was set already, "several" bits; a second scan flags the
// triplets whose row has its "several" bit set, appends them to a list in LDS (bitonic sort by (row, position)) and — item
// side — flags the NEXT batch's triplets against the same bitmap (DF).  No global sort, no scatter: the ids are read
// (L2-resident after the first touch), ~0.1 B of plan per triplet is written.  Tables beyond 2^21 rows are hashed
// (row mod 2^21): a collision only makes a row look shared (it then takes the stash + tile path as a run of one).
//
// Step (bprmf_group_step): one launch per step, three kinds of workgroups:
//   main      one team per triplet in ORIGINAL order reads its three rows; a row that occurs once in the batch is finished
//             in place by that team (U[u] -= lr c (I[p] - I[n]), I[p] -= lr c U[u], I[n] += lr c U[u]); for a shared row
//             the contribution is stashed (ZU[t] = c (I[p] - I[n]), Z[t] = c U[u]);
//   tiles     of the batch BEFORE, over its user and item lists: one team per run sums the stashed rows in list order and
//             rewrites the row once (write-through), then the workgroup signals a sharded counter (the hand-off of
//             wr_bpr.hip's chained launch);
//   deferred  triplets with their DF bit set wait for that counter inside the launch (~2 % of a batch).
// Every table row has exactly one writer per step and a fixed summation order: bitwise reproducible, no float atomics.
#include "wr_common.h"
#include <hip/hip_ext.h>

#include <algorithm>

namespace wr {

constexpr int kGpThreads = 1024;
constexpr unsigned kGpRangeBits = 18;                 // hashed rows per range workgroup: 2^18
constexpr int kGpWords = 1 << (kGpRangeBits - 5);     // words of one bitmap (32 KiB)
constexpr int kGpCap = 8192;                          // list entries per (batch, side, range): 64 KiB of 8-byte keys in LDS
constexpr int kGpBins = 4096;                         // counting-sort bins of a range: 64 hashed rows each
constexpr int kGpMaxBin = 256;                        // longest bin ordered by ranking (else: overflow)
constexpr int kGpMaxRanges = 8;                       // hashed space <= 2^21 rows (gs_tile_of loads 8 lengths)
constexpr int64_t kGpMaxBatch = 1 << 17;              // three flag arrays of B / 32 words in LDS
constexpr int kGpLongRun = 64;                        // a ROW with more occurrences than this sets meta[2] (a hot row: the caller goes
                                                      // back to the sorted plan, whose hot-row path is made for it)
constexpr int kGpMetaWords = 16;
#ifndef WR_GP_DBG
#define WR_GP_DBG 0      // timing-only variants (A/B builds, never shipped; all stay inside the arrays): 1 no ordering, 2 no second
#endif                 // scan, 4 no flag words out, 8 no first scan, 16 no deferral scan

struct GroupLayout {
    int64_t nb, B, fw;
    int R_u, R_i, cap_u, cap_i;     // ranges per batch and side; entries a range's list segment holds in the plan
    unsigned mask_u, mask_i, hbits_u, hbits_i;
    int64_t flags, ucnt, icnt, ul_row, ul_src, il_row, il_src, total;   // offsets in int32 words; meta at 0
    int64_t zero_words;                                                  // meta + flags + counts: cleared before a build
};

static inline int ranges_for(int64_t n_rows, unsigned *mask, unsigned *hbits, int *cap) {
    unsigned bits = kGpRangeBits;
    while (bits < kGpRangeBits + 3 && (int64_t(1) << bits) < n_rows) ++bits;
    *mask = (1u << bits) - 1u;
    *hbits = bits;
    const int R = 1 << (bits - kGpRangeBits);
    *cap = kGpCap;
    return R;
}

static bool group_layout(int64_t n, int64_t B, int64_t n_users, int64_t n_items, GroupLayout &L) {
    if (n <= 0 || n >= (int64_t(1) << 31) || B <= 0 || B > kGpMaxBatch) return false;
    if (n_users <= 0 || n_users >= (int64_t(1) << 31) || n_items <= 0 || n_items >= (int64_t(1) << 31)) return false;
    L.nb = (n + B - 1) / B;
    L.B = B;
    L.fw = (B + 31) / 32;
    L.R_u = ranges_for(n_users, &L.mask_u, &L.hbits_u, &L.cap_u);
    L.R_i = ranges_for(n_items, &L.mask_i, &L.hbits_i, &L.cap_i);
    L.flags = kGpMetaWords;
    L.ucnt = L.flags + L.nb * L.fw * 4;
    L.icnt = L.ucnt + align_up(L.nb * L.R_u, 4);
    L.zero_words = L.icnt + align_up(L.nb * L.R_i, 4);
    L.ul_row = L.zero_words;
    L.ul_src = L.ul_row + L.nb * L.R_u * (int64_t)L.cap_u;
    L.il_row = L.ul_src + L.nb * L.R_u * (int64_t)L.cap_u;
    L.il_src = L.il_row + L.nb * L.R_i * (int64_t)L.cap_i;
    L.total = L.il_src + L.nb * L.R_i * (int64_t)L.cap_i;
    return true;
}

struct GpDev {
    int *meta;            // [0] id out of range, [1] a list overflowed, [2] a row with more than kGpLongRun occurrences
    unsigned *flags;      // [nb][fw][4] = {US, PS, NS, DF}
    int *ucnt, *icnt;     // [nb][R]
    int *ul_row, *ul_src, *il_row, *il_src;   // [nb][R][cap]
    int R_u, R_i, cap_u, cap_i;
    unsigned mask_u, mask_i, hbits_u, hbits_i;
    int fw;
};

static inline GpDev group_dev(int32_t *plan, const GroupLayout &L) {
    return GpDev{plan, reinterpret_cast<unsigned *>(plan + L.flags), plan + L.ucnt, plan + L.icnt, plan + L.ul_row,
                 plan + L.ul_src, plan + L.il_row, plan + L.il_src, L.R_u, L.R_i, L.cap_u, L.cap_i, L.mask_u, L.mask_i,
                 L.hbits_u, L.hbits_i, (int)L.fw};
}

// f(value, position) over a[0 .. cnt): 16-byte loads when the array is aligned (batch starts are, for batch sizes that are
// multiples of 4)
template <typename F>
__device__ __forceinline__ void gp_scan(const int *__restrict__ a, int cnt, F f) {
    if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) {
        const int n4 = cnt >> 2;
        const int4 *a4 = reinterpret_cast<const int4 *>(a);
        int i = threadIdx.x;
        // four loads in flight per thread: the loop is a chain of global round trips, not of arithmetic
        for (; i + 3 * kGpThreads < n4; i += 4 * kGpThreads) {
            int4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a4[i + q * kGpThreads];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int at = 4 * (i + q * kGpThreads);
                f(v[q].x, at);
                f(v[q].y, at + 1);
                f(v[q].z, at + 2);
                f(v[q].w, at + 3);
            }
        }
        for (; i < n4; i += kGpThreads) {
            const int4 v = a4[i];
            f(v.x, 4 * i);
            f(v.y, 4 * i + 1);
            f(v.z, 4 * i + 2);
            f(v.w, 4 * i + 3);
        }
        for (int j = (n4 << 2) + threadIdx.x; j < cnt; j += kGpThreads) f(a[j], j);
    } else {
        for (int j = threadIdx.x; j < cnt; j += kGpThreads) f(a[j], j);
    }
}

// One workgroup = (batch, side, range of 2^18 hashed rows).  LDS: region X (64 KiB: the "seen" bitmap in its first half, then
// the list of shared (row, source) keys), region M (32 KiB: the "several" bitmap, then the bins' counters and the keys'
// indices grouped by bin), three flag arrays of B / 32 words.
__global__ __launch_bounds__(kGpThreads) void group_plan_kernel(const int *__restrict__ u, const int *__restrict__ p,
                                                                 const int *__restrict__ n, int64_t n_total, int B, int nb,
                                                                 int n_users, int n_items, GpDev L) {
    // every LDS word lives in the dynamic region, 16-byte aligned (cdna_hip_programming.md, Guideline 17)
    extern __shared__ __attribute__((aligned(16))) unsigned gp_lds[];
    unsigned *seen = gp_lds;                                                       // words [0, kGpWords) of region X
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(gp_lds);     // region X later: kGpCap 8-byte keys
    unsigned *multi = gp_lds + 2 * kGpCap;                                         // region M
    unsigned *f0 = multi + kGpWords, *f1 = f0 + L.fw, *f2 = f1 + L.fw;
    int &n_list = *reinterpret_cast<int *>(f2 + L.fw);
    const int per = L.R_u + L.R_i;
    // the workgroups of one batch read the same ids: keep them on one XCD's L2
    const unsigned lb = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int b = (int)(lb / (unsigned)per), rr = (int)(lb % (unsigned)per);
    const bool item = rr >= L.R_u;
    const unsigned r = (unsigned)(item ? rr - L.R_u : rr);
    const int64_t base = (int64_t)b * B;
    const int Bb = (int)((base + B <= n_total) ? B : (n_total - base));
    const unsigned n_rows = (unsigned)(item ? n_items : n_users);
    const unsigned mask = item ? L.mask_i : L.mask_u;
    // A hashed row h belongs to range h & (R - 1) and has bit h >> rb of that range's bitmaps (rb = log2 R): consecutive ids
    // — the popular items of an id space sorted by popularity — are dealt round the ranges, and round the bins of the ordering
    // below (bin = low 12 bits of the bit index), instead of filling one list and one bin.
    const unsigned rb = (item ? L.hbits_i : L.hbits_u) - kGpRangeBits, rsel = (1u << rb) - 1u;
    for (int i = threadIdx.x; i < kGpWords; i += kGpThreads) {
        seen[i] = 0u;
        multi[i] = 0u;
    }
    for (int i = threadIdx.x; i < 3 * L.fw; i += kGpThreads) f0[i] = 0u;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    // scan A: "seen" and "several" bits of this range
    auto mark = [&](int row, int) {
        if ((unsigned)row >= n_rows) {
            L.meta[0] = 1;
            return;
        }
        const unsigned h = (unsigned)row & mask;
        if ((h & rsel) != r) return;
        const unsigned bit = h >> rb, m = 1u << (bit & 31u);
        const unsigned old = atomicOr(&seen[bit >> 5], m);
        if (old & m) atomicOr(&multi[bit >> 5], m);
    };
    if (!(WR_GP_DBG & 8)) {
        if (item) {
            gp_scan(p + base, Bb, mark);
            gp_scan(n + base, Bb, mark);
        } else {
            gp_scan(u + base, Bb, mark);
        }
    }
    __syncthreads();
    // scan B: flag the shared occurrences and collect their sources (the list takes over the "seen" region); an id out of
    // range was reported by scan A and its hashed bit is inside the bitmap anyway.
    // (Measured on MI355X, 1M x 1M, B = 65,536, us of plan per batch — this form 2.76; the LDS operations of four ids issued
    // together before any is waited for: 2.63; a wave per 64 consecutive positions, flag words from ballots by plain stores
    // and one list atomic per wave: 3.82 — four 4-byte loads and ballots cost more than the 16-byte load's atomics.  Of the
    // 2.7: launch + zeroing 0.6, first scan 0.55, second scan 1.25 (0.65 of it visits that find nothing), ordering 0.3.)
    auto several = [&](int row) -> bool {
        const unsigned h = (unsigned)row & mask;
        if ((h & rsel) != r) return false;
        const unsigned bit = h >> rb;
        return (multi[bit >> 5] >> (bit & 31u)) & 1u;
    };
    auto append = [&](int row, unsigned src) {
        const int pos = atomicAdd(&n_list, 1);
        if (pos < kGpCap) keys[pos] = ((unsigned long long)(unsigned)row << 32) | src;
    };
    if (!(WR_GP_DBG & 2)) {
        const bool next = b + 1 < nb && !(WR_GP_DBG & 16);
        const int64_t nbase = base + B;
        const int Bn = next ? (int)((nbase + B <= n_total) ? B : (n_total - nbase)) : 0;
        // the next batch's triplets that read a row this batch's tiles rewrite
        auto defer = [&](int row, int t) {
            if (several(row)) atomicOr(&f2[t >> 5], 1u << (t & 31));
        };
        if (item) {
            gp_scan(p + base, Bb, [&](int row, int t) {
                if (several(row)) {
                    atomicOr(&f0[t >> 5], 1u << (t & 31));
                    append(row, (unsigned)t << 1);
                }
            });
            gp_scan(n + base, Bb, [&](int row, int t) {
                if (several(row)) {
                    atomicOr(&f1[t >> 5], 1u << (t & 31));
                    append(row, ((unsigned)t << 1) | 1u);
                }
            });
            gp_scan(p + nbase, Bn, defer);
            gp_scan(n + nbase, Bn, defer);
        } else {
            gp_scan(u + base, Bb, [&](int row, int t) {
                if (several(row)) {
                    atomicOr(&f0[t >> 5], 1u << (t & 31));
                    append(row, (unsigned)t << 1);
                }
            });
            gp_scan(u + nbase, Bn, defer);
        }
    }
    __syncthreads();
    // flags out: the ranges of a batch OR their words together (agent-scope, no return value)
    if (!(WR_GP_DBG & 4)) {
        unsigned *fb = L.flags + (int64_t)b * L.fw * 4;
        for (int w = threadIdx.x; w < L.fw; w += kGpThreads) {
            if (f0[w]) __hip_atomic_fetch_or(fb + 4 * w + (item ? 1 : 0), f0[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (item && f1[w]) __hip_atomic_fetch_or(fb + 4 * w + 2, f1[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f2[w])
                __hip_atomic_fetch_or(fb + 4 * (L.fw + w) + 3, f2[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int cap = item ? L.cap_i : L.cap_u, R = item ? L.R_i : L.R_u;
    int m = n_list;
    if (m > cap) {
        if (threadIdx.x == 0) L.meta[1] = 1;
        m = cap;
    }
    // Order the m keys by (bin, row, source) without a comparison sort: a counting sort over 4,096 bins of the range's rows
    // places the keys' INDICES (16 bits each) grouped by bin, and a key's final position is its bin's start + the number of
    // smaller keys in its bin (keys are unique; a bin holds ~1 key).  Equal rows end up adjacent, in source order: what the
    // tiles need.  (A bitonic sort of the same keys took 2.1 of the 4.7 us per batch of the first version.)
    int *cnt = reinterpret_cast<int *>(multi);                                      // the "several" bitmap is done with
    unsigned short *out16 = reinterpret_cast<unsigned short *>(multi + kGpBins);
    int *wave_tot = reinterpret_cast<int *>(f0);                                    // the flag arrays have been written out
    auto bin_of = [&](unsigned long long k) -> int { return (int)(((((unsigned)(k >> 32)) & mask) >> rb) & (kGpBins - 1)); };
    int *lrow = (item ? L.il_row : L.ul_row) + ((int64_t)b * R + r) * cap, *lsrc = (item ? L.il_src : L.ul_src) + ((int64_t)b * R + r) * cap;
    __syncthreads();      // every thread has read n_list and written its flag words out
    for (int i = threadIdx.x; i < kGpBins; i += kGpThreads) cnt[i] = 0;
    __syncthreads();
    if (!(WR_GP_DBG & 1)) {
        for (int j = threadIdx.x; j < m; j += kGpThreads) atomicAdd(&cnt[bin_of(keys[j])], 1);
        __syncthreads();
        // exclusive scan of the kGpBins counters: thread t owns counters [4 t, 4 t + 4)
        constexpr int kOwn = kGpBins / kGpThreads;
        const int c0 = (int)threadIdx.x * kOwn;
        int local = 0;
#pragma unroll
        for (int j = 0; j < kOwn; ++j) local += cnt[c0 + j];
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
#pragma unroll
        for (int j = 0; j < kOwn; ++j) {
            const int c = cnt[c0 + j];
            cnt[c0 + j] = run;
            run += c;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += kGpThreads) out16[atomicAdd(&cnt[bin_of(keys[j])], 1)] = (unsigned short)j;
        __syncthreads();
        // cnt[bin] is now the END of the bin; its start is the end of the bin before
        for (int j = threadIdx.x; j < m; j += kGpThreads) {
            const unsigned long long k = keys[j];
            const int bin = bin_of(k);
            const int lo = bin ? cnt[bin - 1] : 0, hi = cnt[bin];
            if (hi - lo > kGpMaxBin) {
                L.meta[1] = 1;
                continue;
            }
            int pos = lo, same = 0;
            for (int jj = lo; jj < hi; ++jj) {
                const unsigned long long o = keys[out16[jj]];
                pos += o < k ? 1 : 0;
                same += (unsigned)(o >> 32) == (unsigned)(k >> 32) ? 1 : 0;
            }
            if (same > kGpLongRun) L.meta[2] = 1;
            lrow[pos] = (int)(k >> 32);
            lsrc[pos] = (int)(unsigned)k;
        }
    } else {
        for (int j = threadIdx.x; j < m; j += kGpThreads) {
            lrow[j] = (int)(keys[j] >> 32);
            lsrc[j] = (int)(unsigned)keys[j];
        }
    }
    if (threadIdx.x == 0) (item ? L.icnt : L.ucnt)[(int64_t)b * R + r] = m;
}

// The reference hands int64 index columns (src/models/BaseModel.py:96-127); the group plan and its step read int32.  One
// pass narrows a range of the three columns; an id outside [0, 2^31) becomes -1 (the plan then reports it as out of range
// instead of letting it alias a valid row).
__global__ __launch_bounds__(kBlock) void narrow_ids_kernel(const int64_t *__restrict__ u, const int64_t *__restrict__ p,
                                                             const int64_t *__restrict__ n, int *__restrict__ u32,
                                                             int *__restrict__ p32, int *__restrict__ n32, int64_t cnt) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cnt) return;
    const int64_t a = u[i], b = p[i], c = n[i];
    u32[i] = (a >= 0 && a < (int64_t(1) << 31)) ? (int)a : -1;
    p32[i] = (b >= 0 && b < (int64_t(1) << 31)) ? (int)b : -1;
    n32[i] = (c >= 0 && c < (int64_t(1) << 31)) ? (int)c : -1;
}

// ----------------------------------------------------------------------------------------------- step
constexpr int kGsTile = 32;      // list entries per tile: ~16 runs, one per team of the workgroup
constexpr int kGsAhead = 8;      // entries staged beyond the tile
constexpr int kGsShards = 64;
constexpr int kGsShardStride = 16;
constexpr int kGsStepWords = kGsShards * kGsShardStride;
constexpr int kGsDefChunk = 32;   // flag words (1,024 positions) a deferred workgroup collects at a time
constexpr int kGsLdsInts = 1024 + 8;   // one LDS buffer shared by the workgroup kinds: tiles 2*40+33, deferred list 1024
constexpr unsigned kGsSpinLimit = 1u << 22;

struct GsBatch {
    const int *u, *p, *n;
    const uint4 *flags;
    const int *ul_row, *ul_src, *ul_cnt, *il_row, *il_src, *il_cnt;
    int B, R_u, R_i, cap_u, cap_i;
};

struct GsArgs {
    float *U, *I;
    int D;
    float lr;
    GsBatch cur;                 // phase 1: this batch
    float *Z, *ZU, *partials;    // stashes of this batch: z = c U[u] per triplet with a shared item row, g = c (I[p] - I[n]) per
    float denom;                 // triplet with a shared user row
    GsBatch prev;                // tiles: the batch before (nIT + nUT > 0)
    const float *Zp, *ZUp, *partials_prev;
    int n_partials_prev;
    float denom_prev;
    float *loss_prev;
    int nA, nIT, nUT, nDS, tiles_at, side_at;
    int chained;                 // the tiles of `prev` ride in this launch: DF bits are honoured
    unsigned *done, *timeout;
    // row-sharded step (multi-GPU, whisprrec_amd/sharded.py): item rows >= nL of the item table are rows RECEIVED from their
    // owners for this step (slots); they are read like any row, but instead of being rewritten their gradient row (the sum
    // of +-c U[u] over the batch) goes to Gs[row - nL], to be returned to the owner.  Gs == NULL: not sharded.
    float *Gs;
    int nL;
};

template <int NV>
__device__ __forceinline__ Row<NV> row_zero() {
    Row<NV> r;
#pragma unroll
    for (int k = 0; k < NV; ++k) r.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
}

// One triplet, everything it contributes (plain SGD, l2 = 0: w -= lr g).  Rows that occur once in the batch are read by
// this team only and are finished in place: U[u] with g_u = c (I[p] - I[n]), I[p] with +c U[u], I[n] with -c U[u].  For a row
// with several occurrences the contribution is stashed instead (ZU[t] = g_u; Z[t] = c U[u] for either item side) and the
// tiles that ride in the NEXT launch sum the stashed rows per run and rewrite the row once.
template <int T, int NV, bool FULL>
__device__ __forceinline__ void gs_single(const GsArgs &a, int t, int u, int p, int n, bool us, bool ps, bool ns, int lane,
                                          float &terms) {
    const Row<NV> ur = load_row<T, NV, FULL>(a.U, u, a.D, lane);
    const Row<NV> pr = load_row<T, NV, FULL>(a.I, p, a.D, lane);
    const Row<NV> nr = load_row<T, NV, FULL>(a.I, n, a.D, lane);
    const float sp = team_sum<T>(dot_partial<NV>(ur, pr));
    const float sn = team_sum<T>(dot_partial<NV>(ur, nr));
    float term, c;
    bpr_terms(sp, sn, a.denom, term, c);
    terms += term;
    Row<NV> g, z;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k] = make_float4(c * (pr.v[k].x - nr.v[k].x), c * (pr.v[k].y - nr.v[k].y), c * (pr.v[k].z - nr.v[k].z),
                             c * (pr.v[k].w - nr.v[k].w));
        z.v[k] = make_float4(c * ur.v[k].x, c * ur.v[k].y, c * ur.v[k].z, c * ur.v[k].w);
    }
    const float lr = a.lr;
    if (!us) {
        Row<NV> w;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            w.v[k] = make_float4(ur.v[k].x - lr * g.v[k].x, ur.v[k].y - lr * g.v[k].y, ur.v[k].z - lr * g.v[k].z,
                                 ur.v[k].w - lr * g.v[k].w);
        store_row<T, NV, FULL>(a.U, u, a.D, lane, w);
    } else {
        store_row<T, NV, FULL>(a.ZU, t, a.D, lane, g);
    }
    if (!ps) {
        if (a.Gs != nullptr && p >= a.nL) {       // a received row: its gradient goes back to the owner
            store_row<T, NV, FULL>(a.Gs, p - a.nL, a.D, lane, z);
        } else {
            Row<NV> w;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                w.v[k] = make_float4(pr.v[k].x - lr * z.v[k].x, pr.v[k].y - lr * z.v[k].y, pr.v[k].z - lr * z.v[k].z,
                                     pr.v[k].w - lr * z.v[k].w);
            store_row<T, NV, FULL>(a.I, p, a.D, lane, w);
        }
    }
    if (!ns) {
        if (a.Gs != nullptr && n >= a.nL) {
            Row<NV> zn;
#pragma unroll
            for (int k = 0; k < NV; ++k) zn.v[k] = make_float4(-z.v[k].x, -z.v[k].y, -z.v[k].z, -z.v[k].w);
            store_row<T, NV, FULL>(a.Gs, n - a.nL, a.D, lane, zn);
        } else {
            Row<NV> w;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                w.v[k] = make_float4(nr.v[k].x - lr * (-z.v[k].x), nr.v[k].y - lr * (-z.v[k].y), nr.v[k].z - lr * (-z.v[k].z),
                                     nr.v[k].w - lr * (-z.v[k].w));
            store_row<T, NV, FULL>(a.I, n, a.D, lane, w);
        }
    }
    if (ps || ns) store_row<T, NV, FULL>(a.Z, t, a.D, lane, z);
}

// wave 0 of the workgroup waits until every tile workgroup of this launch has signalled; then the workgroup may read the
// rows those tiles stored (write-through) with plain loads.  The poll is bounded: on expiry the sticky word is set.
__device__ __forceinline__ void gs_wait_tiles(const GsArgs &a) {
    if (threadIdx.x < 64) {
        const int j = (int)threadIdx.x;
        const int n_sig = a.nIT + a.nUT;
        const unsigned want = j < kGsShards ? (unsigned)((n_sig - j + kGsShards - 1) / kGsShards) : 0u;
        const unsigned *shard = a.done + (j < kGsShards ? j : 0) * kGsShardStride;
        unsigned spins = 0;
        for (;;) {
            const bool ok = __hip_atomic_load(shard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kGsSpinLimit) {
                if (j == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// tile q of a batch side's lists -> (segment, first entry, entries of the segment); false when q is beyond the last tile.
// All segment lengths are fetched together (one round trip, not one per segment).
__device__ __forceinline__ bool gs_tile_of(const int *__restrict__ cnt, int R, int cap, int q, int &seg, int &e0, int &len) {
    int c[kGpMaxRanges];
#pragma unroll
    for (int s = 0; s < kGpMaxRanges; ++s) c[s] = cnt[s < R ? s : 0];      // unconditional loads, one wait for all of them
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
#pragma unroll
    for (int s = 0; s < kGpMaxRanges; ++s)
        if (s >= R) c[s] = 0;
    bool found = false;
#pragma unroll
    for (int s = 0; s < kGpMaxRanges; ++s) {
        const int cs = min(c[s], cap);
        const int nt = (cs + kGsTile - 1) / kGsTile;
        if (!found && q < nt) {
            seg = s;
            e0 = q * kGsTile;
            len = cs;
            found = true;
        }
        if (!found) q -= nt;
    }
    return found;
}

// Tiles over one side's lists of one batch (users: table U, stash ZU; items: table I, stash Z).  One team per run (entries
// of equal row): the row and the first stashed contributions are requested together, contributions are summed in list
// order (fixed) and the row is rewritten once — write-through when WT.  A source is (t << 1) | negative: the stash row t,
// subtracted for the negative side of an item.  A tile is FETCHED (one list entry per thread, into registers) and FINISHED
// in two calls, so that a workgroup can fetch its tile early and finish it behind other work.
struct GsTile {
    const int *lrow, *lsrc;
    int e0, len;          // first entry of the tile in its segment, entries of the segment
    int er, es, before;   // this thread's staged entry (row, source); thread 0: the row in front of the tile
};

__device__ __forceinline__ void gs_tile_fetch(GsTile &tl, const int *__restrict__ l_row, const int *__restrict__ l_src, int seg,
                                              int cap, int e0, int len, int B) {
    tl.lrow = l_row + (int64_t)seg * cap;
    tl.lsrc = l_src + (int64_t)seg * cap;
    tl.e0 = e0;
    tl.len = len;
    tl.er = -1;
    tl.es = 0;
    tl.before = -1;
    const int e = e0 + (int)threadIdx.x;
    if (threadIdx.x < kGsTile + kGsAhead && e < len) {
        tl.er = tl.lrow[e];      // raw: nothing here may USE a loaded value (the loads fly beside the triplets' row loads)
        tl.es = tl.lsrc[e];
    }
    if (threadIdx.x == 0 && e0 > 0) tl.before = tl.lrow[e0 - 1];
}

template <int T, int NV, bool FULL, bool WT>
__device__ __forceinline__ void gs_tile_finish(float *__restrict__ W, int D, float lr, int B, const float *__restrict__ Z,
                                               const GsTile &tl, int *__restrict__ lds, float *__restrict__ Gs = nullptr,
                                               int nL = 0) {
    int *rows_t = lds, *src_t = rows_t + (kGsTile + kGsAhead), *heads = src_t + (kGsTile + kGsAhead);
    int &n_heads = heads[kGsTile];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    __syncthreads();   // whatever used the LDS buffer before is done with it
    if (threadIdx.x == 0) n_heads = 0;
    if (threadIdx.x < kGsTile + kGsAhead) {
        rows_t[threadIdx.x] = tl.er;
        src_t[threadIdx.x] = min(max(tl.es, 0), 2 * B - 1);   // plan arrays are index data: stay inside the stash whatever they hold
    }
    __syncthreads();
    if (threadIdx.x < kGsTile) {
        const int r = rows_t[threadIdx.x];
        const int bf = threadIdx.x == 0 ? tl.before : rows_t[threadIdx.x - 1];
        if (r >= 0 && bf != r) heads[atomicAdd(&n_heads, 1)] = threadIdx.x;
    }
    __syncthreads();
    const int nh = n_heads;
    for (int h = threadIdx.x / T; h < nh; h += TEAMS) {
        const int j0 = heads[h];
        const int r = rows_t[j0];
        const bool slot = Gs != nullptr && r >= nL;     // row-sharded step: a received row — its summed gradient goes to Gs
        Row<NV> ir = row_zero<NV>();
        if (!slot) ir = load_row<T, NV, FULL>(W, r, D, lane);
        const int s0 = src_t[j0];
        const bool two = rows_t[j0 + 1] == r;     // kGsAhead >= 1: in the staged window
        const int s1 = src_t[j0 + 1];
        const Row<NV> z0 = load_row<T, NV, FULL>(Z, s0 >> 1, D, lane);
        Row<NV> z1 = row_zero<NV>();
        if (two) z1 = load_row<T, NV, FULL>(Z, s1 >> 1, D, lane);
        Row<NV> g;
        const float g0 = (s0 & 1) ? -1.0f : 1.0f, g1 = (s1 & 1) ? -1.0f : 1.0f;   // d/dI[p] = +c U, d/dI[n] = -c U
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            g.v[k].x = fmaf(g0, z0.v[k].x, 0.f);
            g.v[k].y = fmaf(g0, z0.v[k].y, 0.f);
            g.v[k].z = fmaf(g0, z0.v[k].z, 0.f);
            g.v[k].w = fmaf(g0, z0.v[k].w, 0.f);
        }
        if (two) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                g.v[k].x = fmaf(g1, z1.v[k].x, g.v[k].x);
                g.v[k].y = fmaf(g1, z1.v[k].y, g.v[k].y);
                g.v[k].z = fmaf(g1, z1.v[k].z, g.v[k].z);
                g.v[k].w = fmaf(g1, z1.v[k].w, g.v[k].w);
            }
            // third and later occurrences: from LDS inside the staged window, then (long runs) from the list itself — two
            // loops, so that no load has to choose between an LDS and a global address
            // (four stashed rows requested per trip for the long runs of skewed ids: Zipf(0.5) 34.0 -> 30.9 us per launch, but
            // the uniform case 22.5 -> 25.5 — the four extra rows of registers cost the common path its occupancy; not adopted)
            auto add_z = [&](int src) {
                const Row<NV> z = load_row<T, NV, FULL>(Z, src >> 1, D, lane);
                const float sgn = (src & 1) ? -1.0f : 1.0f;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    g.v[k].x = fmaf(sgn, z.v[k].x, g.v[k].x);
                    g.v[k].y = fmaf(sgn, z.v[k].y, g.v[k].y);
                    g.v[k].z = fmaf(sgn, z.v[k].z, g.v[k].z);
                    g.v[k].w = fmaf(sgn, z.v[k].w, g.v[k].w);
                }
            };
            int j = j0 + 2;
            for (; j < kGsTile + kGsAhead && rows_t[j] == r; ++j) add_z(src_t[j]);
            if (j == kGsTile + kGsAhead)
                for (int e = tl.e0 + j; e < tl.len && tl.lrow[e] == r; ++e) add_z(min(max(tl.lsrc[e], 0), 2 * B - 1));
        }
        if (slot) {
            store_row<T, NV, FULL>(Gs, r - nL, D, lane, g);
            continue;
        }
        Row<NV> wv;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            wv.v[k] = make_float4(ir.v[k].x - lr * g.v[k].x, ir.v[k].y - lr * g.v[k].y, ir.v[k].z - lr * g.v[k].z,
                                  ir.v[k].w - lr * g.v[k].w);
        if constexpr (WT) store_row_wt<T, NV, FULL>(W, r, D, lane, wv);
        else store_row<T, NV, FULL>(W, r, D, lane, wv);
    }
}

#ifndef WR_GS_WAVES
#define WR_GS_WAVES 8
#endif
#ifndef WR_GS_DBG
#define WR_GS_DBG 0     // timing-only variants (A/B builds, never shipped): 1 tiles only signal, 2 no deferred workgroups' work,
#endif                  // 4 no wait, 8 main teams ignore the flag words

template <int T, int NV, bool FULL, bool WT>
__device__ __forceinline__ void gs_list_tiles(float *__restrict__ W, int D, float lr, const int *__restrict__ l_row,
                                              const int *__restrict__ l_src, const int *__restrict__ l_cnt, int R, int cap,
                                              int B, const float *__restrict__ Z, int w, int nw, int *__restrict__ lds,
                                              float *__restrict__ Gs = nullptr, int nL = 0) {
    for (int q = w;; q += nw) {      // workgroup `w` of `nw` takes tiles w, w + nw, ...
        int seg, e0, len;
        if (!gs_tile_of(l_cnt, R, cap, q, seg, e0, len)) break;
        GsTile tl;
        gs_tile_fetch(tl, l_row, l_src, seg, cap, e0, len, B);
        gs_tile_finish<T, NV, FULL, WT>(W, D, lr, B, Z, tl, lds, Gs, nL);
    }
}

// One launch = the triplets of batch `cur` + the tiles of batch `prev`.  Workgroups, by blockIdx.x:
//   [main 0 .. tiles_at) [item tiles: nIT] [user tiles: nUT] [main tiles_at .. side_at) [deferred: nDS] [main side_at .. nA)
// (Measured and not adopted: every tile finished by one of the first main workgroups, its list entries fetched beside the
// triplets' rows — 27.2 against 25.7 us per launch at 1M x 1M x 64, B = 65,536: the tile's row loads and the drain of its
// write-through stores then sit behind the triplets in the SAME workgroup, and a workgroup's lifetime is what counts.)
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock, NV == 1 ? WR_GS_WAVES : 1) void bprmf_group_step(GsArgs a) {
    __shared__ float scratch[kBlock / 64];
    __shared__ __attribute__((aligned(16))) int lds[kGsLdsInts];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    int b = (int)blockIdx.x;
    if (b >= a.tiles_at && b < a.tiles_at + a.nIT + a.nUT) {
        const int w = b - a.tiles_at;
        const bool wt = a.nA > 0;      // tiles that ride beside the next batch's triplets hand their rows over
        if (!(WR_GS_DBG & 1)) {
            if (w < a.nIT) {
                if (wt) gs_list_tiles<T, NV, FULL, true>(a.I, a.D, a.lr, a.prev.il_row, a.prev.il_src, a.prev.il_cnt, a.prev.R_i, a.prev.cap_i,
                                                         a.prev.B, a.Zp, w, a.nIT, lds, a.Gs, a.nL);
                else gs_list_tiles<T, NV, FULL, false>(a.I, a.D, a.lr, a.prev.il_row, a.prev.il_src, a.prev.il_cnt, a.prev.R_i, a.prev.cap_i,
                                                       a.prev.B, a.Zp, w, a.nIT, lds, a.Gs, a.nL);
            } else {
                if (wt) gs_list_tiles<T, NV, FULL, true>(a.U, a.D, a.lr, a.prev.ul_row, a.prev.ul_src, a.prev.ul_cnt, a.prev.R_u, a.prev.cap_u,
                                                         a.prev.B, a.ZUp, w - a.nIT, a.nUT, lds);
                else gs_list_tiles<T, NV, FULL, false>(a.U, a.D, a.lr, a.prev.ul_row, a.prev.ul_src, a.prev.ul_cnt, a.prev.R_u, a.prev.cap_u,
                                                       a.prev.B, a.ZUp, w - a.nIT, a.nUT, lds);
            }
        }
        if (wt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its write-through stores have left
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(a.done + (w % kGsShards) * kGsShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (w == 0 && a.loss_prev != nullptr) {   // fold the loss partials of the step before (fixed order)
            const float s0 = strided_partial_sum(a.partials_prev, a.n_partials_prev);
            const float s = block_sum(s0, scratch);
            if (threadIdx.x == 0) a.loss_prev[0] = s / a.denom_prev;
        }
        return;
    }
    if (b >= a.tiles_at) b -= a.nIT + a.nUT;
    float terms = 0.f;
    int slot;   // index of this workgroup's loss partial
    if (b >= a.side_at && b < a.side_at + a.nDS) {
        // deferred triplets: this workgroup's slice of the flag words, in position order.  The list and the first
        // triplets' ids are fetched BEFORE the wait for the tiles; only the rows are read behind it.
        const int w = b - a.side_at;
        int *dl = lds, &total_s = lds[kGsDefChunk * 32];
        const int fw = (a.cur.B + 31) / 32;
        const int per = (fw + a.nDS - 1) / a.nDS;
        const int w_end = min(fw, (w + 1) * per);
        bool waited = false;
        for (int c0 = w * per; c0 < w_end; c0 += kGsDefChunk) {
            const int wi = c0 + (int)threadIdx.x;
            unsigned m = 0u, us = 0u, ps = 0u, ns = 0u;
            if (threadIdx.x < kGsDefChunk && wi < w_end) {
                const uint4 f = a.cur.flags[wi];
                m = f.w;
                us = f.x;
                ps = f.y;
                ns = f.z;
            }
            // exclusive prefix of the popcounts (first wave; lanes beyond the chunk hold 0)
            int cnt = __popc(m), incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if ((int)(threadIdx.x & 63) >= d) incl += v;
            }
            if (threadIdx.x == 63) total_s = incl;
            int at = incl - cnt;
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                dl[at++] = ((wi * 32 + bit) << 3) | (((us >> bit) & 1u) ? 4 : 0) | (((ps >> bit) & 1u) ? 1 : 0) |
                           (((ns >> bit) & 1u) ? 2 : 0);
            }
            __syncthreads();
            const int total = (WR_GS_DBG & 2) ? 0 : total_s;
            // the ids of a team's first two triplets are requested before the wait
            const int e0 = (int)threadIdx.x / T, e1 = e0 + TEAMS;
            int t0 = 0, u0 = 0, p0 = 0, n0 = 0, t1 = 0, u1 = 0, p1 = 0, n1 = 0;
            if (e0 < total) {
                t0 = min(dl[e0] >> 3, a.cur.B - 1);
                u0 = a.cur.u[t0];
                p0 = a.cur.p[t0];
                n0 = a.cur.n[t0];
            }
            if (e1 < total) {
                t1 = min(dl[e1] >> 3, a.cur.B - 1);
                u1 = a.cur.u[t1];
                p1 = a.cur.p[t1];
                n1 = a.cur.n[t1];
            }
            if (!waited) {
                if (!(WR_GS_DBG & 4)) gs_wait_tiles(a);      // ends with a workgroup barrier
                waited = true;
            }
            if (e0 < total) gs_single<T, NV, FULL>(a, t0, u0, p0, n0, dl[e0] & 4, dl[e0] & 1, dl[e0] & 2, lane, terms);
            if (e1 < total) gs_single<T, NV, FULL>(a, t1, u1, p1, n1, dl[e1] & 4, dl[e1] & 1, dl[e1] & 2, lane, terms);
            for (int e = e1 + TEAMS; e < total; e += TEAMS) {
                const int t = min(dl[e] >> 3, a.cur.B - 1);
                gs_single<T, NV, FULL>(a, t, a.cur.u[t], a.cur.p[t], a.cur.n[t], dl[e] & 4, dl[e] & 1, dl[e] & 2, lane, terms);
            }
            __syncthreads();
        }
        slot = a.nA + w;
    } else {
        if (b >= a.side_at) b -= a.nDS;
        const int t = b * TEAMS + (int)threadIdx.x / T;
        // the flag word and the three ids go out together, on a clamped position and with no branch between them: left to
        // itself the compiler sinks the id loads behind the test of the flag word (two more dependent round trips in front
        // of the row loads, and a workgroup's lifetime is what bounds the bytes in flight)
        const int tc = min(t, a.cur.B - 1);
        uint4 f = a.cur.flags[tc >> 5];
        int u = a.cur.u[tc], p = a.cur.p[tc], n = a.cur.n[tc];
#if WR_GS_DBG & 8
        f = make_uint4(0u, 0u, 0u, 0u);      // timing only: no flag word (every row finished in place: wrong tables)
#endif
        asm volatile("" : "+v"(u), "+v"(p), "+v"(n), "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w));
        const unsigned bit = 1u << (t & 31);
        if (t < a.cur.B && !(a.chained && (f.w & bit)))
            gs_single<T, NV, FULL>(a, t, u, p, n, f.x & bit, f.y & bit, f.z & bit, lane, terms);
        slot = b;
    }
    if (lane != 0) terms = 0.f;   // every lane of a team holds the same terms: count them once
    const float sum = block_sum(terms, scratch);
    if (threadIdx.x == 0) a.partials[slot] = sum;
}

static inline int gs_teams_per_block(int D) { return D >= 64 ? kBlock / 16 : (D == 32 ? kBlock / 8 : (D == 16 ? kBlock / 4 : (D == 8 ? kBlock / 2 : (D == 4 ? kBlock : kBlock / 16)))); }

static inline int64_t gs_ws_one(int64_t B, int32_t D) {
    const int64_t nA = (B + gs_teams_per_block(D) - 1) / gs_teams_per_block(D);
    return 2 * align_up(B * (int64_t)D * 4, 256) + align_up((nA + 512) * 4, 256);
}

static inline bool gs_shape_ok(const float *U, const float *I, int32_t D) {
    return (D * 4) % 128 == 0 && (reinterpret_cast<uintptr_t>(U) & 127u) == 0 && (reinterpret_cast<uintptr_t>(I) & 127u) == 0;
}

#ifndef WR_GS_TILES_AT
#define WR_GS_TILES_AT 0     // where the tile workgroups sit among the main ones (in 1/16 of the main grid)
#endif
#ifndef WR_GS_SIDE_AT
#define WR_GS_SIDE_AT 12     // where the deferred workgroups sit among the main ones (in 1/16 of the main grid).  They poll from
#endif                       // the moment they are dispatched: A/B on MI355X, 1M x 1M x 64, B = 65,536, us per launch at 2 / 6 / 8 /
                             // 10 / 12 / 14 sixteenths: 25.2 / 23.8 / 23.2 / 22.5 / 22.3 / 21.7-21.9 (12-14 tie within noise; the
                             // ~1,150 deferred triplets of a batch take 64 workgroups two rounds of teams, ~4 us)
#ifndef WR_GS_NUT
#define WR_GS_NUT 256
#endif
#ifndef WR_GS_NDS
#define WR_GS_NDS 64
#endif
#ifndef WR_GS_NIT
#define WR_GS_NIT 1024
#endif

template <int T, int NV, bool FULL>
static int32_t launch_group_steps(float *U, float *I, int32_t D, const int32_t *u, const int32_t *p, const int32_t *n,
                                  int64_t n_triplets, const GroupLayout &L, const GpDev &G, int64_t first_batch,
                                  int64_t n_batches, float lr, float *loss_out, void *workspace, uint32_t *sync,
                                  int64_t sync_words, hipStream_t stream, void *const *events, int n_cu,
                                  float *grad_slots = nullptr, int64_t n_local_items = 0, int64_t global_batch = 0) {
    const int64_t B = L.B;
    const int64_t ws_one = gs_ws_one(B, D), zb = align_up(B * (int64_t)D * 4, 256);
    char *ws = reinterpret_cast<char *>(workspace);
    float *Zs[2] = {reinterpret_cast<float *>(ws), reinterpret_cast<float *>(ws + ws_one)};
    float *ZUs[2] = {reinterpret_cast<float *>(ws + zb), reinterpret_cast<float *>(ws + ws_one + zb)};
    float *Ps[2] = {reinterpret_cast<float *>(ws + 2 * zb), reinterpret_cast<float *>(ws + ws_one + 2 * zb)};
    constexpr int TEAMS = kBlock / T;
    WR_HIP(hipMemsetAsync(sync, 0, (size_t)(n_batches * kGsStepWords) * 4, stream));
    uint32_t *timeout = sync + (sync_words - 4);
    // the workgroups that wait inside a launch (deferred triplets) must stay below the resident workgroup slots whatever
    // the device: at most a quarter of a workgroup per CU; nothing they wait for (the tiles) ever waits
    int nDS = WR_GS_NDS;
    while (nDS > n_cu / 4 && nDS > 1) nDS >>= 1;
    auto batch_of = [&](int64_t b) {
        const int64_t off = b * B;
        const int Bk = (int)((off + B <= n_triplets) ? B : (n_triplets - off));
        return GsBatch{u + off, p + off, n + off, reinterpret_cast<const uint4 *>(G.flags + b * L.fw * 4),
                       G.ul_row + b * L.R_u * (int64_t)L.cap_u, G.ul_src + b * L.R_u * (int64_t)L.cap_u, G.ucnt + b * L.R_u,
                       G.il_row + b * L.R_i * (int64_t)L.cap_i, G.il_src + b * L.R_i * (int64_t)L.cap_i, G.icnt + b * L.R_i,
                       Bk, L.R_u, L.R_i, L.cap_u, L.cap_i};
    };
    auto ev = [&](int64_t k, int j) { return events ? reinterpret_cast<hipEvent_t>(events[2 * k + j]) : (hipEvent_t) nullptr; };
    int n_partials_prev = 0;
    for (int64_t k = 0; k <= n_batches; ++k) {
        GsArgs a{};
        a.U = U;
        a.I = I;
        a.D = D;
        a.lr = lr;
        a.done = sync + (k < n_batches ? k : 0) * kGsStepWords;
        a.timeout = timeout;
        a.Gs = grad_slots;
        a.nL = (int)n_local_items;
        const bool have_cur = k < n_batches, have_prev = k > 0;
        if (have_cur) {
            a.cur = batch_of(first_batch + k);
            a.Z = Zs[k & 1];
            a.ZU = ZUs[k & 1];
            a.partials = Ps[k & 1];
            a.denom = global_batch > 0 ? (float)global_batch : (float)a.cur.B;
            a.nA = (int)((a.cur.B + TEAMS - 1) / TEAMS);
            a.nDS = have_prev ? nDS : 0;
            a.chained = have_prev ? 1 : 0;
        }
        if (have_prev) {
            a.prev = batch_of(first_batch + k - 1);
            a.Zp = Zs[(k - 1) & 1];
            a.ZUp = ZUs[(k - 1) & 1];
            a.partials_prev = Ps[(k - 1) & 1];
            a.n_partials_prev = n_partials_prev;
            a.denom_prev = global_batch > 0 ? (float)global_batch : (float)a.prev.B;
            a.loss_prev = loss_out ? loss_out + (k - 1) : nullptr;
            // one tile (32 list entries) per workgroup where the lists are as long as uniform ids make them; longer lists
            // are walked in strides
            a.nIT = (int)std::min<int64_t>(WR_GS_NIT, (2 * B + kGsTile - 1) / kGsTile + L.R_i);
            a.nUT = (int)std::min<int64_t>(WR_GS_NUT, (B + kGsTile - 1) / kGsTile + L.R_u);
        }
        a.tiles_at = (int)((int64_t)a.nA * WR_GS_TILES_AT / 16);
        a.side_at = std::max(a.tiles_at, (int)((int64_t)a.nA * WR_GS_SIDE_AT / 16));
        const dim3 grid((unsigned)(a.nA + a.nIT + a.nUT + a.nDS));
        hipEvent_t e0 = have_cur ? ev(k, 0) : nullptr, e1 = have_cur ? ev(k, 1) : nullptr;
        if (e0 != nullptr || e1 != nullptr)
            hipExtLaunchKernelGGL((bprmf_group_step<T, NV, FULL>), grid, dim3(kBlock), 0, stream, e0, e1, 0, a);
        else
            hipLaunchKernelGGL((bprmf_group_step<T, NV, FULL>), grid, dim3(kBlock), 0, stream, a);
        WR_LAUNCH_CHECK("bprmf_group_step");
        n_partials_prev = a.nA + a.nDS;
    }
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_group_plan_words(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items) {
    GroupLayout L;
    if (!group_layout(n_triplets, batch_size, n_users, n_items, L)) return 0;   // 0: not applicable (use the sorted plan)
    return L.total;
}

int32_t wr_group_plan_layout(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items, int64_t *out) {
    WR_REQUIRE(out != nullptr, WR_E_NULL, "out is NULL");
    GroupLayout L;
    WR_REQUIRE(group_layout(n_triplets, batch_size, n_users, n_items, L), WR_E_RANGE,
               "group plan not applicable to n=%lld, batch=%lld", (long long)n_triplets, (long long)batch_size);
    const int64_t v[16] = {L.nb, L.fw, L.R_u, L.R_i, (int64_t)L.mask_u, (int64_t)L.mask_i, L.flags, L.ucnt, L.icnt, L.ul_row,
                           L.ul_src, L.il_row, L.il_src, L.total, L.cap_u, L.cap_i};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
    return WR_OK;
}

int32_t wr_narrow_ids_i64(const int64_t *u, const int64_t *p, const int64_t *n, int32_t *u32, int32_t *p32, int32_t *n32,
                          int64_t count, void *stream_) {
    WR_REQUIRE(u && p && n && u32 && p32 && n32, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(count >= 0 && count < (int64_t(1) << 40), WR_E_SHAPE, "count out of range");
    if (count == 0) return WR_OK;
    hipLaunchKernelGGL(narrow_ids_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream_), u, p, n, u32, p32, n32, count);
    WR_LAUNCH_CHECK("narrow_ids_kernel");
    return WR_OK;
}

int32_t wr_group_plan_build(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size,
                            int64_t n_users, int64_t n_items, int32_t *plan, int64_t plan_words, void *stream_) {
    WR_REQUIRE(u && p && n && plan, WR_E_NULL, "index arrays / plan must not be NULL");
    GroupLayout L;
    WR_REQUIRE(group_layout(n_triplets, batch_size, n_users, n_items, L), WR_E_RANGE,
               "group plan not applicable to n=%lld, batch=%lld", (long long)n_triplets, (long long)batch_size);
    WR_REQUIRE(aligned16(plan) && plan_words >= L.total, WR_E_WORKSPACE, "group plan: %lld words < %lld", (long long)plan_words,
               (long long)L.total);
    WR_REQUIRE(L.nb * (L.R_u + L.R_i) < (int64_t(1) << 31), WR_E_SHAPE, "group plan: too many batches in one plan");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    WR_HIP(hipMemsetAsync(plan, 0, (size_t)L.zero_words * 4, stream));
    const size_t lds = (size_t)(2 * kGpCap + kGpWords + 3 * L.fw + 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(group_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((2 * kGpCap + kGpWords + 3 * ((kGpMaxBatch + 31) / 32) + 4) * 4)));
        attr_set = true;
    }
    const GpDev G = group_dev(plan, L);
    hipLaunchKernelGGL(group_plan_kernel, dim3((unsigned)(L.nb * (L.R_u + L.R_i))), dim3(kGpThreads), lds, stream, u, p, n,
                       n_triplets, (int)batch_size, (int)L.nb, (int)n_users, (int)n_items, G);
    WR_LAUNCH_CHECK("group_plan_kernel");
    return WR_OK;
}

int64_t wr_bprmf_group_workspace_bytes(int64_t batch_size, int32_t D) { return 2 * gs_ws_one(batch_size, D); }

int64_t wr_bprmf_group_sync_words(int64_t n_batches) { return n_batches < 0 ? WR_E_SHAPE : (n_batches + 1) * kGsStepWords + 4; }

int32_t wr_bprmf_group_supported(const float *user_tab, const float *item_tab, int32_t D) {
    return (user_tab && item_tab && D >= 4 && D <= 1024 && D % 4 == 0 && gs_shape_ok(user_tab, item_tab, D)) ? 1 : 0;
}

int32_t wr_bprmf_run_sgd_group(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                               const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size,
                               const int32_t *plan, int64_t plan_words, int64_t first_batch, int64_t n_batches, float lr,
                               float *loss_out, void *const *events, void *workspace, int64_t workspace_bytes, int32_t *sync,
                               int64_t sync_words, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    WR_REQUIRE(u && p && n && plan && sync, WR_E_NULL, "index arrays / plan / sync words must not be NULL");
    WR_REQUIRE(gs_shape_ok(user_tab, item_tab, D), WR_E_ALIGN,
               "wr_bprmf_run_sgd_group: rows must be whole 128-B lines (D %% 32 == 0, tables 128-B aligned); D = %d", (int)D);
    GroupLayout L;
    WR_REQUIRE(group_layout(n_triplets, batch_size, n_users, n_items, L), WR_E_RANGE,
               "group plan not applicable to n=%lld, batch=%lld", (long long)n_triplets, (long long)batch_size);
    WR_REQUIRE(plan_words >= L.total, WR_E_WORKSPACE, "group plan: %lld words < %lld", (long long)plan_words, (long long)L.total);
    WR_REQUIRE(first_batch >= 0 && n_batches >= 0 && first_batch + n_batches <= L.nb, WR_E_SHAPE,
               "batches [%lld,%lld) exceed the plan's %lld", (long long)first_batch, (long long)(first_batch + n_batches),
               (long long)L.nb);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= 2 * gs_ws_one(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_run_sgd_group: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)(2 * gs_ws_one(batch_size, D)));
    WR_REQUIRE(aligned16(sync) && sync_words >= (n_batches + 1) * kGsStepWords + 4, WR_E_WORKSPACE,
               "wr_bprmf_run_sgd_group: %lld sync words < %lld", (long long)sync_words,
               (long long)((n_batches + 1) * kGsStepWords + 4));
    if (n_batches == 0) return WR_OK;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        WR_HIP(hipGetDevice(&dev));
        WR_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const GpDev G = group_dev(const_cast<int32_t *>(plan), L);
#define WR_CALL_GS(T_, NV_, FULL_)                                                                                          \
    return launch_group_steps<T_, NV_, FULL_>(user_tab, item_tab, D, u, p, n, n_triplets, L, G, first_batch, n_batches, lr, \
                                              loss_out, workspace, reinterpret_cast<uint32_t *>(sync), sync_words, stream,  \
                                              events, n_cu)
    WR_DISPATCH_D(D, WR_CALL_GS);
#undef WR_CALL_GS
    return WR_OK;
}

int32_t wr_bprmf_shard_step_group(float *user_shard, int64_t n_user_rows, float *item_ext, int64_t n_ext_rows,
                                  int64_t n_local_items, int32_t D, const int32_t *vu, const int32_t *vp, const int32_t *vn,
                                  int64_t n_triplets, int64_t batch_size, const int32_t *plan, int64_t plan_words, int64_t batch,
                                  int64_t global_batch, float lr, float *grad_slots, float *loss_partial, void *workspace,
                                  int64_t workspace_bytes, int32_t *sync, int64_t sync_words, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_shard, n_user_rows, D, "user_shard")) != WR_OK) return rc;
    if ((rc = check_table(item_ext, n_ext_rows, D, "item_ext")) != WR_OK) return rc;
    WR_REQUIRE(vu && vp && vn && plan && sync && grad_slots, WR_E_NULL, "index arrays / plan / sync words / grad_slots must not be NULL");
    WR_REQUIRE(aligned16(grad_slots), WR_E_ALIGN, "grad_slots is not 16-byte aligned");
    WR_REQUIRE(n_local_items >= 0 && n_local_items <= n_ext_rows, WR_E_SHAPE, "n_local_items %lld outside [0, %lld]",
               (long long)n_local_items, (long long)n_ext_rows);
    WR_REQUIRE(gs_shape_ok(user_shard, item_ext, D), WR_E_ALIGN,
               "wr_bprmf_shard_step_group: rows must be whole 128-B lines (D %% 32 == 0, tables 128-B aligned); D = %d", (int)D);
    GroupLayout L;
    WR_REQUIRE(group_layout(n_triplets, batch_size, n_user_rows, n_ext_rows, L), WR_E_RANGE,
               "group plan not applicable to n=%lld, batch=%lld", (long long)n_triplets, (long long)batch_size);
    WR_REQUIRE(plan_words >= L.total, WR_E_WORKSPACE, "group plan: %lld words < %lld", (long long)plan_words, (long long)L.total);
    WR_REQUIRE(batch >= 0 && batch < L.nb, WR_E_SHAPE, "batch %lld outside the plan's %lld", (long long)batch, (long long)L.nb);
    WR_REQUIRE(global_batch >= batch_size, WR_E_SHAPE, "global_batch %lld < local batch %lld", (long long)global_batch,
               (long long)batch_size);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= 2 * gs_ws_one(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_shard_step_group: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)(2 * gs_ws_one(batch_size, D)));
    WR_REQUIRE(aligned16(sync) && sync_words >= 2 * kGsStepWords + 4, WR_E_WORKSPACE, "wr_bprmf_shard_step_group: %lld sync words < %lld",
               (long long)sync_words, (long long)(2 * kGsStepWords + 4));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const GpDev G = group_dev(const_cast<int32_t *>(plan), L);
    // one batch = the launch of its triplets + the launch of its own tiles (the tiles cannot ride in the next step's launch:
    // the received rows and the gradient slots belong to THIS step's exchange)
#define WR_CALL_GSS(T_, NV_, FULL_)                                                                                        \
    return launch_group_steps<T_, NV_, FULL_>(user_shard, item_ext, D, vu, vp, vn, n_triplets, L, G, batch, 1, lr,         \
                                              loss_partial, workspace, reinterpret_cast<uint32_t *>(sync), sync_words,     \
                                              stream, nullptr, 256, grad_slots, n_local_items, global_batch)
    WR_DISPATCH_D(D, WR_CALL_GSS);
#undef WR_CALL_GSS
    return WR_OK;
}

}  // extern "C"
