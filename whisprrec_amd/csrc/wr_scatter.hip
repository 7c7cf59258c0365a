// wr_scatter.hip — scatter-add of rows into a table of ANY size without a sort of the positions (gfx950 / MI355X).
//
// Reference path restated (paths relative to the reference root): the backward of an embedding lookup —
// src/models/sequential/SASRec.py:60,84,105-106 (`nn.Embedding(padding_idx=0)` gathered at B x T positions; autograd's
// index_put with accumulate) — is  table[idx[k]] += alpha * src[k]  for all k, duplicates summed.  The multi-GPU step of
// repo:whisprrec_amd/sharded.py applies the gradient rows its peers send back the same way.
//
// wr_rows.hip orders the positions by destination row for this: an LDS counting sort for tables of at most 16 K rows, a
// radix sort of (row, position) pairs (rocPRIM, a dozen launches, ~35 us for 45 K pairs) beyond.  This file takes the
// sort away for the large tables, with the idea of the group plan of wr_group.hip: most rows of a large table occur ONCE
// among the positions and need no ordering at all — only the rows that recur do.
//
// Row plan (index work only; one launch for any number of independent segments of positions):
//   flags   one bit per position: its destination row occurs several times in the segment;
//   lists   those positions only, as (row, position) ordered by (row, position), in R ranges of rows per segment.
// Two launches.  (1) every tile of 2,048 positions deals its (row, position) pairs to the buckets of their ranges (ranks
// from an LDS histogram, one global atomic per tile and range) — each position is visited once.  (2) a workgroup owns
// (segment, range of rows): it loads its bucket (a few thousand pairs) into LDS, sets "seen" / "several" bits of the
// range's rows, flags the pairs whose row recurs and orders those by (row, position) with a counting sort over bins of
// rows + the rank inside a bin (counting the smaller keys: pairs are unique).  The bucket's arrival order (atomics) does not
// reach the result.  A bucket that overflows (more than 8,192 positions in one range: skewed ids) sends its range to the
// slow form of (2): the workgroup scans the whole segment itself.
// Apply (one launch): a team per position adds its row if the row is not shared (read the table row, add, write: the row
// has no other writer); a team per list entry that heads a run sums the run's rows in position order and adds the sum.
// Every table row has one writer and a fixed summation order: no float atomics, the same bits as the sorted path.
// A range whose list would exceed its capacity (more than 8,192 shared positions in one range: a few rows that draw most
// of the positions) is not listed: its shared positions are summed by brute force (every position looks for an earlier one
// of its row, the first one walks the rest) — slow, never wrong; meta[1] tells the caller.
#include "wr_common.h"

namespace wr {

constexpr int kScThreads = 1024;
constexpr unsigned kScRangeBits = 18;                  // rows (or hashed rows) of one range: at most 2^18
constexpr int kScWords = 1 << (kScRangeBits - 5);      // words of one bitmap (32 KiB)
constexpr int kScCap = 8192;                           // list entries per (segment, range): 64 KiB of 8-byte keys in LDS
constexpr int kScBins = 4096;
constexpr int kScMaxRanges = 64;
constexpr int64_t kScMaxSeg = 1 << 18;                 // positions of one segment (flag words in LDS: 32 KiB)
constexpr int kScMetaWords = 16;
constexpr int kScPerRange = 2048;
#ifndef WR_SC_DBG
#define WR_SC_DBG 0      // timing-only variants (A/B builds, never shipped): 1 stop after the bucket load, 2 after the bitmaps,
#endif                 // 4 no flag atomics, 8 stop before the ranking                      // positions per range the layout aims at (all of them may be shared)

struct ScLayout {
    int64_t nseg, stride, fw;
    int R;
    unsigned shift, hbits;
    int64_t flags, cnt, bcnt, lists, total, zero_words;    // offsets in int32 words; meta at 0
};

static bool sc_layout(int64_t nseg, int64_t stride, int64_t n_rows, ScLayout &L) {
    if (nseg <= 0 || stride <= 0 || stride > kScMaxSeg || n_rows <= 0 || n_rows >= (int64_t(1) << 31)) return false;
    L.nseg = nseg;
    L.stride = stride;
    L.fw = (stride + 31) / 32;
    int64_t want = (stride + kScPerRange - 1) / kScPerRange;
    want = want < 1 ? 1 : (want > kScMaxRanges ? kScMaxRanges : want);
    const int64_t rows_per = (n_rows + want - 1) / want;
    unsigned shift = 5;
    while ((int64_t(1) << shift) < rows_per) ++shift;
    L.shift = shift;
    L.hbits = shift < kScRangeBits ? shift : kScRangeBits;
    L.R = (int)(((n_rows - 1) >> shift) + 1);
    if (L.R > kScMaxRanges) return false;
    L.flags = kScMetaWords;
    L.cnt = L.flags + align_up(nseg * L.fw, 4);
    L.bcnt = L.cnt + align_up(nseg * L.R, 4);
    L.zero_words = L.bcnt + align_up(nseg * L.R, 4);
    L.lists = L.zero_words;                                  // per (segment, range): 2 * kScCap words — the bucket's 8-byte
    L.total = L.lists + nseg * L.R * (int64_t)(2 * kScCap);  // pairs first, then rows [kScCap] | positions [kScCap]
    return true;
}

struct ScDev {
    int *meta;
    unsigned *flags;     // [nseg][fw]
    int *cnt;            // [nseg][R]: shared positions of the range (more than kScCap: not listed)
    int *bcnt;           // [nseg][R]: positions of the range (bucket fill; more than kScCap: the bucket is incomplete)
    int *lists;          // [nseg][R][2 * kScCap]
    int R, fw;
    unsigned shift, hbits;
};

static inline ScDev sc_dev(int32_t *plan, const ScLayout &L) {
    return ScDev{plan, reinterpret_cast<unsigned *>(plan + L.flags), plan + L.cnt, plan + L.bcnt, plan + L.lists, L.R, (int)L.fw,
                 L.shift, L.hbits};
}

// f(row, position) over a[0 .. cnt): 16-byte loads (two ids) when the segment is aligned, four in flight per thread
template <typename F>
__device__ __forceinline__ void sc_scan(const int64_t *__restrict__ a, int cnt, F f) {
    if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) {
        const int n2 = cnt >> 1;
        const longlong2 *a2 = reinterpret_cast<const longlong2 *>(a);
        int i = threadIdx.x;
        for (; i + 3 * kScThreads < n2; i += 4 * kScThreads) {
            longlong2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a2[i + q * kScThreads];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int at = 2 * (i + q * kScThreads);
                f(v[q].x, at);
                f(v[q].y, at + 1);
            }
        }
        for (; i < n2; i += kScThreads) {
            const longlong2 v = a2[i];
            f(v.x, 2 * i);
            f(v.y, 2 * i + 1);
        }
        if ((cnt & 1) && threadIdx.x == 0) f(a[cnt - 1], cnt - 1);
    } else {
        for (int j = threadIdx.x; j < cnt; j += kScThreads) f(a[j], j);
    }
}

constexpr int kScTile = 2048;          // positions per workgroup of the bucket pass
constexpr int kScTilePer = kScTile / kBlock;

__device__ __forceinline__ int sc_seg_len(const int *__restrict__ seg_len, int s, int64_t n_total, int64_t stride) {
    int64_t len = seg_len != nullptr ? (int64_t)seg_len[s] : n_total - (int64_t)s * stride;
    return (int)(len < 0 ? 0 : (len > stride ? stride : len));
}

// (1) the pairs of a tile go to the buckets of their ranges
__global__ __launch_bounds__(kBlock) void scatter_bucket_kernel(const int64_t *__restrict__ idx, int64_t stride,
                                                                 const int *__restrict__ seg_len, int64_t n_total, int tiles_per_seg,
                                                                 int64_t n_rows, int64_t padding_idx, ScDev L) {
    __shared__ int hist[kScMaxRanges], base[kScMaxRanges];
    const int s = (int)(blockIdx.x / (unsigned)tiles_per_seg), tile = (int)(blockIdx.x % (unsigned)tiles_per_seg);
    const int len = sc_seg_len(seg_len, s, n_total, stride);
    if (tile * kScTile >= len) return;
    const int64_t *a = idx + (int64_t)s * stride;
    if (threadIdx.x < kScMaxRanges) hist[threadIdx.x] = 0;
    int64_t row[kScTilePer];
#pragma unroll
    for (int j = 0; j < kScTilePer; ++j) {
        const int k = tile * kScTile + (int)threadIdx.x + j * kBlock;
        row[j] = k < len ? a[k] : -1;
        if (row[j] >= n_rows || row[j] == padding_idx) row[j] = -1;
    }
    __syncthreads();
    int rank[kScTilePer];
#pragma unroll
    for (int j = 0; j < kScTilePer; ++j)
        if (row[j] >= 0) rank[j] = atomicAdd(&hist[(int)(row[j] >> L.shift)], 1);
    __syncthreads();
    if ((int)threadIdx.x < L.R) {
        const int h = hist[threadIdx.x];
        base[threadIdx.x] = h ? atomicAdd(&L.bcnt[(int64_t)s * L.R + threadIdx.x], h) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kScTilePer; ++j)
        if (row[j] >= 0) {
            const int r = (int)(row[j] >> L.shift), at = base[r] + rank[j];
            if (at < kScCap) {
                unsigned long long *bucket = reinterpret_cast<unsigned long long *>(L.lists + ((int64_t)s * L.R + r) * (2 * kScCap));
                bucket[at] = ((unsigned long long)(unsigned)row[j] << 32) | (unsigned)(tile * kScTile + (int)threadIdx.x + j * kBlock);
            }
        }
}

// exclusive scan of kScBins counters in place by kScThreads threads (thread t owns four); returns the total to everyone
__device__ __forceinline__ int sc_scan_bins(int *cnt, int *wave_tot) {
    constexpr int kOwn = kScBins / kScThreads;
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
    int run = incl - local, total = 0;
    for (int w = 0; w < kScThreads / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) run += wave_tot[w];
        total += wave_tot[w];
    }
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
        const int c = cnt[c0 + j];
        cnt[c0 + j] = run;
        run += c;
    }
    __syncthreads();
    return total;
}

// (2) One workgroup = (segment, range of rows).  LDS: region X (64 KiB: the keys), region M (32 KiB: the "several" bitmap),
// region S (32 KiB: the "seen" bitmap, then the keys' indices grouped by bin), 20 words of scan totals, region C (16 KiB: the
// bins' counters).
// Slow form (the bucket overflowed): "seen" lives in the first half of X until the list takes X over, the segment's flag
// words in S, counters + indices in M.
__global__ __launch_bounds__(kScThreads) void scatter_plan_kernel(const int64_t *__restrict__ idx, int64_t stride,
                                                                   const int *__restrict__ seg_len, int64_t n_total, int nseg,
                                                                   int64_t n_rows, int64_t padding_idx, ScDev L) {
    extern __shared__ __attribute__((aligned(16))) unsigned sc_lds[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(sc_lds);
    unsigned *multi = sc_lds + 2 * kScCap;
    unsigned *regS = multi + kScWords;
    int *wave_tot = reinterpret_cast<int *>(regS + kScWords);               // 16 words
    int &n_list = wave_tot[16];
    int *cntC = wave_tot + 20;                                               // region C (16 KiB): the bins' counters
    const unsigned lb = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int s = (int)(lb / (unsigned)L.R);
    const unsigned r = lb % (unsigned)L.R;
    const int len = sc_seg_len(seg_len, s, n_total, stride);
    const unsigned bmask = (1u << L.hbits) - 1u;
    const int words = (int)((bmask >> 5) + 1u);
    const unsigned bshift = L.hbits > 12u ? L.hbits - 12u : 0u;
    auto bin_of = [&](unsigned long long k) -> int { return (int)(((unsigned)(k >> 32) & bmask) >> bshift); };
    int *lists = L.lists + ((int64_t)s * L.R + r) * (2 * kScCap);
    int *lrow = lists, *lpos = lists + kScCap;
    unsigned *fb = L.flags + (int64_t)s * L.fw;
    // the bucket's pairs are requested before its fill is known (all kScCap slots exist; what lies beyond the fill is not used)
    constexpr int kPer = kScCap / kScThreads;
    const unsigned long long *bucket = reinterpret_cast<const unsigned long long *>(lists);
    unsigned long long kb[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) kb[q] = bucket[threadIdx.x + q * kScThreads];
    const int nbk = L.bcnt[(int64_t)s * L.R + r];
    if (nbk <= kScCap) {
        // ---- the range's pairs are all in its bucket
        unsigned *seen = regS;
        for (int i = threadIdx.x; i < words; i += kScThreads) {
            seen[i] = 0u;
            multi[i] = 0u;
        }
        int *cnt = cntC;
        for (int i = threadIdx.x; i < kScBins; i += kScThreads) cnt[i] = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if ((int)threadIdx.x + q * kScThreads < nbk) keys[threadIdx.x + q * kScThreads] = kb[q];
        __syncthreads();
        if (WR_SC_DBG & 1) return;
        for (int j = threadIdx.x; j < nbk; j += kScThreads) {
            const unsigned bit = (unsigned)(keys[j] >> 32) & bmask, m = 1u << (bit & 31u);
            const unsigned old = atomicOr(&seen[bit >> 5], m);
            if (old & m) atomicOr(&multi[bit >> 5], m);
        }
        __syncthreads();
        if (WR_SC_DBG & 2) return;
        unsigned short *out16 = reinterpret_cast<unsigned short *>(regS);     // "seen" is done with
        auto kept = [&](unsigned long long k) -> bool {
            const unsigned bit = (unsigned)(k >> 32) & bmask;
            return (multi[bit >> 5] >> (bit & 31u)) & 1u;
        };
        for (int j = threadIdx.x; j < nbk; j += kScThreads) {
            const unsigned long long k = keys[j];
            if (!kept(k)) continue;
            atomicAdd(&cnt[bin_of(k)], 1);
            const unsigned pos = (unsigned)k;
            if (!(WR_SC_DBG & 4)) __hip_atomic_fetch_or(fb + (pos >> 5), 1u << (pos & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int m = sc_scan_bins(cnt, wave_tot);
        for (int j = threadIdx.x; j < nbk; j += kScThreads) {
            const unsigned long long k = keys[j];
            if (kept(k)) out16[atomicAdd(&cnt[bin_of(k)], 1)] = (unsigned short)j;
        }
        __syncthreads();
        if (WR_SC_DBG & 8) return;
        for (int j = threadIdx.x; j < nbk; j += kScThreads) {                 // cnt[bin] is now the END of the bin
            const unsigned long long k = keys[j];
            if (!kept(k)) continue;
            const int bin = bin_of(k);
            const int lo = bin ? cnt[bin - 1] : 0, hi = cnt[bin];
            int pos = lo;
            for (int jj = lo; jj < hi; ++jj) pos += keys[out16[jj]] < k ? 1 : 0;   // a hot row's bin is long: still exact
            lrow[pos] = (int)(k >> 32);                                       // (the bucket is in LDS: its memory is the list's)
            lpos[pos] = (int)(unsigned)k;
        }
        if (threadIdx.x == 0) L.cnt[(int64_t)s * L.R + r] = m;
        return;
    }
    // ---- slow form: scan the whole segment
    unsigned *seen = sc_lds;
    unsigned *fl = regS;
    const int64_t *a = idx + (int64_t)s * stride;
    for (int i = threadIdx.x; i < words; i += kScThreads) {
        seen[i] = 0u;
        multi[i] = 0u;
    }
    const int fwn = (len + 31) >> 5;
    for (int i = threadIdx.x; i < fwn; i += kScThreads) fl[i] = 0u;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    auto mine = [&](int64_t row) -> bool {
        return (unsigned)((uint64_t)row >> L.shift) == r && row < n_rows && row != padding_idx;
    };
    sc_scan(a, len, [&](int64_t row, int) {
        if (!mine(row)) return;
        const unsigned bit = (unsigned)row & bmask, m = 1u << (bit & 31u);
        const unsigned old = atomicOr(&seen[bit >> 5], m);
        if (old & m) atomicOr(&multi[bit >> 5], m);
    });
    __syncthreads();
    sc_scan(a, len, [&](int64_t row, int k) {
        if (!mine(row)) return;
        const unsigned bit = (unsigned)row & bmask;
        if (!((multi[bit >> 5] >> (bit & 31u)) & 1u)) return;
        atomicOr(&fl[k >> 5], 1u << (k & 31));
        const int pos = atomicAdd(&n_list, 1);
        if (pos < kScCap) keys[pos] = ((unsigned long long)(unsigned)row << 32) | (unsigned)k;
    });
    __syncthreads();
    for (int w = threadIdx.x; w < fwn; w += kScThreads)
        if (fl[w]) __hip_atomic_fetch_or(fb + w, fl[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int m = n_list;
    if (threadIdx.x == 0) {
        L.cnt[(int64_t)s * L.R + r] = m;
        if (m > kScCap) L.meta[1] = 1;
    }
    if (m > kScCap) return;                                          // workgroup-uniform: the apply kernel takes the slow way
    int *cnt = reinterpret_cast<int *>(multi);
    unsigned short *out16 = reinterpret_cast<unsigned short *>(multi + kScBins);
    __syncthreads();
    for (int i = threadIdx.x; i < kScBins; i += kScThreads) cnt[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += kScThreads) atomicAdd(&cnt[bin_of(keys[j])], 1);
    __syncthreads();
    sc_scan_bins(cnt, wave_tot);
    for (int j = threadIdx.x; j < m; j += kScThreads) out16[atomicAdd(&cnt[bin_of(keys[j])], 1)] = (unsigned short)j;
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += kScThreads) {
        const unsigned long long k = keys[j];
        const int bin = bin_of(k);
        const int lo = bin ? cnt[bin - 1] : 0, hi = cnt[bin];
        int pos = lo;
        for (int jj = lo; jj < hi; ++jj) pos += keys[out16[jj]] < k ? 1 : 0;
        lrow[pos] = (int)(k >> 32);
        lpos[pos] = (int)(unsigned)k;
    }
}

template <int NV>
__device__ __forceinline__ void row_add(Row<NV> &acc, const Row<NV> &x) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        acc.v[k].x += x.v[k].x; acc.v[k].y += x.v[k].y; acc.v[k].z += x.v[k].z; acc.v[k].w += x.v[k].w;
    }
}

template <int T, int NV, bool FULL>
__device__ __forceinline__ void row_apply(float *__restrict__ tab, int64_t row, int D, int lane, float alpha, const Row<NV> &acc) {
    Row<NV> g = load_row<T, NV, FULL>(tab, row, D, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k].x = fmaf(alpha, acc.v[k].x, g.v[k].x); g.v[k].y = fmaf(alpha, acc.v[k].y, g.v[k].y);
        g.v[k].z = fmaf(alpha, acc.v[k].z, g.v[k].z); g.v[k].w = fmaf(alpha, acc.v[k].w, g.v[k].w);
    }
    store_row<T, NV, FULL>(tab, row, D, lane, g);
}

// Workgroups [0, nS): one team per position (rows that occur once; shared rows of an unlisted range by brute force);
// workgroups [nS, gridDim.x): the lists, one team per entry, the head of a run sums it.
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void scatter_apply_kernel(float *__restrict__ tab, int D, int64_t n_rows, int64_t padding_idx,
                                                                const int64_t *__restrict__ idx, int n,
                                                                const float *__restrict__ src, float alpha,
                                                                const unsigned *__restrict__ flags, const int *__restrict__ cnt,
                                                                const int *__restrict__ lists, int R, unsigned shift, int nS) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T, team = threadIdx.x / T;
    Row<NV> acc;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((int)blockIdx.x < nS) {
        const int k = (int)blockIdx.x * TEAMS + team;
        if (k >= n) return;
        const int64_t row = idx[k];
        if (row < 0 || row >= n_rows || row == padding_idx) return;
        const bool shared = (flags[k >> 5] >> (k & 31)) & 1u;
        if (!shared) {
            row_add<NV>(acc, load_row<T, NV, FULL>(src, k, D, lane));
            row_apply<T, NV, FULL>(tab, row, D, lane, alpha, acc);
            return;
        }
        if (cnt[(unsigned)((uint64_t)row >> shift)] <= kScCap) return;       // listed: a list team sums it
        // unlisted range: the first position of the row sums all of them, in position order
        const unsigned tshift = (unsigned)((threadIdx.x & 63u) / T * T);
        const unsigned long long tmask = (T == 64) ? ~0ull : ((1ull << T) - 1ull);
        for (int j0 = 0; j0 < k; j0 += T) {
            const int j = j0 + lane;
            const bool hit = j < k && idx[j] == row;
            if ((__ballot(hit) >> tshift) & tmask) return;
        }
        for (int j0 = k; j0 < n; j0 += T) {
            const int j = j0 + lane;
            const bool hit = j < n && idx[j] == row;
            unsigned long long mk = (__ballot(hit) >> tshift) & tmask;
            while (mk) {
                const int l = __builtin_ctzll(mk);
                mk &= mk - 1;
                row_add<NV>(acc, load_row<T, NV, FULL>(src, j0 + l, D, lane));
            }
        }
        row_apply<T, NV, FULL>(tab, row, D, lane, alpha, acc);
        return;
    }
    __shared__ int tstart[kScMaxRanges + 1];
    if (threadIdx.x == 0) {
        int at = 0;
        for (int r = 0; r < R; ++r) {
            tstart[r] = at;
            const int c = cnt[r];
            at += c <= kScCap ? (c + TEAMS - 1) / TEAMS : 0;
        }
        tstart[R] = at;
    }
    __syncthreads();
    const int total = tstart[R], nT = (int)gridDim.x - nS;
    for (int q = (int)blockIdx.x - nS; q < total; q += nT) {
        int r = 0;
        while (tstart[r + 1] <= q) ++r;
        const int c = cnt[r], e = (q - tstart[r]) * TEAMS + team;
        if (e >= c) continue;
        const int *lr = lists + (int64_t)r * (2 * kScCap), *lp = lr + kScCap;
        const int row = lr[e];
        if (e != 0 && lr[e - 1] == row) continue;                            // not the head of its run
#pragma unroll
        for (int k = 0; k < NV; ++k) acc.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = e; j < c && lr[j] == row; j += 4) {                     // four contributions requested together
            bool ok[4];
            int sp[4];
            Row<NV> sr[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                ok[f] = j + f < c && lr[j + f] == row;
                sp[f] = ok[f] ? lp[j + f] : 0;
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (ok[f]) sr[f] = load_row<T, NV, FULL>(src, sp[f], D, lane);
            bool more = true;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                more = more && ok[f];
                if (!more) break;
                row_add<NV>(acc, sr[f]);
            }
            if (!more) break;
        }
        row_apply<T, NV, FULL>(tab, row, D, lane, alpha, acc);
    }
}

static inline int sc_teams(int D) {
    if (D >= 64) return kBlock / 16;
    if (D == 32) return kBlock / 8;
    if (D == 16) return kBlock / 4;
    if (D == 8) return kBlock / 2;
    if (D == 4) return kBlock;
    return kBlock / 16;
}

static int32_t sc_build(const int64_t *idx, int64_t nseg, int64_t stride, const int32_t *seg_len, int64_t n_total,
                        int64_t n_rows, int64_t padding_idx, int32_t *plan, const ScLayout &L, hipStream_t stream) {
    WR_HIP(hipMemsetAsync(plan, 0, (size_t)L.zero_words * 4, stream));
    const int tiles = (int)((L.stride + kScTile - 1) / kScTile);
    hipLaunchKernelGGL(scatter_bucket_kernel, dim3((unsigned)(nseg * tiles)), dim3(kBlock), 0, stream, idx, stride, seg_len, n_total,
                       tiles, n_rows, padding_idx, sc_dev(plan, L));
    WR_LAUNCH_CHECK("scatter_bucket_kernel");
    const size_t lds = (size_t)(2 * kScCap + 2 * kScWords + 20 + kScBins) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(scatter_plan_kernel, dim3((unsigned)(nseg * L.R)), dim3(kScThreads), lds, stream, idx, stride, seg_len,
                       n_total, (int)nseg, n_rows, padding_idx, sc_dev(plan, L));
    WR_LAUNCH_CHECK("scatter_plan_kernel");
    return WR_OK;
}

static int32_t sc_apply(float *tab, int64_t n_rows, int32_t D, const int64_t *idx_seg, int64_t n, int64_t padding_idx,
                        const float *src, float alpha, const int32_t *plan, const ScLayout &L, int64_t seg, hipStream_t stream) {
    const ScDev G = sc_dev(const_cast<int32_t *>(plan), L);
    const int tpb = sc_teams(D);
    const int nS = (int)((n + tpb - 1) / tpb);
    int64_t nT = (n + tpb - 1) / tpb + L.R;                 // tiles if every position were listed
    nT = nT > 2048 ? 2048 : nT;                             // (one tile per workgroup, up to 32 K of them: 24.0 against 21.6 us)
    const unsigned *fl = G.flags + seg * L.fw;
    const int *cnt = G.cnt + seg * L.R, *lists = G.lists + seg * L.R * (int64_t)(2 * kScCap);
#define WR_CALL_SA(T_, NV_, FULL_)                                                                                              \
    hipLaunchKernelGGL((scatter_apply_kernel<T_, NV_, FULL_>), dim3((unsigned)(nS + nT)), dim3(kBlock), 0, stream, tab, D, n_rows, \
                       padding_idx, idx_seg, (int)n, src, alpha, fl, cnt, lists, L.R, L.shift, nS)
    WR_DISPATCH_D(D, WR_CALL_SA);
#undef WR_CALL_SA
    WR_LAUNCH_CHECK("scatter_apply_kernel");
    return WR_OK;
}

// (Tried for the SMALL tables that wr_rows.hip serves with its two-launch LDS counting sort — 3,706 rows, 45 K positions,
// 18.1 us: ONE launch in which a 1,024-thread workgroup owns 64 rows, scans all positions' indices in passes, compacts its
// matches in position order and lets every team sum its row.  21.6 us: a workgroup that tests 45 K indices is bound by
// VALU issue on its one CU — 2 us per pass of 8 K positions, 12 us of scans before the first row is added.)
// wr_scatter_add_rows' path for tables beyond the LDS counting sort's reach (wr_rows.hip): plan + apply in `workspace`
int64_t scatter_planned_words(int64_t n, int64_t n_rows) {
    ScLayout L;
    return sc_layout(1, n, n_rows, L) ? L.total : 0;
}

int32_t scatter_add_planned_once(float *tab, int64_t n_rows, int32_t D, const int64_t *idx, const float *src, int64_t n,
                                 int64_t padding_idx, float alpha, int32_t *plan, hipStream_t stream) {
    ScLayout L;
    WR_REQUIRE(sc_layout(1, n, n_rows, L), WR_E_RANGE, "row plan not applicable to n=%lld", (long long)n);
    int32_t rc;
    if ((rc = sc_build(idx, 1, n, nullptr, n, n_rows, padding_idx, plan, L, stream)) != WR_OK) return rc;
    return sc_apply(tab, n_rows, D, idx, n, padding_idx, src, alpha, plan, L, 0, stream);
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_scatter_plan_words(int64_t n_segments, int64_t seg_stride, int64_t n_rows) {
    ScLayout L;
    if (!sc_layout(n_segments, seg_stride, n_rows, L)) return 0;     // 0: not applicable (segments of more than 2^18 positions)
    return L.total;
}

int32_t wr_scatter_plan_build(const int64_t *idx, int64_t n_segments, int64_t seg_stride, const int32_t *seg_len, int64_t n_rows,
                              int64_t padding_idx, int32_t *plan, int64_t plan_words, void *stream_) {
    WR_REQUIRE(idx && plan, WR_E_NULL, "idx / plan must not be NULL");
    ScLayout L;
    WR_REQUIRE(sc_layout(n_segments, seg_stride, n_rows, L), WR_E_RANGE, "row plan not applicable to %lld segments of %lld positions",
               (long long)n_segments, (long long)seg_stride);
    WR_REQUIRE(aligned16(plan) && plan_words >= L.total, WR_E_WORKSPACE, "row plan: %lld words < %lld", (long long)plan_words,
               (long long)L.total);
    WR_REQUIRE(n_segments * L.R < (int64_t(1) << 31), WR_E_SHAPE, "row plan: too many segments in one plan");
    return sc_build(idx, n_segments, seg_stride, seg_len, n_segments * seg_stride, n_rows, padding_idx, plan, L,
                    reinterpret_cast<hipStream_t>(stream_));
}

int32_t wr_scatter_add_planned(float *table, int64_t n_rows, int32_t D, const int64_t *idx, int64_t n_segments, int64_t seg_stride,
                               int64_t segment, int64_t n, int64_t padding_idx, const float *src, float alpha,
                               const int32_t *plan, int64_t plan_words, void *stream_) {
    int32_t rc;
    if ((rc = check_table(table, n_rows, D, "table")) != WR_OK) return rc;
    WR_REQUIRE(idx && src && plan, WR_E_NULL, "idx / src / plan must not be NULL");
    WR_REQUIRE(aligned16(src), WR_E_ALIGN, "src is not 16-byte aligned");
    ScLayout L;
    WR_REQUIRE(sc_layout(n_segments, seg_stride, n_rows, L), WR_E_RANGE, "row plan not applicable to %lld segments of %lld positions",
               (long long)n_segments, (long long)seg_stride);
    WR_REQUIRE(plan_words >= L.total, WR_E_WORKSPACE, "row plan: %lld words < %lld", (long long)plan_words, (long long)L.total);
    WR_REQUIRE(segment >= 0 && segment < n_segments && n >= 0 && n <= seg_stride, WR_E_SHAPE,
               "segment %lld of %lld with %lld of %lld positions", (long long)segment, (long long)n_segments, (long long)n,
               (long long)seg_stride);
    if (n == 0) return WR_OK;
    return sc_apply(table, n_rows, D, idx + segment * seg_stride, n, padding_idx, src, alpha, plan, L, segment,
                    reinterpret_cast<hipStream_t>(stream_));
}

}  // extern "C"
