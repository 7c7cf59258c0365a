// wr_plan_fast.hip — hand-written batch-plan builder: bucket scatter + per-bucket LDS counting sort.
//
// Produces EXACTLY the arrays of the generic builder (wr_plan.hip, radix sort over composite keys): triplets of each
// batch stably sorted by user, item occurrences sorted by (item, positive-before-negative, sorted triplet index), bit
// 31 of tp/tn flagging item rows with several occurrences.  It replaces the reference's host-side batching
// (src/helpers/BaseRunner.py:188-193, src/models/BaseModel.py:96-127) like the generic one; tests compare the two
// bit for bit.
//
// Why: the generic radix sort moves every (key,value) pair 4x through HBM at ~2 TB/s; here every pair is written
// once into a (batch, row-range) bucket and sorted inside LDS.
//   F1 user scatter : tiles of 4096 triplets; (user<<32 | original index) appended to bucket (batch, user >> shift_u);
//                     ranks inside a tile from LDS atomics, ONE global atomic per (tile, bucket) reserves the range, the
//                     tile is grouped by bucket in LDS and written out in runs of consecutive slots
//   F2 user sort    : one workgroup per bucket, composites in registers: counting sort into bins of the low row bits
//                     (LDS), sorted position = bin start + number of smaller composites in the bin; every thread
//                     writes tu/tp/tn/torig for its own composites at the bucket's prefix
//   F3 item scatter : same tiling over the batch's [positives | negatives] occurrences:
//                     (item<<32 | side<<31 | sorted index) appended to bucket (batch, item >> shift_i)
//   F4 item sort    : like F2; writes oc_item/oc_src, flags rows with several occurrences in tp/tn
// 256 buckets per batch (1024 for batches beyond 128 K), any table size.  Buckets have a fixed capacity (2x the mean + 64);
// if any bucket overflows flags[1] is set and the caller must rebuild with the generic builder — never a wrong plan.
// Equal-width row ranges overflow on popularity-skewed ids (already a Zipf(0.3) popularity over id-sorted items does); with
// a bucket map (wr_bucket_side: row ranges of equal expected load, heavy rows split by position into sub-buckets; see
// SideDev / BinMap below) the same four kernels handle power-law ids and emit the same arrays.  Ties are impossible (composites are unique), so
// the unstable bucket placement does not leak into the result.
#include "wr_common.h"

namespace wr {

#ifndef WR_PLAN_CAPX
#define WR_PLAN_CAPX 2
#endif
constexpr int kCapFactor = WR_PLAN_CAPX;   // bucket capacity = kCapFactor x the mean bucket population + 64
constexpr int kMaxBuckets = 1024;  // row-range buckets per batch: 256 up to B = 128 K, 1024 beyond
constexpr int kMaxCap = 4608;    // largest bucket handled by one workgroup (36 KiB of 8-byte composites in LDS, 18 per thread in registers): B = 1,048,576 needs 4,160

struct FastLayout {
    int nbk_u, nbk_i;         // buckets per batch on each side (equal-width mode: a power of two; mapped mode: from the map)
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

// map_u / map_i: bucket counts of a bucket map (0: equal-width row ranges on that side)
static bool fast_layout(int64_t n, int64_t B, int64_t n_users, int64_t n_items, int map_u, int map_i, FastLayout &L) {
    if (n <= 0 || n >= (int64_t(1) << 31) || B <= 0 || B > (int64_t(1) << 24)) return false;
    if (n_users <= 0 || n_users >= (int64_t(1) << 31) || n_items <= 0 || n_items >= (int64_t(1) << 31)) return false;
    if (map_u < 0 || map_u > kMaxBuckets || map_i < 0 || map_i > kMaxBuckets) return false;
    L.nb = (n + B - 1) / B;
    L.user_bits = bits_for_rows(n_users);
    L.item_bits = bits_for_rows(n_items);
#ifndef WR_PLAN_BBITS
#define WR_PLAN_BBITS 8
#endif
    const unsigned bbits = B > 131072 ? 10 : WR_PLAN_BBITS;
    const int nominal = 1 << bbits;   // the capacity follows the nominal bucket count: a map balances the load to that mean
    if ((map_u || map_i) && bbits != WR_PLAN_BBITS) return false;   // maps are built for the 256-bucket regime
    L.nbk_u = map_u ? map_u : nominal;
    L.nbk_i = map_i ? map_i : nominal;
    L.shift_u = L.user_bits > bbits ? L.user_bits - bbits : 0;
    L.shift_i = L.item_bits > bbits ? L.item_bits - bbits : 0;
    const int64_t cu = kCapFactor * ((B + nominal - 1) / nominal) + 64;
    const int64_t ci = kCapFactor * ((2 * B + nominal - 1) / nominal) + 64;
    if (cu > kMaxCap || ci > kMaxCap) return false;
    L.cap_u = (int)cu;
    L.cap_i = (int)ci;
    L.cnt_bytes = align_up(L.nb * (int64_t)(L.nbk_u + L.nbk_i) * 4, 256);
    L.ubuf_bytes = align_up(L.nb * L.nbk_u * (int64_t)L.cap_u * 8, 256);
    L.ibuf_bytes = align_up(L.nb * L.nbk_i * (int64_t)L.cap_i * 8, 256);
    L.total = L.cnt_bytes + L.ubuf_bytes + L.ibuf_bytes;
    return true;
}

// One side (users or items) of the bucket assignment as the kernels see it.  Equal-width mode (row_bucket == nullptr):
// bucket = row >> shift.  Mapped mode (wr_bucket_side, include/whisprrec_hip.h): bucket = row_bucket[row] & 0xffff; a heavy
// row — one that would overflow a bucket on its own — owns `row_bucket[row] >> 16` consecutive sub-buckets, which split
// its occurrences by position in the batch (the order inside a row IS the position order, so the ranges keep it).
struct SideDev {
    int nbk;
    unsigned shift;
    const int *row_bucket, *start, *rows, *sub;
    __device__ __forceinline__ int bucket_of(uint32_t row, uint32_t pos, uint32_t npos) const {
        if (row_bucket == nullptr) return (int)(row >> shift);
        const uint32_t e = (uint32_t)row_bucket[row];
        const uint32_t nsub = e >> 16;
        return (int)(e & 0xffffu) + (nsub ? (int)(((uint64_t)pos * nsub) / npos) : 0);
    }
};

#ifndef WR_PLAN_TILE
#define WR_PLAN_TILE 4096
#endif
constexpr int kTile = WR_PLAN_TILE;       // elements of one batch handled by one scatter workgroup
#ifndef WR_PLAN_BINSHIFT
#define WR_PLAN_BINSHIFT 2
#endif
constexpr unsigned kBinShift = WR_PLAN_BINSHIFT;   // counting-sort bins per bucket = capacity >> kBinShift
constexpr int kMaxGroup = 256;    // longest bin ordered by ranking (m reads per composite of a bin of m; else: overflow)

// Appends a tile's composites to their buckets.  Ranks inside the tile come from LDS atomics and ONE global atomic per
// non-empty (tile, bucket) reserves the range; the composites are then grouped by bucket in LDS and written out slot by
// slot, so that consecutive lanes fill consecutive slots of the same bucket (128-B runs at 16 composites per (tile,
// bucket)).  Writing each composite straight from the thread that loaded it — 64 lanes, 64 buckets, 64 separate 8-byte
// stores per wave instruction — took 18 of the kernel's 27 us (timing-only variants: no stores 9 us, no global atomics
// 27 us).  `shift` recovers the bucket from a composite (row id in the high word).  Placement order inside a bucket is
// arbitrary; the bucket sort fixes it.
template <int PER_THREAD, bool MAPPED, typename KeyFn>
__device__ __forceinline__ void tile_scatter(int n_local, int nbk, unsigned shift, int *__restrict__ cnt_global,
                                             unsigned long long *__restrict__ buf, int cap, int *__restrict__ flags,
                                             KeyFn key_of) {
    __shared__ int hist[kMaxBuckets];    // per bucket: count, then offset of the bucket's group inside the staged tile
    __shared__ int delta[kMaxBuckets];   // per bucket: (reserved slot base in the bucket) - (offset inside the tile)
    __shared__ int wave_tot[kBlock / 64];
    __shared__ unsigned long long stage[kTile];
    __shared__ unsigned short stage_bk[MAPPED ? kTile : 1];   // mapped mode: the bucket is not a function of the row bits
    for (int j = threadIdx.x; j < nbk; j += kBlock) hist[j] = 0;
    __syncthreads();
    unsigned long long key[PER_THREAD];
    int bucket[PER_THREAD], rank[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int e = threadIdx.x + k * kBlock;
        bucket[k] = -1;
        if (e < n_local) {
            key[k] = key_of(e, bucket[k]);
            rank[k] = atomicAdd(&hist[bucket[k]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counts (thread t owns buckets [t*per, (t+1)*per)) + the global reservations
    const int per = (nbk + kBlock - 1) / kBlock;
    const int j0 = threadIdx.x * per;
    int local = 0;
    for (int j = 0; j < per; ++j)
        if (j0 + j < nbk) local += hist[j0 + j];
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
        if (j0 + j < nbk) {
            const int c = hist[j0 + j];
            const int reserved = c ? atomicAdd(&cnt_global[j0 + j], c) : 0;
            hist[j0 + j] = run;
            delta[j0 + j] = reserved - run;
            run += c;
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k)
        if (bucket[k] >= 0) {
            stage[hist[bucket[k]] + rank[k]] = key[k];
            if (MAPPED) stage_bk[hist[bucket[k]] + rank[k]] = (unsigned short)bucket[k];
        }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < n_local; sidx += kBlock) {
        const unsigned long long kv = stage[sidx];
        const int bk = MAPPED ? (int)stage_bk[sidx] : (int)((unsigned)(kv >> 32) >> shift);
        const int slot = delta[bk] + sidx;
        // slot < 0 cannot happen with the reservations above; the test keeps every timing-only variant of this kernel
        // (e.g. one built without the global atomics, whose "reservations" are garbage) inside the bucket arrays
        if (slot >= 0 && slot < cap) buf[(int64_t)bk * cap + slot] = kv;
        else flags[1] = 1;
    }
}

template <typename Idx, bool MAPPED>
__global__ __launch_bounds__(kBlock) void fast_user_scatter(const Idx *__restrict__ u, int64_t n, int64_t B, int tiles_per_batch,
                                                             SideDev side, int64_t n_users, int cap_u, int *__restrict__ cnt_u,
                                                             unsigned long long *__restrict__ ubuf, int *__restrict__ flags) {
    const int64_t b = blockIdx.x / tiles_per_batch;
    const int tile = blockIdx.x % tiles_per_batch;
    const int64_t lo = b * B + (int64_t)tile * kTile;
    const int64_t batch_end = (b * B + B < n) ? (b * B + B) : n;
    const int n_local = (int)((lo + kTile <= batch_end) ? kTile : (batch_end > lo ? batch_end - lo : 0));
    const int nbk = side.nbk;
    tile_scatter<kTile / kBlock, MAPPED>(n_local, nbk, side.shift, cnt_u + b * nbk, ubuf + b * nbk * (int64_t)cap_u, cap_u, flags,
                                         [&](int e, int &bucket) {
                                             const int64_t i = lo + e;
                                             int64_t uu = (int64_t)u[i];
                                             if (uu < 0 || uu >= n_users) {
                                                 flags[0] = 1;
                                                 uu = 0;
                                             }
                                             bucket = side.bucket_of((uint32_t)uu, (uint32_t)(i - b * B), (uint32_t)B);
                                             return ((unsigned long long)uu << 32) | (unsigned long long)(uint32_t)i;
                                         });
}

// Item occurrences of a tile of user-sorted triplets: composite (item, side, sorted index) — positives before negatives
// for equal items, then by sorted triplet index: the order a stable sort of [positives | negatives] gives.
template <bool MAPPED>
__global__ __launch_bounds__(kBlock) void fast_item_scatter(const int *__restrict__ tp, const int *__restrict__ tn, int64_t n,
                                                             int64_t B, int tiles_per_batch, SideDev side, int64_t n_items,
                                                             int cap_i, int *__restrict__ cnt_i,
                                                             unsigned long long *__restrict__ ibuf, int *__restrict__ flags) {
    const int64_t b = blockIdx.x / tiles_per_batch;
    const int tile = blockIdx.x % tiles_per_batch;
    const int64_t Bb = ((b * B + B < n) ? B : (n - b * B));
    const int64_t lo2 = (int64_t)tile * kTile;            // offset into the batch's 2*Bb occurrences: [pos | neg]
    const int n_local = (int)((lo2 + kTile <= 2 * Bb) ? kTile : (2 * Bb > lo2 ? 2 * Bb - lo2 : 0));
    const int nbk = side.nbk;
    tile_scatter<kTile / kBlock, MAPPED>(n_local, nbk, side.shift, cnt_i + b * nbk, ibuf + b * nbk * (int64_t)cap_i, cap_i, flags,
                                         [&](int e, int &bucket) {
                                             const int64_t o = lo2 + e;
                                             const int sd = o >= Bb;
                                             const int tloc = (int)(sd ? o - Bb : o);
                                             int item = (sd ? tn : tp)[b * B + tloc];
                                             // positions the user stage dropped after a bucket overflow were never written:
                                             // keep every access in range (the plan is already flagged invalid)
                                             if (item < 0 || item >= n_items) {
                                                 flags[1] = 1;
                                                 item = 0;
                                             }
                                             // position key of an occurrence: positives [0, B), negatives [B, 2B)
                                             bucket = side.bucket_of((uint32_t)item, (uint32_t)(sd * B + tloc), (uint32_t)(2 * B));
                                             return ((unsigned long long)(uint32_t)item << 32) | ((unsigned long long)sd << 31) |
                                                    (unsigned long long)(uint32_t)tloc;
                                         });
}

#ifndef WR_SORT_BLOCK
#define WR_SORT_BLOCK 256
#endif
constexpr int kSortBlock = WR_SORT_BLOCK;   // threads of a bucket-sort workgroup (one bucket each)

// One bucket per workgroup; every thread keeps its (up to PER) composites in registers from load to output.
//   1. the capped counts of the buckets before this one are summed (the bucket's prefix inside its batch) while the bin
//      counters are cleared; the composites are loaded and histogrammed on the top `bin_bits` of the `low_bits` row bits
//      that vary inside a bucket (about two composites per bin: fewer counters to clear and scan than composites);
//      `early(k, composite)` runs right behind the load — the user sort issues its p[] / n[] gathers there, so that they
//      fly during the LDS phases instead of forming a fourth memory round trip at the end (22 of 66 us);
//   2. exclusive scan of the bin counters, placement of the composites into `out` grouped by bin (LDS atomics);
//   3. rank_in_bin(): every composite counts the smaller composites of its bin (unique composites: the counts are the
//      ranks) — its sorted position — and its thread writes the outputs for that position straight from registers.
//      One thread per COMPOSITE with independent LDS reads: one thread per BIN running an insertion sort (serial,
//      dependent LDS traffic, the rest of the wave idle behind the longest bin) took 21 of the item sort's 70 us.
// Bins longer than kMaxGroup raise the overflow flag (degenerate batch: the caller rebuilds with the generic builder).
// A workgroup's lifetime is a chain of barriers and memory round trips, and 16 K workgroups per plan chunk wait on it.
struct BinMap {
    int kind;   // 0: equal-width row range; 1: row range of a bucket map; 2: sub-bucket of a heavy row (bins over positions)
    unsigned down, mask;                    // kind 0: bin = (row >> down) & mask
    uint32_t start, nbin1;                  // kind 1: first row of the bucket; nbin - 1
    unsigned long long scale;               // kind 1 / 2: ceil(nbin * 2^32 / extent): bin = (offset * scale) >> 32 (monotone)
    uint32_t nsub, sidx, npos, pos_base, half;   // kind 2: position key = low word - pos_base (users), side * half + tloc (items)
    __device__ __forceinline__ int of(unsigned long long kv) const {
        const uint32_t row = (uint32_t)(kv >> 32);
        if (kind == 0) return (int)((row >> down) & mask);
        unsigned long long off;
        if (kind == 1) {
            off = row - start;
        } else {
            const uint32_t low = (uint32_t)kv;
            const uint32_t pos = half ? (low >> 31) * half + (low & 0x7fffffffu) : low - pos_base;
            off = (unsigned long long)pos * nsub - (unsigned long long)sidx * npos;   // in [0, npos): this sub-bucket's share
        }
        const uint32_t bin = (uint32_t)((off * scale) >> 32);
        return (int)(bin < nbin1 ? bin : nbin1);
    }
};

// bin mapping of bucket `bucket` of batch `b`; item_side: composites carry (side, sorted index) in the low word
__device__ __forceinline__ BinMap make_binmap(const SideDev &side, int bucket, unsigned bin_bits, int64_t b, int64_t B,
                                              bool item_side) {
    BinMap bm{};
    const uint32_t nbin = 1u << bin_bits;
    bm.nbin1 = nbin - 1u;
    if (side.row_bucket == nullptr) {
        bm.kind = 0;
        bm.down = side.shift - bin_bits;
        bm.mask = nbin - 1u;
        return bm;
    }
    const uint32_t sub = (uint32_t)side.sub[bucket];
    if (sub == 0) {
        bm.kind = 1;
        bm.start = (uint32_t)side.start[bucket];
        const unsigned long long width = (unsigned long long)max(side.rows[bucket], 1);
        bm.scale = (((unsigned long long)nbin << 32) + width - 1) / width;
        return bm;
    }
    bm.kind = 2;
    bm.nsub = sub >> 16;
    bm.sidx = sub & 0xffffu;
    bm.npos = (uint32_t)(item_side ? 2 * B : B);
    bm.half = item_side ? (uint32_t)B : 0u;
    bm.pos_base = (uint32_t)(b * B);
    bm.scale = (((unsigned long long)nbin << 32) + bm.npos - 1) / bm.npos;
    return bm;
}

template <int PER, typename Early>
__device__ __forceinline__ int bucket_bins(const unsigned long long *__restrict__ src, const int *__restrict__ cnt_batch,
                                           int bucket, int cap, unsigned long long *__restrict__ out, int *__restrict__ cnt,
                                           int *__restrict__ wave_tot, int *__restrict__ wave_pre, int count, int nbin,
                                           const BinMap bm, unsigned long long (&kv)[PER], Early early) {
    int a = 0;
    for (int j = threadIdx.x; j < bucket; j += kSortBlock) a += min(cnt_batch[j], cap);
    for (int j = threadIdx.x; j < nbin; j += kSortBlock) cnt[j] = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);   // integer sum: order irrelevant
    if ((threadIdx.x & 63) == 0) wave_pre[threadIdx.x >> 6] = a;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int j = threadIdx.x + k * kSortBlock;
        kv[k] = (j < count) ? src[j] : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if ((int)threadIdx.x + k * kSortBlock < count) {
            early(k, kv[k]);
            atomicAdd(&cnt[bm.of(kv[k])], 1);
        }
    }
    __syncthreads();
    int prefix = 0;
#pragma unroll
    for (int w = 0; w < kSortBlock / 64; ++w) prefix += wave_pre[w];
    // exclusive scan of cnt[0..nbin): thread t owns counters [t*per, (t+1)*per)
    const int per = (nbin + kSortBlock - 1) / kSortBlock;
    const int c0 = threadIdx.x * per;
    int local = 0;
    for (int j = 0; j < per; ++j)
        if (c0 + j < nbin) local += cnt[c0 + j];
    int incl = local;  // inclusive scan over the 64 lanes of the wave
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
        if (c0 + j < nbin) {
            const int c = cnt[c0 + j];
            cnt[c0 + j] = run;
            run += c;
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if ((int)threadIdx.x + k * kSortBlock < count) out[atomicAdd(&cnt[bm.of(kv[k])], 1)] = kv[k];
    __syncthreads();
    return prefix;
}

// After placement cnt[b] is the END of bin b; its start is the end of bin b-1.  Returns the sorted position of `kv` inside
// the bucket (-1: its bin is longer than kMaxGroup) and the number of composites of the bin with the same row id.
__device__ __forceinline__ int rank_in_bin(const unsigned long long *__restrict__ out, const int *__restrict__ cnt,
                                           const BinMap bm, unsigned long long kv, int &same_row) {
    const int bin = bm.of(kv);
    const int lo = bin ? cnt[bin - 1] : 0;
    const int hi = cnt[bin];
    same_row = 1;
    if (hi - lo == 1) return lo;
    if (hi - lo > kMaxGroup) return -1;
    int pos = lo, same = 0;
    for (int j = lo; j < hi; ++j) {
        const unsigned long long o = out[j];
        pos += (o < kv) ? 1 : 0;
        same += ((unsigned)(o >> 32) == (unsigned)(kv >> 32)) ? 1 : 0;
    }
    same_row = same;
    return pos;
}

template <typename Idx, int PER>
__global__ __launch_bounds__(kSortBlock) void fast_user_sort(const Idx *__restrict__ p, const Idx *__restrict__ nn, int64_t n,
                                                          int64_t B, SideDev side, int64_t n_items, int cap_u, unsigned bin_bits,
                                                          const int *__restrict__ cnt_u, const unsigned long long *__restrict__ ubuf,
                                                          int *__restrict__ tu, int *__restrict__ tp, int *__restrict__ tn,
                                                          int *__restrict__ torig, int *__restrict__ flags) {
    extern __shared__ unsigned long long lds[];  // out[cap] | cnt[1 << bin_bits] (ints)
    __shared__ int wave_tot[kSortBlock / 64], wave_pre[kSortBlock / 64];
    unsigned long long *out = lds;
    int *cnt = reinterpret_cast<int *>(lds + cap_u);
    // the buckets of one batch gather p[] / n[] from the same 2 x 4B x B bytes: keep them on one XCD's L2
    const int nbk = side.nbk;
    const unsigned lb = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int64_t b = lb / nbk;
    const int bucket = lb % nbk;
    const int count = min(cnt_u[lb], cap_u);
    if (count == 0) return;
    const BinMap bm = make_binmap(side, bucket, bin_bits, b, B, false);
    unsigned long long kv[PER];
    Idx pv[PER], nv[PER];
    const int prefix = bucket_bins<PER>(ubuf + (int64_t)lb * cap_u, cnt_u + b * nbk, bucket, cap_u, out, cnt, wave_tot, wave_pre,
                                        count, 1 << bin_bits, bm, kv, [&](int k, unsigned long long c) {
                                            pv[k] = p[(uint32_t)c];
                                            nv[k] = nn[(uint32_t)c];
                                        });
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if ((int)threadIdx.x + k * kSortBlock >= count) continue;
        int same;
        const int r = rank_in_bin(out, cnt, bm, kv[k], same);
        if (r < 0) {
            flags[1] = 1;
            continue;
        }
        const int64_t t = b * B + prefix + r;  // position in the user-sorted batch
        int64_t pi = (int64_t)pv[k], ni = (int64_t)nv[k];
        if (pi < 0 || pi >= n_items || ni < 0 || ni >= n_items) {
            flags[0] = 1;
            pi = (pi < 0 || pi >= n_items) ? 0 : pi;
            ni = (ni < 0 || ni >= n_items) ? 0 : ni;
        }
        tu[t] = (int)(kv[k] >> 32);
        tp[t] = (int)pi;
        tn[t] = (int)ni;
        if (torig) torig[t] = (int)(uint32_t)kv[k];
    }
}

// MARKS: the workgroup also writes its bucket's words of the batch's bitmap "item row has several occurrences in the batch"
// (what wr_bprmf_plan_overlap_marks otherwise builds with one global atomicOr per such occurrence: 41 us per 64-batch plan
// at the headline shape).  An equal-width bucket is the row range [bucket << shift, (bucket + 1) << shift): with shift >= 5
// its bitmap words belong to this workgroup alone — bits are collected in LDS and the non-zero words stored plainly (the
// bitmap was zeroed before the launch).
constexpr int kMarkWordsMax = 2048;   // words of a bucket's row range: shift <= 16
template <int PER, bool MARKS>
__global__ __launch_bounds__(kSortBlock) void fast_item_sort(int64_t n, int64_t B, SideDev side, int cap_i, unsigned bin_bits,
                                                          const int *__restrict__ cnt_i, const unsigned long long *__restrict__ ibuf,
                                                          int *__restrict__ oc_item, int *__restrict__ oc_src, int *__restrict__ tp,
                                                          int *__restrict__ tn, int *__restrict__ flags,
                                                          unsigned *__restrict__ bitmap, int64_t bitmap_words) {
    extern __shared__ unsigned long long lds[];  // out[cap] | cnt[1 << bin_bits] (ints)
    __shared__ int wave_tot[kSortBlock / 64], wave_pre[kSortBlock / 64];
    __shared__ unsigned mark_words[MARKS ? kMarkWordsMax : 1];
    unsigned long long *out = lds;
    int *cnt = reinterpret_cast<int *>(lds + cap_i);
    // the buckets of one batch set flag bits all over the batch's tp[] / tn[]: keep them on one XCD's L2
    const int nbk = side.nbk;
    const unsigned lb = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int64_t b = lb / nbk;
    const int bucket = lb % nbk;
    const int count = min(cnt_i[lb], cap_i);
    if (count == 0) return;
    const unsigned side_shift = side.shift;
    const int n_mark_words = MARKS ? (1 << (side_shift - 5)) : 0;
    if constexpr (MARKS)
        for (int i = threadIdx.x; i < n_mark_words; i += kSortBlock) mark_words[i] = 0u;   // bucket_bins below has the barrier
    const BinMap bm = make_binmap(side, bucket, bin_bits, b, B, true);
    // a heavy row's occurrences are spread over its sub-buckets: "several occurrences" is decided over all of them
    int row_total = 0;
    if (bm.kind == 2)
        for (uint32_t q = 0; q < bm.nsub; ++q) row_total += min(cnt_i[b * nbk + bucket - (int)bm.sidx + (int)q], cap_i);
    unsigned long long kv[PER];
    const int prefix = bucket_bins<PER>(ibuf + (int64_t)lb * cap_i, cnt_i + b * nbk, bucket, cap_i, out, cnt, wave_tot, wave_pre,
                                        count, 1 << bin_bits, bm, kv, [](int, unsigned long long) {});
    const int64_t base = 2 * b * B + prefix;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if ((int)threadIdx.x + k * kSortBlock >= count) continue;
        int same;
        const int r = rank_in_bin(out, cnt, bm, kv[k], same);
        if (r < 0) {
            flags[1] = 1;
            continue;
        }
        const int item = (int)(kv[k] >> 32);
        const uint32_t lo = (uint32_t)kv[k];
        const int side = (int)(lo >> 31);
        const int tloc = (int)(lo & 0x7fffffffu);
        oc_item[base + r] = item;
        oc_src[base + r] = (tloc << 1) | side;
        // equal items always share a bucket AND a bin: another composite of the bin with this item = "several occurrences"
        if (same > 1 || row_total > 1) {
            int *dst = side ? tn : tp;
            dst[b * B + tloc] |= (int)0x80000000;  // one writer per (triplet, side)
            if constexpr (MARKS) {
                const unsigned in_bucket = (unsigned)item & ((1u << side_shift) - 1u);
                atomicOr(&mark_words[in_bucket >> 5], 1u << (in_bucket & 31u));
            }
        }
    }
    if constexpr (MARKS) {
        __syncthreads();
        const int64_t w0 = ((int64_t)bucket << side_shift) >> 5;
        for (int i = threadIdx.x; i < n_mark_words; i += kSortBlock) {
            const unsigned w = mark_words[i];
            if (w != 0u && w0 + i < bitmap_words) bitmap[b * bitmap_words + w0 + i] = w;
        }
    }
}

static int32_t check_side(const wr_bucket_side *m, const char *what) {
    if (m == nullptr) return WR_OK;
    WR_REQUIRE(m->n_buckets >= 1 && m->n_buckets <= kMaxBuckets, WR_E_RANGE, "bucket map (%s): %d buckets (1..%d)", what,
               (int)m->n_buckets, kMaxBuckets);
    WR_REQUIRE(m->row_bucket && m->bucket_start && m->bucket_rows && m->bucket_sub, WR_E_NULL, "bucket map (%s): NULL array", what);
    return WR_OK;
}

static inline SideDev side_dev(const wr_bucket_side *m, int nbk, unsigned shift) {
    if (m == nullptr) return SideDev{nbk, shift, nullptr, nullptr, nullptr, nullptr};
    return SideDev{nbk, 0u, m->row_bucket, m->bucket_start, m->bucket_rows, m->bucket_sub};
}

// in-build marks (fast_item_sort<.., true>): equal-width item buckets whose row range is a whole number of bitmap words
static inline bool fast_marks_ok(const FastLayout &L, bool mapped_items) {
    return !mapped_items && L.shift_i >= 5 && L.shift_i <= 16;
}

template <typename Idx>
static int32_t plan_build_fast(const Idx *u, const Idx *p, const Idx *nn, int64_t n, int64_t B, int64_t n_users,
                               int64_t n_items, const wr_bucket_side *map_u, const wr_bucket_side *map_i, int32_t *tu,
                               int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                               void *workspace, int64_t workspace_bytes, void *stream_, int32_t *bitmap = nullptr) {
    WR_REQUIRE(u && p && nn, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(tu && tp && tn && oc_item && oc_src && flags, WR_E_NULL, "plan output arrays / flags must not be NULL");
    int32_t rc;
    if ((rc = check_side(map_u, "users")) != WR_OK) return rc;
    if ((rc = check_side(map_i, "items")) != WR_OK) return rc;
    FastLayout L;
    WR_REQUIRE(fast_layout(n, B, n_users, n_items, map_u ? map_u->n_buckets : 0, map_i ? map_i->n_buckets : 0, L), WR_E_RANGE,
               "fast plan builder not applicable to n=%lld, batch=%lld (use the generic builder)", (long long)n, (long long)B);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= L.total, WR_E_WORKSPACE,
               "fast plan workspace %lld B < %lld B", (long long)workspace_bytes, (long long)L.total);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char *ws = reinterpret_cast<char *>(workspace);
    int *cnt_u = reinterpret_cast<int *>(ws);
    int *cnt_i = cnt_u + L.nb * L.nbk_u;
    unsigned long long *ubuf = reinterpret_cast<unsigned long long *>(ws + L.cnt_bytes);
    unsigned long long *ibuf = reinterpret_cast<unsigned long long *>(ws + L.cnt_bytes + L.ubuf_bytes);
    WR_HIP(hipMemsetAsync(cnt_u, 0, (size_t)(L.nb * (int64_t)(L.nbk_u + L.nbk_i) * 4), stream));
    const int tiles_u = (int)((B + kTile - 1) / kTile), tiles_i = (int)((2 * B + kTile - 1) / kTile);
    const unsigned gb_u = (unsigned)(L.nb * L.nbk_u), gb_i = (unsigned)(L.nb * L.nbk_i);
    const SideDev su = side_dev(map_u, L.nbk_u, L.shift_u), si = side_dev(map_i, L.nbk_i, L.shift_i);
    if (map_u)
        hipLaunchKernelGGL((fast_user_scatter<Idx, true>), dim3((unsigned)(L.nb * tiles_u)), dim3(kBlock), 0, stream, u, n, B,
                           tiles_u, su, n_users, L.cap_u, cnt_u, ubuf, flags);
    else
        hipLaunchKernelGGL((fast_user_scatter<Idx, false>), dim3((unsigned)(L.nb * tiles_u)), dim3(kBlock), 0, stream, u, n, B,
                           tiles_u, su, n_users, L.cap_u, cnt_u, ubuf, flags);
    WR_LAUNCH_CHECK("fast_user_scatter");
    // bins per bucket: capacity / 4 (capacity = twice the mean bucket population -> about two composites per bin;
    // A/B at the headline shape: 1 per bin 7.0 us of plan per batch, 2 per bin 6.35, 4 per bin 6.6); with equal-width
    // buckets never more bins than distinct low-bit patterns
    auto bins_for = [](int cap, unsigned shift) {
        unsigned b = 0;
        while ((1 << b) < cap) ++b;
        b = b > kBinShift + 4 ? b - kBinShift : b;
        return b < shift ? b : shift;
    };
    const unsigned bb_u = bins_for(L.cap_u, map_u ? 31u : L.shift_u), bb_i = bins_for(L.cap_i, map_i ? 31u : L.shift_i);
    const size_t lds_u = (size_t)L.cap_u * 8 + ((size_t)4 << bb_u);
    const size_t lds_i = (size_t)L.cap_i * 8 + ((size_t)4 << bb_i);
    // composites per thread of a bucket workgroup (registers): instantiations for the capacities that occur
    const int per_u = (L.cap_u + kSortBlock - 1) / kSortBlock, per_i = (L.cap_i + kSortBlock - 1) / kSortBlock;
#define WR_USER_SORT(PER_)                                                                                                 \
    hipLaunchKernelGGL((fast_user_sort<Idx, PER_>), dim3(gb_u), dim3(kSortBlock), lds_u, stream, p, nn, n, B, su, n_items,   \
                       L.cap_u, bb_u, cnt_u, ubuf, tu, tp, tn, torig, flags)
    if (per_u <= 3) WR_USER_SORT(3);
    else if (per_u <= 5) WR_USER_SORT(5);
    else if (per_u <= 9) WR_USER_SORT(9);
    else WR_USER_SORT(kMaxCap / kSortBlock);
#undef WR_USER_SORT
    WR_LAUNCH_CHECK("fast_user_sort");
    if (map_i)
        hipLaunchKernelGGL((fast_item_scatter<true>), dim3((unsigned)(L.nb * tiles_i)), dim3(kBlock), 0, stream, tp, tn, n, B,
                           tiles_i, si, n_items, L.cap_i, cnt_i, ibuf, flags);
    else
        hipLaunchKernelGGL((fast_item_scatter<false>), dim3((unsigned)(L.nb * tiles_i)), dim3(kBlock), 0, stream, tp, tn, n, B,
                           tiles_i, si, n_items, L.cap_i, cnt_i, ibuf, flags);
    WR_LAUNCH_CHECK("fast_item_scatter");
    const int64_t bitmap_words = (n_items + 31) / 32;
    if (bitmap != nullptr) {
        WR_REQUIRE(fast_marks_ok(L, map_i != nullptr), WR_E_RANGE, "fast plan builder: no in-build marks for this shape");
        WR_HIP(hipMemsetAsync(bitmap, 0, (size_t)(L.nb * bitmap_words * 4), stream));
    }
#define WR_ITEM_SORT(PER_)                                                                                                 \
    do {                                                                                                                   \
        if (bitmap != nullptr)                                                                                             \
            hipLaunchKernelGGL((fast_item_sort<PER_, true>), dim3(gb_i), dim3(kSortBlock), lds_i, stream, n, B, si, L.cap_i, \
                               bb_i, cnt_i, ibuf, oc_item, oc_src, tp, tn, flags, reinterpret_cast<unsigned *>(bitmap),     \
                               bitmap_words);                                                                              \
        else                                                                                                               \
            hipLaunchKernelGGL((fast_item_sort<PER_, false>), dim3(gb_i), dim3(kSortBlock), lds_i, stream, n, B, si,         \
                               L.cap_i, bb_i, cnt_i, ibuf, oc_item, oc_src, tp, tn, flags, (unsigned *)nullptr,             \
                               (int64_t)0);                                                                                \
    } while (0)
    if (per_i <= 3) WR_ITEM_SORT(3);
    else if (per_i <= 5) WR_ITEM_SORT(5);
    else if (per_i <= 9) WR_ITEM_SORT(9);
    else WR_ITEM_SORT(kMaxCap / kSortBlock);
#undef WR_ITEM_SORT
    WR_LAUNCH_CHECK("fast_item_sort");
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_bprmf_plan_fast_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items) {
    FastLayout L;
    if (!fast_layout(n_triplets, batch_size, n_users, n_items, 0, 0, L)) return 0;  // 0: not applicable, use the generic builder
    return L.total;
}

int64_t wr_bprmf_plan_fast_mapped_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items,
                                                  int32_t n_buckets_users, int32_t n_buckets_items) {
    FastLayout L;
    if (!fast_layout(n_triplets, batch_size, n_users, n_items, n_buckets_users, n_buckets_items, L)) return 0;
    return L.total;
}

int32_t wr_bprmf_plan_build_fast_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, nullptr, nullptr, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream);
}

int32_t wr_bprmf_plan_build_fast_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, nullptr, nullptr, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream);
}

int32_t wr_bprmf_plan_fast_marks_supported(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items) {
    FastLayout L;
    return (fast_layout(n_triplets, batch_size, n_users, n_items, 0, 0, L) && fast_marks_ok(L, false)) ? 1 : 0;
}

int32_t wr_bprmf_plan_build_fast_marks_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                           int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                           int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                           void *workspace, int64_t workspace_bytes, int32_t *bitmap, void *stream) {
    WR_REQUIRE(bitmap != nullptr, WR_E_NULL, "bitmap must not be NULL");
    return plan_build_fast<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, nullptr, nullptr, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream, bitmap);
}

int32_t wr_bprmf_plan_build_fast_marks_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                           int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                           int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                           void *workspace, int64_t workspace_bytes, int32_t *bitmap, void *stream) {
    WR_REQUIRE(bitmap != nullptr, WR_E_NULL, "bitmap must not be NULL");
    return plan_build_fast<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, nullptr, nullptr, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream, bitmap);
}

int32_t wr_bprmf_plan_build_fast_mapped_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                            int64_t batch_size, int64_t n_users, int64_t n_items,
                                            const wr_bucket_side *map_users, const wr_bucket_side *map_items, int32_t *tu,
                                            int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src,
                                            int32_t *flags, void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int64_t>(u, p, n, n_triplets, batch_size, n_users, n_items, map_users, map_items, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream);
}

int32_t wr_bprmf_plan_build_fast_mapped_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                            int64_t batch_size, int64_t n_users, int64_t n_items,
                                            const wr_bucket_side *map_users, const wr_bucket_side *map_items, int32_t *tu,
                                            int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src,
                                            int32_t *flags, void *workspace, int64_t workspace_bytes, void *stream) {
    return plan_build_fast<int32_t>(u, p, n, n_triplets, batch_size, n_users, n_items, map_users, map_items, tu, tp, tn, torig,
                                    oc_item, oc_src, flags, workspace, workspace_bytes, stream);
}

}  // extern "C"
