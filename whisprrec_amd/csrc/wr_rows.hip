// wr_rows.hip — row gather / deterministic scatter-add, whole-table optimizers, CSR SpMM, small reductions.
//
// Reference call sites restated (paths relative to the reference root):
//   nn.Embedding forward / embedding_dense_backward with padding_idx   src/models/sequential/SASRec.py:60,84,105-106
//   torch.optim.SGD / Adam .step() over full tables                      src/helpers/BaseRunner.py:120-124,199
//   torch.sparse.mm(norm_adj, E) + layer mean                            src/models/general/LightGCN.py:139,142-143
//   EmbLoss (un-squared Frobenius norms)                                  src/utils/loss.py:94-98
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "wr_common.h"

namespace wr {

// ------------------------------------------------------------------------------------------------ gather
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float *__restrict__ tab, int D, int64_t n_rows,
                                                              const int64_t *__restrict__ idx, int64_t n,
                                                              float *__restrict__ out) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int64_t k = (int64_t)blockIdx.x * TEAMS + threadIdx.x / T;
    if (k >= n) return;
    int64_t r = idx[k];
    if (r < 0 || r >= n_rows) r = 0;  // callers validate; never read out of bounds
    const Row<NV> v = load_row<T, NV, FULL>(tab, r, D, lane);
    store_row<T, NV, FULL>(out, k, D, lane, v);
}

// ------------------------------------------------------------------------------------------------ scatter-add
__global__ __launch_bounds__(kBlock) void scatter_keys_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t n_rows,
                                                               int64_t padding_idx, uint32_t *__restrict__ keys,
                                                               uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t r = idx[i];
    // padding (and out-of-range) entries sort to the end under key n_rows and are skipped
    keys[i] = (r == padding_idx || r < 0 || r >= n_rows) ? (uint32_t)n_rows : (uint32_t)r;
    vals[i] = (uint32_t)i;
}

template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void scatter_add_sorted_kernel(float *__restrict__ grad, int D, uint32_t n_rows,
                                                                     const uint32_t *__restrict__ keys,
                                                                     const uint32_t *__restrict__ perm, int64_t n,
                                                                     const float *__restrict__ src, float alpha) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int64_t q0 = (int64_t)blockIdx.x * TEAMS + threadIdx.x / T;
    if (q0 >= n) return;
    const uint32_t r = keys[q0];
    if (r >= n_rows) return;
    if (q0 != 0 && keys[q0 - 1] == r) return;  // not the head of its run
    Row<NV> acc;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // four contributions per trip: their positions, then their rows, are requested together and added in run order
    // (one contribution per trip made a run of m rows a chain of 2 m dependent round trips)
    for (int64_t q = q0; q < n && keys[q] == r; q += 4) {
        bool ok[4];
        uint32_t sp[4];
        Row<NV> sr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            ok[f] = q + f < n && keys[q + f] == r;
            sp[f] = ok[f] ? perm[q + f] : 0u;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (ok[f]) sr[f] = load_row<T, NV, FULL>(src, sp[f], D, lane);
        bool more = true;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            more = more && ok[f];
            if (!more) break;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                acc.v[k].x += sr[f].v[k].x; acc.v[k].y += sr[f].v[k].y; acc.v[k].z += sr[f].v[k].z; acc.v[k].w += sr[f].v[k].w;
            }
        }
        if (!more) break;
    }
    Row<NV> g = load_row<T, NV, FULL>(grad, r, D, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k].x = fmaf(alpha, acc.v[k].x, g.v[k].x); g.v[k].y = fmaf(alpha, acc.v[k].y, g.v[k].y);
        g.v[k].z = fmaf(alpha, acc.v[k].z, g.v[k].z); g.v[k].w = fmaf(alpha, acc.v[k].w, g.v[k].w);
    }
    store_row<T, NV, FULL>(grad, r, D, lane, g);
}

// ---- destination lists for small tables (SASRec's item table on ml-1m: 3,706 rows, 45 K gathered rows per step) ---------
// A radix sort of (row, position) pairs per call (rocPRIM: a dozen launches, ~35 us for 45 K pairs) is replaced by two
// launches whenever the table has at most kSmallRows rows:
//   S1 one workgroup per tile of kSortTile positions: counting sort of the tile by destination row in LDS (histogram,
//      exclusive scan, placement through LDS cursors) — per tile the number of positions of every row, the start of the
//      row's segment, and the positions grouped by row.  The order INSIDE a segment is whatever the LDS atomics gave;
//   S2 one team per destination row collects its positions from all tiles into an LDS list, SORTS the list (positions are
//      unique integers, so the sorted list does not depend on S1's placement order: ascending original position, the order
//      the stable radix sort produced) and adds the listed rows eight per trip, in list order.  No float atomics, bitwise
//      reproducible, same bits as the radix-sorted path.
constexpr int kSortTile = 1024;
constexpr int kSmallRows = 16383;     // table rows + 1 sentinel (padding / out-of-range positions)
constexpr int kLongSeg = 16;          // = the shortest per-team list of S2 (kRowListPerLane x 1 lane)

__global__ __launch_bounds__(kBlock) void small_sort_tiles_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t n_rows,
                                                                   int64_t padding_idx, uint32_t *__restrict__ pos_grouped,
                                                                   int *__restrict__ hist, int *__restrict__ offs) {
    extern __shared__ int sm[];                 // cnt[nk] | cur[nk] | out[kSortTile]
    __shared__ int wave_tot[kBlock / 64];
    const int nk = (int)n_rows + 1;
    int *cnt = sm, *cur = sm + nk, *out = sm + 2 * nk;
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    constexpr int PER = kSortTile / kBlock;
    for (int j = threadIdx.x; j < nk; j += kBlock) cnt[j] = 0;
    int key[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = base + threadIdx.x + k * kBlock;
        key[k] = -1;
        if (i < n) {
            const int64_t r = idx[i];
            key[k] = (r == padding_idx || r < 0 || r >= n_rows) ? (int)n_rows : (int)r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (key[k] >= 0) atomicAdd(&cnt[key[k]], 1);            // integer counts: order irrelevant
    __syncthreads();
    // exclusive scan of cnt[0..nk) into cur: thread t owns counters [t*per, (t+1)*per)
    const int per = (nk + kBlock - 1) / kBlock;
    const int c0 = threadIdx.x * per;
    int local = 0;
    for (int j = 0; j < per; ++j)
        if (c0 + j < nk) local += cnt[c0 + j];
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
        if (c0 + j < nk) {
            cur[c0 + j] = run;
            run += cnt[c0 + j];
        }
    __syncthreads();
    for (int j = threadIdx.x; j < nk; j += kBlock) {             // segment starts, before the cursors move
        hist[(int64_t)blockIdx.x * nk + j] = cnt[j];
        offs[(int64_t)blockIdx.x * nk + j] = cur[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (key[k] >= 0) out[atomicAdd(&cur[key[k]], 1)] = threadIdx.x + k * kBlock;
    __syncthreads();
    // Segments longer than kLongSeg are put in ascending position order here (S2 streams them instead of listing them):
    // one after the other by the whole workgroup, every position counting the smaller ones of its segment.
    __shared__ int long_keys[kSortTile / kLongSeg];
    __shared__ int n_long;
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < nk; j += kBlock)
        if (cnt[j] > kLongSeg) long_keys[atomicAdd(&n_long, 1)] = j;   // at most kSortTile / kLongSeg of them; order irrelevant
    __syncthreads();
    const int nl = n_long;
    for (int q = 0; q < nl; ++q) {
        const int j = long_keys[q];
        const int m = cnt[j], s0 = cur[j] - m;                           // the cursor stands at the segment's end
        int v[PER], rank[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = threadIdx.x + k * kBlock;
            v[k] = e < m ? out[s0 + e] : 0;
            rank[k] = 0;
            if (e < m)
                for (int x = 0; x < m; ++x) rank[k] += out[s0 + x] < v[k] ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (threadIdx.x + k * kBlock < m) out[s0 + rank[k]] = v[k];
        __syncthreads();
    }
    for (int e = threadIdx.x; e < kSortTile; e += kBlock)
        if (base + e < n) pos_grouped[base + e] = (uint32_t)(base + out[e]);
}

constexpr int kRowListPerLane = 16;     // list slots per team = 16 x (lanes of a team): 256 at D >= 64

template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void scatter_add_tiles_kernel(float *__restrict__ grad, int D, int n_rows, int nk, int n_tiles,
                                                                    const uint32_t *__restrict__ pos_grouped,
                                                                    const int *__restrict__ hist, const int *__restrict__ offs,
                                                                    const float *__restrict__ src, float alpha) {
    constexpr int TEAMS = kBlock / T;
    constexpr int kRowList = kRowListPerLane * T;
    __shared__ uint32_t lists[TEAMS][2][kRowList];
    const int lane = threadIdx.x % T, team = threadIdx.x / T;
    const int r = blockIdx.x * TEAMS + team;
    if (r >= n_rows) return;
    uint32_t *list = lists[team][0], *sorted = lists[team][1];
    Row<NV> acc;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_rows = [&](const uint32_t (&sp)[8], int m) {      // m <= 8 listed rows: requested together, added in order
        Row<NV> sr[8];
#pragma unroll
        for (int f = 0; f < 8; ++f)
            if (f < m) sr[f] = load_row<T, NV, FULL>(src, sp[f], D, lane);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (f >= m) break;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                acc.v[k].x += sr[f].v[k].x; acc.v[k].y += sr[f].v[k].y; acc.v[k].z += sr[f].v[k].z; acc.v[k].w += sr[f].v[k].w;
            }
        }
    };
    // ascending order of m collected positions, by counting the smaller ones (unique values), then the sum in that order
    auto flush = [&](int m) {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < m; i += T) {
            const uint32_t v = list[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += list[j] < v ? 1 : 0;
            sorted[rank] = v;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < m; i += 8) {
            uint32_t sp[8];
            const int mm = m - i < 8 ? m - i : 8;
#pragma unroll
            for (int f = 0; f < 8; ++f) sp[f] = f < mm ? sorted[i + f] : 0u;
            add_rows(sp, mm);
        }
        __builtin_amdgcn_wave_barrier();
    };
    int filled = 0;                                            // team-uniform
    bool any = false;
    for (int t0 = 0; t0 < n_tiles; t0 += T) {
        const int t = t0 + lane;
        const int c = t < n_tiles ? hist[(int64_t)t * nk + r] : 0;
        const int o = t < n_tiles ? offs[(int64_t)t * nk + r] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < T; d <<= 1) {
            const int v = __shfl_up(incl, d, T);
            if (lane >= d) incl += v;
        }
        const int total = __shfl(incl, T - 1, T);
        if (total == 0) continue;
        any = true;
        if (total > kRowList) {                                  // a row that fills tiles: one tile's segment at a time
            if (filled) { flush(filled); filled = 0; }
            for (int j = 0; j < T; ++j) {
                const int cj = __shfl(c, j, T), oj = __shfl(o, j, T);
                const int64_t tb = (int64_t)(t0 + j) * kSortTile;
                if (cj <= kLongSeg) {                            // short segment: unordered, through the list
                    for (int e = lane; e < cj; e += T) list[e] = pos_grouped[tb + oj + e];
                    if (cj) flush(cj);
                    continue;
                }
                for (int i0 = 0; i0 < cj; i0 += 8) {             // long segment: S1 left it in ascending order
                    uint32_t sp[8];
                    const int m = cj - i0 < 8 ? cj - i0 : 8;
#pragma unroll
                    for (int f = 0; f < 8; ++f) sp[f] = f < m ? pos_grouped[tb + oj + i0 + f] : 0u;
                    add_rows(sp, m);
                }
            }
            continue;
        }
        if (filled + total > kRowList) { flush(filled); filled = 0; }
        const int at = filled + incl - c;
        for (int e = 0; e < c; ++e) list[at + e] = pos_grouped[(int64_t)t * kSortTile + o + e];
        filled += total;
    }
    if (filled) flush(filled);
    if (!any) return;
    Row<NV> g = load_row<T, NV, FULL>(grad, r, D, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k].x = fmaf(alpha, acc.v[k].x, g.v[k].x); g.v[k].y = fmaf(alpha, acc.v[k].y, g.v[k].y);
        g.v[k].z = fmaf(alpha, acc.v[k].z, g.v[k].z); g.v[k].w = fmaf(alpha, acc.v[k].w, g.v[k].w);
    }
    store_row<T, NV, FULL>(grad, r, D, lane, g);
}

struct ScatterLayout {
    int64_t arr_bytes;
    size_t temp;
    unsigned end_bit;
    int64_t total;
    bool small;                 // counting sort in LDS tiles (table of at most kSmallRows rows)
    bool planned;               // row plan of wr_scatter.hip (no sort): larger tables, at most 2^18 positions
    int64_t n_tiles, comp_bytes, hist_bytes, base_bytes;
};

static int32_t scatter_layout(int64_t n, int64_t n_rows, ScatterLayout &L) {
    WR_REQUIRE(n > 0 && n < (int64_t(1) << 31), WR_E_SHAPE, "scatter: n=%lld out of range", (long long)n);
    WR_REQUIRE(n_rows > 0 && n_rows < (int64_t(1) << 31), WR_E_SHAPE, "scatter: n_rows out of range");
    L.arr_bytes = align_up(n * 4, 256);
    L.n_tiles = (n + kSortTile - 1) / kSortTile;
    L.small = n_rows <= kSmallRows && L.n_tiles * (n_rows + 1) <= (int64_t(1) << 24);
    if (L.small) {
        L.comp_bytes = align_up(L.n_tiles * kSortTile * 4, 256);
        L.hist_bytes = align_up(L.n_tiles * (n_rows + 1) * 4, 256);
        L.base_bytes = 0;
        L.temp = 0;
        L.end_bit = 0;
        L.total = L.comp_bytes + 2 * L.hist_bytes;
        L.planned = false;
        return WR_OK;
    }
    const int64_t pw = scatter_planned_words(n, n_rows);
    L.planned = pw > 0;
    if (L.planned) {
        L.temp = 0;
        L.end_bit = 0;
        L.total = pw * 4;
        return WR_OK;
    }
    L.end_bit = 1;
    while ((int64_t(1) << L.end_bit) < n_rows + 1) ++L.end_bit;
    uint32_t *k = nullptr;
    size_t t = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, t, k, k, k, k, (size_t)n, 0u, L.end_bit, (hipStream_t)0);
    if (e != hipSuccess) return fail_hip(e, "rocprim::radix_sort_pairs (size query)");
    L.temp = (size_t)align_up((int64_t)t, 256);
    L.total = 4 * L.arr_bytes + (int64_t)L.temp;
    return WR_OK;
}

// ------------------------------------------------------------------------------------------------ optimizers
// One thread per float4; row = (4*i)/D selects the stamp.
__global__ __launch_bounds__(kBlock) void sgd_decay_untouched_kernel(float4 *__restrict__ w, int64_t n4, int D4,
                                                                      const int *__restrict__ stamp, int step_id, float lr,
                                                                      float l2) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        if (stamp[i / D4] == step_id) continue;
        float4 v = w[i];  // g' = 0 + l2*w ; w -= lr*g'
        v.x = v.x - lr * (l2 * v.x); v.y = v.y - lr * (l2 * v.y); v.z = v.z - lr * (l2 * v.z); v.w = v.w - lr * (l2 * v.w);
        w[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void sgd_dense_kernel(float4 *__restrict__ w, const float4 *__restrict__ g, int64_t n4,
                                                            int D4, const int *__restrict__ stamp, int step_id, float lr,
                                                            float l2) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const bool live = (stamp == nullptr) || (stamp[i / D4] == step_id);
        if (!live && l2 == 0.f) continue;
        float4 v = w[i];
        float4 gg = live ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (l2 != 0.f) { gg.x = fmaf(l2, v.x, gg.x); gg.y = fmaf(l2, v.y, gg.y); gg.z = fmaf(l2, v.z, gg.z); gg.w = fmaf(l2, v.w, gg.w); }
        v.x -= lr * gg.x; v.y -= lr * gg.y; v.z -= lr * gg.z; v.w -= lr * gg.w;
        w[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void adam_dense_kernel(float4 *__restrict__ w, float4 *__restrict__ m,
                                                             float4 *__restrict__ v, const float4 *__restrict__ g, int64_t n4,
                                                             int D4, const int *__restrict__ stamp, int step_id, float l2,
                                                             float b1, float b2, float eps, float step_size,
                                                             float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const bool live = (stamp == nullptr) || (stamp[i / D4] == step_id);
        float4 ww = w[i], mm = m[i], vv = v[i];
        const float4 gg = live ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        adam_elem(ww.x, mm.x, vv.x, gg.x, l2, b1, b2, eps, step_size, bc2_sqrt);
        adam_elem(ww.y, mm.y, vv.y, gg.y, l2, b1, b2, eps, step_size, bc2_sqrt);
        adam_elem(ww.z, mm.z, vv.z, gg.z, l2, b1, b2, eps, step_size, bc2_sqrt);
        adam_elem(ww.w, mm.w, vv.w, gg.w, l2, b1, b2, eps, step_size, bc2_sqrt);
        w[i] = ww; m[i] = mm; v[i] = vv;
    }
}

// The same pass with the step number in DEVICE memory: consts[2t], consts[2t+1] = step_size, 1/sqrt(bias_correction2) of step
// t = *step_dev.  Nothing in the launch depends on the step, so a captured hipGraph of a training step can be replayed.
__global__ __launch_bounds__(kBlock) void adam_dense_dev_kernel(float4 *__restrict__ w, float4 *__restrict__ m,
                                                                 float4 *__restrict__ v, const float4 *__restrict__ g,
                                                                 int64_t n4, int D4, const int *__restrict__ stamp,
                                                                 const int *__restrict__ step_id_dev, float l2, float b1,
                                                                 float b2, float eps, const float *__restrict__ consts,
                                                                 const int *__restrict__ step_dev) {
    const int t = step_dev[0];
    const float step_size = consts[2 * t], inv_bc2 = consts[2 * t + 1];
    const int step_id = step_id_dev != nullptr ? step_id_dev[0] : 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const bool live = (stamp == nullptr) || (stamp[i / D4] == step_id);
        float4 ww = w[i], mm = m[i], vv = v[i];
        const float4 gg = live ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        adam_elem(ww.x, mm.x, vv.x, gg.x, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.y, mm.y, vv.y, gg.y, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.z, mm.z, vv.z, gg.z, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.w, mm.w, vv.w, gg.w, l2, b1, b2, eps, step_size, inv_bc2);
        w[i] = ww; m[i] = mm; v[i] = vv;
    }
}

// Two tables in one launch (a model's user and item embeddings): workgroups [0, blocks_a) take table a, the rest table b.
struct AdamTab {
    float4 *w, *m, *v;
    const float4 *g;
    int64_t n4;
};

__global__ __launch_bounds__(kBlock) void adam_dense_dev_pair_kernel(AdamTab a, AdamTab b, int blocks_a, float l2, float b1,
                                                                      float b2, float eps, const float *__restrict__ consts,
                                                                      const int *__restrict__ step_dev) {
    const int t = step_dev[0];
    const float step_size = consts[2 * t], inv_bc2 = consts[2 * t + 1];
    const bool second = (int)blockIdx.x >= blocks_a;                 // workgroup-uniform
    const AdamTab &tab = second ? b : a;
    const int64_t first = (int64_t)((int)blockIdx.x - (second ? blocks_a : 0)) * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)(second ? (int)gridDim.x - blocks_a : blocks_a) * kBlock;
    for (int64_t i = first; i < tab.n4; i += stride) {
        float4 ww = tab.w[i], mm = tab.m[i], vv = tab.v[i];
        const float4 gg = tab.g[i];
        adam_elem(ww.x, mm.x, vv.x, gg.x, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.y, mm.y, vv.y, gg.y, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.z, mm.z, vv.z, gg.z, l2, b1, b2, eps, step_size, inv_bc2);
        adam_elem(ww.w, mm.w, vv.w, gg.w, l2, b1, b2, eps, step_size, inv_bc2);
        tab.w[i] = ww; tab.m[i] = mm; tab.v[i] = vv;
    }
}

__global__ void counter_add_kernel(int *__restrict__ c, int delta) { c[0] += delta; }

__global__ __launch_bounds__(kBlock) void axpy_kernel(float4 *__restrict__ y, const float4 *__restrict__ x, int64_t n4,
                                                       float alpha, int overwrite, float *__restrict__ ytail,
                                                       const float *__restrict__ xtail, int tail) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 a = x[i];
        float4 b = overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : y[i];
        b.x = fmaf(alpha, a.x, b.x); b.y = fmaf(alpha, a.y, b.y); b.z = fmaf(alpha, a.z, b.z); b.w = fmaf(alpha, a.w, b.w);
        y[i] = b;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) {
        const float b = overwrite ? 0.f : ytail[threadIdx.x];
        ytail[threadIdx.x] = fmaf(alpha, xtail[threadIdx.x], b);
    }
}

static inline unsigned stream_grid(int64_t n4) {
    int64_t g = (n4 + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ CSR SpMM
// One team per output row: y[i,:] = sum_k val[k] * X[col[k],:], optionally acc[i,:] += y[i,:].
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(int64_t n_rows, const int64_t *__restrict__ row_ptr,
                                                           const int *__restrict__ col, const float *__restrict__ val,
                                                           const float *__restrict__ X, int D, float *__restrict__ Y,
                                                           float *__restrict__ acc) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int64_t i = (int64_t)blockIdx.x * TEAMS + threadIdx.x / T;
    if (i >= n_rows) return;
    Row<NV> s;
#pragma unroll
    for (int k = 0; k < NV; ++k) s.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t k1 = row_ptr[i + 1];
    for (int64_t k = row_ptr[i]; k < k1; ++k) {
        const float a = val[k];
        const Row<NV> x = load_row<T, NV, FULL>(X, col[k], D, lane);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            s.v[c].x = fmaf(a, x.v[c].x, s.v[c].x); s.v[c].y = fmaf(a, x.v[c].y, s.v[c].y);
            s.v[c].z = fmaf(a, x.v[c].z, s.v[c].z); s.v[c].w = fmaf(a, x.v[c].w, s.v[c].w);
        }
    }
    store_row<T, NV, FULL>(Y, i, D, lane, s);
    if (acc != nullptr) {
        Row<NV> a = load_row<T, NV, FULL>(acc, i, D, lane);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            a.v[c].x += s.v[c].x; a.v[c].y += s.v[c].y; a.v[c].z += s.v[c].z; a.v[c].w += s.v[c].w;
        }
        store_row<T, NV, FULL>(acc, i, D, lane, a);
    }
}

// Load-balanced form: the rows are cut into chunks of at most L non-zeros (the adjacency is static, the cut is made once
// on the host).  One team per chunk, four neighbour rows requested per round.  A row with a single chunk is written
// directly; the chunks of a longer row go to `partials` and are summed in chunk order by the second kernel, so the
// result is reproducible and a power-law hub (thousands of neighbours) no longer serialises on one team.
// The running layer sum of LightGCN's propagation (acc = E0 + A E0 + A^2 E0 + ..., then the mean): acc[row] = (base[row] +
// y[row]) * scale, base = acc itself or — first layer — the product's input X (saves the copy of E0 into acc), scale = 1
// or — last layer — 1 / (layers + 1) (saves the scaling pass).  The same operations in the same order as copy / += / scale.
template <int T, int NV, bool FULL>
__device__ __forceinline__ void acc_update(float *__restrict__ acc, const float *__restrict__ acc_src, float acc_scale, int row,
                                           int D, int lane, const Row<NV> &y) {
    Row<NV> a = load_row<T, NV, FULL>(acc_src != nullptr ? acc_src : acc, row, D, lane);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        a.v[q].x = (a.v[q].x + y.v[q].x) * acc_scale; a.v[q].y = (a.v[q].y + y.v[q].y) * acc_scale;
        a.v[q].z = (a.v[q].z + y.v[q].z) * acc_scale; a.v[q].w = (a.v[q].w + y.v[q].w) * acc_scale;
    }
    store_row<T, NV, FULL>(acc, row, D, lane, a);
}

// Rows cut into several chunks are put together in two levels, so that a hub row with hundreds of chunks is not one team's
// chain of hundreds of dependent adds.  Level 1: inside every aligned group of kGroup chunk slots, the first chunk of a row
// adds that row's other chunks of the group into its own partial (in place; the groups' segments are disjoint).  Level 2: the
// row's head adds the group leaders (its own partial, then the ones at the following group boundaries) and writes the row.
// Fixed order, no atomics.
constexpr int kGroup = 16;

#ifndef WR_SPMM_COMBINE_FLY
#define WR_SPMM_COMBINE_FLY 8
#endif
template <int T, int NV, bool FULL>
__device__ __forceinline__ Row<NV> sum_partials(const float *__restrict__ partials, const int *__restrict__ chunk_row, int row,
                                                int first, int next, int step, int end, int D, int lane) {
    Row<NV> s = load_row<T, NV, FULL>(partials, first, D, lane);
    constexpr int kFly = WR_SPMM_COMBINE_FLY;   // partial rows in flight per trip, added in order
    for (int j = next; j < end && chunk_row[j] == row; j += kFly * step) {
        Row<NV> x[kFly];
        bool ok[kFly];
#pragma unroll
        for (int f = 0; f < kFly; ++f) ok[f] = j + f * step < end && chunk_row[j + f * step] == row;
#pragma unroll
        for (int f = 0; f < kFly; ++f)
            if (ok[f]) x[f] = load_row<T, NV, FULL>(partials, j + f * step, D, lane);
#pragma unroll
        for (int f = 0; f < kFly; ++f) {
            if (!ok[f]) continue;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                s.v[q].x += x[f].v[q].x; s.v[q].y += x[f].v[q].y; s.v[q].z += x[f].v[q].z; s.v[q].w += x[f].v[q].w;
            }
        }
    }
    return s;
}

#ifndef WR_SPMM_FLY
#define WR_SPMM_FLY 8
#endif
// neighbour rows requested per round and team.  The kernel is bound by its chain of dependent L2 round trips (index trip,
// then one trip per round), not by L2 bandwidth: A/B on the ml-1m-shaped graph (scripts/ab_spmm.py), 4 -> 8 per round.
constexpr int kSpmmFly = (WR_SPMM_FLY);

// FUSE: no combine launch.  A chunk of a row cut into several stores its partial write-through, waits for the store and
// counts itself in at the row's counter; the team that arrives last (the count tells it) makes the other teams' partials
// visible with one agent-scope acquire and adds all the row's partials in chunk order — the bits of spmm_combine_kernel<ONE>,
// whoever does it — then zeroes the counter for the next product.  Needs rows of whole 128-byte lines (D % 32 == 0): a
// partial row shares no line with another team's.  row_span[2r], row_span[2r + 1]: first chunk and number of chunks of row r.
template <int T, int NV, bool FULL, bool FUSE = false>
__global__ __launch_bounds__(kBlock) void spmm_chunk_kernel(int n_chunks, const int64_t *__restrict__ chunk_ptr,
                                                             const int *__restrict__ chunk_row, const int *__restrict__ col,
                                                             const float *__restrict__ val, const float *__restrict__ X, int D,
                                                             float *__restrict__ Y, float *__restrict__ acc,
                                                             float *__restrict__ partials,
                                                             const signed char *__restrict__ row_mode,
                                                             const float *__restrict__ acc_src, float acc_scale,
                                                             const int *__restrict__ row_span = nullptr,
                                                             unsigned *__restrict__ counters = nullptr) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int c = blockIdx.x * TEAMS + threadIdx.x / T;
    if (c >= n_chunks) return;
    const int row = chunk_row[c];
    // row_mode (hybrid product, wr_spmm_mfma.hip): 0 write the row, 1 add to what the dense tiles left in Y[row],
    // 2 the row belongs to the dense tiles entirely
    const int mode = row_mode != nullptr ? (int)row_mode[row] : 0;
    if (mode == 2) return;
    const bool multi = (c > 0 && chunk_row[c - 1] == row) || (c + 1 < n_chunks && chunk_row[c + 1] == row);
    Row<NV> s;
#pragma unroll
    for (int k = 0; k < NV; ++k) s.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // The chunk's column ids and values are fetched by the team's lanes in one go (2*T entries per trip, lane l takes
    // entries l and l+T) and handed round with shuffles: one index round trip per 2*T non-zeros instead of one per four,
    // and four row gathers in flight.  Accumulation order = CSR order, as in the sequential loop.
    const int64_t k1 = chunk_ptr[c + 1];
    for (int64_t base = chunk_ptr[c]; base < k1; base += 2 * T) {
        const int cnt = (int)((k1 - base < 2 * T) ? (k1 - base) : 2 * T);
        const int c_lo = (lane < cnt) ? col[base + lane] : 0, c_hi = (lane + T < cnt) ? col[base + T + lane] : 0;
        const float a_lo = (lane < cnt) ? val[base + lane] : 0.f, a_hi = (lane + T < cnt) ? val[base + T + lane] : 0.f;
        for (int j = 0; j < cnt; j += kSpmmFly) {
            int cj[kSpmmFly];
            float aj[kSpmmFly];
            Row<NV> x[kSpmmFly];
#pragma unroll
            for (int f = 0; f < kSpmmFly; ++f) {   // entries past the chunk get value 0 and row 0: they add +0 in order
                const int e = j + f;
                const int src_lane = e < T ? e : e - T;
                const int cl = __shfl(c_lo, src_lane, T), ch = __shfl(c_hi, src_lane, T);
                const float al = __shfl(a_lo, src_lane, T), ah = __shfl(a_hi, src_lane, T);
                cj[f] = e < T ? cl : ch;
                aj[f] = e < cnt ? (e < T ? al : ah) : 0.f;
                if (e >= cnt) cj[f] = -1;
            }
#pragma unroll
            for (int f = 0; f < kSpmmFly; ++f)
                if (cj[f] >= 0) x[f] = load_row<T, NV, FULL>(X, cj[f], D, lane);
#pragma unroll
            for (int f = 0; f < kSpmmFly; ++f) {
                if (cj[f] < 0) continue;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    s.v[q].x = fmaf(aj[f], x[f].v[q].x, s.v[q].x); s.v[q].y = fmaf(aj[f], x[f].v[q].y, s.v[q].y);
                    s.v[q].z = fmaf(aj[f], x[f].v[q].z, s.v[q].z); s.v[q].w = fmaf(aj[f], x[f].v[q].w, s.v[q].w);
                }
            }
        }
    }
    if (multi) {
        if constexpr (FUSE) {
            const int first = row_span[2 * row], cnt = row_span[2 * row + 1];
            store_row_wt<T, NV, FULL>(partials, c, D, lane, s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores of this wave have left
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(counters + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __shfl(old, 0, T);
            if ((int)old != cnt - 1) return;                     // every team of the row leaves here but the last to arrive
            if (lane == 0) __hip_atomic_store(counters + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s = sum_partials<T, NV, FULL>(partials, chunk_row, row, first, first + 1, 1, n_chunks, D, lane);
        } else {
            store_row<T, NV, FULL>(partials, c, D, lane, s);
            return;
        }
    }
    if (mode == 1) {
        const Row<NV> y0 = load_row<T, NV, FULL>(Y, row, D, lane);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            s.v[q].x += y0.v[q].x; s.v[q].y += y0.v[q].y; s.v[q].z += y0.v[q].z; s.v[q].w += y0.v[q].w;
        }
    }
    store_row<T, NV, FULL>(Y, row, D, lane, s);
    if (acc != nullptr) acc_update<T, NV, FULL>(acc, acc_src, acc_scale, row, D, lane, s);
}

template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void spmm_combine_groups_kernel(int n_chunks, const int *__restrict__ chunk_row,
                                                                      float *__restrict__ partials, int D) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int c = blockIdx.x * TEAMS + threadIdx.x / T;
    if (c >= n_chunks) return;
    const int row = chunk_row[c];
    const bool leader = (c % kGroup == 0) || chunk_row[c - 1] != row;
    const int end = (c / kGroup + 1) * kGroup < n_chunks ? (c / kGroup + 1) * kGroup : n_chunks;
    if (!leader || c + 1 >= end || chunk_row[c + 1] != row) return;   // nothing of this row follows inside the group
    const Row<NV> s = sum_partials<T, NV, FULL>(partials, chunk_row, row, c, c + 1, 1, end, D, lane);
    store_row<T, NV, FULL>(partials, c, D, lane, s);
}

// ONE: no group level — the head adds all the row's chunks itself (eight partial rows in flight).  One launch less per
// product; right while no row has more than a few dozen chunks (the caller decides: wr_spmm_csr_chunked_levels).
template <int T, int NV, bool FULL, bool ONE>
__global__ __launch_bounds__(kBlock) void spmm_combine_kernel(int n_chunks, const int *__restrict__ chunk_row,
                                                               const float *__restrict__ partials, int D, float *__restrict__ Y,
                                                               float *__restrict__ acc,
                                                               const signed char *__restrict__ row_mode,
                                                               const float *__restrict__ acc_src, float acc_scale) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int c = blockIdx.x * TEAMS + threadIdx.x / T;
    if (c >= n_chunks) return;
    const int row = chunk_row[c];
    const bool head = (c == 0 || chunk_row[c - 1] != row) && (c + 1 < n_chunks && chunk_row[c + 1] == row);
    if (!head) return;
    const int mode = row_mode != nullptr ? (int)row_mode[row] : 0;
    if (mode == 2) return;
    // the head's own partial holds its group's sum; the row's other group leaders sit at the following group boundaries
    Row<NV> s = ONE ? sum_partials<T, NV, FULL>(partials, chunk_row, row, c, c + 1, 1, n_chunks, D, lane)
                    : sum_partials<T, NV, FULL>(partials, chunk_row, row, c, (c / kGroup + 1) * kGroup, kGroup, n_chunks, D, lane);
    if (mode == 1) {
        const Row<NV> y0 = load_row<T, NV, FULL>(Y, row, D, lane);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            s.v[q].x += y0.v[q].x; s.v[q].y += y0.v[q].y; s.v[q].z += y0.v[q].z; s.v[q].w += y0.v[q].w;
        }
    }
    store_row<T, NV, FULL>(Y, row, D, lane, s);
    if (acc != nullptr) acc_update<T, NV, FULL>(acc, acc_src, acc_scale, row, D, lane, s);
}

// ------------------------------------------------------------------------------------------------ EmbLoss
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void embloss_sumsq_kernel(const float *__restrict__ U, const float *__restrict__ I, int D,
                                                                const int64_t *__restrict__ u, const int64_t *__restrict__ p,
                                                                const int64_t *__restrict__ n, int B,
                                                                float *__restrict__ partials, int n_blocks) {
    __shared__ float scratch[kBlock / 64];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int b = blockIdx.x * TEAMS + threadIdx.x / T;
    float su = 0.f, sp = 0.f, sn = 0.f;
    if (b < B) {
        const Row<NV> ur = load_row<T, NV, FULL>(U, u[b], D, lane);
        const Row<NV> pr = load_row<T, NV, FULL>(I, p[b], D, lane);
        const Row<NV> nr = load_row<T, NV, FULL>(I, n[b], D, lane);
        su = dot_partial<NV>(ur, ur);
        sp = dot_partial<NV>(pr, pr);
        sn = dot_partial<NV>(nr, nr);
    }
    const float a = block_sum(su, scratch);
    __syncthreads();
    const float c = block_sum(sp, scratch);
    __syncthreads();
    const float e = block_sum(sn, scratch);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = a;
        partials[n_blocks + blockIdx.x] = c;
        partials[2 * n_blocks + blockIdx.x] = e;
    }
}

__global__ __launch_bounds__(kBlock) void embloss_finish_kernel(const float *__restrict__ partials, int n_blocks,
                                                                 float *__restrict__ sq3) {
    __shared__ float scratch[kBlock / 64];
    for (int j = 0; j < 3; ++j) {
        float a = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += kBlock) a += partials[j * n_blocks + i];
        const float s = block_sum(a, scratch);
        if (threadIdx.x == 0) sq3[j] = s;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ LightGCN loss, fused
// LightGCN.predict's per-batch tail (reference src/models/general/LightGCN.py:156-175) in one pass over the batch: the BPR
// terms on the PROPAGATED rows (two dots, -log(1e-10 + sigmoid)) and the squared norms of the three gathered EGO rows
// (EmbLoss, loss.py:94-98); a one-workgroup second kernel folds the per-block partials in a fixed order into
// loss = mean(term) + reg_weight * (||U0[u]||_F + ||I0[p]||_F + ||I0[n]||_F) / B   and keeps the three sums of squares for
// the backward pass.  Two launches where bpr_fwd + finish_loss + embloss_sumsq + embloss_finish + five small tensor
// operations were nine.
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void lightgcn_tail_fwd_kernel(const float *__restrict__ Ua, const float *__restrict__ Ia,
                                                                    const float *__restrict__ U0, const float *__restrict__ I0,
                                                                    int D, const int64_t *__restrict__ u,
                                                                    const int64_t *__restrict__ p, const int64_t *__restrict__ n,
                                                                    int B, float *__restrict__ partials, int n_blocks,
                                                                    int64_t n_users, int64_t n_items) {
    __shared__ float scratch[kBlock / 64];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int b = blockIdx.x * TEAMS + threadIdx.x / T;
    float term = 0.f, su = 0.f, sp = 0.f, sn = 0.f;
    if (b < B) {
        // ids are clamped into the tables: an id out of range is REPORTED by the batch plan (nn.Embedding would raise), and
        // no kernel ever reads outside a table on the way to that report
        const int64_t ub = min(max(u[b], (int64_t)0), n_users - 1), pb = min(max(p[b], (int64_t)0), n_items - 1),
                      nb = min(max(n[b], (int64_t)0), n_items - 1);
        const Row<NV> ua = load_row<T, NV, FULL>(Ua, ub, D, lane), pa = load_row<T, NV, FULL>(Ia, pb, D, lane),
                      na = load_row<T, NV, FULL>(Ia, nb, D, lane);
        const Row<NV> u0 = load_row<T, NV, FULL>(U0, ub, D, lane), p0 = load_row<T, NV, FULL>(I0, pb, D, lane),
                      n0 = load_row<T, NV, FULL>(I0, nb, D, lane);
        const float spos = team_sum<T>(dot_partial<NV>(ua, pa));
        const float sneg = team_sum<T>(dot_partial<NV>(ua, na));
        float tt, cc;
        bpr_terms(spos, sneg, (float)B, tt, cc);
        term = lane == 0 ? tt : 0.f;
        su = dot_partial<NV>(u0, u0);
        sp = dot_partial<NV>(p0, p0);
        sn = dot_partial<NV>(n0, n0);
    }
    const float a = block_sum(term, scratch);
    __syncthreads();
    const float c = block_sum(su, scratch);
    __syncthreads();
    const float e = block_sum(sp, scratch);
    __syncthreads();
    const float f = block_sum(sn, scratch);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = a;
        partials[n_blocks + blockIdx.x] = c;
        partials[2 * n_blocks + blockIdx.x] = e;
        partials[3 * n_blocks + blockIdx.x] = f;
    }
}

__global__ __launch_bounds__(kBlock) void lightgcn_tail_finish_kernel(const float *__restrict__ partials, int n_blocks, float B,
                                                                       float reg_weight, float *__restrict__ loss,
                                                                       float *__restrict__ sq3) {
    __shared__ float scratch[kBlock / 64];
    __shared__ float tot[4];
    for (int j = 0; j < 4; ++j) {
        float a = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += kBlock) a += partials[j * n_blocks + i];
        const float s = block_sum(a, scratch);
        if (threadIdx.x == 0) tot[j] = s;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sq3[0] = tot[1]; sq3[1] = tot[2]; sq3[2] = tot[3];
        const float reg = (sqrtf(tot[1]) + sqrtf(tot[2]) + sqrtf(tot[3])) / B;
        loss[0] = tot[0] / B + reg_weight * reg;
    }
}

// Gradient of EmbLoss w.r.t. the gathered ego rows, accumulated per table row using a batch plan: every occurrence of row
// r in block k contributes (reg_weight / (B * ||block_k||_F)) * row, so a run of m occurrences adds m times that.
// Both tables in one launch: workgroups [0, user_blocks) take the user rows (runs of tu), the rest the item rows (runs of
// oc_item; positives and negatives have different norms).
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void embloss_grad_kernel(const float *__restrict__ user_tab, float *__restrict__ grad_user,
                                                               const int *__restrict__ tu, int n_user, int user_blocks,
                                                               const float *__restrict__ item_tab, float *__restrict__ grad_item,
                                                               const int *__restrict__ oc_item, const int *__restrict__ oc_src,
                                                               int n_item, int D, const float *__restrict__ sq3,
                                                               float reg_over_B) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int side = (int)blockIdx.x >= user_blocks ? 1 : 0;              // workgroup-uniform
    const float *__restrict__ tab = side ? item_tab : user_tab;
    float *__restrict__ grad = side ? grad_item : grad_user;
    const int *__restrict__ keys = side ? oc_item : tu;
    const int *__restrict__ src = oc_src;
    const int n = side ? n_item : n_user;
    const int q0 = ((int)blockIdx.x - (side ? user_blocks : 0)) * TEAMS + threadIdx.x / T;
    if (q0 >= n) return;
    const int r = keys[q0];
    if (q0 > 0 && keys[q0 - 1] == r) return;  // not the head of its run
    int m_a = 0, m_b = 0;                      // users: all in m_a; items: positives in m_a, negatives in m_b
    for (int q = q0; q < n && keys[q] == r; ++q) {
        if (side == 1 && (src[q] & 1)) ++m_b; else ++m_a;
    }
    const float na = sqrtf(sq3[side == 0 ? 0 : 1]), nb = sqrtf(sq3[2]);
    float coef = 0.f;
    if (m_a > 0 && na > 0.f) coef += (float)m_a * (reg_over_B / na);
    if (m_b > 0 && nb > 0.f) coef += (float)m_b * (reg_over_B / nb);
    const Row<NV> x = load_row<T, NV, FULL>(tab, r, D, lane);
    Row<NV> g = load_row<T, NV, FULL>(grad, r, D, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k].x = fmaf(coef, x.v[k].x, g.v[k].x); g.v[k].y = fmaf(coef, x.v[k].y, g.v[k].y);
        g.v[k].z = fmaf(coef, x.v[k].z, g.v[k].z); g.v[k].w = fmaf(coef, x.v[k].w, g.v[k].w);
    }
    store_row<T, NV, FULL>(grad, r, D, lane, g);
}

static inline int teams_per_block_for(int D) {
    if (D >= 64) return kBlock / 16;
    if (D == 32) return kBlock / 8;
    if (D == 16) return kBlock / 4;
    if (D == 8) return kBlock / 2;
    if (D == 4) return kBlock;
    return kBlock / 16;
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_gather_rows(const float *tab, int64_t n_rows, int32_t D, const int64_t *idx, int64_t n, float *out,
                       void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    WR_REQUIRE(idx && out, WR_E_NULL, "idx/out must not be NULL");
    WR_REQUIRE(aligned16(out), WR_E_ALIGN, "out is not 16-byte aligned");
    WR_REQUIRE(n >= 0 && n < (int64_t(1) << 31), WR_E_SHAPE, "n out of range");
    if (n == 0) return WR_OK;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const unsigned grid = (unsigned)((n + tpb - 1) / tpb);
#define WR_CALL_G(T_, NV_, FULL_) \
    hipLaunchKernelGGL((gather_rows_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream, tab, D, n_rows, idx, n, out)
    WR_DISPATCH_D(D, WR_CALL_G);
#undef WR_CALL_G
    WR_LAUNCH_CHECK("gather_rows_kernel");
    return WR_OK;
}

int64_t wr_scatter_add_workspace_bytes(int64_t n, int64_t n_rows) {
    ScatterLayout L;
    const int32_t rc = scatter_layout(n, n_rows, L);
    if (rc != WR_OK) return rc < 0 ? rc : -(int64_t)rc;
    return L.total;
}

int32_t wr_scatter_add_rows(float *grad, int64_t n_rows, int32_t D, const int64_t *idx, const float *src, int64_t n,
                            int64_t padding_idx, float alpha, void *workspace, int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(grad, n_rows, D, "grad")) != WR_OK) return rc;
    WR_REQUIRE(idx && src, WR_E_NULL, "idx/src must not be NULL");
    WR_REQUIRE(aligned16(src), WR_E_ALIGN, "src is not 16-byte aligned");
    if (n == 0) return WR_OK;
    ScatterLayout L;
    if ((rc = scatter_layout(n, n_rows, L)) != WR_OK) return rc;
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= L.total, WR_E_WORKSPACE,
               "scatter workspace %lld B < %lld B", (long long)workspace_bytes, (long long)L.total);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char *ws = reinterpret_cast<char *>(workspace);
    uint32_t *keyB = nullptr, *valB = nullptr;
    const int tpb = teams_per_block_for(D);
    if (L.small) {
        uint32_t *comp = reinterpret_cast<uint32_t *>(ws);
        int *hist = reinterpret_cast<int *>(ws + L.comp_bytes);
        int *offs = reinterpret_cast<int *>(ws + L.comp_bytes + L.hist_bytes);
        const int nk = (int)n_rows + 1;
        const size_t lds = ((size_t)kSortTile + 2 * (size_t)nk) * 4;
        if (lds > 64 * 1024)
            WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_sort_tiles_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(small_sort_tiles_kernel, dim3((unsigned)L.n_tiles), dim3(kBlock), lds, stream, idx, n, n_rows,
                           padding_idx, comp, hist, offs);
        WR_LAUNCH_CHECK("small_sort_tiles_kernel");
        const unsigned grid_rows = (unsigned)((n_rows + tpb - 1) / tpb);
#define WR_CALL_ST(T_, NV_, FULL_)                                                                                       \
    hipLaunchKernelGGL((scatter_add_tiles_kernel<T_, NV_, FULL_>), dim3(grid_rows), dim3(kBlock), 0, stream, grad, D,      \
                       (int)n_rows, nk, (int)L.n_tiles, comp, hist, offs, src, alpha)
        WR_DISPATCH_D(D, WR_CALL_ST);
#undef WR_CALL_ST
        WR_LAUNCH_CHECK("scatter_add_tiles_kernel");
        return WR_OK;
    } else if (L.planned) {
        return scatter_add_planned_once(grad, n_rows, D, idx, src, n, padding_idx, alpha, reinterpret_cast<int32_t *>(ws), stream);
    } else {
        uint32_t *keyA = reinterpret_cast<uint32_t *>(ws);
        keyB = reinterpret_cast<uint32_t *>(ws + L.arr_bytes);
        uint32_t *valA = reinterpret_cast<uint32_t *>(ws + 2 * L.arr_bytes);
        valB = reinterpret_cast<uint32_t *>(ws + 3 * L.arr_bytes);
        void *temp = ws + 4 * L.arr_bytes;
        size_t temp_bytes = L.temp;
        const unsigned g1 = (unsigned)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(scatter_keys_kernel, dim3(g1), dim3(kBlock), 0, stream, idx, n, n_rows, padding_idx, keyA, valA);
        WR_LAUNCH_CHECK("scatter_keys_kernel");
        WR_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keyA, keyB, valA, valB, (size_t)n, 0u, L.end_bit, stream));
    }
    const unsigned grid = (unsigned)((n + tpb - 1) / tpb);
#define WR_CALL_S(T_, NV_, FULL_)                                                                                    \
    hipLaunchKernelGGL((scatter_add_sorted_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream, grad, D,     \
                       (uint32_t)n_rows, keyB, valB, n, src, alpha)
    WR_DISPATCH_D(D, WR_CALL_S);
#undef WR_CALL_S
    WR_LAUNCH_CHECK("scatter_add_sorted_kernel");
    return WR_OK;
}

int32_t wr_apply_rows_sorted(float *tab, int64_t n_rows, int32_t D, const int32_t *sorted_rows, const int32_t *perm,
                             const float *src, int64_t n, float alpha, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    WR_REQUIRE(n >= 0 && n < (int64_t(1) << 31), WR_E_SHAPE, "n out of range");
    if (n == 0) return WR_OK;
    WR_REQUIRE(sorted_rows && perm && src, WR_E_NULL, "sorted_rows/perm/src must not be NULL");
    WR_REQUIRE(aligned16(src), WR_E_ALIGN, "src is not 16-byte aligned");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const unsigned grid = (unsigned)((n + tpb - 1) / tpb);
#define WR_CALL_AS(T_, NV_, FULL_)                                                                                   \
    hipLaunchKernelGGL((scatter_add_sorted_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream, tab, D,      \
                       (uint32_t)n_rows, reinterpret_cast<const uint32_t *>(sorted_rows),                            \
                       reinterpret_cast<const uint32_t *>(perm), n, src, alpha)
    WR_DISPATCH_D(D, WR_CALL_AS);
#undef WR_CALL_AS
    WR_LAUNCH_CHECK("scatter_add_sorted_kernel");
    return WR_OK;
}

int32_t wr_sgd_decay_untouched(float *tab, int64_t n_rows, int32_t D, const int32_t *stamp, int32_t step_id, float lr,
                               float l2, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    WR_REQUIRE(stamp, WR_E_NULL, "stamp must not be NULL");
    if (l2 == 0.f) return WR_OK;
    const int64_t n4 = n_rows * (D / 4);
    hipLaunchKernelGGL(sgd_decay_untouched_kernel, dim3(stream_grid(n4)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream_), reinterpret_cast<float4 *>(tab), n4, D / 4, stamp, step_id,
                       lr, l2);
    WR_LAUNCH_CHECK("sgd_decay_untouched_kernel");
    return WR_OK;
}

int32_t wr_sgd_dense(float *tab, int64_t n_rows, int32_t D, const float *grad, const int32_t *stamp, int32_t step_id,
                     float lr, float l2, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    if ((rc = check_table(grad, n_rows, D, "grad")) != WR_OK) return rc;
    const int64_t n4 = n_rows * (D / 4);
    hipLaunchKernelGGL(sgd_dense_kernel, dim3(stream_grid(n4)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_),
                       reinterpret_cast<float4 *>(tab), reinterpret_cast<const float4 *>(grad), n4, D / 4, stamp, step_id,
                       lr, l2);
    WR_LAUNCH_CHECK("sgd_dense_kernel");
    return WR_OK;
}

int32_t wr_adam_dense(float *tab, float *exp_avg, float *exp_avg_sq, int64_t n_rows, int32_t D, const float *grad,
                      const int32_t *stamp, int32_t step_id, int64_t adam_step, float lr, float l2, float beta1,
                      float beta2, float eps, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg, n_rows, D, "exp_avg")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg_sq, n_rows, D, "exp_avg_sq")) != WR_OK) return rc;
    if ((rc = check_table(grad, n_rows, D, "grad")) != WR_OK) return rc;
    WR_REQUIRE(adam_step >= 1, WR_E_RANGE, "adam_step must be >= 1");
    float step_size, bc2_sqrt;   // bc2_sqrt holds 1/sqrt(bias_correction2)
    adam_step_consts(adam_step, lr, beta1, beta2, &step_size, &bc2_sqrt);
    const int64_t n4 = n_rows * (D / 4);
    hipLaunchKernelGGL(adam_dense_kernel, dim3(stream_grid(n4)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_),
                       reinterpret_cast<float4 *>(tab), reinterpret_cast<float4 *>(exp_avg),
                       reinterpret_cast<float4 *>(exp_avg_sq), reinterpret_cast<const float4 *>(grad), n4, D / 4, stamp,
                       step_id, l2, beta1, beta2, eps, step_size, bc2_sqrt);
    WR_LAUNCH_CHECK("adam_dense_kernel");
    return WR_OK;
}

int32_t wr_adam_dense_dev(float *tab, float *exp_avg, float *exp_avg_sq, int64_t n_rows, int32_t D, const float *grad,
                          const float *consts, int64_t n_consts, const int32_t *step_dev, float l2, float beta1, float beta2,
                          float eps, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg, n_rows, D, "exp_avg")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg_sq, n_rows, D, "exp_avg_sq")) != WR_OK) return rc;
    if ((rc = check_table(grad, n_rows, D, "grad")) != WR_OK) return rc;
    WR_REQUIRE(consts != nullptr && step_dev != nullptr && n_consts >= 2, WR_E_NULL, "consts / step_dev is NULL");
    const int64_t n4 = n_rows * (D / 4);
    hipLaunchKernelGGL(adam_dense_dev_kernel, dim3(stream_grid(n4)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_),
                       reinterpret_cast<float4 *>(tab), reinterpret_cast<float4 *>(exp_avg),
                       reinterpret_cast<float4 *>(exp_avg_sq), reinterpret_cast<const float4 *>(grad), n4, D / 4, nullptr,
                       nullptr, l2, beta1, beta2, eps, consts, step_dev);
    WR_LAUNCH_CHECK("adam_dense_dev_kernel");
    return WR_OK;
}

int32_t wr_adam_dense_dev_pair(float *tab_a, float *exp_avg_a, float *exp_avg_sq_a, int64_t n_rows_a, const float *grad_a,
                               float *tab_b, float *exp_avg_b, float *exp_avg_sq_b, int64_t n_rows_b, const float *grad_b,
                               int32_t D, const float *consts, int64_t n_consts, const int32_t *step_dev, float l2,
                               float beta1, float beta2, float eps, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab_a, n_rows_a, D, "tab_a")) != WR_OK) return rc;
    if ((rc = check_table(tab_b, n_rows_b, D, "tab_b")) != WR_OK) return rc;
    WR_REQUIRE(exp_avg_a && exp_avg_sq_a && grad_a && exp_avg_b && exp_avg_sq_b && grad_b, WR_E_NULL, "NULL state or gradient");
    WR_REQUIRE(aligned16(exp_avg_a) && aligned16(exp_avg_sq_a) && aligned16(grad_a) && aligned16(exp_avg_b) &&
                   aligned16(exp_avg_sq_b) && aligned16(grad_b), WR_E_ALIGN, "state/gradient not 16-byte aligned");
    WR_REQUIRE(consts != nullptr && step_dev != nullptr && n_consts >= 2, WR_E_NULL, "consts / step_dev is NULL");
    const int64_t n4a = n_rows_a * (D / 4), n4b = n_rows_b * (D / 4);
    const unsigned ga = stream_grid(n4a), gb = stream_grid(n4b);
    const AdamTab a{reinterpret_cast<float4 *>(tab_a), reinterpret_cast<float4 *>(exp_avg_a), reinterpret_cast<float4 *>(exp_avg_sq_a),
                    reinterpret_cast<const float4 *>(grad_a), n4a};
    const AdamTab b{reinterpret_cast<float4 *>(tab_b), reinterpret_cast<float4 *>(exp_avg_b), reinterpret_cast<float4 *>(exp_avg_sq_b),
                    reinterpret_cast<const float4 *>(grad_b), n4b};
    hipLaunchKernelGGL(adam_dense_dev_pair_kernel, dim3(ga + gb), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_), a, b,
                       (int)ga, l2, beta1, beta2, eps, consts, step_dev);
    WR_LAUNCH_CHECK("adam_dense_dev_pair_kernel");
    return WR_OK;
}

int32_t wr_counter_add(int32_t *counter, int32_t delta, void *stream_) {
    WR_REQUIRE(counter != nullptr, WR_E_NULL, "counter is NULL");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream_), counter, delta);
    WR_LAUNCH_CHECK("counter_add_kernel");
    return WR_OK;
}

int32_t wr_spmm_csr(int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const float *val, const float *X,
                    int32_t D, float *Y, float *acc, void *stream_) {
    int32_t rc;
    if ((rc = check_table(X, n_rows, D, "X")) != WR_OK) return rc;
    if ((rc = check_table(Y, n_rows, D, "Y")) != WR_OK) return rc;
    WR_REQUIRE(row_ptr && col && val, WR_E_NULL, "CSR arrays must not be NULL");
    WR_REQUIRE(X != Y, WR_E_SHAPE, "spmm: X and Y must not alias");
    WR_REQUIRE(acc == nullptr || aligned16(acc), WR_E_ALIGN, "acc is not 16-byte aligned");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const unsigned grid = (unsigned)((n_rows + tpb - 1) / tpb);
#define WR_CALL_M(T_, NV_, FULL_)                                                                                     \
    hipLaunchKernelGGL((spmm_csr_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream, n_rows, row_ptr, col, val, \
                       X, D, Y, acc)
    WR_DISPATCH_D(D, WR_CALL_M);
#undef WR_CALL_M
    WR_LAUNCH_CHECK("spmm_csr_kernel");
    return WR_OK;
}

static int32_t spmm_chunked_impl(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                 const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                                 float *partials, const signed char *row_mode, void *stream_, int levels = 2,
                                 bool acc_from_x = false, float acc_scale = 1.0f) {
    int32_t rc;
    WR_REQUIRE(levels == 1 || levels == 2, WR_E_RANGE, "combine levels must be 1 or 2");
    if ((rc = check_table(X, n_rows, D, "X")) != WR_OK) return rc;
    if ((rc = check_table(Y, n_rows, D, "Y")) != WR_OK) return rc;
    WR_REQUIRE(chunk_ptr && chunk_row && col && val && partials, WR_E_NULL, "chunked CSR arrays must not be NULL");
    WR_REQUIRE(n_chunks >= n_rows && n_chunks < (int64_t(1) << 31), WR_E_SHAPE, "every row needs at least one chunk");
    WR_REQUIRE(X != Y, WR_E_SHAPE, "spmm: X and Y must not alias");
    WR_REQUIRE(aligned16(partials) && (acc == nullptr || aligned16(acc)), WR_E_ALIGN, "partials/acc not 16-byte aligned");
    WR_REQUIRE(!acc_from_x || (acc != nullptr && acc != X), WR_E_SHAPE, "acc_from_x needs an acc buffer other than X");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const unsigned grid = (unsigned)((n_chunks + tpb - 1) / tpb);
    const float *acc_src = acc_from_x ? X : nullptr;
#define WR_CALL_MC(T_, NV_, FULL_)                                                                                      \
    do {                                                                                                                \
        hipLaunchKernelGGL((spmm_chunk_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream, (int)n_chunks,      \
                           chunk_ptr, chunk_row, col, val, X, D, Y, acc, partials, row_mode, acc_src, acc_scale);       \
        if (levels == 1) {                                                                                              \
            hipLaunchKernelGGL((spmm_combine_kernel<T_, NV_, FULL_, true>), dim3(grid), dim3(kBlock), 0, stream,         \
                               (int)n_chunks, chunk_row, partials, D, Y, acc, row_mode, acc_src, acc_scale);            \
        } else {                                                                                                        \
            hipLaunchKernelGGL((spmm_combine_groups_kernel<T_, NV_, FULL_>), dim3(grid), dim3(kBlock), 0, stream,        \
                               (int)n_chunks, chunk_row, partials, D);                                                  \
            hipLaunchKernelGGL((spmm_combine_kernel<T_, NV_, FULL_, false>), dim3(grid), dim3(kBlock), 0, stream,        \
                               (int)n_chunks, chunk_row, partials, D, Y, acc, row_mode, acc_src, acc_scale);            \
        }                                                                                                               \
    } while (0)
    WR_DISPATCH_D(D, WR_CALL_MC);
#undef WR_CALL_MC
    WR_LAUNCH_CHECK("spmm_chunk_kernel");
    return WR_OK;
}

int32_t wr_spmm_csr_chunked(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                            const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                            float *partials, void *stream_) {
    return spmm_chunked_impl(n_rows, n_chunks, chunk_ptr, chunk_row, col, val, X, D, Y, acc, partials, nullptr, stream_);
}

int32_t wr_spmm_csr_chunked_modes(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                  const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                                  float *partials, const int8_t *row_mode, void *stream_) {
    WR_REQUIRE(row_mode != nullptr, WR_E_NULL, "row_mode is NULL");
    return spmm_chunked_impl(n_rows, n_chunks, chunk_ptr, chunk_row, col, val, X, D, Y, acc, partials,
                             reinterpret_cast<const signed char *>(row_mode), stream_);
}

int32_t wr_spmm_csr_chunked_levels(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                   const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                                   float *partials, const int8_t *row_mode, int32_t levels, int32_t acc_from_x,
                                   float acc_scale, void *stream_) {
    return spmm_chunked_impl(n_rows, n_chunks, chunk_ptr, chunk_row, col, val, X, D, Y, acc, partials,
                             reinterpret_cast<const signed char *>(row_mode), stream_, (int)levels, acc_from_x != 0, acc_scale);
}

int32_t wr_spmm_fused_supported(int32_t D, const float *partials) {
    return (D >= 32 && D % 32 == 0 && partials != nullptr && (reinterpret_cast<uintptr_t>(partials) & 127) == 0) ? 1 : 0;
}

int32_t wr_spmm_csr_chunked_fused(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                  const int32_t *row_span, const int32_t *col, const float *val, const float *X, int32_t D,
                                  float *Y, float *acc, float *partials, const int8_t *row_mode, int32_t acc_from_x,
                                  float acc_scale, uint32_t *counters, void *stream_) {
    int32_t rc;
    if ((rc = check_table(X, n_rows, D, "X")) != WR_OK) return rc;
    if ((rc = check_table(Y, n_rows, D, "Y")) != WR_OK) return rc;
    WR_REQUIRE(chunk_ptr && chunk_row && row_span && col && val && partials && counters, WR_E_NULL,
               "chunked CSR arrays, row_span and counters must not be NULL");
    WR_REQUIRE(n_chunks >= n_rows && n_chunks < (int64_t(1) << 31), WR_E_SHAPE, "every row needs at least one chunk");
    WR_REQUIRE(X != Y, WR_E_SHAPE, "spmm: X and Y must not alias");
    WR_REQUIRE(wr_spmm_fused_supported(D, partials), WR_E_SHAPE,
               "fused combine: rows must be whole 128-byte lines (D %% 32 == 0, partials 128-byte aligned); D=%d", D);
    WR_REQUIRE(acc == nullptr || aligned16(acc), WR_E_ALIGN, "acc not 16-byte aligned");
    WR_REQUIRE(!acc_from_x || (acc != nullptr && acc != X), WR_E_SHAPE, "acc_from_x needs an acc buffer other than X");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const unsigned grid = (unsigned)((n_chunks + tpb - 1) / tpb);
    const float *acc_src = acc_from_x ? X : nullptr;
    const signed char *rm = reinterpret_cast<const signed char *>(row_mode);
#define WR_CALL_MF(T_, NV_, FULL_)                                                                                       \
    hipLaunchKernelGGL((spmm_chunk_kernel<T_, NV_, FULL_, true>), dim3(grid), dim3(kBlock), 0, stream, (int)n_chunks,      \
                       chunk_ptr, chunk_row, col, val, X, D, Y, acc, partials, rm, acc_src, acc_scale, row_span, counters)
    WR_DISPATCH_D(D, WR_CALL_MF);
#undef WR_CALL_MF
    WR_LAUNCH_CHECK("spmm_chunk_kernel (fused combine)");
    return WR_OK;
}

int32_t wr_axpy(float *y, const float *x, int64_t numel, float alpha, int32_t overwrite, void *stream_) {
    WR_REQUIRE(y && x, WR_E_NULL, "x/y must not be NULL");
    WR_REQUIRE(aligned16(y) && aligned16(x), WR_E_ALIGN, "x/y not 16-byte aligned");
    WR_REQUIRE(numel >= 0, WR_E_SHAPE, "numel < 0");
    if (numel == 0) return WR_OK;
    const int64_t n4 = numel / 4;
    const int tail = (int)(numel - n4 * 4);
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n4)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_),
                       reinterpret_cast<float4 *>(y), reinterpret_cast<const float4 *>(x), n4, alpha, (int)overwrite,
                       y + n4 * 4, x + n4 * 4, tail);
    WR_LAUNCH_CHECK("axpy_kernel");
    return WR_OK;
}

int32_t wr_embloss_sumsq(const float *user_tab, const float *item_tab, int32_t D, const int64_t *u, const int64_t *p,
                         const int64_t *n, int64_t B, float *sq3, void *workspace, int64_t workspace_bytes,
                         void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, 1, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, 1, D, "item_tab")) != WR_OK) return rc;
    WR_REQUIRE(u && p && n && sq3, WR_E_NULL, "index arrays / output must not be NULL");
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "B out of range");
    const int tpb = teams_per_block_for(D);
    const int nblk = (int)((B + tpb - 1) / tpb);
    WR_REQUIRE(workspace && workspace_bytes >= (int64_t)nblk * 12, WR_E_WORKSPACE, "embloss workspace %lld B < %lld B",
               (long long)workspace_bytes, (long long)nblk * 12);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    float *partials = reinterpret_cast<float *>(workspace);
#define WR_CALL_E(T_, NV_, FULL_)                                                                                       \
    hipLaunchKernelGGL((embloss_sumsq_kernel<T_, NV_, FULL_>), dim3(nblk), dim3(kBlock), 0, stream, user_tab, item_tab, D, \
                       u, p, n, (int)B, partials, nblk)
    WR_DISPATCH_D(D, WR_CALL_E);
#undef WR_CALL_E
    WR_LAUNCH_CHECK("embloss_sumsq_kernel");
    hipLaunchKernelGGL(embloss_finish_kernel, dim3(1), dim3(kBlock), 0, stream, partials, nblk, sq3);
    WR_LAUNCH_CHECK("embloss_finish_kernel");
    return WR_OK;
}

int64_t wr_lightgcn_loss_workspace_bytes(int64_t B) { return align_up(((B + 15) / 16 + 1) * 16, 256); }

int32_t wr_lightgcn_loss(const float *user_all, const float *item_all, const float *user_ego, const float *item_ego,
                         int64_t n_users, int64_t n_items, int32_t D, const int64_t *u, const int64_t *p, const int64_t *n,
                         int64_t B, float reg_weight, float *loss, float *sq3, void *workspace, int64_t workspace_bytes,
                         void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_all, n_users, D, "user_all")) != WR_OK) return rc;
    if ((rc = check_table(item_all, n_items, D, "item_all")) != WR_OK) return rc;
    if ((rc = check_table(user_ego, n_users, D, "user_ego")) != WR_OK) return rc;
    if ((rc = check_table(item_ego, n_items, D, "item_ego")) != WR_OK) return rc;
    WR_REQUIRE(u && p && n && loss && sq3, WR_E_NULL, "index arrays / outputs must not be NULL");
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "B out of range");
    const int tpb = teams_per_block_for(D);
    const int nblk = (int)((B + tpb - 1) / tpb);
    WR_REQUIRE(workspace && workspace_bytes >= (int64_t)nblk * 16, WR_E_WORKSPACE, "lightgcn loss workspace %lld B < %lld B",
               (long long)workspace_bytes, (long long)nblk * 16);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    float *partials = reinterpret_cast<float *>(workspace);
#define WR_CALL_LT(T_, NV_, FULL_)                                                                                         \
    hipLaunchKernelGGL((lightgcn_tail_fwd_kernel<T_, NV_, FULL_>), dim3(nblk), dim3(kBlock), 0, stream, user_all, item_all,  \
                       user_ego, item_ego, D, u, p, n, (int)B, partials, nblk, n_users, n_items)
    WR_DISPATCH_D(D, WR_CALL_LT);
#undef WR_CALL_LT
    WR_LAUNCH_CHECK("lightgcn_tail_fwd_kernel");
    hipLaunchKernelGGL(lightgcn_tail_finish_kernel, dim3(1), dim3(kBlock), 0, stream, partials, nblk, (float)B, reg_weight, loss,
                       sq3);
    WR_LAUNCH_CHECK("lightgcn_tail_finish_kernel");
    return WR_OK;
}

int32_t wr_embloss_grad(const float *user_tab, const float *item_tab, int32_t D, const int32_t *tu, const int32_t *oc_item,
                        const int32_t *oc_src, int64_t B, const float *sq3, float reg_weight, float *grad_user,
                        float *grad_item, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, 1, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, 1, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(grad_user, 1, D, "grad_user")) != WR_OK) return rc;
    if ((rc = check_table(grad_item, 1, D, "grad_item")) != WR_OK) return rc;
    WR_REQUIRE(tu && oc_item && oc_src && sq3, WR_E_NULL, "embloss_grad: NULL argument");
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "B out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tpb = teams_per_block_for(D);
    const float rob = reg_weight / (float)B;
    const unsigned ub = (unsigned)((B + tpb - 1) / tpb), ib = (unsigned)((2 * B + tpb - 1) / tpb);
#define WR_CALL_EG(T_, NV_, FULL_)                                                                                        \
    hipLaunchKernelGGL((embloss_grad_kernel<T_, NV_, FULL_>), dim3(ub + ib), dim3(kBlock), 0, stream, user_tab, grad_user, \
                       tu, (int)B, (int)ub, item_tab, grad_item, oc_item, oc_src, (int)(2 * B), D, sq3, rob)
    WR_DISPATCH_D(D, WR_CALL_EG);
#undef WR_CALL_EG
    WR_LAUNCH_CHECK("embloss_grad_kernel");
    return WR_OK;
}

}  // extern "C"
