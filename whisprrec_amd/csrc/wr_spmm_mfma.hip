// wr_spmm_mfma.hip — the dense part of LightGCN's normalised-adjacency product on the matrix cores (gfx950).
//
// Reference: ego = torch.sparse.mm(norm_adj, ego) (src/models/general/LightGCN.py:139; norm_adj is a DENSE N x N fp32
// matrix at runtime, :114-121).  The bipartite rating graph is sparse on average (1-3 %) but not at the head of the item
// popularity: on an ml-1m-shaped graph the 128 most rated items are rated by 12-40 % of the users and hold a fifth of
// all non-zeros.  The block  (all users) x (head items)  of the adjacency — and its transpose — is therefore kept DENSE
// (zeros stored) and multiplied with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an exact fma chain in k order, so the
// 1e-5 parity bar holds; it issues at the fp32 vector rate, what it buys is 256 B of L2 gather traffic less per non-zero
// and the matrix pipe running BESIDE the gather kernel's memory pipe); everything else stays on the CSR gather kernels
// (wr_rows.hip).  One wave = one 32-row output tile x D columns; the four waves of a workgroup split the tile's K range
// and are summed through LDS in wave order; a K range too long for one workgroup (head-item rows: K = all users) is split
// over gridDim.y workgroups whose partial tiles are added in split order by spmm_dense_combine_kernel.  No atomics.
//
// Operand layout (MI355X fragment layout, cdna_hip_programming.md section 3): A operand lane l = A[i = l % 32][k = l / 32],
// B operand lane l = B[k = l / 32][j = l % 32], accumulator register v of lane l = C[(v & 3) + 8 * (v >> 2) + 4 * (l / 32)][l % 32].
// The dense block is stored per tile as A_T[tile][k][32 rows], so a wave's A load is two 128-B runs.
#include "wr_common.h"

namespace wr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDenseWaves = kBlock / 64;   // waves per workgroup = K slices per split
constexpr int kDenseUnroll = 8;            // k-pairs per stage: the loads of a stage are issued together
constexpr int kDenseMaxKw = 256;           // longest K slice of one wave (its column ids are kept in 4 registers)

// One stage = kDenseUnroll k-pairs: per pair one A value and NJ B values per lane.  The column ids of the wave's whole K
// slice sit in registers (ccache, lane l of register j = column kb + 64 j + l), so a stage's loads depend on nothing
// loaded inside the loop and the NEXT stage's loads are issued before the current stage's MFMAs (two register stages).
template <int NJ>
struct DenseStage {
    float a[kDenseUnroll];
    float b[kDenseUnroll][NJ];
};

template <int NJ>
__device__ __forceinline__ void dense_load_stage(DenseStage<NJ> &st, const float *__restrict__ At, const float *__restrict__ X,
                                                 const int (&ccache)[kDenseMaxKw / 64], int k, int kb, int half, int col) {
    constexpr int D = 32 * NJ;
#pragma unroll
    for (int q = 0; q < kDenseUnroll; ++q) {
        const int kk = k + 2 * q + half;
        const int rel = kk - kb;
        int c = __shfl(ccache[0], rel & 63, 64);
#pragma unroll
        for (int j = 1; j < kDenseMaxKw / 64; ++j) {
            const int cj = __shfl(ccache[j], rel & 63, 64);
            c = (rel >> 6) == j ? cj : c;
        }
        st.a[q] = At[(int64_t)kk * 32];
#pragma unroll
        // a padding column (cols[k] < 0) contributes exact zeros whatever X holds: 0 * inf would be a NaN in every row of the tile
        for (int j = 0; j < NJ; ++j) st.b[q][j] = c >= 0 ? X[(int64_t)c * D + 32 * j + col] : 0.f;
    }
}

// One launch carries up to two groups of tiles (LightGCN: the user-row tiles, K = head items, one split; and the
// head-item-row tiles, K = all users, many splits): workgroups [0, g0.n_tiles * g0.n_splits) belong to group 0.
struct DenseGroup {
    const float *A_T;
    const int *cols, *rows;
    float *partials;
    int K_pad, k_per_split, n_tiles, n_splits;
};

template <int NJ>   // D = 32 * NJ
__global__ __launch_bounds__(kBlock) void spmm_dense_tiles_kernel(DenseGroup g0, DenseGroup g1, const float *__restrict__ X,
                                                                   float *__restrict__ Y) {
    constexpr int D = 32 * NJ;
    extern __shared__ float red[];   // [kDenseWaves][32][D]
    const int n0 = g0.n_tiles * g0.n_splits;
    const bool first = (int)blockIdx.x < n0;
    const DenseGroup &g = first ? g0 : g1;
    const int job = first ? (int)blockIdx.x : (int)blockIdx.x - n0;
    const int n_splits = g.n_splits;
    const int tile = job / n_splits, split = job - tile * n_splits;
    const float *__restrict__ A_T = g.A_T;
    const int *__restrict__ cols = g.cols;
    const int *__restrict__ rows = g.rows;
    float *__restrict__ partials = g.partials;
    const int K_pad = g.K_pad, k_per_split = g.k_per_split;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int kw = k_per_split / kDenseWaves;                 // multiple of 2 * kDenseUnroll, <= kDenseMaxKw (host-checked)
    const int kb = split * k_per_split + wave * kw, ke = kb + kw;
    const float *At = A_T + ((int64_t)tile * K_pad) * 32 + col;
    int ccache[kDenseMaxKw / 64];
#pragma unroll
    for (int j = 0; j < kDenseMaxKw / 64; ++j) ccache[j] = (64 * j + lane < kw) ? cols[kb + 64 * j + lane] : 0;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    DenseStage<NJ> s0, s1;
    dense_load_stage<NJ>(s0, At, X, ccache, kb, kb, half, col);
    for (int k = kb; k < ke; k += 4 * kDenseUnroll) {
        const bool more1 = k + 2 * kDenseUnroll < ke;
        if (more1) dense_load_stage<NJ>(s1, At, X, ccache, k + 2 * kDenseUnroll, kb, half, col);
#pragma unroll
        for (int q = 0; q < kDenseUnroll; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0.a[q], s0.b[q][j], acc[j], 0, 0, 0);
        if (!more1) break;
        if (k + 4 * kDenseUnroll < ke) dense_load_stage<NJ>(s0, At, X, ccache, k + 4 * kDenseUnroll, kb, half, col);
#pragma unroll
        for (int q = 0; q < kDenseUnroll; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1.a[q], s1.b[q][j], acc[j], 0, 0, 0);
    }
    // the four K slices, added in wave order
    float *mine = red + (int64_t)wave * 32 * D;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * half;
            mine[row * D + 32 * j + col] = acc[j][v];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * D; e += kBlock) {
        float s = red[e];
#pragma unroll
        for (int w = 1; w < kDenseWaves; ++w) s += red[w * 32 * D + e];
        const int r = e / D, d = e - r * D;
        if (n_splits == 1) {
            const int node = rows[tile * 32 + r];
            if (node >= 0) Y[(int64_t)node * D + d] = s;
        } else {
            partials[(((int64_t)tile * n_splits + split) * 32 + r) * D + d] = s;
        }
    }
}

// rows whose K range was split over several workgroups: add the split tiles in split order, write the row, add it to
// the layer sum (`acc`, LightGCN.py:142-143's running mean numerator)
__global__ __launch_bounds__(kBlock) void spmm_dense_combine_kernel(const float *__restrict__ partials, int n_tiles, int n_splits,
                                                                     int D, const int *__restrict__ rows, float *__restrict__ Y,
                                                                     float *__restrict__ acc) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t per_tile = 32 * (int64_t)D;
    if (e >= n_tiles * per_tile) return;
    const int tile = (int)(e / per_tile);
    const int64_t in = e - tile * per_tile;
    const int r = (int)(in / D), d = (int)(in - (int64_t)r * D);
    const int node = rows[tile * 32 + r];
    if (node < 0) return;
    // eight split tiles requested per trip, added in split order
    const float *p0 = partials + ((int64_t)tile * n_splits) * per_tile + in;
    float s = 0.f;
    for (int sp = 0; sp < n_splits; sp += 8) {
        float v[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) v[f] = sp + f < n_splits ? p0[(int64_t)(sp + f) * per_tile] : 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f)
            if (sp + f < n_splits) s += v[f];
    }
    Y[(int64_t)node * D + d] = s;
    if (acc != nullptr) acc[(int64_t)node * D + d] += s;
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_spmm_dense_partials_bytes(int64_t n_tiles, int64_t K_pad, int64_t k_per_split, int32_t D) {
    if (n_tiles <= 0 || K_pad <= 0 || k_per_split <= 0 || K_pad % k_per_split != 0) return WR_E_SHAPE;
    const int64_t n_splits = K_pad / k_per_split;
    return n_splits > 1 ? n_tiles * n_splits * 32 * (int64_t)D * 4 : 0;
}

static int32_t check_group(const wr_dense_group *g, int32_t D, const char *what) {
    WR_REQUIRE(g->A_T && g->cols && g->rows, WR_E_NULL, "dense tiles (%s): NULL array", what);
    WR_REQUIRE(g->n_tiles > 0 && g->n_tiles < 65536 && g->K_pad > 0 && g->k_per_split > 0 && g->K_pad % g->k_per_split == 0 &&
                   g->k_per_split % (kDenseWaves * 2 * kDenseUnroll) == 0 && g->k_per_split <= kDenseWaves * kDenseMaxKw &&
                   g->K_pad / g->k_per_split < 65536,
               WR_E_SHAPE, "dense tiles (%s): K_pad=%lld must be a multiple of k_per_split=%lld, itself a multiple of %d and <= %d",
               what, (long long)g->K_pad, (long long)g->k_per_split, kDenseWaves * 2 * kDenseUnroll, kDenseWaves * kDenseMaxKw);
    const int64_t n_splits = g->K_pad / g->k_per_split;
    WR_REQUIRE(n_splits == 1 || (g->partials != nullptr && aligned16(g->partials)), WR_E_NULL,
               "dense tiles (%s): split K needs partials", what);
    return WR_OK;
}

static inline DenseGroup to_dev(const wr_dense_group *g) {
    if (g == nullptr) return DenseGroup{nullptr, nullptr, nullptr, nullptr, 0, 1, 0, 1};
    return DenseGroup{g->A_T, g->cols, g->rows, g->partials, (int)g->K_pad, (int)g->k_per_split, (int)g->n_tiles,
                      (int)(g->K_pad / g->k_per_split)};
}

int32_t wr_spmm_dense_tiles(const wr_dense_group *unsplit, const wr_dense_group *split, const float *X, int64_t n_nodes,
                            int32_t D, float *Y, float *acc, void *stream_) {
    int32_t rc;
    if ((rc = check_table(X, n_nodes, D, "X")) != WR_OK) return rc;
    if ((rc = check_table(Y, n_nodes, D, "Y")) != WR_OK) return rc;
    WR_REQUIRE(unsplit != nullptr || split != nullptr, WR_E_NULL, "dense tiles: no group given");
    WR_REQUIRE(D == 32 || D == 64 || D == 96 || D == 128, WR_E_SHAPE, "dense tiles: D must be 32, 64, 96 or 128 (got %d)", D);
    WR_REQUIRE(X != Y, WR_E_SHAPE, "spmm: X and Y must not alias");
    if (unsplit != nullptr) {
        if ((rc = check_group(unsplit, D, "unsplit")) != WR_OK) return rc;
        WR_REQUIRE(unsplit->K_pad == unsplit->k_per_split, WR_E_SHAPE, "dense tiles: the first group must have one split");
    }
    if (split != nullptr) {
        if ((rc = check_group(split, D, "split")) != WR_OK) return rc;
        WR_REQUIRE(split->K_pad > split->k_per_split, WR_E_SHAPE, "dense tiles: the second group must have several splits");
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const size_t lds = (size_t)kDenseWaves * 32 * D * 4;
    const DenseGroup g0 = to_dev(unsplit), g1 = to_dev(split);
    const dim3 grid((unsigned)(g0.n_tiles * g0.n_splits + g1.n_tiles * g1.n_splits));
#define WR_DENSE(NJ_)                                                                                                   \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_dense_tiles_kernel<NJ_>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        hipLaunchKernelGGL((spmm_dense_tiles_kernel<NJ_>), grid, dim3(kBlock), lds, stream, g0, g1, X, Y);               \
    } while (0)
    if (D == 32) WR_DENSE(1);
    else if (D == 64) WR_DENSE(2);
    else if (D == 96) WR_DENSE(3);
    else WR_DENSE(4);
#undef WR_DENSE
    WR_LAUNCH_CHECK("spmm_dense_tiles_kernel");
    if (split != nullptr) {
        const int64_t total = g1.n_tiles * 32 * (int64_t)D;
        hipLaunchKernelGGL(spmm_dense_combine_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           g1.partials, g1.n_tiles, g1.n_splits, D, g1.rows, Y, acc);
        WR_LAUNCH_CHECK("spmm_dense_combine_kernel");
    }
    return WR_OK;
}

}  // extern "C"
