// wr_eval.hip — full-ranking evaluation without the [n_eval, n_items] host matrix.
//
// Reference: BaseRunner.interface + evaluate_method (src/helpers/BaseRunner.py:218-258, 50-92) with
// BPRMF/LightGCN.full_predict (src/models/general/BPRMF.py:82-91): scores = U[user] @ I^T for every item, the user's
// train/dev/test items set to -inf (BaseRunner.py:246-255), rank of the ground-truth = its position in the descending
// order.  Only the rank is needed by every metric (HR/NDCG/RECALL/PRECISION@k), and
//     rank_i = 1 + #{ j not masked for user_i : score(i, j) > score(i, target_i) }.
// This is the one GEMM-shaped piece of the path, so it runs on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate = a k-ordered fmaf chain, MI355X_MICROARCH.md "Matrix cores"), one 32x32 score tile per wave per item tile,
// compared against the target score (computed with the same k-ordered chain, so the target's own column can never count)
// and counted in registers.  A workgroup = 4 waves = 128 evaluation rows sharing each 32-item tile through LDS.
#include "wr_common.h"

namespace wr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kEvalRows = 128;   // evaluation rows per workgroup (32 per wave)
constexpr int kEvalChunk = 2048; // items per workgroup along grid.y

// score of the ground-truth item, as the k-ordered fmaf chain the MFMA accumulates
__global__ __launch_bounds__(kBlock) void eval_target_kernel(const float *__restrict__ U, const float *__restrict__ I, int D,
                                                              const int64_t *__restrict__ eu, const int64_t *__restrict__ et,
                                                              int64_t n, float *__restrict__ tscore) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *a = U + eu[i] * (int64_t)D, *b = I + et[i] * (int64_t)D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(a[k], b[k], s);
    tscore[i] = s;
}

__global__ __launch_bounds__(kBlock) void eval_rank_kernel(const float *__restrict__ U, const float *__restrict__ I, int D,
                                                            int64_t n_items, const int64_t *__restrict__ eu,
                                                            const float *__restrict__ tscore, int64_t n,
                                                            const int64_t *__restrict__ mask_ptr, const int *__restrict__ mask_idx,
                                                            int *__restrict__ rank_cnt) {
    extern __shared__ float lds[];
    const int ldw = D + 1;                       // padded row: conflict-free column reads
    float *ue = lds;                             // [kEvalRows][ldw]
    float *it = lds + kEvalRows * ldw;           // [32][ldw]
    unsigned *rowmask = reinterpret_cast<unsigned *>(it + 32 * ldw);   // [kEvalRows]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t e0 = (int64_t)blockIdx.x * kEvalRows;
    const int64_t c0 = (int64_t)blockIdx.y * kEvalChunk;
    // stage the evaluation users' rows
    for (int idx = threadIdx.x; idx < kEvalRows * D; idx += kBlock) {
        const int r = idx / D, k = idx - r * D;
        const int64_t e = e0 + r;
        ue[r * ldw + k] = (e < n) ? U[eu[e] * (int64_t)D + k] : 0.f;
    }
    // per evaluation row (threads 0..127): cursor into the user's ascending mask list, positioned at this chunk
    int64_t cur = 0, cend = 0;
    if (threadIdx.x < kEvalRows && mask_ptr != nullptr && e0 + threadIdx.x < n) {
        const int64_t uu = eu[e0 + threadIdx.x];
        int64_t lo = mask_ptr[uu], hi = mask_ptr[uu + 1];
        cend = hi;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)mask_idx[mid] < c0) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    // this lane's 16 accumulator rows inside its wave's 32-row slab and their target scores
    float trow[16];
    int cnt[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int64_t e = e0 + wave * 32 + row;
        trow[reg] = (e < n) ? tscore[e] : 3.4e38f;   // rows past the end never count
        cnt[reg] = 0;
    }
    const int col = lane & 31, half = lane >> 5;
    const float *arow = ue + (wave * 32 + col) * ldw + half;   // A[i = lane&31][k = lane>>5]
    const float *brow = it + col * ldw + half;                  // B[k = lane>>5][j = lane&31]
    for (int64_t j0 = c0; j0 < c0 + kEvalChunk && j0 < n_items; j0 += 32) {
        __syncthreads();                                        // previous tile fully consumed (and ue staged)
        for (int idx = threadIdx.x; idx < 32 * D; idx += kBlock) {
            const int r = idx / D, k = idx - r * D;
            it[r * ldw + k] = (j0 + r < n_items) ? I[(j0 + r) * (int64_t)D + k] : 0.f;
        }
        if (threadIdx.x < kEvalRows) {                          // which of the tile's 32 items are masked for this row
            unsigned m = 0;
            while (cur < cend && (int64_t)mask_idx[cur] < j0 + 32) {
                if ((int64_t)mask_idx[cur] >= j0) m |= 1u << (unsigned)(mask_idx[cur] - j0);
                ++cur;
            }
            rowmask[threadIdx.x] = m;
        }
        __syncthreads();
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k0 = 0; k0 < D; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k0], brow[k0], acc, 0, 0, 0);
        const bool col_ok = j0 + col < n_items;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const bool masked = (rowmask[wave * 32 + row] >> col) & 1u;
            cnt[reg] += (col_ok && !masked && acc[reg] > trow[reg]) ? 1 : 0;
        }
    }
    // sum over the 32 item columns (lanes of one half), one integer atomic per row and workgroup
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        int v = cnt[reg];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int64_t e = e0 + wave * 32 + row;
        if (col == 0 && e < n && v) atomicAdd(&rank_cnt[e], v);
    }
}

// Same computation for D = 2*KS <= 64 with the A operand (this wave's 32 evaluation rows) held in KS registers for the
// whole chunk, item tiles of 64 rows double-buffered in LDS (the next tile's global loads are in flight while the
// current one feeds the matrix cores; one barrier per tile instead of two per 32 items), and one LDS read per MFMA.
#ifndef WR_EVAL_TILE
#define WR_EVAL_TILE 64
#endif
constexpr int kEvalTile = WR_EVAL_TILE;

// 3 workgroups per CU (<= 168 VGPRs, no spill): A/B on MI355X 2 / 3 / 4 per CU = 101 / 111 / 86 TFLOP/s at 100K x 100K x 64.
template <int KS>
__global__ __launch_bounds__(kBlock, 3) void eval_rank_kernel_rega(const float *__restrict__ U, const float *__restrict__ I,
                                                                 int64_t n_items, const int64_t *__restrict__ eu,
                                                                 const float *__restrict__ tscore, int64_t n,
                                                                 const int64_t *__restrict__ mask_ptr,
                                                                 const int *__restrict__ mask_idx, int *__restrict__ rank_cnt) {
    constexpr int D = 2 * KS, LDW = D + 1, D4 = D / 4;
    constexpr int NLOAD = (kEvalTile * D4 + kBlock - 1) / kBlock;        // float4 loads per thread and tile
    __shared__ float it[2][kEvalTile * LDW];
    __shared__ unsigned rowmask[2][kEvalTile / 32][kEvalRows];
    __shared__ unsigned anymask[2][kBlock / 64];   // does any row of slab w have a masked item in the tile?
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * kEvalRows;
    const int64_t c0 = (int64_t)blockIdx.y * kEvalChunk;
    const int64_t c1 = (c0 + kEvalChunk < n_items) ? c0 + kEvalChunk : n_items;
    // A[i = lane&31][k = 2s + (lane>>5)] of this wave's slab
    float a[KS];
    {
        const int64_t e = e0 + wave * 32 + col;
        const float *urow = U + ((e < n) ? eu[e] : 0) * (int64_t)D + half;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = (e < n) ? urow[2 * s] : 0.f;
    }
    int64_t cur = 0, cend = 0;
    if (threadIdx.x < kEvalRows && mask_ptr != nullptr && e0 + threadIdx.x < n) {
        const int64_t uu = eu[e0 + threadIdx.x];
        int64_t lo = mask_ptr[uu], hi = mask_ptr[uu + 1];
        cend = hi;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)mask_idx[mid] < c0) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    // the next masked item of this row waits in a register: a tile without masked items (almost all of them) costs no
    // memory access on the way to the barrier
    int nxt = (cur < cend) ? mask_idx[cur] : 0x7fffffff;
    float trow[16];
    int cnt[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int64_t e = e0 + wave * 32 + row;
        trow[reg] = (e < n) ? tscore[e] : 3.4e38f;
        cnt[reg] = 0;
    }
    float4 stage[NLOAD];
    auto fetch = [&](int64_t j0) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int f = threadIdx.x + i * kBlock;
            const int r = f / D4, k4 = f - r * D4;
            stage[i] = (f < kEvalTile * D4 && j0 + r < n_items)
                           ? reinterpret_cast<const float4 *>(I + (j0 + r) * (int64_t)D)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto deposit = [&](int buf, int64_t j0) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int f = threadIdx.x + i * kBlock;
            if (f < kEvalTile * D4) {
                const int r = f / D4, k4 = f - r * D4;
                float *dst = &it[buf][r * LDW + 4 * k4];
                dst[0] = stage[i].x; dst[1] = stage[i].y; dst[2] = stage[i].z; dst[3] = stage[i].w;
            }
        }
        if (threadIdx.x < kEvalRows) {           // which of the tile's items are masked for this row
            unsigned m[kEvalTile / 32];
#pragma unroll
            for (int c = 0; c < kEvalTile / 32; ++c) m[c] = 0;
            while ((int64_t)nxt < j0 + kEvalTile) {
                const int64_t d = (int64_t)nxt - j0;
                if (d >= 0) m[d >> 5] |= 1u << (unsigned)(d & 31);
                ++cur;
                nxt = (cur < cend) ? mask_idx[cur] : 0x7fffffff;
            }
            unsigned any = 0;
#pragma unroll
            for (int c = 0; c < kEvalTile / 32; ++c) {
                rowmask[buf][c][threadIdx.x] = m[c];
                any |= m[c];
            }
            // rows 0..127 sit in waves 0 and 1: slab w = rows 32w..32w+31 = lanes 32(w&1).. of wave w>>1
            const unsigned long long bal = __ballot(any != 0);
            if (lane == 0) {
                anymask[buf][2 * wave] = (unsigned)(bal & 0xffffffffull) != 0;
                anymask[buf][2 * wave + 1] = (unsigned)(bal >> 32) != 0;
            }
        }
    };
    fetch(c0);
    deposit(0, c0);
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = c0; j0 < c1; j0 += kEvalTile, buf ^= 1) {
        const bool more = j0 + kEvalTile < c1;
        if (more) fetch(j0 + kEvalTile);                        // global loads fly while the matrix cores work
        {   // the tile's 32-item column blocks as independent accumulator chains: a dependent MFMA waits for its
            // predecessor's result, an independent one issues right behind it
            constexpr int C = kEvalTile / 32;
            const float *bcol = &it[buf][col * LDW + half];          // B[k = 2s + (lane>>5)][j = lane&31] of block 0
            f32x16 acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bcol[c * 32 * LDW + 2 * s], acc[c], 0, 0, 0);
            }
            if (anymask[buf][wave] == 0 && j0 + kEvalTile <= n_items) {   // wave-uniform: nothing masked, tile inside the table
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
#pragma unroll
                    for (int c = 0; c < C; ++c) cnt[reg] += acc[c][reg] > trow[reg] ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const bool ok = j0 + c * 32 + col < n_items;
                        const bool m = (rowmask[buf][c][wave * 32 + row] >> col) & 1u;
                        cnt[reg] += (ok && !m && acc[c][reg] > trow[reg]) ? 1 : 0;
                    }
                }
            }
        }
        if (more) deposit(buf ^ 1, j0 + kEvalTile);             // the other buffer was last read one barrier ago
        __syncthreads();
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        int v = cnt[reg];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int64_t e = e0 + wave * 32 + row;
        if (col == 0 && e < n && v) atomicAdd(&rank_cnt[e], v);
    }
}

__global__ __launch_bounds__(kBlock) void eval_finish_kernel(int *__restrict__ rank, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) rank[i] += 1;
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_rank_eval(const float *user_mat, int64_t n_user_rows, const float *item_tab, int64_t n_items, int32_t D,
                     const int64_t *eval_user, const int64_t *eval_target, int64_t n, const int64_t *mask_ptr,
                     const int32_t *mask_idx, int32_t *rank, float *target_score, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_mat, n_user_rows, D, "user_mat")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    WR_REQUIRE(eval_user && eval_target && rank && target_score, WR_E_NULL, "rank_eval: NULL argument");
    WR_REQUIRE((mask_ptr == nullptr) == (mask_idx == nullptr), WR_E_NULL, "rank_eval: mask_ptr and mask_idx go together");
    WR_REQUIRE(n >= 0 && n < (int64_t(1) << 31), WR_E_SHAPE, "rank_eval: n out of range");
    // D outside {8,16,32,64}: the LDS-operand kernel stages (kEvalRows + 32) rows of D + 1 floats + kEvalRows counters; a
    // workgroup gets at most 160 KiB (163,840 B) on gfx950 -> D <= 252
    const size_t lds_generic = ((size_t)(kEvalRows + 32) * (D + 1) + kEvalRows) * 4;
    WR_REQUIRE(D == 64 || D == 32 || D == 16 || D == 8 || lds_generic <= 160 * 1024, WR_E_RANGE,
               "rank_eval supports D <= 252 (LDS staging: %lld B needed, 163840 B per workgroup); got D=%d",
               (long long)lds_generic, D);
    if (n == 0) return WR_OK;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    WR_HIP(hipMemsetAsync(rank, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(eval_target_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, user_mat, item_tab,
                       D, eval_user, eval_target, n, target_score);
    WR_LAUNCH_CHECK("eval_target_kernel");
    const dim3 grid((unsigned)((n + kEvalRows - 1) / kEvalRows), (unsigned)((n_items + kEvalChunk - 1) / kEvalChunk));
    if (D == 64 || D == 32 || D == 16 || D == 8) {   // A operand in registers, double-buffered item tiles
#define WR_EVAL_REGA(KS_)                                                                                             \
    hipLaunchKernelGGL(eval_rank_kernel_rega<KS_>, grid, dim3(kBlock), 0, stream, user_mat, item_tab, n_items, eval_user, \
                       target_score, n, mask_ptr, mask_idx, rank)
        if (D == 64) WR_EVAL_REGA(32);
        else if (D == 32) WR_EVAL_REGA(16);
        else if (D == 16) WR_EVAL_REGA(8);
        else WR_EVAL_REGA(4);
#undef WR_EVAL_REGA
    } else {
        const size_t lds = lds_generic;
        if (lds > 64 * 1024)
            WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(eval_rank_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(eval_rank_kernel, grid, dim3(kBlock), lds, stream, user_mat, item_tab, D, n_items, eval_user,
                           target_score, n, mask_ptr, mask_idx, rank);
    }
    WR_LAUNCH_CHECK("eval_rank_kernel");
    hipLaunchKernelGGL(eval_finish_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rank, n);
    WR_LAUNCH_CHECK("eval_finish_kernel");
    return WR_OK;
}

}  // extern "C"
