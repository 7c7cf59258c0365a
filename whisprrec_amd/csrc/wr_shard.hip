// wr_shard.hip — index work of the row-sharded step (multi-GPU; whisprrec_amd/sharded.py): which item rows of a batch live on
// which rank, and where a requester keeps the rows it receives.  The reference is single-device: this is new
// design.  Semantics to keep: ONE BaseRunner.fit iteration (src/helpers/BaseRunner.py:196-199) over the union of the ranks'
// batches, negatives drawn from ALL items (src/models/BaseModel.py:168,174) — so a rank's batch touches item rows of every
// rank, and each step exchanges rows (owner -> requester) and gradient rows (requester -> owner).
//
// Layout: item row i lives on rank i % G at local row i / G (cyclic); a rank owns the users of its triplets.
// wr_shard_route — one workgroup per step, index work only:
//   * a LOCAL item (i % G == rank) keeps its local row as its "virtual" id: the step kernels read and rewrite the shard in place;
//   * the step's DISTINCT remote items get slots 0, 1, ... in ascending (owner, local row) order — a bitmap of the routed
//     keys owner * M + row in LDS (passes of 2^19 keys), prefix popcounts, slot = number of set bits below the key — and the
//     virtual id nL + slot: row `slot` of the buffer the owners' rows are received into, right behind the shard's own rows;
//   * per owner, the list of requested local rows (ascending) is written into a padded send buffer [owner][step][C] with
//     its length: the index exchange of a whole chunk of steps is then two fixed-size all-to-alls, no host-side sizes.
// wr_shard_pack — after that exchange, per step the rows this rank must SERVE, requester by requester, as one list.
#include "wr_common.h"

namespace wr {

constexpr int kSrThreads = 1024;
constexpr unsigned kSrPassBits = 19;                  // routed keys per pass: 64 KiB of bitmap + 64 KiB of per-word prefixes
constexpr int kSrWords = 1 << (kSrPassBits - 5);
constexpr int kSrMaxWorld = 64;

template <typename F>
__device__ __forceinline__ void sr_scan(const int *__restrict__ a, int cnt, F f) {
    for (int j = threadIdx.x; j < cnt; j += kSrThreads) f(a[j], j);
}

__global__ __launch_bounds__(kSrThreads) void shard_route_kernel(const int *__restrict__ u, const int *__restrict__ p,
                                                                  const int *__restrict__ n, int64_t n_total, int B, int nb, int G,
                                                                  int rank, int n_users, int n_items, int M, int nL, int C,
                                                                  int *__restrict__ vu, int *__restrict__ vp, int *__restrict__ vn,
                                                                  int *__restrict__ send_rows, int *__restrict__ send_cnt,
                                                                  int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned sr_lds[];
    unsigned *bm = sr_lds;                                       // kSrWords
    int *pre = reinterpret_cast<int *>(sr_lds + kSrWords);       // kSrWords: set bits in the words before (inside the pass)
    int *ownoff = pre + kSrWords, *owncnt = ownoff + kSrMaxWorld, *wave_tot = owncnt + kSrMaxWorld;
    int &pass_total = wave_tot[kSrThreads / 64];
    const int k = (int)blockIdx.x;
    const int64_t base = (int64_t)k * B;
    const int Bb = (int)((base + B <= n_total) ? B : (n_total - base));
    for (int o = threadIdx.x; o < kSrMaxWorld; o += kSrThreads) {
        ownoff[o] = 0;
        owncnt[o] = 0;
    }
    // users: every triplet of this rank's batches must belong to a user it owns
    sr_scan(u + base, Bb, [&](int uu, int t) {
        if ((unsigned)uu >= (unsigned)n_users || uu % G != rank) {
            err[0] = 1;
            uu = rank;
        }
        vu[base + t] = uu / G;
    });
    const int64_t nkeys = (int64_t)G * M;
    int slot_base = 0;
    for (int64_t k0 = 0; k0 < nkeys; k0 += (int64_t(1) << kSrPassBits)) {
        __syncthreads();
        for (int i = threadIdx.x; i < kSrWords; i += kSrThreads) bm[i] = 0u;
        __syncthreads();
        auto mark = [&](int i, int) {
            if ((unsigned)i >= (unsigned)n_items) {
                err[0] = 1;
                return;
            }
            const int o = i % G;
            if (o == rank) return;
            const int64_t key = (int64_t)o * M + i / G;
            if (key < k0 || key >= k0 + (int64_t(1) << kSrPassBits)) return;
            const unsigned bit = (unsigned)(key - k0);
            atomicOr(&bm[bit >> 5], 1u << (bit & 31u));
        };
        sr_scan(p + base, Bb, mark);
        sr_scan(n + base, Bb, mark);
        __syncthreads();
        // exclusive prefix of the words' popcounts: thread t owns words [16 t, 16 t + 16)
        constexpr int kOwn = kSrWords / kSrThreads;
        const int w0 = (int)threadIdx.x * kOwn;
        int local = 0;
#pragma unroll
        for (int j = 0; j < kOwn; ++j) local += __popc(bm[w0 + j]);
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
            pre[w0 + j] = run;
            run += __popc(bm[w0 + j]);
        }
        if (threadIdx.x == kSrThreads - 1) pass_total = run;
        __syncthreads();
        // an owner's first slot: the set bits below its first key (M is a multiple of 32: a word belongs to one owner)
        for (int o = threadIdx.x; o < G; o += kSrThreads) {
            const int64_t first = (int64_t)o * M;
            if (first >= k0 && first < k0 + (int64_t(1) << kSrPassBits)) ownoff[o] = slot_base + pre[(first - k0) >> 5];
        }
        __syncthreads();
        // virtual ids of the occurrences
        auto assign = [&](int *__restrict__ out, const int *__restrict__ ids) {
            sr_scan(ids + base, Bb, [&](int i, int t) {
                if ((unsigned)i >= (unsigned)n_items) {
                    if (k0 == 0) out[base + t] = 0;
                    return;
                }
                const int o = i % G, row = i / G;
                if (o == rank) {
                    if (k0 == 0) out[base + t] = row;
                    return;
                }
                const int64_t key = (int64_t)o * M + row;
                if (key < k0 || key >= k0 + (int64_t(1) << kSrPassBits)) return;
                const unsigned bit = (unsigned)(key - k0), w = bit >> 5;
                out[base + t] = nL + slot_base + pre[w] + __popc(bm[w] & ((1u << (bit & 31u)) - 1u));
            });
        };
        assign(vp, p);
        assign(vn, n);
        // the request lists: thread per word, one entry per set bit
        for (int w = threadIdx.x; w < kSrWords; w += kSrThreads) {
            unsigned m = bm[w];
            if (m == 0u) continue;
            const int64_t key0 = k0 + (int64_t)w * 32;
            const int o = (int)(key0 / M);
            const int row0 = (int)(key0 - (int64_t)o * M);
            int idx = slot_base + pre[w] - ownoff[o];
            atomicAdd(&owncnt[o], __popc(m));
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                if (idx < C) send_rows[((int64_t)o * nb + k) * C + idx] = row0 + bit;
                else err[1] = 1;
                ++idx;
            }
        }
        __syncthreads();
        slot_base += pass_total;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < G; o += kSrThreads) send_cnt[(int64_t)o * nb + k] = owncnt[o];
}

// per step k: serve_rows[k][...] = the rows requested by rank 0, then by rank 1, ... (int64: what wr_gather_rows and
// wr_scatter_add_rows take), serve_off[k][s] = where requester s's rows start, serve_off[k][G] = their number
__global__ __launch_bounds__(256) void shard_pack_kernel(const int *__restrict__ recv_rows, const int *__restrict__ recv_cnt, int nb,
                                                          int G, int C, int64_t *__restrict__ serve_rows, int64_t stride,
                                                          int *__restrict__ serve_off, int n_local_rows, int *__restrict__ err) {
    __shared__ int off[kSrMaxWorld + 1];
    const int k = (int)blockIdx.x;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int s = 0; s < G; ++s) {
            off[s] = run;
            run += min(max(recv_cnt[(int64_t)s * nb + k], 0), C);
        }
        off[G] = run;
    }
    __syncthreads();
    for (int s = threadIdx.x; s <= G; s += 256) serve_off[(int64_t)k * (G + 1) + s] = off[s];
    for (int s = 0; s < G; ++s) {
        const int c = off[s + 1] - off[s];
        const int *src = recv_rows + ((int64_t)s * nb + k) * C;
        for (int j = threadIdx.x; j < c; j += 256) {
            int row = src[j];
            if ((unsigned)row >= (unsigned)n_local_rows) {      // a peer asked for a row this shard does not have
                err[0] = 1;
                row = 0;
            }
            serve_rows[(int64_t)k * stride + off[s] + j] = row;
        }
    }
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_shard_route(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size, int32_t world,
                       int32_t rank, int64_t n_users, int64_t n_items, int64_t rows_per_owner, int64_t n_local_items,
                       int64_t list_cap, int32_t *vu, int32_t *vp, int32_t *vn, int32_t *send_rows, int32_t *send_cnt,
                       int32_t *err, void *stream_) {
    WR_REQUIRE(u && p && n && vu && vp && vn && send_rows && send_cnt && err, WR_E_NULL, "wr_shard_route: NULL argument");
    WR_REQUIRE(world >= 1 && world <= kSrMaxWorld && rank >= 0 && rank < world, WR_E_RANGE, "world %d / rank %d (at most %d ranks)",
               (int)world, (int)rank, kSrMaxWorld);
    WR_REQUIRE(n_triplets > 0 && n_triplets < (int64_t(1) << 31) && batch_size > 0 && batch_size <= (int64_t(1) << 24), WR_E_SHAPE,
               "bad sizes");
    WR_REQUIRE(n_users > 0 && n_users < (int64_t(1) << 31) && n_items > 0 && n_items < (int64_t(1) << 31), WR_E_SHAPE, "bad table sizes");
    WR_REQUIRE(rows_per_owner % 32 == 0 && rows_per_owner * world >= n_items && rows_per_owner * world < (int64_t(1) << 40), WR_E_RANGE,
               "rows_per_owner must be a multiple of 32 with world * rows_per_owner >= n_items");
    WR_REQUIRE(list_cap >= 1 && list_cap < (int64_t(1) << 31) && n_local_items >= 0, WR_E_RANGE, "bad list capacity");
    const int64_t nb = (n_triplets + batch_size - 1) / batch_size;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const size_t lds = (size_t)(2 * kSrWords + 2 * kSrMaxWorld + kSrThreads / 64 + 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        WR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(shard_route_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(shard_route_kernel, dim3((unsigned)nb), dim3(kSrThreads), lds, stream, u, p, n, n_triplets, (int)batch_size,
                       (int)nb, (int)world, (int)rank, (int)n_users, (int)n_items, (int)rows_per_owner, (int)n_local_items,
                       (int)list_cap, vu, vp, vn, send_rows, send_cnt, err);
    WR_LAUNCH_CHECK("shard_route_kernel");
    return WR_OK;
}

int32_t wr_shard_pack(const int32_t *recv_rows, const int32_t *recv_cnt, int64_t n_batches, int32_t world, int64_t list_cap,
                      int64_t n_local_items, int64_t *serve_rows, int64_t serve_stride, int32_t *serve_off, int32_t *err,
                      void *stream_) {
    WR_REQUIRE(recv_rows && recv_cnt && serve_rows && serve_off && err, WR_E_NULL, "wr_shard_pack: NULL argument");
    WR_REQUIRE(world >= 1 && world <= kSrMaxWorld && n_batches > 0 && n_batches < (int64_t(1) << 31), WR_E_RANGE, "bad world / batches");
    WR_REQUIRE(list_cap >= 1 && serve_stride >= list_cap * world && n_local_items >= 0, WR_E_RANGE, "bad capacities");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)n_batches), dim3(256), 0, stream, recv_rows, recv_cnt, (int)n_batches,
                       (int)world, (int)list_cap, serve_rows, serve_stride, serve_off, (int)n_local_items, err);
    WR_LAUNCH_CHECK("shard_pack_kernel");
    return WR_OK;
}

}  // extern "C"
