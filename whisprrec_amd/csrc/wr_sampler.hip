// wr_sampler.hip — on-device epoch preparation: negative sampling with rejection against the user's train set, and the
// epoch shuffle (second half of the file).
//
// Device counterpart of GeneralModel.Dataset.actions_before_epoch (reference src/models/BaseModel.py:167-177): one negative
// per training row, uniform over [1, n_items) — item 0 is never drawn, like the reference (:168) — redrawn while it is in
// the user's train set (:172-174).  The reference consumes NumPy's MT19937 stream row by row on the host (minutes at
// 100 M rows); this kernel uses a counter-based generator keyed by (seed, epoch, row, attempt), so rows are independent
// and the result does not depend on scheduling.  It is NOT stream-compatible with NumPy (documented in DESIGN.md): the
// host path in whisprrec_amd/host.py keeps the bit-exact NumPy sampler for small-scale parity; the oracle restates THIS
// generator in NumPy and tests compare bit for bit.
#include "wr_common.h"

namespace wr {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__host__ __device__ __forceinline__ uint32_t draw_item(uint64_t seed, uint64_t epoch, uint64_t row, uint32_t attempt,
                                                       uint32_t n_items) {
    const uint64_t x = mix64(mix64(seed ^ (epoch * 0x9E3779B97F4A7C15ull)) ^ mix64(row * 0xD1B54A32D192ED03ull + attempt));
    // unbiased-enough multiply-high reduction of the top 32 bits onto [0, n_items-1), then shift to [1, n_items)
    return 1u + (uint32_t)(((x >> 32) * (uint64_t)(n_items - 1)) >> 32);
}

__device__ __forceinline__ bool clicked(const int *__restrict__ idx, int64_t lo, int64_t hi, int item) {
    while (lo < hi) {  // binary search in the user's ascending item list
        const int64_t mid = (lo + hi) >> 1;
        const int v = idx[mid];
        if (v == item) return true;
        if (v < item) lo = mid + 1; else hi = mid;
    }
    return false;
}

constexpr uint32_t kMaxAttempts = 64;

template <typename Idx>
__global__ __launch_bounds__(kBlock) void sample_negatives_kernel(const Idx *__restrict__ users, int64_t n, int64_t n_users,
                                                                   uint32_t n_items, const int64_t *__restrict__ ptr,
                                                                   const int *__restrict__ idx, uint64_t seed, uint64_t epoch,
                                                                   Idx *__restrict__ neg, int *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t u = (int64_t)users[i];
    if (u < 0 || u >= n_users) {
        if (err) *err = 1;
        u = 0;
    }
    const int64_t lo = ptr[u], hi = ptr[u + 1];
    uint32_t attempt = 0;
    uint32_t cand = draw_item(seed, epoch, (uint64_t)i, attempt, n_items);
    while (clicked(idx, lo, hi, (int)cand)) {
        if (++attempt < kMaxAttempts) {
            cand = draw_item(seed, epoch, (uint64_t)i, attempt, n_items);
        } else {
            // a user who clicked (almost) everything: walk forward cyclically over [1, n_items) from the last draw
            cand = (cand + 1u >= n_items) ? 1u : cand + 1u;
            if (attempt >= kMaxAttempts + n_items) {  // every item clicked: keep the reference's range, flag it
                if (err) *err = 2;
                break;
            }
        }
    }
    neg[i] = (Idx)cand;
}

// ----------------------------------------------------------------------------------------------- epoch shuffle
// Device counterpart of DataLoader(shuffle=True) (reference src/helpers/BaseRunner.py:188-193): the epoch's rows in a fresh
// random order.  The reference draws torch.randperm on the host; torch.randperm on the device sorts 100 M random keys
// (14 ms at C2 scale, a fifth of the whole epoch).  Here the order is a keyed bijection evaluated per row — no sort, no order
// array: out[i] = in[perm(i)], perm = alternating Feistel network on ceil(log2 n) bits (the two halves take turns being
// XORed with a splitmix64 hash of the other half, the round number and the (seed, epoch) key; kShuffleRounds rounds) with
// cycle walking (re-apply until the value is < n; the domain is < 2n, so < 2 evaluations on average).  Rows are
// independent; the result equals oracle.epoch_permutation bit for bit.  Like the device sampler it is NOT the reference's
// RNG stream: the bit-exact host path stays available.
constexpr int kShuffleRounds = 8;

__host__ __device__ __forceinline__ uint64_t shuffle_index(uint64_t i, uint64_t n, unsigned bits, uint64_t key) {
    const unsigned a = bits >> 1, b = bits - a;              // widths of the high and the low half (b >= a >= 1)
    const uint64_t mask_a = (uint64_t(1) << a) - 1, mask_b = (uint64_t(1) << b) - 1;
    uint64_t x = i;
    do {
        uint64_t hi = x >> b, lo = x & mask_b;
#pragma unroll
        for (int r = 0; r < kShuffleRounds; ++r) {
            if ((r & 1) == 0) hi ^= mix64(key ^ (lo * 0xD1B54A32D192ED03ull + (uint64_t)r)) & mask_a;
            else lo ^= mix64(key ^ (hi * 0xD1B54A32D192ED03ull + (uint64_t)r)) & mask_b;
        }
        x = (hi << b) | lo;
    } while (x >= n);
    return x;
}

static inline unsigned shuffle_bits(int64_t n) {
    unsigned bits = 2;
    while (bits < 62 && (int64_t(1) << bits) < n) ++bits;
    return bits;
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void epoch_shuffle_kernel(const Idx *__restrict__ c0, const Idx *__restrict__ c1,
                                                                const Idx *__restrict__ c2, int64_t n, unsigned bits,
                                                                uint64_t key, Idx *__restrict__ o0, Idx *__restrict__ o1,
                                                                Idx *__restrict__ o2, int64_t *__restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t j = (int64_t)shuffle_index((uint64_t)i, (uint64_t)n, bits, key);
    if (c0) o0[i] = c0[j];
    if (c1) o1[i] = c1[j];
    if (c2) o2[i] = c2[j];
    if (order) order[i] = j;
}

template <typename Idx>
static int32_t epoch_shuffle(const Idx *c0, const Idx *c1, const Idx *c2, int64_t n, uint64_t seed, uint64_t epoch, Idx *o0,
                             Idx *o1, Idx *o2, int64_t *order, void *stream) {
    WR_REQUIRE(n >= 0 && n < (int64_t(1) << 62), WR_E_SHAPE, "shuffle: n=%lld out of range", (long long)n);
    WR_REQUIRE((c0 == nullptr) == (o0 == nullptr) && (c1 == nullptr) == (o1 == nullptr) && (c2 == nullptr) == (o2 == nullptr),
               WR_E_NULL, "shuffle: every input column needs its output column");
    WR_REQUIRE(c0 || c1 || c2 || order, WR_E_NULL, "shuffle: nothing to do");
    WR_REQUIRE((c0 == nullptr || c0 != o0) && (c1 == nullptr || c1 != o1) && (c2 == nullptr || c2 != o2), WR_E_NULL,
               "shuffle: in-place permutation is not supported");
    if (n == 0) return WR_OK;
    const uint64_t key = mix64(mix64(seed ^ (epoch * 0x9E3779B97F4A7C15ull)) ^ 0x5DEECE66Dull);
    hipLaunchKernelGGL((epoch_shuffle_kernel<Idx>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c0, c1, c2, n, shuffle_bits(n), key, o0, o1, o2, order);
    WR_LAUNCH_CHECK("epoch_shuffle_kernel");
    return WR_OK;
}

// ----------------------------------------------------------------------------------------------- fused, by range
// Sampler and shuffle in one pass over a RANGE of the epoch's output rows: output row i takes source row j = perm(i) and the
// negative wr_sample_negatives would have drawn for row j (the generator is keyed by the source row) — bit for bit the
// columns of wr_sample_negatives followed by wr_epoch_shuffle, without the intermediate negatives array, and chunk by
// chunk: the step stream prepares the rows of plan chunk c+1 on the plan stream while chunk c trains.
template <typename Idx>
__global__ __launch_bounds__(kBlock) void epoch_prepare_range_kernel(const Idx *__restrict__ users, const Idx *__restrict__ items,
                                                                      int64_t n, int64_t n_users, uint32_t n_items,
                                                                      const int64_t *__restrict__ ptr, const int *__restrict__ idx,
                                                                      uint64_t seed, uint64_t epoch, unsigned bits, uint64_t key,
                                                                      int64_t first, int64_t count, Idx *__restrict__ out_u,
                                                                      Idx *__restrict__ out_p, Idx *__restrict__ out_n,
                                                                      int64_t *__restrict__ order, int *__restrict__ err) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const int64_t j = (int64_t)shuffle_index((uint64_t)(first + k), (uint64_t)n, bits, key);
    const Idx uj = users[j];
    int64_t u = (int64_t)uj;
    if (u < 0 || u >= n_users) {
        if (err) *err = 1;
        u = 0;
    }
    const int64_t lo = ptr[u], hi = ptr[u + 1];
    uint32_t attempt = 0;
    uint32_t cand = draw_item(seed, epoch, (uint64_t)j, attempt, n_items);
    while (clicked(idx, lo, hi, (int)cand)) {
        if (++attempt < kMaxAttempts) {
            cand = draw_item(seed, epoch, (uint64_t)j, attempt, n_items);
        } else {
            cand = (cand + 1u >= n_items) ? 1u : cand + 1u;
            if (attempt >= kMaxAttempts + n_items) {
                if (err) *err = 2;
                break;
            }
        }
    }
    out_u[k] = uj;
    out_p[k] = items[j];
    out_n[k] = (Idx)cand;
    if (order) order[k] = j;
}

template <typename Idx>
static int32_t epoch_prepare_range(const Idx *users, const Idx *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, Idx *out_u, Idx *out_p, Idx *out_n, int64_t *order,
                                   int32_t *err_flag, void *stream) {
    WR_REQUIRE(users && items && clicked_ptr && clicked_idx && out_u && out_p && out_n, WR_E_NULL, "epoch prepare: NULL argument");
    WR_REQUIRE(n > 0 && n < (int64_t(1) << 62) && n_users > 0 && n_items >= 2 && n_items < (int64_t(1) << 31), WR_E_SHAPE,
               "epoch prepare: bad sizes");
    WR_REQUIRE(first >= 0 && count >= 0 && first + count <= n, WR_E_RANGE, "epoch prepare: rows [%lld,%lld) outside the epoch's %lld",
               (long long)first, (long long)(first + count), (long long)n);
    if (count == 0) return WR_OK;
    const uint64_t key = mix64(mix64(seed ^ (epoch * 0x9E3779B97F4A7C15ull)) ^ 0x5DEECE66Dull);
    hipLaunchKernelGGL((epoch_prepare_range_kernel<Idx>), dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), users, items, n, n_users, (uint32_t)n_items, clicked_ptr, clicked_idx,
                       seed, epoch, shuffle_bits(n), key, first, count, out_u, out_p, out_n, order, err_flag);
    WR_LAUNCH_CHECK("epoch_prepare_range_kernel");
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_sample_negatives_i64(const int64_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int64_t *neg_items,
                                int32_t *err_flag, void *stream) {
    WR_REQUIRE(users && clicked_ptr && clicked_idx && neg_items, WR_E_NULL, "sampler: NULL argument");
    WR_REQUIRE(n >= 0 && n_users > 0 && n_items >= 2 && n_items < (int64_t(1) << 31), WR_E_SHAPE, "sampler: bad sizes");
    if (n == 0) return WR_OK;
    hipLaunchKernelGGL((sample_negatives_kernel<int64_t>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), users, n, n_users, (uint32_t)n_items, clicked_ptr, clicked_idx,
                       seed, epoch, neg_items, err_flag);
    WR_LAUNCH_CHECK("sample_negatives_kernel");
    return WR_OK;
}

int32_t wr_sample_negatives_i32(const int32_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int32_t *neg_items,
                                int32_t *err_flag, void *stream) {
    WR_REQUIRE(users && clicked_ptr && clicked_idx && neg_items, WR_E_NULL, "sampler: NULL argument");
    WR_REQUIRE(n >= 0 && n_users > 0 && n_items >= 2 && n_items < (int64_t(1) << 31), WR_E_SHAPE, "sampler: bad sizes");
    if (n == 0) return WR_OK;
    hipLaunchKernelGGL((sample_negatives_kernel<int32_t>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), users, n, n_users, (uint32_t)n_items, clicked_ptr, clicked_idx,
                       seed, epoch, neg_items, err_flag);
    WR_LAUNCH_CHECK("sample_negatives_kernel");
    return WR_OK;
}

int32_t wr_epoch_prepare_range_i64(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                   int64_t *order_out, int32_t *err_flag, void *stream) {
    return epoch_prepare_range<int64_t>(users, items, n, n_users, n_items, clicked_ptr, clicked_idx, seed, epoch, first, count,
                                        out_users, out_pos, out_neg, order_out, err_flag, stream);
}

int32_t wr_epoch_prepare_range_i32(const int32_t *users, const int32_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                   int64_t *order_out, int32_t *err_flag, void *stream) {
    return epoch_prepare_range<int32_t>(users, items, n, n_users, n_items, clicked_ptr, clicked_idx, seed, epoch, first, count,
                                        out_users, out_pos, out_neg, order_out, err_flag, stream);
}

int32_t wr_epoch_shuffle_i64(const int64_t *col0, const int64_t *col1, const int64_t *col2, int64_t n, uint64_t seed,
                             uint64_t epoch, int64_t *out0, int64_t *out1, int64_t *out2, int64_t *order_out, void *stream) {
    return epoch_shuffle<int64_t>(col0, col1, col2, n, seed, epoch, out0, out1, out2, order_out, stream);
}

int32_t wr_epoch_shuffle_i32(const int32_t *col0, const int32_t *col1, const int32_t *col2, int64_t n, uint64_t seed,
                             uint64_t epoch, int32_t *out0, int32_t *out1, int32_t *out2, int64_t *order_out, void *stream) {
    return epoch_shuffle<int32_t>(col0, col1, col2, n, seed, epoch, out0, out1, out2, order_out, stream);
}

}  // extern "C"
