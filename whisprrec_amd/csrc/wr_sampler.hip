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

// The same membership test against a HASH SET of the (user, item) pairs (wr_pairset_build): open addressing, linear probing,
// 64-bit keys (user << 32 | item), at most a third full.  A test is one random 64-byte sector (probes that follow stay in
// it, mostly) where the binary search in the user's list reads ~7 — the sampler is bound by exactly that traffic at 100 M
// rows (45 GB of sectors per epoch).  Same answers, so the drawn negatives do not change.
constexpr unsigned long long kPairEmpty = ~0ull;

struct ClickedLists {     // per-user ascending item lists (CSR)
    const int64_t *ptr;
    const int *idx;
    int64_t lo, hi;
    __device__ __forceinline__ void open(int64_t u) { lo = ptr[u]; hi = ptr[u + 1]; }
    __device__ __forceinline__ bool has(int64_t, int item) const { return clicked(idx, lo, hi, item); }
};

struct ClickedPairs {     // hash set of the pairs
    const unsigned long long *table;
    uint64_t mask;
    __device__ __forceinline__ void open(int64_t) {}
    __device__ __forceinline__ bool has(int64_t u, int item) const {
        const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)item;
        uint64_t slot = mix64(key) & mask;
        for (uint64_t probes = 0; probes <= mask; ++probes) {      // the table is never full: an empty slot ends the walk
            const unsigned long long v = table[slot];
            if (v == key) return true;
            if (v == kPairEmpty) return false;
            slot = (slot + 1) & mask;
        }
        return false;
    }
};

constexpr uint32_t kMaxAttempts = 64;

// the negative of source row `row` of user u (shared by the two kernels below)
template <typename Clicked>
__device__ __forceinline__ uint32_t draw_negative(Clicked &cl, int64_t u, uint64_t seed, uint64_t epoch, uint64_t row,
                                                  uint32_t n_items, int *__restrict__ err) {
    cl.open(u);
    uint32_t attempt = 0;
    uint32_t cand = draw_item(seed, epoch, row, attempt, n_items);
    while (cl.has(u, (int)cand)) {
        if (++attempt < kMaxAttempts) {
            cand = draw_item(seed, epoch, row, attempt, n_items);
        } else {
            // a user who clicked (almost) everything: walk forward cyclically over [1, n_items) from the last draw
            cand = (cand + 1u >= n_items) ? 1u : cand + 1u;
            if (attempt >= kMaxAttempts + n_items) {  // every item clicked: keep the reference's range, flag it
                if (err) *err = 2;
                break;
            }
        }
    }
    return cand;
}

template <typename Idx, typename Clicked>
__global__ __launch_bounds__(kBlock) void sample_negatives_kernel(const Idx *__restrict__ users, int64_t n, int64_t n_users,
                                                                   uint32_t n_items, Clicked cl, uint64_t seed, uint64_t epoch,
                                                                   Idx *__restrict__ neg, int *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t u = (int64_t)users[i];
    if (u < 0 || u >= n_users) {
        if (err) *err = 1;
        u = 0;
    }
    neg[i] = (Idx)draw_negative(cl, u, seed, epoch, (uint64_t)i, n_items, err);
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
// PACKED: the source rows as ONE 8-byte word each, (user << 32) | item — one random memory sector per output row where the
// two columns cost two (the kernel is bound by its random sectors: source row(s) + the membership probe); `users` then
// points to the packed words and `items` is not read.
template <typename Idx, typename Clicked, bool PACKED = false>
__global__ __launch_bounds__(kBlock) void epoch_prepare_range_kernel(const Idx *__restrict__ users, const Idx *__restrict__ items,
                                                                      int64_t n, int64_t n_users, uint32_t n_items, Clicked cl,
                                                                      uint64_t seed, uint64_t epoch, unsigned bits, uint64_t key,
                                                                      int64_t first, int64_t count, Idx *__restrict__ out_u,
                                                                      Idx *__restrict__ out_p, Idx *__restrict__ out_n,
                                                                      int64_t *__restrict__ order, int *__restrict__ err) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const int64_t j = (int64_t)shuffle_index((uint64_t)(first + k), (uint64_t)n, bits, key);
    Idx uj, ij;
    if constexpr (PACKED) {
        const unsigned long long w = reinterpret_cast<const unsigned long long *>(users)[j];
        uj = (Idx)(int32_t)(uint32_t)(w >> 32);
        ij = (Idx)(int32_t)(uint32_t)w;
    } else {
        uj = users[j];
        ij = items[j];
    }
    int64_t u = (int64_t)uj;
    if (u < 0 || u >= n_users) {
        if (err) *err = 1;
        u = 0;
    }
    const uint32_t cand = draw_negative(cl, u, seed, epoch, (uint64_t)j, n_items, err);
    out_u[k] = uj;
    out_p[k] = ij;
    out_n[k] = (Idx)cand;
    if (order) order[k] = j;
}

static inline bool pairset_ok(const void *table, int64_t capacity) {
    return table != nullptr && capacity >= 2 && (capacity & (capacity - 1)) == 0;
}

template <typename Idx>
static int32_t epoch_prepare_range(const Idx *users, const Idx *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, Idx *out_u, Idx *out_p, Idx *out_n, int64_t *order,
                                   int32_t *err_flag, void *stream, const uint64_t *pair_table = nullptr,
                                   int64_t pair_capacity = 0, bool packed = false) {
    WR_REQUIRE(users && (items || packed) && out_u && out_p && out_n, WR_E_NULL, "epoch prepare: NULL argument");
    WR_REQUIRE(!packed || pair_table != nullptr, WR_E_NULL, "epoch prepare: packed source rows go with a pair set");
    WR_REQUIRE((clicked_ptr && clicked_idx) || pairset_ok(pair_table, pair_capacity), WR_E_NULL,
               "epoch prepare: needs the clicked lists or a pair set (capacity a power of two)");
    WR_REQUIRE(n > 0 && n < (int64_t(1) << 62) && n_users > 0 && n_items >= 2 && n_items < (int64_t(1) << 31), WR_E_SHAPE,
               "epoch prepare: bad sizes");
    WR_REQUIRE(first >= 0 && count >= 0 && first + count <= n, WR_E_RANGE, "epoch prepare: rows [%lld,%lld) outside the epoch's %lld",
               (long long)first, (long long)(first + count), (long long)n);
    if (count == 0) return WR_OK;
    const uint64_t key = mix64(mix64(seed ^ (epoch * 0x9E3779B97F4A7C15ull)) ^ 0x5DEECE66Dull);
    const dim3 grid((unsigned)((count + kBlock - 1) / kBlock));
    if (packed)
        hipLaunchKernelGGL((epoch_prepare_range_kernel<Idx, ClickedPairs, true>), grid, dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), users, items, n, n_users, (uint32_t)n_items,
                           ClickedPairs{reinterpret_cast<const unsigned long long *>(pair_table), (uint64_t)pair_capacity - 1},
                           seed, epoch, shuffle_bits(n), key, first, count, out_u, out_p, out_n, order, err_flag);
    else if (pair_table != nullptr)
        hipLaunchKernelGGL((epoch_prepare_range_kernel<Idx, ClickedPairs>), grid, dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), users, items, n, n_users, (uint32_t)n_items,
                           ClickedPairs{reinterpret_cast<const unsigned long long *>(pair_table), (uint64_t)pair_capacity - 1},
                           seed, epoch, shuffle_bits(n), key, first, count, out_u, out_p, out_n, order, err_flag);
    else
        hipLaunchKernelGGL((epoch_prepare_range_kernel<Idx, ClickedLists>), grid, dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), users, items, n, n_users, (uint32_t)n_items,
                           ClickedLists{clicked_ptr, clicked_idx, 0, 0}, seed, epoch, shuffle_bits(n), key, first, count, out_u,
                           out_p, out_n, order, err_flag);
    WR_LAUNCH_CHECK("epoch_prepare_range_kernel");
    return WR_OK;
}

// one 16-lane team per user: the user's items go into the set (atomicCAS on the slot, linear probing)
__global__ __launch_bounds__(kBlock) void pairset_insert_kernel(const int64_t *__restrict__ ptr, const int *__restrict__ idx,
                                                                 int64_t n_users, unsigned long long *__restrict__ table,
                                                                 uint64_t mask, int *__restrict__ err) {
    const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
    if (u >= n_users) return;
    const int lane = threadIdx.x % 16;
    for (int64_t k = ptr[u] + lane; k < ptr[u + 1]; k += 16) {
        const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)idx[k];
        uint64_t slot = mix64(key) & mask;
        uint64_t probes = 0;
        for (; probes <= mask; ++probes) {
            const unsigned long long prev = atomicCAS(&table[slot], kPairEmpty, key);
            if (prev == kPairEmpty || prev == key) break;
            slot = (slot + 1) & mask;
        }
        if (probes > mask && err) *err = 3;        // table full (the caller sized it too small)
    }
}

template <typename Idx>
static int32_t sample_negatives(const Idx *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed,
                                uint64_t epoch, Idx *neg_items, int32_t *err_flag, void *stream) {
    WR_REQUIRE(users && neg_items, WR_E_NULL, "sampler: NULL argument");
    WR_REQUIRE((clicked_ptr && clicked_idx) || pairset_ok(pair_table, pair_capacity), WR_E_NULL,
               "sampler: needs the clicked lists or a pair set (capacity a power of two)");
    WR_REQUIRE(n >= 0 && n_users > 0 && n_items >= 2 && n_items < (int64_t(1) << 31), WR_E_SHAPE, "sampler: bad sizes");
    if (n == 0) return WR_OK;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (pair_table != nullptr)
        hipLaunchKernelGGL((sample_negatives_kernel<Idx, ClickedPairs>), grid, dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), users, n, n_users, (uint32_t)n_items,
                           ClickedPairs{reinterpret_cast<const unsigned long long *>(pair_table), (uint64_t)pair_capacity - 1},
                           seed, epoch, neg_items, err_flag);
    else
        hipLaunchKernelGGL((sample_negatives_kernel<Idx, ClickedLists>), grid, dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), users, n, n_users, (uint32_t)n_items,
                           ClickedLists{clicked_ptr, clicked_idx, 0, 0}, seed, epoch, neg_items, err_flag);
    WR_LAUNCH_CHECK("sample_negatives_kernel");
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_sample_negatives_i64(const int64_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int64_t *neg_items,
                                int32_t *err_flag, void *stream) {
    WR_REQUIRE(clicked_ptr && clicked_idx, WR_E_NULL, "sampler: NULL argument");
    return sample_negatives<int64_t>(users, n, n_users, n_items, clicked_ptr, clicked_idx, nullptr, 0, seed, epoch, neg_items,
                                     err_flag, stream);
}

int32_t wr_sample_negatives_i32(const int32_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int32_t *neg_items,
                                int32_t *err_flag, void *stream) {
    WR_REQUIRE(clicked_ptr && clicked_idx, WR_E_NULL, "sampler: NULL argument");
    return sample_negatives<int32_t>(users, n, n_users, n_items, clicked_ptr, clicked_idx, nullptr, 0, seed, epoch, neg_items,
                                     err_flag, stream);
}

int64_t wr_pairset_capacity(int64_t n_pairs) {
    if (n_pairs < 0 || n_pairs >= (int64_t(1) << 40)) return WR_E_SHAPE;
    int64_t cap = 1024;
    while (cap < 3 * n_pairs) cap <<= 1;      // at most a third full
    return cap;
}

int32_t wr_pairset_build(const int64_t *clicked_ptr, const int32_t *clicked_idx, int64_t n_users, uint64_t *table,
                         int64_t capacity, int32_t *err_flag, void *stream_) {
    WR_REQUIRE(clicked_ptr && clicked_idx && table, WR_E_NULL, "pair set: NULL argument");
    WR_REQUIRE(n_users > 0 && n_users < (int64_t(1) << 31) && pairset_ok(table, capacity), WR_E_SHAPE,
               "pair set: capacity must be a power of two");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    WR_HIP(hipMemsetAsync(table, 0xFF, (size_t)capacity * 8, stream));
    hipLaunchKernelGGL(pairset_insert_kernel, dim3((unsigned)((n_users * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       clicked_ptr, clicked_idx, n_users, reinterpret_cast<unsigned long long *>(table), (uint64_t)capacity - 1,
                       err_flag);
    WR_LAUNCH_CHECK("pairset_insert_kernel");
    return WR_OK;
}

int32_t wr_sample_negatives_set_i64(const int64_t *users, int64_t n, int64_t n_users, int64_t n_items, const uint64_t *pair_table,
                                    int64_t pair_capacity, uint64_t seed, uint64_t epoch, int64_t *neg_items,
                                    int32_t *err_flag, void *stream) {
    WR_REQUIRE(pair_table != nullptr, WR_E_NULL, "sampler: pair set is NULL");
    return sample_negatives<int64_t>(users, n, n_users, n_items, nullptr, nullptr, pair_table, pair_capacity, seed, epoch,
                                     neg_items, err_flag, stream);
}

int32_t wr_sample_negatives_set_i32(const int32_t *users, int64_t n, int64_t n_users, int64_t n_items, const uint64_t *pair_table,
                                    int64_t pair_capacity, uint64_t seed, uint64_t epoch, int32_t *neg_items,
                                    int32_t *err_flag, void *stream) {
    WR_REQUIRE(pair_table != nullptr, WR_E_NULL, "sampler: pair set is NULL");
    return sample_negatives<int32_t>(users, n, n_users, n_items, nullptr, nullptr, pair_table, pair_capacity, seed, epoch,
                                     neg_items, err_flag, stream);
}

int32_t wr_epoch_prepare_range_set_i64(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                       const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                       int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                       int64_t *order_out, int32_t *err_flag, void *stream) {
    WR_REQUIRE(pair_table != nullptr, WR_E_NULL, "epoch prepare: pair set is NULL");
    return epoch_prepare_range<int64_t>(users, items, n, n_users, n_items, nullptr, nullptr, seed, epoch, first, count, out_users,
                                        out_pos, out_neg, order_out, err_flag, stream, pair_table, pair_capacity);
}

int32_t wr_epoch_prepare_range_set_i32(const int32_t *users, const int32_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                       const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                       int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                       int64_t *order_out, int32_t *err_flag, void *stream) {
    WR_REQUIRE(pair_table != nullptr, WR_E_NULL, "epoch prepare: pair set is NULL");
    return epoch_prepare_range<int32_t>(users, items, n, n_users, n_items, nullptr, nullptr, seed, epoch, first, count, out_users,
                                        out_pos, out_neg, order_out, err_flag, stream, pair_table, pair_capacity);
}

int32_t wr_epoch_prepare_range_packed_i64(const uint64_t *packed_rows, int64_t n, int64_t n_users, int64_t n_items,
                                          const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                          int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                          int64_t *order_out, int32_t *err_flag, void *stream) {
    WR_REQUIRE(packed_rows != nullptr && pair_table != nullptr, WR_E_NULL, "epoch prepare: packed rows / pair set is NULL");
    return epoch_prepare_range<int64_t>(reinterpret_cast<const int64_t *>(packed_rows), nullptr, n, n_users, n_items, nullptr,
                                        nullptr, seed, epoch, first, count, out_users, out_pos, out_neg, order_out, err_flag,
                                        stream, pair_table, pair_capacity, true);
}

int32_t wr_epoch_prepare_range_packed_i32(const uint64_t *packed_rows, int64_t n, int64_t n_users, int64_t n_items,
                                          const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                          int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                          int64_t *order_out, int32_t *err_flag, void *stream) {
    WR_REQUIRE(packed_rows != nullptr && pair_table != nullptr, WR_E_NULL, "epoch prepare: packed rows / pair set is NULL");
    return epoch_prepare_range<int32_t>(reinterpret_cast<const int32_t *>(packed_rows), nullptr, n, n_users, n_items, nullptr,
                                        nullptr, seed, epoch, first, count, out_users, out_pos, out_neg, order_out, err_flag,
                                        stream, pair_table, pair_capacity, true);
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
