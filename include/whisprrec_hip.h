/*
 * whisprrec_hip.h — C-ABI of libwhisprrec_hip.so: the MI355X (gfx950) implementation of WhisprRec's
 * embedding-CF training hot path.
 *
 * The reference (HeyWeCome/WhisprRec) is pure Python and has no FFI; its "operator boundary" for this
 * path is the chain of PyTorch calls cited per entry point below (paths relative to the reference
 * root).  A maintainer binds this library with ctypes (INTEGRATION.md shows the stub); every symbol is
 * plain C: raw device pointers, sizes, a hipStream_t passed as void*.
 *
 * Conventions
 *   - return value: 0 = ok, <0 = argument error (WR_E_*), >0 = hipError_t of the failing HIP call.
 *     wr_last_error() returns a thread-local, human-readable message for the last non-zero return.
 *   - ownership: every table / index / output / workspace buffer is allocated and owned by the caller
 *     (e.g. a PyTorch-ROCm tensor, passed as tensor.data_ptr()).  The library allocates nothing.
 *   - asynchrony: every call only enqueues work on `stream` (hipStream_t; NULL = default stream) and
 *     returns; no call synchronises the device.  Calls on one stream execute in order.
 *   - tables are row-major fp32 [n_rows, D], rows contiguous (row stride = D floats), 16-byte aligned
 *     base.  D must be a multiple of 4 and <= 1024.
 *   - indices at the reference boundary are int64 (src/models/BaseModel.py:121); the sorted "plan"
 *     arrays produced and consumed inside the library are int32.
 *   - semantics are batch-synchronous: every gradient of a step is computed from the pre-step tables
 *     (src/helpers/BaseRunner.py:197-199), duplicates inside a batch are summed, and results do not
 *     depend on scheduling (no float atomics: bitwise reproducible run to run).
 */
#ifndef WHISPRREC_HIP_H
#define WHISPRREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WR_ABI_VERSION 1

#define WR_OK 0
#define WR_E_NULL (-1)     /* required pointer is NULL            */
#define WR_E_SHAPE (-2)    /* bad size / D / batch                */
#define WR_E_WORKSPACE (-3)/* workspace too small                 */
#define WR_E_ALIGN (-4)    /* pointer not 16-byte aligned         */
#define WR_E_RANGE (-5)    /* value out of supported range        */

int32_t wr_abi_version(void);
const char *wr_last_error(void);
/* number of CUs, wavefront size and gcnArchName of the current device (sanity: expects gfx950) */
int32_t wr_device_info(int32_t *n_cu, int32_t *wave_size, char *arch, int32_t arch_len);

/* ---------------------------------------------------------------------------------------------------
 * K1-K3  BPRMF.predict forward  — src/models/general/BPRMF.py:69-80, src/utils/loss.py:37-39
 *   pos[b] = <U[u_b], I[p_b]>, neg[b] = <U[u_b], I[n_b]>, loss = -mean(log(1e-10 + sigmoid(pos-neg)))
 *   coef[b] = dloss/dpos[b]  (what loss.backward(), BaseRunner.py:198, feeds the three gathers)
 * pos_score / neg_score / coef may be NULL.  loss: 1 float on device.
 * workspace: >= wr_bpr_fwd_workspace_bytes(B) bytes.
 * Also serves LightGCN.predict (LightGCN.py:156-163) and SASRec.predict (SASRec.py:105-111) on
 * propagated / encoded rows: `user_tab` is then any [n_users, D] fp32 matrix.
 * --------------------------------------------------------------------------------------------------- */
int64_t wr_bpr_fwd_workspace_bytes(int64_t B);
int32_t wr_bpr_fwd(const float *user_tab, int64_t n_users, const float *item_tab, int64_t n_items, int32_t D,
                   const int64_t *u, const int64_t *p, const int64_t *n, int64_t B, float *pos_score, float *neg_score,
                   float *coef, float *loss, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Batch plan — the on-device counterpart of DataLoader batching (BaseRunner.py:188-193) +
 * collate_batch (BaseModel.py:96-127) for the fused step: for each consecutive batch of `batch_size`
 * triplets (last one may be short, no drop_last — BaseRunner.py:201), triplets are stably sorted by
 * user and the 2*B (item, source) occurrences are stably sorted by item, so that each table row of a
 * step has exactly one owner.
 *   tu,tp,tn [n]  : triplets of batch k at [k*batch_size, ...), sorted by user; bit 31 of tp[t] / tn[t] is
 *                   set when that item row has more than one occurrence in the batch (row = low 31 bits)
 *   torig    [n]  : original position (0..n) of each sorted triplet (may be NULL)
 *   oc_item  [2n] : occurrences of batch k at [2*k*batch_size, ...), sorted by item row
 *   oc_src   [2n] : (local sorted triplet index << 1) | (1 if the occurrence is the NEGATIVE item)
 * Index inputs are int64 (reference layout) or int32.  Every index is range-checked on device against
 * n_users / n_items; *err_flag (int32 on device, caller-zeroed) is set to 1 on violation.
 * --------------------------------------------------------------------------------------------------- */
int64_t wr_bprmf_plan_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items);
int32_t wr_bprmf_plan_build_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                void *workspace, int64_t workspace_bytes, void *stream);
int32_t wr_bprmf_plan_build_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                void *workspace, int64_t workspace_bytes, void *stream);

/* Small batches (batch_size <= wr_bprmf_plan_small_max_batch() = 4,096; the reference's default is 2,048): the plan of a
 * batch is built by ONE workgroup in LDS, one launch for all batches, no workspace, nothing to read back but err_flag — the
 * builder of the LightGCN step (a plan per optimizer step, hipGraph-capturable).  Tables small enough for one LDS counter
 * per row (8 * max(n_users, n_items) + 16 * P bytes <= 128 KB, P = batch size rounded up to a power of two: ml-scale) take a
 * counting sort (count, scan, scatter, order every row's segment by position); the others two bitonic sorts of unique
 * composites.  Either way the arrays are wr_bprmf_plan_build_*'s, bit for bit.  WR_PLAN_SMALL=bitonic in the environment
 * keeps the sorting form (A/B runs). */
int64_t wr_bprmf_plan_small_max_batch(void);
int32_t wr_bprmf_plan_build_small_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                      int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                      int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                      void *stream);
int32_t wr_bprmf_plan_build_small_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                      int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                      int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *err_flag,
                                      void *stream);

/* Hand-written builder producing bit-identical plan arrays: bucket scatter + per-bucket LDS bitonic sort (one HBM
 * round trip per pair instead of the radix sort's four).  Applies when wr_bprmf_plan_fast_workspace_bytes(...) > 0
 * (batch sizes up to ~1 M, any table size).  flags: int32[2] on device, caller-zeroed; flags[0] = index out of range (as err_flag
 * above), flags[1] = a bucket overflowed (heavily skewed ids) -> the plan is INVALID and the caller must rebuild it
 * with wr_bprmf_plan_build_*.  Asynchronous like every other call; the caller reads the flags after the stream work. */
int64_t wr_bprmf_plan_fast_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items);
int32_t wr_bprmf_plan_build_fast_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream);
int32_t wr_bprmf_plan_build_fast_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                     int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                     int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                     void *workspace, int64_t workspace_bytes, void *stream);

/* wr_bprmf_plan_build_fast_* that also writes, per batch, the bitmap "item row has several occurrences in the batch"
 * (bitmap [n_batches * ceil(n_items/32)], zeroed and filled here): the workgroup that sorts a bucket of item rows owns the
 * bucket's bitmap words, so no global atomics are needed (wr_bprmf_plan_overlap_marks spends one per such occurrence).
 * Applies to equal-width buckets spanning whole bitmap words: wr_bprmf_plan_fast_marks_supported() == 1 (WR_E_RANGE
 * otherwise).  After a bucket overflow (flags[1]) the bitmap is as incomplete as the plan. */
int32_t wr_bprmf_plan_fast_marks_supported(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items);
int32_t wr_bprmf_plan_build_fast_marks_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                           int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                           int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                           void *workspace, int64_t workspace_bytes, int32_t *bitmap, void *stream);
int32_t wr_bprmf_plan_build_fast_marks_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                           int64_t batch_size, int64_t n_users, int64_t n_items, int32_t *tu, int32_t *tp,
                                           int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src, int32_t *flags,
                                           void *workspace, int64_t workspace_bytes, int32_t *bitmap, void *stream);

/* Bucket map for the hand-written plan builder on skewed ids.  The builder's first stage appends every triplet (item
 * occurrence) to a (batch, row-range) bucket of FIXED capacity (twice the mean load + 64).  Without a map the ranges have
 * equal width, and popularity-skewed ids overflow them (flags[1], rebuild with wr_bprmf_plan_build_*).  A map makes the
 * ranges equal in expected LOAD instead: row_bucket[row] & 0xffff is the bucket of the row (buckets ascend with the rows,
 * so the plan arrays are the same, bit for bit); a row heavy enough to fill a bucket on its own owns row_bucket[row] >> 16
 * consecutive sub-buckets that split its occurrences by position in the batch.  Per bucket: first row, number of rows,
 * and for the sub-buckets of a heavy row (index among them) | (their number << 16), else 0.  All arrays on the device.
 * whisprrec_amd/hip_ops.py (BucketMap) derives a map from an epoch's id columns; either side may be NULL (equal widths). */
typedef struct wr_bucket_side {
    int32_t n_buckets;             /* 1..1024 */
    const int32_t *row_bucket;     /* [n_rows] */
    const int32_t *bucket_start;   /* [n_buckets] */
    const int32_t *bucket_rows;    /* [n_buckets] */
    const int32_t *bucket_sub;     /* [n_buckets] */
} wr_bucket_side;

int64_t wr_bprmf_plan_fast_mapped_workspace_bytes(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items,
                                                  int32_t n_buckets_users, int32_t n_buckets_items);
int32_t wr_bprmf_plan_build_fast_mapped_i64(const int64_t *u, const int64_t *p, const int64_t *n, int64_t n_triplets,
                                            int64_t batch_size, int64_t n_users, int64_t n_items,
                                            const wr_bucket_side *map_users, const wr_bucket_side *map_items, int32_t *tu,
                                            int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src,
                                            int32_t *flags, void *workspace, int64_t workspace_bytes, void *stream);
int32_t wr_bprmf_plan_build_fast_mapped_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets,
                                            int64_t batch_size, int64_t n_users, int64_t n_items,
                                            const wr_bucket_side *map_users, const wr_bucket_side *map_items, int32_t *tu,
                                            int32_t *tp, int32_t *tn, int32_t *torig, int32_t *oc_item, int32_t *oc_src,
                                            int32_t *flags, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Epoch preparation on the device — GeneralModel.Dataset.actions_before_epoch (src/models/BaseModel.py:167-177):
 * one negative per training row, uniform over [1, n_items) (item 0 is never drawn, :168), redrawn while it is in the
 * user's train set (:172-174).  clicked_ptr int64 [n_users+1] / clicked_idx int32 (ascending per user) is the CSR of
 * train_clicked_set.  Counter-based generator keyed by (seed, epoch, row, attempt): rows are independent, the result
 * is reproducible and equals oracle.sample_negatives_counter bit for bit; it is NOT NumPy's MT19937 stream (the
 * bit-exact NumPy sampler stays available on the host for small-scale parity).  err_flag (int32, device, caller-
 * zeroed): 1 = user id out of range, 2 = some user has clicked every item.
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_sample_negatives_i64(const int64_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int64_t *neg_items,
                                int32_t *err_flag, void *stream);
int32_t wr_sample_negatives_i32(const int32_t *users, int64_t n, int64_t n_users, int64_t n_items, const int64_t *clicked_ptr,
                                const int32_t *clicked_idx, uint64_t seed, uint64_t epoch, int32_t *neg_items,
                                int32_t *err_flag, void *stream);

/* Sampler and shuffle fused, for a RANGE [first, first + count) of the epoch's output rows (batch order): output row i takes
 * source row j = perm(i) of (users, items) and the negative wr_sample_negatives draws for row j — bit for bit the columns of
 * wr_sample_negatives followed by wr_epoch_shuffle (same seed, epoch), no intermediate array.  out_* [count] receive rows
 * first.. (the caller passes pointers to that position); order_out [count] (may be NULL) the source rows.  The step stream
 * prepares plan chunk c+1's rows this way beside chunk c's steps (hip_ops.PipelinedSgd, `prep`). */
int32_t wr_epoch_prepare_range_i64(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                   int64_t *order_out, int32_t *err_flag, void *stream);
int32_t wr_epoch_prepare_range_i32(const int32_t *users, const int32_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                   const int64_t *clicked_ptr, const int32_t *clicked_idx, uint64_t seed, uint64_t epoch,
                                   int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                   int64_t *order_out, int32_t *err_flag, void *stream);

/* The same membership test — "has this user clicked this item" (BaseModel.py:172-174) — against a HASH SET of the (user, item)
 * pairs instead of the per-user lists: open addressing with linear probing over 64-bit keys (user << 32 | item), capacity a
 * power of two, at most a third full.  One test then reads one random 64-byte sector where the binary search reads ~7: the
 * sampler is bound by that traffic at 100 M rows.  Same answers, hence the same negatives, bit for bit.
 *   wr_pairset_capacity(n_pairs)  -> entries (uint64 each) the table needs;
 *   wr_pairset_build              fills `table` [capacity] from the clicked lists (once per training frame);
 *                                 err_flag (may be NULL) is set to 3 if the table was too small;
 *   wr_sample_negatives_set_*, wr_epoch_prepare_range_set_*   the two entry points above with the set in place of the lists. */
int64_t wr_pairset_capacity(int64_t n_pairs);
int32_t wr_pairset_build(const int64_t *clicked_ptr, const int32_t *clicked_idx, int64_t n_users, uint64_t *table,
                         int64_t capacity, int32_t *err_flag, void *stream);
int32_t wr_sample_negatives_set_i64(const int64_t *users, int64_t n, int64_t n_users, int64_t n_items, const uint64_t *pair_table,
                                    int64_t pair_capacity, uint64_t seed, uint64_t epoch, int64_t *neg_items,
                                    int32_t *err_flag, void *stream);
int32_t wr_sample_negatives_set_i32(const int32_t *users, int64_t n, int64_t n_users, int64_t n_items, const uint64_t *pair_table,
                                    int64_t pair_capacity, uint64_t seed, uint64_t epoch, int32_t *neg_items,
                                    int32_t *err_flag, void *stream);
int32_t wr_epoch_prepare_range_set_i64(const int64_t *users, const int64_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                       const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                       int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                       int64_t *order_out, int32_t *err_flag, void *stream);
int32_t wr_epoch_prepare_range_set_i32(const int32_t *users, const int32_t *items, int64_t n, int64_t n_users, int64_t n_items,
                                       const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                       int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                       int64_t *order_out, int32_t *err_flag, void *stream);

/* wr_epoch_prepare_range_set_* with the source rows PACKED, one 8-byte word (user << 32) | item per interaction (built once
 * per training frame): one random memory sector per output row for the source instead of two — the kernel is bound by its
 * random sectors (source row + membership probe).  Same columns, bit for bit. */
int32_t wr_epoch_prepare_range_packed_i64(const uint64_t *packed_rows, int64_t n, int64_t n_users, int64_t n_items,
                                          const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                          int64_t first, int64_t count, int64_t *out_users, int64_t *out_pos, int64_t *out_neg,
                                          int64_t *order_out, int32_t *err_flag, void *stream);
int32_t wr_epoch_prepare_range_packed_i32(const uint64_t *packed_rows, int64_t n, int64_t n_users, int64_t n_items,
                                          const uint64_t *pair_table, int64_t pair_capacity, uint64_t seed, uint64_t epoch,
                                          int64_t first, int64_t count, int32_t *out_users, int32_t *out_pos, int32_t *out_neg,
                                          int64_t *order_out, int32_t *err_flag, void *stream);

/* Epoch shuffle on the device — the row order DataLoader(shuffle=True) gives an epoch (src/helpers/BaseRunner.py:188-193):
 * out_k[i] = col_k[perm(i)] for up to three index columns (NULL pairs are skipped), order_out[i] = perm(i) if not NULL.
 * perm is a keyed bijection of [0, n) evaluated per row (alternating Feistel network on ceil(log2 n) bits, splitmix64
 * round function keyed by (seed, epoch), cycle walking): no sort and no order array are needed.  Reproducible, equal to
 * oracle.epoch_permutation bit for bit; NOT the reference's torch.randperm stream (the bit-exact host path stays
 * available, whisprrec_amd/runner.py epoch_order).  Outputs must not alias inputs. */
int32_t wr_epoch_shuffle_i64(const int64_t *col0, const int64_t *col1, const int64_t *col2, int64_t n, uint64_t seed,
                             uint64_t epoch, int64_t *out0, int64_t *out1, int64_t *out2, int64_t *order_out, void *stream);
int32_t wr_epoch_shuffle_i32(const int32_t *col0, const int32_t *col1, const int32_t *col2, int64_t n, uint64_t seed,
                             uint64_t epoch, int32_t *out0, int32_t *out1, int32_t *out2, int64_t *order_out, void *stream);

/* Long runs ("hot rows": a table row with more than 32 occurrences in one batch, e.g. power-law ids).  Optional second
 * part of a plan: every such run is cut into pieces of at most 256 positions so that the step can work on a hot row with
 * many workgroups instead of one 16-lane team (still in a fixed order: reproducible).  Two sides: kind 0 = item rows (runs of
 * oc_item, 2*B positions per batch), kind 1 = user rows (runs of tu, B positions per batch).  Arrays are per batch with the
 * capacities of wr_bprmf_hot_caps(batch_size, kind): piece_q/piece_len [n_batches*cap_pieces] (start inside the batch,
 * length), run_q/run_first/run_np [n_batches*cap_runs] (head position, first piece, piece count).  counts: int32
 * [n_batches*4] on the DEVICE, caller-zeroed: item pieces, item runs, user pieces, user runs per batch.  The caller copies
 * `counts` to the host (it sizes the extra launches) and hands everything to the step calls through wr_hot_runs (NULL = none:
 * long runs are then walked sequentially by one team — correct, but slow on skewed data). */
typedef struct wr_hot_runs {
    const int32_t *piece_q, *piece_len, *run_q, *run_first, *run_np;             /* device, item side */
    const int32_t *u_piece_q, *u_piece_len, *u_run_q, *u_run_first, *u_run_np;   /* device, user side */
    const int32_t *counts_host;                                                   /* HOST, 4 per batch */
    int64_t cap_pieces, cap_runs, cap_u_pieces, cap_u_runs;
} wr_hot_runs;
void wr_bprmf_hot_caps(int64_t batch_size, int32_t kind, int64_t *cap_pieces, int64_t *cap_runs);
int32_t wr_bprmf_plan_hot_runs(const int32_t *keys, int32_t kind, int64_t n_triplets, int64_t batch_size, int32_t *piece_q,
                               int32_t *piece_len, int32_t *run_q, int32_t *run_first, int32_t *run_np, int32_t *counts,
                               void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K1-K5 fused  One BaseRunner.fit iteration for BPRMF with torch.optim.SGD —
 *   zero_grad -> predict -> backward -> step  (BaseRunner.py:196-199, optimizer per :120-124)
 * Two kernels: (A) one 16-lane team per user row: gathers U[u], I[p], I[n], BPR loss + coefficient,
 * user-row gradient reduced over the user's triplets, U[u] updated in place, c_b*U[u] stashed;
 * (B) one team per item row: sums the stashed contributions of its occurrences, I[r] updated in place.
 *   w <- w - lr * (g + l2*w)  for rows in the batch.  Rows NOT in the batch are not touched here: with
 *   l2 != 0 call wr_sgd_decay_untouched afterwards (dense torch.optim.SGD weight_decay semantics).
 * stamp_u / stamp_i (int32 [n_rows], may be NULL when l2 == 0): set to step_id for rows in the batch.
 * loss_out: 1 float on device (may be NULL).  workspace >= wr_bprmf_step_workspace_bytes(B, D).
 * --------------------------------------------------------------------------------------------------- */
int64_t wr_bprmf_step_workspace_bytes(int64_t B, int32_t D);
int32_t wr_bprmf_step_sgd(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                          const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                          const int32_t *oc_src, int64_t B, float lr, float l2, int32_t *stamp_u, int32_t *stamp_i,
                          int32_t step_id, float *loss_out, const wr_hot_runs *hot, void *workspace,
                          int64_t workspace_bytes, void *stream);

/* Runs consecutive steps over batches [first_batch, first_batch + n_batches) of a plan built with
 * `batch_size` over `n_triplets` triplets (the native inner loop of BaseRunner.fit, BaseRunner.py:194-200).
 * loss_out[k] receives the loss of batch first_batch + k (may be NULL).  l2 must be 0 here.
 * phase_events (may be NULL): 4*n_batches hipEvent_t handles, any of them NULL (per-kernel timing).  For step k, events
 * 4k / 4k+1 are attached to the first / last kernel of the user phase as that dispatch's start / stop event, 4k+2 / 4k+3
 * likewise for the item phase: they carry the kernels' own timestamps (what rocprofv3 --kernel-trace reports) and no
 * marker packet is queued.  elapsed(4k, 4k+1) is the user phase of step k, elapsed(4k+1, 4k+3) launch to launch. */
int32_t wr_bprmf_run_sgd(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                         const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                         const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                         int64_t n_batches, float lr, float *loss_out, void *const *phase_events, const wr_hot_runs *hot,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* Chained step stream — the same consecutive steps as wr_bprmf_run_sgd (BaseRunner.py:194-200, l2 = 0, no hot rows) with
 * ONE launch per step on ONE stream: the item phase of step k-1 rides, as extra workgroups, in the launch that carries the
 * user phase of step k.  The runs of batch k that read an item row that item phase rewrites (the plan's deferred runs,
 * wr_bprmf_plan_overlap_marks) are taken by a few workgroups that first wait, inside the launch, for a counter the item
 * workgroups add to after their (write-through) row stores; every other run needs no ordering.  Every table row keeps one
 * writer per step and its summation order: tables bit-identical to wr_bprmf_run_sgd's; the loss of a step differs by the
 * association of its partial sums only (fixed by the plan, bitwise reproducible).
 *   tdef  [n_batches_of_plan * ceil(batch_size/32)] : per batch, bit t%32 of word t/32 = the run headed at sorted position
 *         t is deferred;  def_q [n_batches_of_plan * def_cap] : those positions, ascending;
 *   def_count_host (HOST memory, one int32 per batch of the plan) : their number.  A step whose batch has more than
 *         min(def_cap, def_limit) deferred runs (small item tables: almost every run) is issued as the two ordinary
 *         launches; the first step of a call always is, and the call ends with an ordinary item-phase launch — on return
 *         the stream holds complete steps only.
 *   phase_events (may be NULL): 4 handles per step as for wr_bprmf_run_sgd; [4k], [4k+1] are attached to the launch that
 *         carries the user phase of step k, [4k+2], [4k+3] to a separate item-phase launch of step k where there is one.
 *   workspace >= 2 * wr_bprmf_step_workspace_bytes(batch_size, D).
 *   sync: wr_bprmf_chain_sync_words(n_batches) int32 words of device memory, 16-byte aligned, owned by the caller: one
 *         (sharded) counter per step from word 0 (zeroed by the call) and, at word sync_words - 4, a sticky word the kernels set when a
 *         bounded wait expired (zero it once when allocating; non-zero afterwards = the run is invalid).
 * Rows must be whole 128-byte lines: D % 32 == 0 and both tables 128-byte aligned (wr_bprmf_chain_supported; WR_E_ALIGN). */
int32_t wr_bprmf_chain_supported(const float *user_tab, const float *item_tab, int32_t D);
int64_t wr_bprmf_chain_sync_words(int64_t n_batches);
int32_t wr_bprmf_run_sgd_chain(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                               const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                               const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                               int64_t n_batches, float lr, float *loss_out, const int32_t *tdef, const int32_t *def_q,
                               const int32_t *def_count_host, int64_t def_cap, int64_t def_limit, void *const *phase_events,
                               void *workspace, int64_t workspace_bytes, int32_t *sync, int64_t sync_words, void *stream);

/* Step stream WITHOUT a per-batch sort (wr_group.hip) — the same consecutive steps as wr_bprmf_run_sgd
 * (src/helpers/BaseRunner.py:194-200: zero_grad / predict / backward / SGD.step per batch, l2 = 0), one launch per step.
 * The reference loop does no per-batch index work; neither does this path beyond GROUPING what recurs in a batch:
 *   wr_group_plan_build    leaves the triplets (u, p, n: int32, batch order) where they are and writes, per batch, four
 *       flag bits per triplet (user shared in the batch / positive item shared / negative item shared / an item row is
 *       rewritten by the batch before) and the shared occurrences only, sorted by (row, position), as lists.  `plan` holds
 *       wr_group_plan_words(...) int32 words (0: shape not supported — batch_size > 131,072), 16-byte aligned, caller-owned.
 *       plan[0] != 0: an id was out of range (nn.Embedding would raise IndexError); plan[1] != 0: a list overflowed
 *       (tables small against the batch, or popularity-skewed ids) — the plan is NOT usable, build a sorted plan instead;
 *       plan[2] != 0: some row has more than 64 occurrences in a batch (usable, but slow: prefer the sorted plan's hot-row
 *       path).  Tables beyond 2^21 rows are hashed into 2^21 bits: a collision makes a row look shared, never the reverse.
 *   wr_bprmf_run_sgd_group  the steps of batches [first_batch, first_batch + n_batches) of that plan; `u, p, n` are the
 *       arrays the plan was built from.  The item rows with several occurrences in batch k are rewritten by workgroups
 *       that ride in the launch of step k+1 (write-through stores + an in-launch counter hand-off, as wr_bprmf_run_sgd_chain);
 *       the call ends with a launch of its own for the last batch's, so the stream holds complete steps on return.
 *       Every table row has one writer per step and a fixed summation order (bitwise reproducible); the order differs from
 *       the sorted plan's, so tables agree with wr_bprmf_run_sgd to rounding (1e-7 relative), not bit for bit.
 *       events (may be NULL): 2 handles per step, start / stop of the launch that carries step k's triplets.
 *       workspace >= wr_bprmf_group_workspace_bytes(batch_size, D); sync: wr_bprmf_group_sync_words(n_batches) int32 words,
 *       16-byte aligned; word sync_words - 4 is the sticky "a bounded wait expired" word (zero it once when allocating).
 *       Rows must be whole 128-byte lines (wr_bprmf_group_supported; WR_E_ALIGN otherwise). */
int64_t wr_group_plan_words(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items);
/* out[16] = {n_batches, flag words per batch, user ranges, item ranges, user hash mask, item hash mask, and the offsets (in
 * int32 words from `plan`) of: flags [nb][fw][4], user list lengths [nb][R_u], item list lengths [nb][R_i], user list rows,
 * user list sources, item list rows, item list sources [nb][R][cap]; total words; cap of a user segment; cap of an item
 * segment} — for tools and tests */
int32_t wr_group_plan_layout(int64_t n_triplets, int64_t batch_size, int64_t n_users, int64_t n_items, int64_t *out);
/* int64 index columns (the reference's batch layout, src/models/BaseModel.py:96-127) -> the int32 columns the group plan
 * and its step read; ids outside [0, 2^31) become -1 (reported by the plan as out of range). */
int32_t wr_narrow_ids_i64(const int64_t *u, const int64_t *p, const int64_t *n, int32_t *u32, int32_t *p32, int32_t *n32,
                          int64_t count, void *stream);
int32_t wr_group_plan_build(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size,
                            int64_t n_users, int64_t n_items, int32_t *plan, int64_t plan_words, void *stream);
int64_t wr_bprmf_group_workspace_bytes(int64_t batch_size, int32_t D);
int64_t wr_bprmf_group_sync_words(int64_t n_batches);
int32_t wr_bprmf_group_supported(const float *user_tab, const float *item_tab, int32_t D);
int32_t wr_bprmf_run_sgd_group(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                               const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size,
                               const int32_t *plan, int64_t plan_words, int64_t first_batch, int64_t n_batches, float lr,
                               float *loss_out, void *const *events, void *workspace, int64_t workspace_bytes, int32_t *sync,
                               int64_t sync_words, void *stream);

/* Plan-time marks for wr_bprmf_run_sgd_chain (index work only; call after the plan build, on the same stream).
 *   bitmap [n_batches * ceil(n_items/32)] (out): per batch, bit r = item row r has several occurrences in the batch;
 *   prev_bitmap: that bitmap of the batch BEFORE this plan's first batch, or NULL (then the first batch defers nothing);
 *   tdef, def_q as above (out); def_count [n_batches] (out, device): deferred runs per batch — may exceed def_cap, in which
 *   case only the first def_cap positions were stored and the caller must use wr_bprmf_run_sgd for this plan.
 * Every index is range-checked: safe on a plan whose hand-written builder reported a bucket overflow. */
int32_t wr_bprmf_plan_overlap_marks(const int32_t *tu, const int32_t *tp, const int32_t *tn, int64_t n_triplets,
                                    int64_t batch_size, int64_t n_items, const int32_t *prev_bitmap, int32_t *bitmap,
                                    int32_t *tdef, int32_t *def_q, int64_t def_cap, int32_t *def_count, void *stream);

/* The second half of wr_bprmf_plan_overlap_marks for a `bitmap` that is already complete (written by
 * wr_bprmf_plan_build_fast_marks_*): deferred-run mask, list and counts only. */
int32_t wr_bprmf_plan_overlap_deferred(const int32_t *tu, const int32_t *tp, const int32_t *tn, int64_t n_triplets,
                                       int64_t batch_size, int64_t n_items, const int32_t *prev_bitmap, const int32_t *bitmap,
                                       int32_t *tdef, int32_t *def_q, int64_t def_cap, int32_t *def_count, void *stream);

/* Same two kernels in gradient-emitting mode: instead of updating the tables, writes the reduced
 * gradient rows (embedding_dense_backward of BaseRunner.py:198) to grad_u[r,:] / grad_i[r,:] for rows in
 * the batch and sets stamp[r] = step_id.  Rows not in the batch are not written: a consumer treats
 * stamp[r] != step_id as a zero gradient row (wr_adam_dense / wr_sgd_dense below).  stamp_u / stamp_i may be
 * NULL (a caller that zero-filled grad_u / grad_i and reads them densely). */
int32_t wr_bprmf_grads(const float *user_tab, int64_t n_users, const float *item_tab, int64_t n_items, int32_t D,
                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                       const int32_t *oc_src, int64_t B, float *grad_u, float *grad_i, int32_t *stamp_u,
                       int32_t *stamp_i, int32_t step_id, float *loss_out, const wr_hot_runs *hot, void *workspace,
                       int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K5  optimizers over a whole table — torch.optim.{SGD,Adam}.step as built at BaseRunner.py:120-124.
 * grad rows are valid where stamp[r] == step_id and are zero elsewhere (stamp == NULL: all rows valid).
 * --------------------------------------------------------------------------------------------------- */
/* w <- w - lr*l2*w for rows with stamp[r] != step_id (completes weight decay after wr_bprmf_step_sgd) */
int32_t wr_sgd_decay_untouched(float *tab, int64_t n_rows, int32_t D, const int32_t *stamp, int32_t step_id, float lr,
                               float l2, void *stream);
/* g' = g + l2*w ; w <- w - lr*g' */
int32_t wr_sgd_dense(float *tab, int64_t n_rows, int32_t D, const float *grad, const int32_t *stamp, int32_t step_id,
                     float lr, float l2, void *stream);
/* Adam (amsgrad off): g' = g + l2*w; m = lerp(m,g',1-b1); v = b2 v + (1-b2) g'^2;
 * w <- w - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);  t = adam_step (1-based) */
int32_t wr_adam_dense(float *tab, float *exp_avg, float *exp_avg_sq, int64_t n_rows, int32_t D, const float *grad,
                      const int32_t *stamp, int32_t step_id, int64_t adam_step, float lr, float l2, float beta1,
                      float beta2, float eps, void *stream);

/* wr_adam_dense with the step number in device memory (consts as for wr_adam_rows_lazy: consts[2t] = lr/(1-beta1^t),
 * consts[2t+1] = 1/sqrt(1-beta2^t); t = step_dev[0], 1-based, must be < n_consts): the launch has no step-dependent
 * argument, so a hipGraph that captured a whole training step can be replayed; wr_counter_add advances the counter inside the
 * same graph.  All rows take the gradient (no stamps). */
int32_t wr_adam_dense_dev(float *tab, float *exp_avg, float *exp_avg_sq, int64_t n_rows, int32_t D, const float *grad,
                          const float *consts, int64_t n_consts, const int32_t *step_dev, float l2, float beta1, float beta2,
                          float eps, void *stream);

/* wr_adam_dense_dev for two tables of the same width in ONE launch (a model's user and item embeddings; one graph node and
 * one launch latency fewer per step).  Same arithmetic per element. */
int32_t wr_adam_dense_dev_pair(float *tab_a, float *exp_avg_a, float *exp_avg_sq_a, int64_t n_rows_a, const float *grad_a,
                               float *tab_b, float *exp_avg_b, float *exp_avg_sq_b, int64_t n_rows_b, const float *grad_b,
                               int32_t D, const float *consts, int64_t n_consts, const int32_t *step_dev, float l2,
                               float beta1, float beta2, float eps, void *stream);
int32_t wr_counter_add(int32_t *counter, int32_t delta, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K5 (lazy, exact)  The same optimizers without the table passes.  torch.optim.Adam / SGD(weight_decay) move every
 * row at every step (BaseRunner.py:120-124,199); between two batches that contain it a row's trajectory depends on
 * the row alone, so it is replayed when the row is next needed: last_step[r] = number of the last optimizer step
 * applied to row r (0 = none).  Same element arithmetic, order and per-step bias corrections as wr_adam_dense /
 * wr_sgd_dense: the tables come out bit-identical to the dense kernels'.
 *   step t of a batch:  wr_*_rows_lazy(grad = NULL) on the batch's rows  ->  gradients (wr_bprmf_grads / the fused SGD
 *   step)  ->  wr_adam_rows_lazy(grad) ;  before evaluation / checkpoint: wr_*_catchup_all.
 * keys: row ids of one batch with equal ids adjacent (the plan's tu / oc_item arrays); each distinct row is done once.
 * consts: device array, consts[2s] = lr/(1-beta1^s), consts[2s+1] = 1/sqrt(1-beta2^s) for s < n_consts (fill a host
 * copy with wr_adam_consts — the expressions wr_adam_dense evaluates on the host).
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_adam_consts(int64_t first_step, int64_t n_steps, float lr, float beta1, float beta2, float *consts_host);
/* grad == NULL: replay the rows up to step adam_step-1.  grad != NULL ([n_rows, D], rows of the batch valid): replay
 * up to adam_step-1 if still needed, then apply step adam_step with the gradient row. */
int32_t wr_adam_rows_lazy(float *tab, float *exp_avg, float *exp_avg_sq, int32_t *last_step, int64_t n_rows, int32_t D,
                          const int32_t *keys, int64_t n_keys, const float *grad, int64_t adam_step, const float *consts,
                          int64_t n_consts, float l2, float beta1, float beta2, float eps, void *stream);
/* every row up to and including step adam_step (zero gradients) */
int32_t wr_adam_catchup_all(float *tab, float *exp_avg, float *exp_avg_sq, int32_t *last_step, int64_t n_rows, int32_t D,
                            int64_t adam_step, const float *consts, int64_t n_consts, float l2, float beta1, float beta2,
                            float eps, void *stream);
/* SGD weight decay: replay w <- w - lr*l2*w on the batch's rows up to step-1 and mark them as done for `step` (the fused
 * step wr_bprmf_step_sgd applies step `step` itself, decay included, to exactly these rows).  lr and l2 must not have
 * changed since the rows' last step (call wr_sgd_catchup_all before changing them). */
int32_t wr_sgd_rows_lazy(float *tab, int32_t *last_step, int64_t n_rows, int32_t D, const int32_t *keys, int64_t n_keys,
                         int64_t step, float lr, float l2, void *stream);
int32_t wr_sgd_catchup_all(float *tab, int32_t *last_step, int64_t n_rows, int32_t D, int64_t step, float lr, float l2,
                           void *stream);

/* One Adam step on one batch with the update fused into the step kernels (the fused SGD step's two kernels with Adam where
 * they apply SGD): the team that finishes a row reads its two moment rows, takes torch.optim.Adam's step (adam_elem, the
 * arithmetic of wr_adam_dense) and writes weights and moments back; no gradient table.  Every row of the batch must be up to
 * date through adam_step - 1 (wr_adam_rows_lazy with grad = NULL first); last_* [n_rows] become adam_step for those rows. */
int32_t wr_bprmf_step_adam(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                           float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                           const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src, int64_t B,
                           int64_t adam_step, float lr, float l2, float beta1, float beta2, float eps, float *loss_out,
                           const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes, void *stream);
/* torch.optim.Adagrad (kind = 1) / torch.optim.Adadelta (kind = 2) with weight_decay = 0 fused into the step kernels, for
 * n_batches consecutive batches of a plan (the reference eval's --optimizer into torch.optim.<name>(params, lr,
 * weight_decay=l2), src/helpers/BaseRunner.py:34-37,120-124).  With a zero gradient neither optimizer moves a weight, so the
 * tables are always current: no catch-up before the gradients, no flush before evaluation.
 *   Adagrad : s1_* = state_sum tables (s2_*, last_* unused, may be NULL); exactly sparse.  eps = 1e-10 in torch.
 *   Adadelta: s1_* = square_avg, s2_* = acc_delta, last_* [n_rows] int32 = step of the row's last update (0 initially): a
 *             missed step multiplies both state rows by rho, replayed by the finisher before step t.  rho 0.9, eps 1e-6.
 * step0 = optimizer step number of the first batch (1-based).  Hot rows are handled (pieces + combine) like in the SGD step. */
int32_t wr_bprmf_run_stateful(int32_t kind, float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                              float *s1_u, float *s2_u, float *s1_i, float *s2_i, int32_t *last_u, int32_t *last_i,
                              const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                              const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                              int64_t n_batches, int64_t step0, float lr, float rho, float eps, float *loss_out,
                              const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes, void *stream);
/* The same step with the catch-up FOLDED into the row loads: rows need NOT be up to date — every team that loads a row
 * replays the missed zero-gradient steps last[row]+1 .. adam_step-1 on its register copy (consts: the table of
 * wr_adam_consts, entries 0..adam_step), the finisher applies step adam_step and writes weights and moments once: 3 row
 * reads + 3 row writes per touched row instead of the 12 transfers of wr_adam_rows_lazy + wr_bprmf_step_adam.  Same bits
 * as the dense optimizer.  Batches with hot rows are refused (WR_E_RANGE): use the separate catch-up for those. */
int32_t wr_bprmf_step_adam_folded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                                  float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                                  const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                                  int64_t B, int64_t adam_step, float lr, const float *consts, int64_t n_consts, float l2,
                                  float beta1, float beta2, float eps, float *loss_out, void *workspace,
                                  int64_t workspace_bytes, void *stream);
/* wr_bprmf_run_adam_lazy with wr_bprmf_step_adam_folded for every batch without hot rows (same arguments, same results) */
int32_t wr_bprmf_run_adam_folded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                                 float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                                 const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                                 int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches,
                                 int64_t adam_step0, float lr, const float *consts, int64_t n_consts, float l2, float beta1,
                                 float beta2, float eps, float *loss_out, const wr_hot_runs *hot, void *workspace,
                                 int64_t workspace_bytes, void *stream);
/* wr_bprmf_run_adam_folded as ONE launch per step (round 3): the item phase of step k-1 — the rows that recur in batch k-1:
 * weights, both moments and the step stamp, stored write-through — rides in the launch of step k's user phase; the user runs
 * of batch k that read such a row wait for it inside the launch (the chained launch of wr_bprmf_run_sgd_chain, same marks:
 * tdef / def_q / def_count_host / def_cap / def_limit from wr_bprmf_plan_overlap_deferred).  No batch of the range may have
 * hot rows (the caller takes wr_bprmf_run_adam_folded for such plans).  workspace: 2 x wr_bprmf_step_workspace_bytes; sync:
 * wr_bprmf_chain_sync_words(n_batches).  Same bits as the dense optimizer. */
int32_t wr_bprmf_run_adam_folded_chain(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                       float *m_u, float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i,
                                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                       const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                       int64_t n_batches, int64_t adam_step0, float lr, const float *consts, int64_t n_consts,
                                       float l2, float beta1, float beta2, float eps, float *loss_out, const int32_t *tdef,
                                       const int32_t *def_q, const int32_t *def_count_host, int64_t def_cap, int64_t def_limit,
                                       void *workspace, int64_t workspace_bytes, int32_t *sync, int64_t sync_words,
                                       void *stream);
/* wr_bprmf_run_adam_lazy with a bounded lag: before every step a rotating window of ceil(rows / max_lag) consecutive rows
 * of each table is brought to the previous step (wr_adam_catchup_all on the sub-range), so that no row ever misses more
 * than max_lag steps and the replays of a batch's rows stay short (small batches on big tables: a geometric tail of
 * thousands of missed steps otherwise makes every catch-up launch wait for its unluckiest wave).  Same bits as the dense
 * optimizer.  sweep_pos (HOST memory, in/out): [0] next user row, [1] next item row of the window; start at {0, 0}. */
int32_t wr_bprmf_run_adam_lazy_bounded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                       float *m_u, float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i,
                                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                       const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                       int64_t n_batches, int64_t adam_step0, float lr, const float *consts, int64_t n_consts,
                                       float l2, float beta1, float beta2, float eps, float *loss_out, const wr_hot_runs *hot,
                                       int64_t max_lag, int64_t *sweep_pos, void *workspace, int64_t workspace_bytes,
                                       void *stream);
/* The bounded-lag forms of wr_bprmf_run_sgd_lazy and wr_bprmf_run_stateful (Adadelta), as wr_bprmf_run_adam_lazy_bounded:
 * a rotating window of ceil(rows / max_lag) rows per table takes the zero-gradient steps it missed before every step
 * (wr_sgd_catchup_all / wr_adadelta_decay_all on the sub-range).  Same bits as without the window. */
int32_t wr_bprmf_run_sgd_lazy_bounded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                      int32_t *last_u, int32_t *last_i, int32_t *stamp_u, int32_t *stamp_i, int32_t step_id0,
                                      const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                      const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                      int64_t n_batches, int64_t step0, float lr, float l2, float *loss_out,
                                      const wr_hot_runs *hot, int64_t max_lag, int64_t *sweep_pos, void *workspace,
                                      int64_t workspace_bytes, void *stream);
int32_t wr_bprmf_run_stateful_bounded(int32_t kind, float *user_tab, int64_t n_users, float *item_tab, int64_t n_items,
                                      int32_t D, float *s1_u, float *s2_u, float *s1_i, float *s2_i, int32_t *last_u,
                                      int32_t *last_i, const int32_t *tu, const int32_t *tp, const int32_t *tn,
                                      const int32_t *oc_item, const int32_t *oc_src, int64_t n_triplets, int64_t batch_size,
                                      int64_t first_batch, int64_t n_batches, int64_t step0, float lr, float rho, float eps,
                                      float *loss_out, const wr_hot_runs *hot, int64_t max_lag, int64_t *sweep_pos,
                                      void *workspace, int64_t workspace_bytes, void *stream);
/* Adadelta's zero-gradient steps on rows [0, n_rows): both state rows times rho once per missed step (step - last_step[r]
 * multiplications), last_step[r] = step. */
int32_t wr_adadelta_decay_all(float *square_avg, float *acc_delta, int32_t *last_step, int64_t n_rows, int32_t D, int64_t step,
                              float rho, void *stream);
/* n_batches consecutive optimizer steps over batches [first_batch, first_batch + n_batches) of a plan, issued from native
 * code (the inner loop of BaseRunner.fit, BaseRunner.py:196-199, with the lazy optimizers): per batch
 *   Adam:  wr_adam_rows_lazy(NULL) on U and I rows -> wr_bprmf_step_adam (gradients + Adam on the rows it finishes);
 *   SGD + weight decay:  wr_sgd_rows_lazy on U and I rows -> wr_bprmf_step_sgd.
 * adam_step0 / step0 = optimizer step number of the first batch (1-based); step_id0 = stamp id of the first batch;
 * loss_out[k] receives batch k's loss (may be NULL). */
int32_t wr_bprmf_run_adam_lazy(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                               float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                               const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                               int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches,
                               int64_t adam_step0, float lr, const float *consts, int64_t n_consts, float l2, float beta1,
                               float beta2, float eps, float *loss_out, const wr_hot_runs *hot, void *workspace,
                               int64_t workspace_bytes, void *stream);
int32_t wr_bprmf_run_sgd_lazy(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, int32_t *last_u,
                              int32_t *last_i, int32_t *stamp_u, int32_t *stamp_i, int32_t step_id0, const int32_t *tu,
                              const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                              int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches, int64_t step0,
                              float lr, float l2, float *loss_out, const wr_hot_runs *hot, void *workspace,
                              int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K9  nn.Embedding forward / embedding_dense_backward with padding_idx —
 *     src/models/sequential/SASRec.py:60,84,105-106; also the row-exchange primitive of the
 *     row-sharded multi-GPU step.
 * out[k,:] = tab[idx[k],:]                                        (gather; padding rows are read as stored)
 * grad[idx[k],:] += src[k,:] for idx[k] != padding_idx            (deterministic: one writer per row, position order)
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_gather_rows(const float *tab, int64_t n_rows, int32_t D, const int64_t *idx, int64_t n, float *out,
                       void *stream);
int64_t wr_scatter_add_workspace_bytes(int64_t n, int64_t n_rows);
int32_t wr_scatter_add_rows(float *grad, int64_t n_rows, int32_t D, const int64_t *idx, const float *src, int64_t n,
                            int64_t padding_idx, float alpha, void *workspace, int64_t workspace_bytes, void *stream);

/* The same scatter-add through a ROW PLAN instead of a sort of the positions (wr_scatter.hip; tables of any size, segments
 * of at most 2^18 positions).  The plan is index work only, so a caller that knows the indices of several calls ahead — the
 * row-sharded step knows the rows it will serve for a whole chunk of steps — builds ONE plan for all of them (segment s =
 * idx[s * seg_stride .. + seg_len[s]), seg_len on the device or NULL = every segment full) and applies segment after
 * segment: one launch per call.  Per segment: a flag bit per position (its row recurs in the segment) and the recurring
 * positions ordered by (row, position); rows that occur once are added in place, runs are summed in position order (the
 * bits of wr_scatter_add_rows).  plan[1] != 0 after a build: some range of rows held more than 8,192 recurring positions and
 * is summed by brute force (slow, exact).  wr_scatter_add_rows itself takes this path for tables beyond 16,383 rows.
 * wr_scatter_plan_words returns 0 when the plan does not apply (segments longer than 2^18 positions). */
int64_t wr_scatter_plan_words(int64_t n_segments, int64_t seg_stride, int64_t n_rows);
int32_t wr_scatter_plan_build(const int64_t *idx, int64_t n_segments, int64_t seg_stride, const int32_t *seg_len, int64_t n_rows,
                              int64_t padding_idx, int32_t *plan, int64_t plan_words, void *stream);
int32_t wr_scatter_add_planned(float *table, int64_t n_rows, int32_t D, const int64_t *idx, int64_t n_segments,
                               int64_t seg_stride, int64_t segment, int64_t n, int64_t padding_idx, const float *src,
                               float alpha, const int32_t *plan, int64_t plan_words, void *stream);

/* tab[sorted_rows[q], :] += alpha * sum of src[perm[q'], :] over the run of equal sorted_rows (rows ascending; the
 * order was fixed when the exchange was planned, so the sum is reproducible).  Entries >= n_rows are skipped.
 * Owner-side application of gradient rows received from other shards (alpha = -lr for SGD). */
int32_t wr_apply_rows_sorted(float *tab, int64_t n_rows, int32_t D, const int32_t *sorted_rows, const int32_t *perm,
                             const float *src, int64_t n, float alpha, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Row-sharded step (multi-GPU) on a SORTED batch plan — the form for batches the group plan does not take (popularity-skewed
 * ids, shards small against the batch; wr_bprmf_shard_step_group is the usual one): the same two kernels with the user rows
 * of this shard updated in place (SGD, l2 = 0).  `item_rows` [n_rows, D] = the shard's own item rows [0, n_local_items),
 * rewritten in place, followed by the rows received from their owners for this step (tp/tn/oc_item hold row ids into
 * it); `grad_slots` [n_rows - n_local_items, D] receives the reduced gradient row of every received row (to be returned
 * to the owners).  The loss term and coefficient are scaled by 1/global_batch (the mean runs over all shards' triplets);
 * loss_partial = this shard's share.
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_bprmf_shard_step(float *user_shard, int64_t n_user_rows, float *item_rows, int64_t n_rows, int64_t n_local_items,
                            int32_t D, const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                            const int32_t *oc_src, int64_t B, int64_t global_batch, float lr, float *grad_slots,
                            float *loss_partial, const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes,
                            void *stream);

/* Row-sharded step without a per-batch sort (whisprrec_amd/sharded.py; wr_shard.hip + wr_group.hip) — ONE BaseRunner.fit
 * iteration (src/helpers/BaseRunner.py:196-199) over the union of the ranks' batches, negatives from ALL items
 * (src/models/BaseModel.py:168,174).  Item row i lives on rank i % world at local row i / world; a rank owns the users of its
 * triplets.  A rank's LOCAL item rows are read and rewritten in its shard in place; REMOTE ones are received from their
 * owners into rows [n_local_items, n_local_items + slots) of the same buffer (`item_ext`) and their gradient rows go back.
 *   wr_shard_route   index work for a chunk of batches: vu / vp / vn = the triplets with local user rows and "virtual" item
 *       ids (local row, or n_local_items + slot; a step's distinct remote items get slots 0, 1, ... in ascending (owner, row)
 *       order); send_rows [world][n_batches][list_cap] / send_cnt [world][n_batches] = per owner and batch the requested
 *       local rows (ascending) and their number — exchanged by two fixed-size all-to-alls.  rows_per_owner: a multiple of
 *       32 with world * rows_per_owner >= n_items.  err[0] != 0: an id out of range or a user of another rank; err[1] != 0:
 *       a request list beyond list_cap.
 *   wr_shard_pack    after the exchange (recv_rows / recv_cnt in the same layout, indexed by requester): per batch the rows
 *       to serve, requester by requester (serve_rows [n_batches][serve_stride], int64) and where each requester's rows
 *       start (serve_off [n_batches][world + 1]).  err[0] != 0: a peer asked for a row this shard does not have.
 *   wr_bprmf_shard_step_group   batch `batch` of a group plan built on (vu, vp, vn) with n_items = n_ext_rows: user rows and
 *       local item rows updated in place (SGD, l2 = 0), grad_slots[s] = the summed gradient row of slot s; coefficients and
 *       the loss share are scaled by 1 / global_batch.  Two launches (the triplets, then the batch's own tiles).
 *       workspace / sync as for wr_bprmf_run_sgd_group (sync: at least wr_bprmf_group_sync_words(1) words). */
int32_t wr_shard_route(const int32_t *u, const int32_t *p, const int32_t *n, int64_t n_triplets, int64_t batch_size, int32_t world,
                       int32_t rank, int64_t n_users, int64_t n_items, int64_t rows_per_owner, int64_t n_local_items,
                       int64_t list_cap, int32_t *vu, int32_t *vp, int32_t *vn, int32_t *send_rows, int32_t *send_cnt,
                       int32_t *err, void *stream);
int32_t wr_shard_pack(const int32_t *recv_rows, const int32_t *recv_cnt, int64_t n_batches, int32_t world, int64_t list_cap,
                      int64_t n_local_items, int64_t *serve_rows, int64_t serve_stride, int32_t *serve_off, int32_t *err,
                      void *stream);
int32_t wr_bprmf_shard_step_group(float *user_shard, int64_t n_user_rows, float *item_ext, int64_t n_ext_rows,
                                  int64_t n_local_items, int32_t D, const int32_t *vu, const int32_t *vp, const int32_t *vn,
                                  int64_t n_triplets, int64_t batch_size, const int32_t *plan, int64_t plan_words, int64_t batch,
                                  int64_t global_batch, float lr, float *grad_slots, float *loss_partial, void *workspace,
                                  int64_t workspace_bytes, int32_t *sync, int64_t sync_words, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K6-K7  LightGCN propagation — src/models/general/LightGCN.py:134-148
 *   Y = A X for the CSR form of the normalised bipartite adjacency (the reference multiplies the dense
 *   form of the same matrix, LightGCN.py:120,139).  If acc != NULL: acc += Y fused (running layer sum
 *   for the mean over layers, LightGCN.py:142-143).  row_ptr int64 [n_rows+1], col int32, val fp32.
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_spmm_csr(int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const float *val, const float *X,
                    int32_t D, float *Y, float *acc, void *stream);
/* Load-balanced form for power-law graphs: the rows are pre-cut (once, on the host: the adjacency is static) into
 * consecutive chunks of at most L non-zeros; chunk c covers non-zeros [chunk_ptr[c], chunk_ptr[c+1]) of row chunk_row[c],
 * every row has at least one (possibly empty) chunk, the chunks of a row are consecutive.  partials: fp32 [n_chunks, D]
 * scratch.  Rows with several chunks are summed in chunk order (reproducible). */
int32_t wr_spmm_csr_chunked(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                            const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                            float *partials, void *stream);

/* Hybrid product for graphs with a dense head (popular items): the block (all users) x (head items) of the adjacency and
 * its transpose are kept DENSE and multiplied on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32), the rest stays on
 * the chunked CSR kernels.  Per 32-row output tile the dense block is stored as A_T[tile][k][32] (value of tile row r and
 * column k at [k][r], zeros included), k = 0..K_pad-1 with cols[k] the node id of column k (padding columns: -1 —
 * they contribute exact zeros whatever X holds — or any valid node with zero values); rows[tile*32 + r] is the node id of tile row r (-1: padding row).  K_pad is cut into splits of
 * k_per_split columns (a multiple of 64, at most 1024), one workgroup each; with one split the tile is written to Y (acc must be NULL:
 * such rows also have a CSR part, which adds the layer sum), with several the split tiles go to `partials`
 * (wr_spmm_dense_partials_bytes) and are added in split order, then Y[row] is written and acc[row] += it.  D in
 * {32, 64, 96, 128}.  Summation order differs from CSR order: results agree with wr_spmm_csr to fp32 rounding (1e-6
 * relative on unit-scale data), and are bitwise reproducible.
 * wr_spmm_csr_chunked_modes: the CSR part — row_mode[row] (int8): 0 = write Y[row] (and acc), 1 = ADD to the Y[row] the
 * dense tiles wrote before (then acc), 2 = skip the row (it belongs to the dense tiles). */
int64_t wr_spmm_dense_partials_bytes(int64_t n_tiles, int64_t K_pad, int64_t k_per_split, int32_t D);
typedef struct wr_dense_group {
    const float *A_T;        /* [n_tiles][K_pad][32] */
    const int32_t *cols;     /* [K_pad] node id of column k */
    const int32_t *rows;     /* [n_tiles*32] node id of tile row r, -1 = padding */
    float *partials;         /* >= wr_spmm_dense_partials_bytes(...) when K_pad > k_per_split, else may be NULL */
    int64_t K_pad, k_per_split, n_tiles;
} wr_dense_group;
/* One launch for both tile groups of the hybrid product (either may be NULL): `unsplit` (K_pad == k_per_split: tiles written
 * straight to Y; acc is NOT touched for them) and `split` (K_pad > k_per_split: split tiles to its partials, then Y and acc). */
int32_t wr_spmm_dense_tiles(const wr_dense_group *unsplit, const wr_dense_group *split, const float *X, int64_t n_nodes,
                            int32_t D, float *Y, float *acc, void *stream);
int32_t wr_spmm_csr_chunked_modes(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                  const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                                  float *partials, const int8_t *row_mode, void *stream);
/* The same product with the number of combine levels chosen by the caller (row_mode may be NULL): 2 = chunks of a row are
 * first added inside aligned groups of 16 chunk slots, then the group leaders (a hub row with hundreds of chunks is not one
 * team's chain of hundreds of dependent adds); 1 = the row's first chunk adds all the others itself — one launch less,
 * right while no row has more than a few dozen chunks.  The summation order (hence the rounding) differs between the two;
 * each is fixed and reproducible.
 * acc (may be NULL) is the running layer sum: acc[r] = (base[r] + Y[r]) * acc_scale with base = acc, or = X when
 * acc_from_x != 0 (first layer: no copy of the input into acc beforehand); acc_scale = 1 except for the last layer's
 * 1 / (layers + 1) (LightGCN.py:142-143: mean over the layers). */
int32_t wr_spmm_csr_chunked_levels(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                   const int32_t *col, const float *val, const float *X, int32_t D, float *Y, float *acc,
                                   float *partials, const int8_t *row_mode, int32_t levels, int32_t acc_from_x,
                                   float acc_scale, void *stream);

/* The same product (one combine level) in ONE launch: the chunk of a cut row that finishes last adds the row's partials
 * itself (in chunk order: the bits of the two-launch form) instead of a combine launch doing it.  row_span[2r], [2r + 1]:
 * first chunk and number of chunks of row r; counters: n_rows words, zero before the first call (every call leaves them
 * zero).  Rows must be whole 128-byte lines: wr_spmm_fused_supported(D, partials) != 0 (D % 32 == 0, partials 128-byte
 * aligned); rows cut into more than a few dozen chunks are better served by the two-level form. */
int32_t wr_spmm_fused_supported(int32_t D, const float *partials);
int32_t wr_spmm_csr_chunked_fused(int64_t n_rows, int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row,
                                  const int32_t *row_span, const int32_t *col, const float *val, const float *X, int32_t D,
                                  float *Y, float *acc, float *partials, const int8_t *row_mode, int32_t acc_from_x,
                                  float acc_scale, uint32_t *counters, void *stream);
/* out = alpha * x  /  y += alpha * x  over numel floats (layer-mean scaling, gradient accumulation) */
int32_t wr_axpy(float *y, const float *x, int64_t numel, float alpha, int32_t overwrite, void *stream);
/* EmbLoss pieces (src/utils/loss.py:94-98): sq[0..2] = sum of squares of the gathered U[u], I[p], I[n] blocks */
int32_t wr_embloss_sumsq(const float *user_tab, const float *item_tab, int32_t D, const int64_t *u, const int64_t *p,
                         const int64_t *n, int64_t B, float *sq3, void *workspace, int64_t workspace_bytes,
                         void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K10  Full-ranking evaluation — BaseRunner.interface + evaluate_method (src/helpers/BaseRunner.py:218-258, 50-92) over
 *      full_predict (src/models/general/BPRMF.py:82-91), without materialising the [n, n_items] score matrix:
 *   rank[i] = 1 + #{ j not in mask(eval_user[i]) : <user_mat[eval_user[i]], item_tab[j]>  >  target_score[i] }
 *   target_score[i] = <user_mat[eval_user[i]], item_tab[eval_target[i]]>
 * mask_ptr int64 [n_user_rows+1] / mask_idx int32 (ascending per user): the user's train + dev + test items, which the
 * reference sets to -inf (BaseRunner.py:246-255); both NULL = no masking (--test_all 0).  Scores are computed on the matrix
 * cores with v_mfma_f32_32x32x2_f32 (exact fp32 k-ordered fma chain).  rank: int32 [n]; target_score: fp32 [n] (output).
 * --------------------------------------------------------------------------------------------------- */
int32_t wr_rank_eval(const float *user_mat, int64_t n_user_rows, const float *item_tab, int64_t n_items, int32_t D,
                     const int64_t *eval_user, const int64_t *eval_target, int64_t n, const int64_t *mask_ptr,
                     const int32_t *mask_idx, int32_t *rank, float *target_score, void *stream);

/* LightGCN.predict's per-batch tail in two launches (src/models/general/LightGCN.py:156-175, src/utils/loss.py:37-39,94-98):
 *   loss[0] = mean_b( -log(1e-10 + sigmoid(<Ua[u_b], Ia[p_b]> - <Ua[u_b], Ia[n_b]>)) )
 *             + reg_weight * (||U0[u]||_F + ||I0[p]||_F + ||I0[n]||_F) / B
 * with Ua / Ia the propagated tables and U0 / I0 the ego tables; sq3[3] receives the three sums of squares (the backward pass
 * needs them: wr_embloss_grad).  Partials are folded in a fixed order: bitwise reproducible. */
int64_t wr_lightgcn_loss_workspace_bytes(int64_t B);
int32_t wr_lightgcn_loss(const float *user_all, const float *item_all, const float *user_ego, const float *item_ego,
                         int64_t n_users, int64_t n_items, int32_t D, const int64_t *u, const int64_t *p, const int64_t *n,
                         int64_t B, float reg_weight, float *loss, float *sq3, void *workspace, int64_t workspace_bytes,
                         void *stream);
/* Backward of EmbLoss for one planned batch: grad_user[u,:] += m_u * w/(B*sqrt(sq3[0])) * user_tab[u,:] for a user with m_u
 * triplets in the batch; grad_item[r,:] += (m_pos * w/(B*sqrt(sq3[1])) + m_neg * w/(B*sqrt(sq3[2]))) * item_tab[r,:].
 * tu / oc_item / oc_src are the plan arrays of that batch (oc_item must hold plain row ids), sq3 the device output of
 * wr_embloss_sumsq: no host round trip. */
int32_t wr_embloss_grad(const float *user_tab, const float *item_tab, int32_t D, const int32_t *tu, const int32_t *oc_item,
                        const int32_t *oc_src, int64_t B, const float *sq3, float reg_weight, float *grad_user,
                        float *grad_item, void *stream);

/* LightGCN.predict + its backward as ONE call (src/models/general/LightGCN.py:134-175; loss.backward() of
 * src/helpers/BaseRunner.py:198): propagation of cat(user_tab, item_tab) through `n_layers` products with the chunked CSR
 * adjacency (layer mean folded in; `levels` as for wr_spmm_csr_chunked_levels), BPR + reg_weight * EmbLoss -> loss[0] (shape
 * (1,) like the reference's), and the dense gradient of that loss w.r.t. both tables -> grad [n_users + n_items, D] (user
 * rows first).  The same kernels, in the same order, that the separate entry points run: same bits.  B <=
 * wr_bprmf_plan_small_max_batch().  trusted_indices != 0: u / p / n were range-checked by the caller (no error flag is
 * written); else err_flag[0] != 0 reports an id out of range.  No host round trip: capturable into a hipGraph. */
int64_t wr_lightgcn_step_workspace_bytes(int64_t n_users, int64_t n_items, int32_t D, int64_t n_chunks, int64_t B);
int32_t wr_lightgcn_step(const float *user_tab, const float *item_tab, int64_t n_users, int64_t n_items, int32_t D,
                         int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row, const int32_t *col,
                         const float *val, int32_t levels, int32_t n_layers, const int64_t *u, const int64_t *p, const int64_t *n,
                         int64_t B, float reg_weight, int32_t trusted_indices, float *loss, float *grad, int32_t *err_flag,
                         void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WHISPRREC_HIP_H */
