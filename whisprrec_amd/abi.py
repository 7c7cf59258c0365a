"""ctypes binding of libwhisprrec_hip.so (the C-ABI declared in include/whisprrec_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhisprrec_hip.so")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/whisprrec_hip.h one to one (tests/test_abi.py checks the
# header against this table).  Pointers are passed as integers (tensor.data_ptr()).
SIGNATURES = {
    "wr_abi_version": (c_i32, []),
    "wr_last_error": (ctypes.c_char_p, []),
    "wr_device_info": (c_i32, [c_vp, c_vp, c_vp, c_i32]),
    "wr_bpr_fwd_workspace_bytes": (c_i64, [c_i64]),
    "wr_bpr_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                           c_vp]),
    "wr_bprmf_plan_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "wr_bprmf_plan_build_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_build_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_small_max_batch": (c_i64, []),
    "wr_bprmf_plan_build_small_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp]),
    "wr_bprmf_plan_build_small_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp]),
    "wr_bprmf_plan_fast_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "wr_bprmf_plan_fast_mapped_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64, c_i32, c_i32]),
    "wr_bprmf_plan_build_fast_mapped_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_build_fast_mapped_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_build_fast_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_build_fast_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_sample_negatives_i64": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp,
                                        c_vp]),
    "wr_sample_negatives_i32": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp,
                                        c_vp]),
    "wr_epoch_prepare_range_i64": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_i64, c_i64,
                                           c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_prepare_range_i32": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_i64, c_i64,
                                           c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_pairset_capacity": (c_i64, [c_i64]),
    "wr_pairset_build": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "wr_sample_negatives_set_i64": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp,
                                            c_vp]),
    "wr_sample_negatives_set_i32": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp,
                                            c_vp]),
    "wr_epoch_prepare_range_set_i64": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_i64,
                                               c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_prepare_range_set_i32": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_i64,
                                               c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_prepare_range_packed_i64": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_i64,
                                                  c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_prepare_range_packed_i32": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_i64,
                                                  c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_shuffle_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_epoch_shuffle_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_bprmf_step_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "wr_bprmf_hot_caps": (None, [c_i64, c_i32, c_vp, c_vp]),
    "wr_bprmf_plan_hot_runs": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_bprmf_step_sgd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32,
                                  c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_sgd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64,
                                 c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_chain_supported": (c_i32, [c_vp, c_vp, c_i32]),
    "wr_bprmf_chain_sync_words": (c_i64, [c_i64]),
    "wr_bprmf_run_sgd_chain": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64,
                                       c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64,
                                       c_vp]),
    "wr_lightgcn_step_workspace_bytes": (c_i64, [c_i64, c_i64, c_i32, c_i64, c_i64]),
    "wr_lightgcn_step": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp,
                                 c_i64, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_shard_route": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp,
                               c_vp, c_vp, c_vp, c_vp]),
    "wr_shard_pack": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "wr_bprmf_shard_step_group": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64,
                                          c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "wr_group_plan_words": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "wr_group_plan_layout": (c_i32, [c_i64, c_i64, c_i64, c_i64, c_vp]),
    "wr_narrow_ids_i64": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_group_plan_build": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "wr_bprmf_group_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "wr_bprmf_group_sync_words": (c_i64, [c_i64]),
    "wr_bprmf_group_supported": (c_i32, [c_vp, c_vp, c_i32]),
    "wr_bprmf_run_sgd_group": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                       c_i64, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "wr_bprmf_plan_overlap_deferred": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                               c_vp]),
    "wr_bprmf_plan_fast_marks_supported": (c_i32, [c_i64, c_i64, c_i64, c_i64]),
    "wr_bprmf_plan_build_fast_marks_i64": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                   c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "wr_bprmf_plan_build_fast_marks_i32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                   c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "wr_bprmf_plan_overlap_marks": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                            c_vp]),
    "wr_bprmf_grads": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp,
                               c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_sgd_decay_untouched": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_f32, c_f32, c_vp]),
    "wr_sgd_dense": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_f32, c_f32, c_vp]),
    "wr_adam_dense": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32,
                              c_f32, c_vp]),
    "wr_adam_dense_dev": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "wr_adam_dense_dev_pair": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp,
                                       c_f32, c_f32, c_f32, c_f32, c_vp]),
    "wr_counter_add": (c_i32, [c_vp, c_i32, c_vp]),
    "wr_adam_consts": (c_i32, [c_i64, c_i64, c_f32, c_f32, c_f32, c_vp]),
    "wr_adam_rows_lazy": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32,
                                  c_f32, c_f32, c_vp]),
    "wr_adam_catchup_all": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32,
                                    c_vp]),
    "wr_sgd_rows_lazy": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_i64, c_f32, c_f32, c_vp]),
    "wr_sgd_catchup_all": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_f32, c_vp]),
    "wr_bprmf_step_adam": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                   c_i64, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_stateful": (c_i32, [c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_i64,
                                      c_vp]),
    "wr_bprmf_step_adam_folded": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_i64, c_f32, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_i64,
                                          c_vp]),
    "wr_bprmf_run_adam_folded": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32,
                                         c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_adam_folded_chain": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_i64, c_f32, c_f32,
                                               c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64,
                                               c_vp]),
    "wr_bprmf_run_sgd_lazy_bounded": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp,
                                              c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_vp, c_vp, c_i64,
                                              c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_stateful_bounded": (c_i32, [c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_f32,
                                              c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "wr_adadelta_decay_all": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_vp]),
    "wr_bprmf_run_adam_lazy_bounded": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_i64, c_f32,
                                               c_f32, c_f32, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_adam_lazy": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32,
                                       c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_bprmf_run_sgd_lazy": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                      c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_gather_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "wr_scatter_add_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "wr_scatter_add_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_i64, c_vp]),
    "wr_scatter_plan_words": (c_i64, [c_i64, c_i64, c_i64]),
    "wr_scatter_plan_build": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "wr_scatter_add_planned": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_f32, c_vp, c_i64,
                                       c_vp]),
    "wr_apply_rows_sorted": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp]),
    "wr_bprmf_shard_step": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32,
                                    c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wr_spmm_csr": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "wr_spmm_csr_chunked": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "wr_spmm_dense_partials_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i32]),
    "wr_spmm_dense_tiles": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "wr_spmm_csr_chunked_levels": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                           c_i32, c_f32, c_vp]),
    "wr_spmm_fused_supported": (c_i32, [c_i32, c_vp]),
    "wr_spmm_csr_chunked_fused": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                          c_f32, c_vp, c_vp]),
    "wr_spmm_csr_chunked_modes": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_axpy": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_i32, c_vp]),
    "wr_rank_eval": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wr_lightgcn_loss_workspace_bytes": (c_i64, [c_i64]),
    "wr_lightgcn_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp,
                                 c_i64, c_vp]),
    "wr_embloss_grad": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "wr_embloss_sumsq": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
}



class HotRuns(ctypes.Structure):
    """struct wr_hot_runs of include/whisprrec_hip.h"""
    _fields_ = [("piece_q", c_vp), ("piece_len", c_vp), ("run_q", c_vp), ("run_first", c_vp), ("run_np", c_vp),
                ("u_piece_q", c_vp), ("u_piece_len", c_vp), ("u_run_q", c_vp), ("u_run_first", c_vp), ("u_run_np", c_vp),
                ("counts_host", c_vp), ("cap_pieces", c_i64), ("cap_runs", c_i64), ("cap_u_pieces", c_i64),
                ("cap_u_runs", c_i64)]


class DenseGroup(ctypes.Structure):
    """struct wr_dense_group of include/whisprrec_hip.h"""
    _fields_ = [("A_T", c_vp), ("cols", c_vp), ("rows", c_vp), ("partials", c_vp), ("K_pad", c_i64), ("k_per_split", c_i64),
                ("n_tiles", c_i64)]


class BucketSide(ctypes.Structure):
    """struct wr_bucket_side of include/whisprrec_hip.h"""
    _fields_ = [("n_buckets", c_i32), ("row_bucket", c_vp), ("bucket_start", c_vp), ("bucket_rows", c_vp), ("bucket_sub", c_vp)]


_lib = None


class WhisprRecHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WhisprRecHipError(
                "libwhisprrec_hip.so is missing at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C whisprrec_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7; the process must hold exactly one HIP runtime so that the
        # device pointers and hipStream_t handles PyTorch hands out mean the same thing inside this library.
        # Importing torch first makes the dynamic loader bind our NEEDED libamdhip64.so.7 to that copy.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.wr_abi_version() != 1:
            raise WhisprRecHipError("ABI version mismatch: library %d, binding 1" % handle.wr_abi_version())
        _lib = handle
    return _lib


def last_error():
    return lib().wr_last_error().decode("utf-8", "replace")


def check(rc, what):
    """0 ok; <0 argument error; >0 hipError_t — both raise with the library's message."""
    if rc != 0:
        raise WhisprRecHipError("%s failed (rc=%d): %s" % (what, rc, last_error()))


def check_size(nbytes, what):
    if nbytes < 0:
        raise WhisprRecHipError("%s failed (rc=%d): %s" % (what, nbytes, last_error()))
    return int(nbytes)
