"""ctypes loader for librdmnet_hip.so.  There is no CPU fallback: a missing library or a failing
call raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RDM_LIB_PATH') or os.path.join(_HERE, 'librdmnet_hip.so')  # override: A/B builds

c_void = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_size = ctypes.c_size_t

ABI_VERSION = 2  # = RDM_ABI_VERSION of include/rdmnet_hip.h this binding was written against

# name -> (restype, argtypes); mirrors include/rdmnet_hip.h one to one
SIGNATURES = {
    'rdm_abi_version': (c_int, []),
    'rdm_abi_struct_size': (c_size, [c_int]),
    'rdm_last_error': (ctypes.c_char_p, []),
    'rdm_rehash_schedule': (c_int, [c_i64, c_void, c_void, c_int]),
    'rdm_grid_subsample_workspace_bytes': (c_size, [c_i64, c_int]),
    'rdm_grid_subsample': (c_int, [c_void, c_i64, c_void, c_int, c_f32, c_void, c_void, c_void, c_size,
                                   c_void]),
    'rdm_grid_subsample_form': (c_int, [c_void, c_i64, c_void, c_int, c_f32, c_void, c_void, c_void, c_size,
                                        c_void, c_int]),
    'rdm_radius_neighbors_workspace_bytes': (c_size, [c_i64, c_i64, c_int]),
    'rdm_radius_neighbors': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_void, c_int, c_f32, c_int,
                                     c_void, c_void, c_void, c_void, c_void, c_size, c_void]),
    'rdm_radius_grid_workspace_bytes': (c_size, [c_i64]),
    'rdm_radius_grid_build': (c_int, [c_void, c_i64, c_void, c_int, c_f32, c_void, c_size, c_void]),
    'rdm_radius_grid_query': (c_int, [c_void, c_size, c_i64, c_void, c_i64, c_void, c_int, c_f32, c_int, c_void, c_void,
                                      c_void, c_void, c_void, c_size, c_void]),
    'rdm_gemm_workspace_bytes': (c_size, [c_i64, c_i64, c_int]),
    'rdm_gemm': (c_int, [c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_int, c_void, c_i64, c_i64, c_i64, c_i64,
                         c_i64, c_int, c_void, c_void, c_int, c_void, c_size, c_void]),
    'rdm_gemm_form': (c_int, [c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_int, c_void, c_i64, c_i64, c_i64, c_i64,
                              c_i64, c_int, c_void, c_void, c_int, c_void, c_size, c_int, c_void]),
    'rdm_gemm_last_plan': (c_int, [c_void]),
    'rdm_kpconv_gather': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64,
                                  c_i64, c_void, c_void, c_f32, c_void, c_i64, c_void, c_void]),
    'rdm_kpconv_gather_ordered': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64,
                                          c_i64, c_void, c_void, c_f32, c_void, c_i64, c_void, c_void, c_void]),
    'rdm_kpconv_gather_form': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64,
                                       c_i64, c_void, c_void, c_f32, c_void, c_i64, c_void, c_void, c_int, c_void]),
    'rdm_kpconv_fused_enabled': (c_int, []),
    'rdm_kpconv_fused_supported': (c_int, [c_i64, c_i64]),
    'rdm_kpconv_fused_partial_rows': (c_i64, [c_i64, c_i64]),
    'rdm_kpconv_packed_floats': (c_size, [c_i64, c_i64]),
    'rdm_kpconv_pack_weights': (c_int, [c_void, c_i64, c_i64, c_void]),
    'rdm_kpconv_fused': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64, c_i64, c_void, c_void,
                                 c_f32, c_void, c_void, c_i64, c_void, c_i64, c_void, c_void, c_void]),
    'rdm_kpconv_fused_form': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64, c_i64, c_void, c_void,
                                      c_f32, c_void, c_void, c_i64, c_void, c_i64, c_void, c_void, c_int, c_void]),
    'rdm_kpconv_fused_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'rdm_kpconv_fused_group_norm': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_void, c_void, c_i64, c_i64, c_void,
                                            c_void, c_f32, c_void, c_void, c_i64, c_int, c_void, c_void, c_f32, c_int, c_void, c_i64,
                                            c_void, c_i64, c_void, c_size, c_void, c_void]),
    'rdm_voxel_downsample_workspace_bytes': (c_size, [c_i64]),
    'rdm_voxel_downsample': (c_int, [c_void, c_i64, c_i64, c_int, ctypes.c_double, c_void, c_i64, c_void, c_void, c_void,
                                     c_size, c_void]),
    'rdm_ransac_workspace_bytes': (c_size, [c_int]),
    'rdm_ransac_correspondences': (c_int, [c_void, c_void, c_i64, c_f32, c_int, c_int, ctypes.c_uint64, c_void, c_void, c_void,
                                           c_void, c_void, c_size, c_void]),
    'rdm_neighbor_histogram': (c_int, [c_void, c_i64, c_void, c_int, c_void]),
    'rdm_radius_grid_records': (c_void, [c_void, c_size, c_i64]),
    'rdm_row_positive': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void]),
    'rdm_group_norm_workspace_bytes': (c_size, [c_i64, c_i64]),
    'rdm_group_norm': (c_int, [c_void, c_i64, c_i64, c_i64, c_int, c_void, c_void, c_f32, c_void, c_i64, c_int,
                               c_void, c_i64, c_void, c_void, c_size, c_void]),
    'rdm_group_norm_form': (c_int, [c_void, c_i64, c_i64, c_i64, c_int, c_void, c_void, c_f32, c_void, c_i64, c_int,
                                    c_void, c_i64, c_void, c_void, c_size, c_int, c_void]),
    'rdm_linear_group_norm_workspace_bytes': (c_size, [c_i64, c_i64]),
    'rdm_linear_group_norm': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_void, c_i64, c_i64, c_i64, c_int, c_void, c_void,
                                      c_f32, c_void, c_i64, c_int, c_void, c_i64, c_void, c_i64, c_void, c_void, c_size, c_void]),
    'rdm_linear_group_norm_form': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_void, c_i64, c_i64, c_i64, c_int, c_void, c_void,
                                           c_f32, c_void, c_i64, c_int, c_void, c_i64, c_void, c_i64, c_void, c_void, c_size, c_int, c_void]),
    'rdm_decoder_stage_form': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void,
                                       c_i64, c_int, c_void, c_void, c_f32, c_int, c_void, c_i64, c_void, c_i64, c_void, c_size, c_int, c_void]),
    'rdm_patch_scores': (c_int, [c_void, c_i64, c_i64, c_void, c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_void,
                                 c_void]),
    'rdm_decoder_stage_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'rdm_decoder_stage': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void,
                                  c_i64, c_int, c_void, c_void, c_f32, c_int, c_void, c_i64, c_void, c_i64, c_void, c_size, c_void]),
    'rdm_layer_norm': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_void, c_f32, c_int, c_void,
                               c_i64, c_void]),
    'rdm_linear_layer_norm': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_void,
                                      c_f32, c_int, c_void, c_i64, c_void]),
    'rdm_attention_tail': (c_int, [c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_void, c_void, c_void, c_i64,
                                   c_void, c_void, c_i64, c_void, c_void, c_void, c_f32, c_void, c_i64, c_void]),
    'rdm_attention_tail_packed_floats': (c_size, []),
    'rdm_attention_tail_pack_weights': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_void]),
    'rdm_attention_tail_packed': (c_int, [c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_void, c_void, c_void, c_void, c_void,
                                          c_void, c_void, c_f32, c_void, c_i64, c_void]),
    'rdm_gather_max': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_void, c_i64,
                               c_void]),
    'rdm_upsample_concat': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64,
                                    c_void, c_i64, c_void]),
    'rdm_gather_rows': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_i64, c_void]),
    'rdm_rope': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_void]),
    'rdm_attention': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_int, c_int,
                              c_void]),
    'rdm_attention_bf16': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_int, c_int,
                              c_void]),
    'rdm_attention_self_pair': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_i64, c_i64, c_int, c_int,
                                        c_int, c_void]),
    'rdm_vote_shift': (c_int, [c_void, c_void, c_i64, c_i64, c_f32, c_f32, c_f32, c_void, c_void]),
    'rdm_sigmoid_column': (c_int, [c_void, c_i64, c_i64, c_void, c_void]),
    'rdm_l2_normalize': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_void]),
    'rdm_nms': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_void]),
    'rdm_compact_indices': (c_int, [c_void, c_i64, c_i64, c_void, c_void, c_void]),
    'rdm_point_to_node_workspace_bytes': (c_size, [c_i64, c_i64]),
    'rdm_point_to_node': (c_int, [c_void, c_i64, c_void, c_i64, c_int, c_void, c_void, c_void, c_void, c_void,
                                  c_size, c_void]),
    'rdm_point_to_node_pair': (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_int, c_void, c_void, c_void,
                                       c_void, c_void, c_void, c_void, c_void, c_size, c_void]),
    'rdm_coarse_matching_workspace_bytes': (c_size, [c_i64, c_i64]),
    'rdm_coarse_matching': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_int, c_int, c_void, c_void, c_void,
                                    c_void, c_void, c_size, c_void]),
    'rdm_coarse_matching_features_workspace_bytes': (c_size, [c_i64, c_i64]),
    'rdm_coarse_matching_features': (c_int, [c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_i64, c_void, c_void, c_int, c_int,
                                             c_void, c_void, c_void, c_void, c_void, c_size, c_void]),
    'rdm_sinkhorn': (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_void, c_int, c_void, c_void]),
    'rdm_lgr_workspace_bytes': (c_size, [c_i64]),
    'rdm_lgr': (c_int, [c_void, c_void, c_void, c_void, c_void, c_i64, c_i64, c_f32, c_int, c_int, c_void, c_void,
                        c_void, c_void, c_void, c_void, c_size, c_void]),
    'rdm_engine_create': (c_int, [c_void, c_void]),
    'rdm_engine_destroy': (None, [c_void]),
    'rdm_engine_set_param': (c_int, [c_void, ctypes.c_char_p, c_void, c_void, c_int]),
    'rdm_engine_finalize': (c_int, [c_void]),
    'rdm_engine_share_params': (c_int, [c_void, c_void]),
    'rdm_engine_run': (c_int, [c_void, c_void, c_i64, c_void, c_i64, c_void, c_void]),
    'rdm_engine_collate': (c_int, [c_void, c_void, c_i64, c_void, c_i64, c_void, c_void]),
    'rdm_engine_forward': (c_int, [c_void, c_void, c_void, c_void]),
    'rdm_engine_collate_batch': (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void]),
    'rdm_engine_forward_batched': (c_int, [c_void, c_int, c_void, c_void]),
    'rdm_engine_run_lockstep': (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_int, c_void]),
    'rdm_engine_forward_lockstep': (c_int, [c_void, c_int, c_void, c_void, c_void]),
    'rdm_engine_collate_lockstep': (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    'rdm_lockstep_stats': (None, [c_void, c_int]),
    'rdm_lockstep_stats_dump': (None, []),
    'rdm_lockstep_selftest': (c_int, [c_int, c_void, c_int, c_void, c_int, c_void]),
    'rdm_engine_reserve': (c_int, [c_void, c_size]),
    'rdm_engine_set_wait': (c_int, [c_void, c_int]),
    'rdm_engine_set_pairs_in_flight': (c_int, [c_void, c_int]),
    'rdm_engine_set_overlap': (c_int, [c_void, c_int]),
    'rdm_engine_enable_profile': (c_int, [c_void, c_int]),
    'rdm_engine_get_profile': (c_int, [c_void, c_void, c_int]),
    'rdm_engine_keep_taps': (c_int, [c_void, c_int]),
    'rdm_engine_get_tensor': (c_int, [c_void, ctypes.c_char_p, c_void]),
    'rdm_engine_describe': (c_int, [c_void, c_int, c_void, c_void]),
    'rdm_engine_export': (c_int, [c_void, c_int, c_void, c_void, c_void]),
    'rdm_copy_device': (c_int, [c_void, c_void, c_size, c_void]),
}



_lib = None


def build():
    """Compile the library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    import subprocess
    subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc'), '-j8'], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `make -C rdmnet_amd/csrc` '
                '(or __graft_entry__.build()); there is no CPU fallback')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        got = handle.rdm_abi_version()
        if got != ABI_VERSION:  # a stale .so with other struct layouts would corrupt memory silently
            raise RuntimeError(f'{LIB_PATH} has ABI version {got}, this binding expects {ABI_VERSION}: rebuild it '
                               '(`make -C rdmnet_amd/csrc`)')
        _check_struct_sizes(handle)
        _lib = handle
    return _lib


def _check_struct_sizes(handle):
    """The ctypes mirrors of the structs that cross the C-ABI must have the library's sizes (the C side memsets and fills
    them through the pointers it is given)."""
    from . import engine as E  # (the mirrors live next to their user)
    for which, cls in enumerate((E.EngineConfig, E.EngineResult, E.TensorView, E.KpconvProfile, E.DataDict)):
        want, have = handle.rdm_abi_struct_size(which), ctypes.sizeof(cls)
        if want != have:
            raise RuntimeError(f'{LIB_PATH}: sizeof({cls.__name__}) is {want} in the library, {have} in rdmnet_amd/engine.py')


def check(code, what):
    if code != 0:
        msg = lib().rdm_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed with code {code}: {msg}')


def ptr(t):
    """Device (or host) address of a torch tensor, 0 for None."""
    return 0 if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
