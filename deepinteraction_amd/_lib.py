"""ctypes binding of libdeepinteraction_hip.so (the C ABI of include/deepinteraction_hip.h).

There is NO fallback: if the HIP library is missing or an entry point fails, the
product path raises.  (The CPU oracle under oracle/ is test infrastructure only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdeepinteraction_hip.so')

DI_F32, DI_F16, DI_F16_HL = 0, 1, 2
ABI_VERSION = 2

_c_p, _c_i, _c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes, mirrors include/deepinteraction_hip.h one to one
SIGNATURES = {
    'di_local_attn_fwd': [_c_p] * 4 + [_c_i] * 6 + [_c_f, _c_i, _c_p],
    'di_local_attn_fwd_ex': [_c_p] * 4 + [_c_i] * 6 + [_c_f, _c_i, _c_i, _c_p],
    'di_local_attn_train_fwd': [_c_p] * 5 + [_c_i] * 3 + [_c_f, _c_p],
    'di_local_attn_train_bwd': [_c_p] * 10 + [_c_i] * 3 + [_c_f, _c_p],
    'di_locatt_similar_fwd': [_c_p] * 3 + [_c_i] * 7 + [_c_p],
    'di_locatt_similar_bwd': [_c_p] * 3 + [_c_i] * 8 + [_c_p],
    'di_locatt_weighting_fwd': [_c_p] * 3 + [_c_i] * 7 + [_c_p],
    'di_locatt_weighting_bwd_ori': [_c_p] * 3 + [_c_i] * 7 + [_c_p],
    'di_locatt_weighting_bwd_weight': [_c_p] * 3 + [_c_i] * 7 + [_c_p],
    'di_pointwise_chain_fwd': [_c_p] * 8 + [ctypes.c_longlong] + [_c_i] * 4 + [_c_p],
    'di_pointwise_chain_masked_fwd': [_c_p] * 8 + [ctypes.c_longlong] + [_c_i] * 4 + [_c_p, _c_p, _c_p],
    'di_i2p_build_keys': [_c_p] * 6 + [_c_i] * 8 + [_c_f, _c_f, _c_p],
    'di_i2p_compact_keys': [_c_p] * 3 + [_c_i] * 5 + [_c_p],
    'di_i2p_attn_dense_fwd': [_c_p] * 7 + [_c_i] * 5 + [_c_p],
    'di_i2p_attn_fwd': [_c_p] * 6 + [_c_i] * 7 + [_c_f, ctypes.c_ulonglong, _c_i, _c_p],
    'di_i2p_attn_bwd': [_c_p] * 10 + [_c_i] * 9 + [_c_f, _c_f, _c_f, ctypes.c_ulonglong, _c_i, _c_p],
    'di_i2p_attn_fwd_mass': [_c_p] * 7 + [_c_i] * 7 + [_c_f, ctypes.c_ulonglong, _c_i, _c_p],
    'di_i2p_attn_bwd_mass': [_c_p] * 11 + [_c_i] * 9 + [_c_f, _c_f, _c_f, ctypes.c_ulonglong, _c_i, _c_p],
    'di_bevwarp_gather_bwd': [_c_p] * 8 + [_c_i] * 7 + [_c_p],
    'di_roi_align_bwd': [_c_p] * 3 + [_c_i] * 5 + [_c_f, _c_i, _c_p],
    'di_depth_scatter': [_c_p, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_f, _c_p],
    'di_depth_complete': [_c_p] * 4 + [_c_i] * 3 + [_c_p],
    'di_bevwarp_gather_fwd': [_c_p] * 8 + [_c_i] * 7 + [_c_p],
    'di_ms_deform_attn_fwd': [_c_p, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_i, _c_p],
    'di_ms_deform_attn_hm_fwd': [_c_p, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p],
    'di_local_attn_ring_timeouts_async': [_c_p, _c_p],
    'di_timed_begin': [_c_p, _c_p],
    'di_timed_elapsed_us': [_c_p, _c_p, _c_i, _c_p],
    'di_v2_self_feature': [_c_p] * 9 + [_c_f, _c_f] + [_c_p] * 6 + [_c_f] + [_c_p] * 2 + [_c_i] * 6 + [_c_p],
    'di_pointwise_chain_hm_fwd': [_c_p, _c_p, _c_p, _c_p, ctypes.c_longlong, _c_i, _c_i, _c_p],
    'di_pointwise_multi_warp_hm_fwd': [_c_p] * 7 + [_c_i] * 6 + [_c_p] * 7,
    'di_grid_gather_fwd': [_c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p],
    'di_polar_bev_sample_fwd': [_c_p] * 7 + [_c_i] * 8 + [_c_p],
    'di_mha_small_fwd': [_c_p, _c_i, _c_p, _c_p, _c_i, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_i, _c_p],
    'di_add_layernorm_fwd': [_c_p] * 5 + [ctypes.c_longlong, _c_i, _c_f, _c_i, _c_p],
    'di_ms_deform_attn_bwd': [_c_p, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_i, _c_p],
    'di_grid_gather_bwd': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p],
    'di_polar_bev_sample_bwd': [_c_p] * 6 + [_c_i] * 8 + [_c_p],
    'di_upsample_add_inplace': [_c_p, _c_p] + [_c_i] * 6 + [_c_p],
    'di_bias_act_inplace': [_c_p, _c_p, _c_p, ctypes.c_longlong, _c_i, _c_i, _c_p],
    'di_sparse_mark': [_c_p, _c_i, _c_p, _c_p, _c_p],
    'di_sparse_rowstart': [_c_p, _c_i, _c_p, _c_p, _c_p],
    'di_sparse_nbr': [_c_p, _c_p, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p],
    'di_sparse_conv_fwd': [_c_p] * 6 + [_c_i] * 7 + [_c_p],
    'di_voxel_keys': [_c_p, _c_i, _c_i, _c_p, _c_i, _c_i, _c_i, _c_p, _c_p],
    'di_voxel_heads': [_c_p, _c_i, _c_p, _c_p, _c_p],
    'di_voxel_scatter': [_c_p, _c_i, _c_i, _c_i] + [_c_p] * 5 + [_c_i] * 4 + [_c_p] * 4,
    'di_topk_fwd': [_c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p],
    'di_heatmap_nms': [_c_p] * 3 + [_c_i] * 5 + [ctypes.c_uint, _c_i, _c_p],
    'di_query_geometry': [_c_p] * 10 + [_c_i] * 3 + [_c_f] * 5 + [_c_p],
    'di_query_geometry_ld': [_c_p] * 10 + [_c_i] * 4 + [_c_f] * 5 + [_c_p],
    'di_roi_align_fwd': [_c_p] * 3 + [_c_i] * 5 + [_c_f, _c_i, _c_p],
    'di_mha_decode_fwd': [_c_p] * 4 + [_c_i] * 5 + [_c_f, _c_i, _c_p],
    'di_pointwise_multi_fwd': [_c_p, _c_i] + [_c_p] * 5 + [ctypes.c_longlong, _c_p],
    'di_pointwise_multi_warp_fwd': [_c_p] * 7 + [_c_i] * 6 + [_c_p] * 6,
    'di_ffn_ln_fwd': [_c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, ctypes.c_longlong, _c_p],
    'di_ffn_ln_fwd_ex': [_c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, _c_p, ctypes.c_longlong, _c_p],
    'di_conv3x3_fwd': [_c_p] * 5 + [_c_i] * 7 + [_c_p],
    'di_token_program': [_c_p, _c_i, _c_p, _c_i, _c_i, _c_p],
    'di_token_program_timed': [_c_p, _c_i, _c_p, _c_i, _c_i, _c_p, _c_p],
    'di_token_wide': [_c_p, _c_i, _c_p, _c_p, _c_p, _c_i, _c_p],
    'di_token_splitk': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_p, _c_p],
    'di_dynconv_fwd': [_c_p] * 7 + [_c_i, _c_f, _c_p],
    'di_roi_select': [_c_p] * 7 + [_c_i] * 3 + [_c_p],
    'di_query_init': [_c_p] * 12 + [_c_i] * 5 + [_c_p],
    'di_roi_align_x_fwd': [_c_p] * 3 + [_c_i] * 5 + [_c_f, _c_i, _c_i, _c_p],
    'di_kv_project_fwd': [_c_p] * 5 + [_c_i, _c_i, _c_p],
    'di_local_attn_ring_stamps': [_c_p, _c_p],
    'di_i2p_set_seed_ptr': [_c_p],
    'di_copy_d2d': [_c_p, _c_p, ctypes.c_longlong, _c_p],
    'di_iou3d_lidar': [_c_p, _c_i, _c_i, _c_p, _c_i, _c_i, _c_p, _c_p],
    'di_bn_train_fwd': [_c_p, ctypes.c_longlong, _c_i, _c_i, _c_p, _c_p, _c_f, _c_f, _c_p, _c_p, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p],
    'di_bn_train_bwd': [_c_p, _c_p, ctypes.c_longlong, _c_i, _c_i, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p],
    'di_mha_decode_x_fwd': [_c_p] * 4 + [_c_i] * 3 + [_c_f, _c_p],
    'di_wgrad_f32': [_c_p, _c_p, ctypes.c_longlong, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p],
}
# helpers that return a value instead of an error code
VALUE_FUNCS = {'di_mha_decode_scratch_floats': [_c_i] * 4, 'di_i2p_key_table_bytes': [_c_i] * 4, 'di_i2p_dense_bytes': [_c_i] * 5, 'di_topk_workspace_bytes': [_c_i] * 3,
               'di_token_splitk_workspace_bytes': [_c_i] * 2, 'di_mha_decode_x_ranges': [_c_i] * 3,
               'di_graph_node_count': [_c_p], 'di_local_attn_ring_timeouts': [_c_p], 'di_bn_workspace_floats': [_c_i],
               'di_timed_consumed': [], 'di_wgrad_workspace_floats': [ctypes.c_longlong, _c_i, _c_i]}
_LONGLONG = {'di_i2p_dense_bytes', 'di_wgrad_workspace_floats', 'di_topk_workspace_bytes', 'di_i2p_key_table_bytes', 'di_graph_node_count', 'di_token_splitk_workspace_bytes'}

# ---- the step program of di_token_program (structs of include/deepinteraction_hip.h)
TOK_LOAD, TOK_LOAD_PARTS, TOK_ATTN, TOK_COMBINE, TOK_LINEAR, TOK_ROWOP, TOK_STORE, TOK_HEADS = range(1, 9)
TOK_MAX_STEPS, TOK_MAX_HEADS = 20, 8


class TokStep(ctypes.Structure):
    _fields_ = [('kind', _c_i), ('src', _c_i), ('dst', _c_i), ('aux', _c_i), ('K', _c_i), ('N', _c_i), ('a', _c_i),
                ('b', _c_i), ('f', _c_f), ('role_lo', _c_i), ('role_hi', _c_i), ('rt', _c_i), ('rc', _c_i), ('nch', _c_i),
                ('p0', _c_p), ('p1', _c_p), ('p2', _c_p), ('p3', _c_p), ('ld0', ctypes.c_longlong),
                ('ld1', ctypes.c_longlong), ('roff', ctypes.c_longlong)]


class TokHeads(ctypes.Structure):
    _fields_ = [('w2', _c_p), ('b2', _c_p), ('qpos', _c_p), ('keep', _c_p), ('pos_out', _c_p),
                ('out', _c_p * TOK_MAX_HEADS), ('first', _c_p * TOK_MAX_HEADS), ('cls', _c_i * TOK_MAX_HEADS),
                ('nheads', _c_i), ('center_head', _c_i), ('ldo', _c_i), ('col0', _c_i), ('qpos2', _c_p), ('pos2_out', _c_p)]


_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises HipLibraryError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f'{LIB_PATH} is missing - build it with `python -m deepinteraction_amd.build` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.di_abi_version.restype = _c_i
        L.di_last_error.restype = ctypes.c_char_p
        if L.di_abi_version() != ABI_VERSION:
            raise HipLibraryError(f'ABI version mismatch: library {L.di_abi_version()} != binding {ABI_VERSION}')
        for name, argtypes in list(SIGNATURES.items()) + list(VALUE_FUNCS.items()):
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name in _LONGLONG else _c_i
        _lib = L
    return _lib


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise HipLibraryError(f'{name} failed (rc={rc}): {L.di_last_error().decode()}')
