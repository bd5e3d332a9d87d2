"""ctypes binding of libcprhip.so (the C-ABI declared in include/cpr_hip.h).

The product path fails loudly when the library is missing or a symbol is absent: there is no
eager/PyTorch fallback for any op (a silent fallback would void every parity claim).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libcprhip.so')

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_l = ctypes.c_longlong

# name -> argtypes (restype is always int: 0 ok, <0 error).  Mirrors include/cpr_hip.h one to one.
SIGNATURES = {
    'cpr_version': [],
    'cpr_conv2d_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    'cpr_conv2d_dual_fwd': [_p] * 9 + [_i] * 16 + [_p, _p],
    'cpr_conv1x1_stream_fwd': [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p],
    'cpr_conv_wgrad_bf16_workspace': [_i, _i, _i, _i, _i, _i],
    'cpr_conv_wgrad_bf16_workspace_s': [_i] * 7,
    'cpr_conv_wgrad_bf16': [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_conv_wgrad_bf16_s': [_p, _i, _p, _i, _p, _p] + [_i] * 8 + [_p],
    'cpr_wino_pack_weights': [_p, _p, _i, _i, _i, _p],
    'cpr_conv3x3_wino_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_wino32_pack_weights': [_p, _p, _i, _i, _i, _p],
    'cpr_conv3x3_wino32_slots': [_i, _i],
    'cpr_conv3x3_wino32_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_conv3x3_wino_wgrad_workspace': [_i, _i, _i, _i, _i],
    'cpr_conv3x3_wino_wgrad': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_gn_apply_b8': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'cpr_conv2d_fwd_bf16': [_p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 12 + [_p, _p],
    'cpr_conv2d_bf16_mask_slots': [_i] * 10,
    'cpr_wgrad_bf16_set_tn': [_i],
    'cpr_bn_fold_multi': [_p, _i, _i, _p],
    'cpr_pack_weights_multi': [_p, _i, _i, _p],
    'cpr_pack_weights_bf16_multi': [_p, _i, _i, _p],
    'cpr_conv2d_dgrad_bf16_fused': [_p] * 8 + [_i] * 10 + [_p, _p],
    'cpr_stem7x7s2_bf16': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'cpr_stem7x7s2_pool_bf16': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'cpr_maxpool3x3s2_bf16': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_gn_stats_bf16': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_gn_apply_bf16': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_nchw_to_nhwc4': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_nhwc_to_nchw': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_stem7x7s2_pool_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'cpr_maxpool3x3s2': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_gn_stats': [_p, _p, _i, _i, _i, _i, _p],
    'cpr_gn_finalize': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    'cpr_gn_apply': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'cpr_box_centers': [_p, _p, _i, _p],
    'cpr_logit_project': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'cpr_neg_mask_loss': [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _i, _i, _f, _i, _p, _p],
    'cpr_bag_sample': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p],
    'cpr_grid_bag': [_p, _i, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p],
    'cpr_mil_loss': [_p, _i, _i, _p, _p, _p, _p, _p] + [_i] * 10 + [_f, _i, _f, _i, _i, _f, _f, _f, _i, _p, _p],
    'cpr_refine': [_p, _i, _p, _p, _p, _i, _i] + [_p] * 9 + [_i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _i, _i, _p],
    'cpr_point_assign': [_p, _p, _i, _i, _f, _i, _p, _p, _p, _p],
    'cpr_hungarian_cost': [_p, _i, _p, _i, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _i, _p],
    'cpr_lsa_topk': [_p, _p, _p, _p, _p, _p, _i, _i] + [_p] * 14 + [_i, _i, _p],
    'cpr_topk_desc': [_p, _i, _i, _p, _p, _p],
    'cpr_nms_candidates': [_p, _i, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p],
    'cpr_nms_workspace': [_i],
    'cpr_nms': [_p, _p, _p, _i, _f, _p, _p, _p, _p, _p, _p],
    'cpr_topk_desc_batched': [_p, _i, _i, _i, _p, _p, _p],
    'cpr_nms_candidates_batched': [_p, _i, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p],
    'cpr_tap_sum3x3': [_p, _p, _p, _i, _i, _i, _i, _p],
    'cpr_nms_batched': [_p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _l, _p],
    'cpr_p2p_decode': [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    'cpr_rowmax_sigmoid': [_p, _p, ctypes.c_longlong, _i, _p],
    'cpr_sigmoid': [_p, _p, ctypes.c_longlong, _p],
    'cpr_p2p_loss': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _p],
    'cpr_match_cost': [_p, _i, _p, _i, _p, _p, _p, _i, _i, _p, _i, _i, _p],
    # training step: backward + optimizer
    'cpr_conv2d_wgrad_workspace': [_i] * 7,
    'cpr_conv2d_wgrad': [_p] * 6 + [_i] * 11 + [_p],
    'cpr_gn_bwd': [_p] * 12 + [_i] * 7 + [_p],
    'cpr_gn_bwd_bf16': [_p] * 13 + [_i] * 7 + [_p],
    'cpr_gn_bwd_bf16_dz16': [_p] * 13 + [_i] * 7 + [_p],
    'cpr_upsample_add_bwd': [_p, _p] + [_i] * 7 + [_p],
    'cpr_relu_bwd_colsum_ws': [_l, _i],
    'cpr_relu_bwd_colsum': [_p, _p, _p, _i, _p, _p, _p, _p, _l, _i, _i, _p],
    'cpr_bn_fold_bwd': [_p] * 8 + [_i, _i, _p],
    'cpr_bn_fold_bwd_part': [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p],
    'cpr_part_colsum': [_p, _p, _p, _i, _i, _p],
    'cpr_axpby': [_p, _p, _f, _f, _l, _p],
    'cpr_phase_scatter_add': [_p, _p] + [_i] * 11 + [_p],
    'cpr_zero_insert': [_p, _p] + [_i] * 7 + [_p],
    'cpr_loss_bwd': [_p] * 15 + [_i] * 10 + [_f] * 5 + [_p, _p],
    'cpr_bag_gather_bwd': [_p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _p],
    'cpr_bag_points_gather_bwd': [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    'cpr_loss_bwd_general': [_p] * 10 + [_i] * 15 + [_f, _i, _f, _i, _i, _f, _f, _f, _i, _p, _p],
    # data side
    'cpr_preprocess_u8': [_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'cpr_clip_flip_boxes': [_p, _p, _p, _p, _i, _i, _p],
    'cpr_pack_weights': [_p, _p, _p] + [_i] * 7 + [_p],
    'cpr_pack_weights_bf16': [_p, _p, _p, _p] + [_i] * 5 + [_p],
    'cpr_spin': [ctypes.c_longlong, _p],
    'cpr_bn_fold': [_p, _p, _p, _p, _f, _p, _p, _p, _i, _p],
    'cpr_p2p_loss_bwd': [_p] * 9 + [_i] * 5 + [_f] * 9 + [_p, _i, _i, _p],
    'cpr_grad_sumsq': [_p, _l, _p, _p, _i, _p],
    'cpr_sgd_step': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p],
}

# measurement build only (-DCPR_BENCH_HOOKS -> libcprhip_bench.so, tools/*.py): NOT part of the product library
BENCH_SIGNATURES = {
    'cpr_lsa_phase_clocks': [_p, _i],
    'cpr_wino32_set_debug': [_i, _i, _i],
    'cpr_conv_force_tile': [_i, _i],
    'cpr_conv_set_pipeline': [_i],
    'cpr_conv_set_stream': [_i],
    'cpr_conv_set_ablation': [_i],
    'cpr_wgrad_set_ablation': [_i],
    'cpr_wino_set_variant': [_i, _i],
    'cpr_wino_set_staging': [_i, _i],
    'cpr_conv_set_extra_lds': [_i],
    'cpr_bf16_set_dma': [_i],
    'cpr_bf16_set_wfrag': [_i],
}
BENCH_LIB_PATH = os.path.join(_HERE, 'csrc', 'libcprhip_bench.so')

_lib = None


class CprHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it was not built -- build with
    ``python -m pointtinybenchmark_amd.build`` (hipcc cross-compiles for gfx950 without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    path, sigs = LIB_PATH, dict(SIGNATURES)
    if os.environ.get('CPR_BENCH_HOOKS', '0') == '1':      # tools/*.py: the measurement build with its global switches
        path = BENCH_LIB_PATH
        sigs.update(BENCH_SIGNATURES)
    if not os.path.exists(path):
        raise CprHipError('%s not found: run `python -m pointtinybenchmark_amd.build%s` (no CPU fallback exists)'
                          % (path, ' --bench-hooks' if path == BENCH_LIB_PATH else ''))
    lib = ctypes.CDLL(path)
    for name, argtypes in sigs.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise CprHipError('symbol %s missing from %s (stale build?)' % (name, path))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


def call(name, *args, positive=False):
    """Invoke a C-ABI entry point; non-zero status -> RuntimeError (the reference raises Python
    exceptions / asserts on bad inputs; this is the same contract across the boundary).
    positive=True: a non-negative return is a value (size queries), only negatives are errors."""
    rc = getattr(load(), name)(*args)
    if positive and rc >= 0:
        return rc
    if rc != 0:
        if rc == -1001:
            raise CprHipError('%s: invalid argument' % name)
        if rc == -1002:
            raise CprHipError('%s: unsupported configuration' % name)
        raise CprHipError('%s failed with hipError %d' % (name, -rc))
    return rc
