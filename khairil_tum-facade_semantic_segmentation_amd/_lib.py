"""ctypes binding of libpn2hip.so (C ABI: include/pn2_hip.h).  No torch types cross it.

The product path has NO fallback: if the HIP library is missing or a symbol is absent this
module raises, and every operator in ops.py raises with it."""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libpn2hip.so")

_vp, _ci, _cl, _cd, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_float

# name -> argtypes, in the order of include/pn2_hip.h
SIGNATURES = {
    "pn2_abi_version": [],
    "pn2_error_string": [_ci],
    "pn2_farthest_point_sample": [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_square_distance": [_vp, _vp, _ci, _ci, _ci, _vp, _vp],
    "pn2_ball_query_group": [_cd, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp],
    "pn2_ball_query_group_select": [_ci, _cd, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp],
    "pn2_farthest_point_sample_plan": [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _cd, _ci, _vp, _vp, _vp],
    "pn2_ball_plan_bytes": [_ci, _ci, _ci],
    "pn2_ball_pack_rows": [_vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp],
    "pn2_ball_plan": [_cd, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp],
    "pn2_ball_query_group_planned": [_cd, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp],
    "pn2_index_points": [_vp, _vp, _ci, _ci, _ci, _cl, _vp, _vp, _vp],
    "pn2_index_points_backward": [_vp, _vp, _ci, _ci, _ci, _cl, _ci, _ci, _vp, _vp],
    "pn2_group_points": [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp],
    "pn2_three_nn": [_vp, _vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate": [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp],
    "pn2_three_interpolate_backward": [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp],
    "pn2_mlp_gemm_max_partials": [_ci],
    "pn2_mlp_gemm": [_vp, _ci, _ci, _vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _ci, _vp,
                     _vp, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_bn_finalize": [_vp, _ci, _ci, _cd, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_gemm_pool32": [_vp, _ci, _ci, _vp, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _vp, _vp, _vp,
                            _vp, _vp, _vp],
    "pn2_bn_finalize_out": [_vp, _ci, _ci, _cd, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp,
                            _vp, _vp, _vp, _cl, _vp, _vp, _vp],
    "pn2_bn_eval_coeff": [_ci, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp],
    "pn2_bn_relu_out": [_vp, _cl, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_dw_partials": [_ci, _ci, _ci],
    "pn2_mlp_dw": [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _ci, _ci, _vp, _vp,
                   _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_mlp_bwd_layer_partials": [_ci, _ci, _ci],
    "pn2_mlp_bwd_layer": [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp,
                          _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _vp],
    "pn2_mlp_bwd_post": [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _ci, _cd, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_dw_reduce_many": [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_bn_bwd_reduce_partials": [_cl],
    "pn2_bn_bwd_reduce": [_vp, _ci, _vp, _ci, _cl, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_bn_bwd_finalize": [_vp, _ci, _ci, _cd, _vp, _vp, _vp, _vp, _vp],
    "pn2_copy_pad_cols": [_vp, _ci, _ci, _vp, _ci, _ci, _cl, _vp],
    "pn2_invert_index": [_vp, _ci, _cl, _ci, _vp, _vp, _vp],
    "pn2_three_nn_many": [_ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_invert_index_many": [_ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_gather_sum": [_vp, _cl, _ci, _ci, _vp, _vp, _vp, _cl, _ci, _ci, _ci, _ci, _vp, _vp],
    "pn2_gather_sum_add": [_vp, _cl, _ci, _ci, _vp, _vp, _vp, _cl, _ci, _ci, _ci, _ci, _vp, _vp, _vp],
    "pn2_head_logits": [_vp, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp],
    "pn2_head_logits_partials": [_ci],
    "pn2_head_logits_dropout": [_vp, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp, _cf, _vp],
    "pn2_head_logits_dropout_counted": [_vp, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp, _vp, _cf, _vp],
    "pn2_head_logits_dropout_backward": [_vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp, _cf, _vp],
    "pn2_dropout_mask": [_vp, _cf, _cl, _ci, _vp, _vp],
    "pn2_head_logits_backward": [_vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp],
    "pn2_nll_loss_partials": [_cl],
    "pn2_nll_loss": [_vp, _vp, _vp, _cl, _ci, _cl, _vp, _vp, _vp, _vp, _vp],
    "pn2_nll_loss_ticketed": [_vp, _vp, _vp, _cl, _ci, _cl, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_nll_loss_backward": [_vp, _vp, _vp, _vp, _cl, _ci, _cl, _vp, _vp],
    "pn2_adam_step": [_vp, _vp, _vp, _vp, _cl, _vp, _vp, _cd, _cd, _cd, _cd, _cd, _vp],
    "pn2_adam_step_scattered": [_vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _cd, _cd, _cd, _cd, _cd, _vp],
    "pn2_sample_blocks": [_vp, _vp, _vp, _vp, _vp, _cd, _cd, _cd, _ci, _ci, _ci, _ci, _cd, ctypes.POINTER(ctypes.c_double), _ci, _ci,
                          ctypes.c_ulonglong, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_sample_blocks_multi": [_vp, _ci, _vp, _ci, _ci, _cd, _ci, _ci, ctypes.c_ulonglong, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_tile_windows": [_vp, _vp, _vp, _cd, _cd, _cd, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp],
    "pn2_tile_fill": [_vp, _vp, _vp, _vp, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp, _vp, _vp, _ci, _cl, _ci, _vp,
                      ctypes.c_ulonglong, _vp, _vp, _vp, _vp, _vp],
    "pn2_input_blocks": [_vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_seg_metrics": [_vp, _vp, _cl, _ci, _vp, _vp],
    "pn2_add_vote": [_vp, _vp, _vp, _vp, _cl, _ci, _cl, _vp, _vp, _vp],
}

_lib = None


class Pn2LibraryError(RuntimeError):
    pass


def load():
    """Load libpn2hip.so once.  Raises Pn2LibraryError if it was not built (run
    `python __graft_entry__.py build` or the package's build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so).  Loaded before torch, libpn2hip.so
    # would bind to /opt/rocm's copy instead, and the two runtimes do not share devices or streams: every launch on a torch
    # stream then fails with hipErrorNoDevice (seen with `build(); smoke()` in one process).  With torch's runtime already in
    # the process the loader resolves our dependency to it by SONAME.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise Pn2LibraryError(
            "libpn2hip.so not found at %s: the HIP extension is not built and there is no CPU "
            "fallback (build it with `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise Pn2LibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise Pn2LibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.argtypes = args
        fn.restype = ctypes.c_char_p if name == "pn2_error_string" else (ctypes.c_longlong if name == "pn2_ball_plan_bytes" else _ci)
    _lib = lib
    return lib


ERR_UNSUPPORTED = -3      # PN2_ERR_UNSUPPORTED (include/pn2_hip.h): the entry does not cover these operands, nothing was launched


def check(rc, what):
    if rc != 0:
        msg = load().pn2_error_string(rc)
        raise RuntimeError("%s failed: rc=%d (%s)" % (what, rc, msg.decode() if msg else "?"))
