"""ctypes binding of libunet_hip.so (declared in include/unet_hip.h).

The HIP library is the product path: importing this module fails loudly if the shared
library has not been built (``make -C unet_amd/csrc``); there is no CPU or eager fallback.
"""
import ctypes
import os

# torch first: its wheel ships its own HIP runtime (libamdhip64), and the kernels must register with the runtime
# that owns torch's streams and allocations.  Loaded the other way round, libunet_hip.so pulls in the system runtime,
# the process ends up with two, and every launch on a torch stream fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(_HERE, "libunet_hip.so")   # override: kernel experiments

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_longlong
F = ctypes.c_float
Z = ctypes.c_size_t
D = ctypes.c_double

class DwFinDesc(ctypes.Structure):
    """uh_dwfin_desc: one table of uh_dw_wgrad_finalize_multi"""
    _fields_ = [("partial", P), ("out", P), ("parts", I), ("C", I), ("c_real", I)]


DWFIN_MAX = 32          # UH_DWFIN_MAX


class ReduceDesc(ctypes.Structure):
    """uh_reduce_desc: one table of uh_reduce_multi"""
    _fields_ = [("kind", I), ("rows", I), ("accumulate", I), ("C", I), ("n_out", I), ("N", I), ("T", I), ("K", I),
                ("partial", P), ("out", P), ("cols", P),
                ("split_stride", L), ("on", L), ("ot", L), ("sn", L), ("st", L), ("sk", L)]

    @classmethod
    def colsum(cls, partial, rows, C, out, n_out=None, cols=None, accumulate=0):
        return cls(REDUCE_COLSUM, rows, accumulate, C, C if n_out is None else n_out, 0, 0, 0, partial, out, cols,
                   0, 0, 0, 0, 0, 0)

    @classmethod
    def wgrad(cls, partial, S, split_stride, N, T, K, on, ot, sn, st, sk, out, accumulate=0):
        return cls(REDUCE_WGRAD, S, accumulate, 0, 0, N, T, K, partial, out, None, split_stride, on, ot, sn, st, sk)


REDUCE_MAX = 24         # UH_REDUCE_MAX
REDUCE_COLSUM, REDUCE_WGRAD = 0, 1

# name -> (restype, argtypes); mirrors include/unet_hip.h one to one
SIGNATURES = {
    "uh_last_error": (ctypes.c_char_p, []),
    "uh_version": (I, []),
    "uh_guard_rows": (I, [I]),
    "uh_rowmask_build": (I, [I, I, I, P, P]),
    "uh_pack_weights": (I, [P, P, I, I, I, L, L, L, L, L, I, P]),
    "uh_pack_weights_multi": (I, [P, I, I, P]),
    "uh_conv_gemm": (I, [I, I, I, P, I, I, I, P, I, I, I, P, I, I, P, P, P, P, P, P]),
    "uh_conv_gemm_workspace": (Z, [I, I, I, I, I, I]),
    "uh_conv_gemm_ws": (I, [I, I, I, P, I, I, I, P, I, I, I, P, I, I, P, P, P, P, P, P, Z, P]),
    "uh_conv_gemm_bnbwd": (I, [I, I, I, P, I, I, I, P, I, P, I, I, P, P, I, I, P, P, P, P, P, P]),
    "uh_conv_gemm1_bnbwd": (I, [I, I, I, P, I, I, I, P, I, P, I, I, P, P, I, I, P, P, P, P, P, P]),
    "uh_conv_gemm_mtiles": (I, [I, I, I, I]),
    "uh_conv_wgrad_splits": (I, [I, I, I, I, I, I]),
    "uh_conv_wgrad_workspace": (Z, [I, I, I, I, I, I]),
    "uh_conv_wgrad": (I, [I, I, I, P, I, I, I, P, I, I, I, I, P, P]),
    "uh_wgrad_reduce": (I, [P, I, I, I, I, L, L, L, L, L, P, I, P]),
    "uh_wgrad_reduce_strided": (I, [P, I, L, I, I, I, L, L, L, L, L, P, I, P]),
    "uh_conv_first_fwd": (I, [I, I, I, I, I, P, P, P, I, P, P, P, I, I, P, P]),
    "uh_conv_first_parts": (I, [I, I, I]),
    "uh_conv_first_wgrad": (I, [I, I, I, I, I, P, P, I, I, P, P]),
    "uh_conv_first_wgrad_parts": (I, [I, I, I]),
    "uh_bn_finalize": (I, [P, I, I, I, L, P, P, F, F, P, P, P, P, P, P, P, P]),
    "uh_bn_fold_eval": (I, [I, P, P, P, P, P, F, P, P, P]),
    "uh_bn_relu_apply": (I, [L, I, P, I, I, P, P, P, P, I, I, P]),
    "uh_bn_bwd_parts": (I, [L, I]),
    "uh_bn_relu_bwd_reduce": (I, [L, I, P, I, I, P, I, I, P, P, P, P, P, P]),
    "uh_bn_bwd_finalize": (I, [P, I, I, I, L, P, P, P, I, P]),
    "uh_fin_ws_bytes": (Z, [I, I, I]),
    "uh_fin_counters": (I, []),
    "uh_bn_relu_bwd_reduce_fin": (I, [L, I, P, I, I, P, I, I, P, P, P, P, I, L, P, P, P, P, Z, P, P]),
    "uh_bn_res_relu_bwd_reduce_fin": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, P, P, I, L, P, P, P, P, Z, P, P]),
    "uh_dwconv3x3_fwd_bn": (I, [I, I, I, I, P, I, I, P, I, P, P, P, I, I, P, P, F, F, P, P, P, P, P, P, P, P, Z, P, P]),
    "uh_bn_relu_bwd_apply": (I, [L, I, P, I, I, P, I, I, P, P, P, P, P, P, P, I, I, P]),
    "uh_maxpool2_fwd": (I, [I, I, I, I, P, I, I, P, I, I, P]),
    "uh_maxpool2_bwd": (I, [I, I, I, I, P, I, I, P, I, I, P, I, I, P, I, I, P]),
    "uh_space_to_depth2": (I, [I, I, I, I, P, I, I, P, I, I, P]),
    "uh_depth_to_space2": (I, [I, I, I, I, P, I, I, P, I, I, P]),
    "uh_colsum_parts": (I, [L, I]),
    "uh_colsum": (I, [L, I, P, I, I, P, P]),
    "uh_colsum_finalize": (I, [P, I, I, I, P, I, P]),
    "uh_head_fwd": (I, [I, I, I, I, I, P, I, I, P, P, P, P]),
    "uh_head_fwd_counts": (I, [I, I, I, I, P, I, I, P, P, P, P, I, P, P]),
    "uh_head_bwd_parts": (I, [I, I, I, I]),
    "uh_head_bwd_bnstat": (I, [I, I, I, I, I, P, I, I, P, P, P, I, I, P, P, I, I, P, P, P, P, P, P]),
    "uh_bn_relu_apply_pool2": (I, [I, I, I, I, P, I, I, P, P, P, I, I, P, I, I, P]),
    "uh_maxpool2_bwd_bnstat_parts": (I, [I, I, I, I]),
    "uh_maxpool2_bwd_bnstat": (I, [I, I, I, I, P, I, I, P, I, I, P, I, I, P, I, I, P, I, I, P, P, P, P, P, P]),
    "uh_head_bwd": (I, [I, I, I, I, I, P, I, I, P, P, P, I, I, P, P]),
    "uh_dw_parts": (I, [I, I, I, I]),
    "uh_dwconv3x3_fwd": (I, [I, I, I, I, P, I, I, P, I, P, I, P, P, P, P, I, I, P, P]),
    "uh_bn_res_relu_apply": (I, [L, I, P, I, I, P, I, I, P, P, P, P, I, I, P]),
    "uh_bn_res_relu_apply_s2d": (I, [I, I, I, I, P, I, I, P, I, I, P, P, P, I, I, P, I, I, P]),
    "uh_bn_res_relu_bwd_reduce": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, P, P, P, P]),
    "uh_bn_res_relu_bwd_apply": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, P, P, P, P, P, I, I, P, I, I, P]),
    "uh_dwconv3x3_dgrad": (I, [I, I, I, I, P, I, I, P, I, P, I, I, P, P, I, I, P]),
    "uh_dwconv3x3_wgrad": (I, [I, I, I, I, P, I, I, P, I, I, P, P]),
    "uh_dw_wgrad_finalize": (I, [P, I, I, I, P, P]),
    "uh_dw_wgrad_finalize_multi": (I, [P, I, P]),
    "uh_reduce_multi": (I, [P, I, P]),
    "uh_space_to_depth2_of": (I, [I, I, I, I, I, I, P, I, I, P, I, I, P]),
    "uh_depth_to_space2_to": (I, [I, I, I, I, I, I, P, I, I, P, I, I, P]),
    "uh_depth_to_space2_add": (I, [I, I, I, I, I, I, P, I, I, P, I, I, P, I, I, P]),
    "uh_resize_bilinear_fwd": (I, [I, I, I, I, I, I, P, I, I, P, I, I, P]),
    "uh_resize_bilinear_bwd": (I, [I, I, I, I, I, I, P, I, I, P, I, I, P]),
    "uh_dwblock_bwd_parts": (I, [I, I, I, I]),
    "uh_dwconv3x3_fwd_parts": (I, [I, I, I, I]),
    "uh_dwconv3x3_fwd_takes_apply": (I, [I]),
    "uh_dwconv3x3_fwd_applied": (I, [I, I, I, I, P, I, I, P, I, I, P, P, P, I, I, P, I, P, P, I, I, P, P]),
    "uh_dwblock_bwd": (I, [I, I, I, I, P, I, I, P, I, I, P, I, I, P, I, P, P, P, P, P, P, I, I, P, P]),
    "uh_dwblock_bwd_takes_prev": (I, []),
    "uh_dwblock_bwd_colsum": (I, [I, I, I, I, P, I, I, P, I, I, P, I, I, P, I, P, P, P, P, P, P, I, I, P, P, P]),
    "uh_dwblock_bwd_prev": (I, [I, I, I, I, P, I, I, P, I, I, P, I, I, P, I, P, P, P, P, P, P, I, I, P, P, I, I, P, P, P, P]),
    "uh_gate_fwd": (I, [L, I, P, I, I, P, I, I, P, I, I, P]),
    "uh_gate_bwd": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, I, I, P, I, I, P]),
    "uh_gate_bwd_parts": (I, [L, I]),
    "uh_gate_bwd_colsum": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, I, I, P, I, I, P, P]),
    "uh_add": (I, [L, I, P, I, I, P, I, I, P, I, I, P, I, I, P]),
    "uh_gate8_fwd": (I, [L, I, P, I, I, P, I, I, P, P, P, I, I, P]),
    "uh_gate8_parts": (I, [L]),
    "uh_gate8_bwd": (I, [L, I, P, I, I, P, I, I, P, I, I, P, P, I, I, P, I, I, P, P]),
    "uh_nchw_to_c8": (I, [I, I, I, I, P, P, I, I, P]),
    "uh_adam_step": (I, [L, P, P, P, P, F, F, F, F, F, I, P]),
    "uh_gate8_pack": (I, [I, P, P, P, P, P, P, P, P]),
    "uh_pad_rows_f32": (I, [I, I, I, P, P, P]),
    "uh_sigmoid_clamp": (I, [L, P, F, F, P, P]),
    "uh_sigmoid_clamp_bwd": (I, [L, P, F, F, P, P, P]),
    "uh_colsum_finalize_gather": (I, [P, I, I, I, P, P, P]),
    "uh_adam_step_amp": (I, [L, P, P, P, P, F, F, F, F, F, I, P, P, P, P]),
    "uh_gather_batch_u8": (I, [I, I, I, I, I, P, P, P, P, P]),
    "uh_gather_batch_f32": (I, [I, I, I, I, I, P, P, P, P]),
    "uh_confusion_counts": (I, [I, L, P, P, P, P]),
    "uh_bce_logits_workspace": (Z, [L]),
    "uh_bce_logits": (I, [L, P, P, F, P, P, P, Z, P]),
    "uh_bce_logits_noalloc": (I, [L, P, P, F, P, P, P]),
    "uh_selfreg_pool2": (I, [I, I, I, I, P, I, I, I, P, P]),
    "uh_selfreg_stat_parts": (I, [I, I, I, I]),
    "uh_selfreg_plane_stats": (I, [I, I, I, I, P, I, I, P, P]),
    "uh_selfreg_rank": (I, [I, I, I, I, P, P, P]),
    "uh_selfreg_pair_parts": (I, [I, I, I, I]),
    "uh_selfreg_pair_sums": (I, [I, I, I, I, P, I, I, P, P, I, I, I, I, I, P, P, P, P]),
    "uh_selfreg_accumulate": (I, [P, I, D, D, I, P, P]),
    "uh_selfreg_loss_workspace": (Z, [I, L]),
    "uh_selfreg_loss_fwd": (I, [I, L, P, P, P, F, F, P, P, P, Z, P]),
    "uh_selfreg_loss_bwd": (I, [I, L, P, P, P, P, F, P, P]),
    "uh_soft_erode": (I, [I, I, I, P, P, I, P]),
    "uh_soft_dilate": (I, [I, I, I, P, P, P]),
    "uh_soft_skel_workspace": (Z, [I, I, I]),
    "uh_soft_skel": (I, [I, I, I, P, P, P, P]),
    "uh_cldice_workspace": (Z, [I, I, I]),
    "uh_cldice_forward": (I, [I, I, I, P, P, F, F, P, P, P]),
    "uh_cldice_skel_true": (I, [I, I, I, P, P, P]),
    "uh_cldice_forward_pred": (I, [I, I, I, P, P, F, F, P, P, P]),
    "uh_cldice_backward": (I, [I, I, I, P, P, P, P, P, P]),
    "uh_cldice_read_terms": (I, [P, I, I, I, P, P]),
}

EPI_RAW_STATS, EPI_AFFINE_RELU, EPI_MASKED, EPI_CONVT, EPI_MASKED_BNBWD = 0, 1, 2, 3, 4


class HipLibraryMissing(ImportError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class HipKernelError(RuntimeError):
    pass


def check(status, what=""):
    if status != 0:
        msg = lib.uh_last_error().decode("utf-8", "replace")
        raise HipKernelError(f"{what}: uh_status {status}: {msg}")


def call(name, *args):
    """Call an int-status entry point and raise on error."""
    st = getattr(lib, name)(*args)
    if st != 0:
        check(st, name)
