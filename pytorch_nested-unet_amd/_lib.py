"""ctypes binding of libnunet.so (include/nunet.h). No torch types cross the ABI:
only raw device pointers, sizes and the hipStream_t of torch's current stream.

The product path fails loudly when the HIP extension is missing: there is no
CPU or eager-PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NUNET_LIB_PATH") or os.path.join(_HERE, "libnunet.so")   # override: diagnostic builds only

F32, BF16, F16 = 0, 1, 2
DTYPES = {"fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "fp16": F16, "float16": F16,
          torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}

_vp, _i32, _i64, _u32, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_size_t
BN_SUM_REPLICAS = 8        # NUNET_BN_SUM_REPLICAS in include/nunet.h
FX_WORDS = 2               # NUNET_FX_WORDS: int64 words of one fixed-point accumulator (hi * 2^-20 + lo * 2^-60)
TF_NONE, TF_BN_RELU, TF_BN_RELU_BWD = 0, 1, 2


def fx_zeros(c, device):
    """A zeroed fixed-point per-channel sum buffer [REPLICAS][2][c][FX_WORDS] (int64), see include/nunet.h."""
    return torch.zeros(BN_SUM_REPLICAS * 2 * c * FX_WORDS, dtype=torch.int64, device=device)


def fx_decode(buf, c):
    """[2*c] float64 totals of a fixed-point sum buffer (exact integer sum over the replicas first)."""
    w = buf.detach().cpu().view(BN_SUM_REPLICAS, 2 * c, FX_WORDS).sum(0)
    return w[:, 0].double() * 2.0 ** -20 + w[:, 1].double() * 2.0 ** -60


def fx_encode(values, c, device):
    """Fixed-point buffer holding `values` ([2*c] float64) in replica 0 (what a producing kernel would have left)."""
    v = values.detach().cpu().double().reshape(2 * c) * 2.0 ** 20
    hi = torch.floor(v)
    lo = torch.floor((v - hi) * 2.0 ** 40)
    buf = torch.zeros(BN_SUM_REPLICAS, 2 * c, FX_WORDS, dtype=torch.int64)
    buf[0, :, 0] = hi.to(torch.int64)
    buf[0, :, 1] = lo.to(torch.int64)
    return buf.reshape(-1).to(device)


class ConvDesc(C.Structure):
    _fields_ = [("dtype", _i32), ("N", _i32), ("H", _i32), ("W", _i32),
                ("src0", _vp), ("C0", _i32), ("P0", _i32),
                ("src1", _vp), ("C1", _i32), ("P1", _i32),
                ("wpack", _vp), ("bias", _vp),
                ("dst0", _vp), ("D0", _i32), ("Q0", _i32),
                ("dst1", _vp), ("D1", _i32), ("Q1", _i32),
                ("acc_slot_w", _i32), ("acc0_mask", _u32), ("acc1", _i32),
                ("stats", _vp), ("splitk_ws", _vp), ("splitk_ws_floats", _i64),
                ("bn_y", _vp), ("bn_py", _i32), ("bn_mean_invstd", _vp), ("bn_gamma", _vp), ("bn_beta", _vp), ("bn_sums", _vp),
                ("in_tf", _i32), ("tf_training", _i32), ("tf_y", _vp), ("tf_py", _i32), ("tf_fx", _vp),
                ("tf_gamma", _vp), ("tf_beta", _vp), ("tf_conv_bias", _vp),
                ("tf_running_mean", _vp), ("tf_running_var", _vp), ("tf_nbt", _vp), ("tf_mean_invstd", _vp),
                ("tf_momentum", _f32), ("tf_eps", _f32),
                ("tf_dgamma", _vp), ("tf_dbeta", _vp), ("tf_dbias", _vp), ("tf_store", _vp), ("tf_ps", _i32), ("tile", _i32)]


class WgradDesc(C.Structure):
    _fields_ = [("dtype", _i32), ("N", _i32), ("H", _i32), ("W", _i32),
                ("src0", _vp), ("C0", _i32), ("P0", _i32),
                ("src1", _vp), ("C1", _i32), ("P1", _i32),
                ("dy", _vp), ("Cout", _i32), ("PY", _i32),
                ("dw", _vp), ("slab_stride", _i64), ("max_slabs", _i32), ("target_wgs", _i32), ("dw_floats", _i64), ("item_shape", _i32)]


class BnFwdDesc(C.Structure):
    _fields_ = [("dtype", _i32), ("N", _i32), ("H", _i32), ("W", _i32), ("C", _i32),
                ("y", _vp), ("PY", _i32),
                ("conv_bias", _vp), ("stats", _vp), ("gamma", _vp), ("beta", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("num_batches_tracked", _vp),
                ("save_mean_invstd", _vp),
                ("training", _i32), ("momentum", _f32), ("eps", _f32),
                ("a", _vp), ("PA", _i32), ("pooled", _vp), ("PP", _i32), ("up", _vp), ("PU", _i32)]


class BnBwdDesc(C.Structure):
    _fields_ = [("dtype", _i32), ("N", _i32), ("H", _i32), ("W", _i32), ("C", _i32),
                ("da", _vp), ("PDA", _i32), ("y", _vp), ("PY", _i32),
                ("mean_invstd", _vp), ("gamma", _vp), ("beta", _vp),
                ("sums", _vp), ("dgamma", _vp), ("dbeta", _vp), ("dbias", _vp),
                ("dy", _vp), ("PDY", _i32)]


class BnrDesc(C.Structure):
    _fields_ = [("y", _vp), ("PY", _i32), ("mean_invstd", _vp), ("gamma", _vp), ("beta", _vp), ("sums", _vp)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", _i32), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class PlanCfg(C.Structure):
    _fields_ = [("N", _i32), ("H", _i32), ("W", _i32),
                ("input_channels", _i32), ("num_classes", _i32), ("deep_supervision", _i32),
                ("dtype", _i32), ("unet", _i32)]


# every symbol include/nunet.h declares: name -> (restype, argtypes)
_SIG = {
    "nunet_version": (_i32, []),
    "nunet_last_error": (C.c_char_p, []),
    "nunet_conv3x3_fwd": (_i32, [C.POINTER(ConvDesc), _vp]),
    "nunet_conv3x3_wgrad": (_i32, [C.POINTER(WgradDesc), _vp]),
    "nunet_conv3x3_wgrad_pair": (_i32, [C.POINTER(WgradDesc), C.POINTER(WgradDesc), _vp]),
    "nunet_conv3x3_wgrad_slabs": (_i32, [C.POINTER(WgradDesc)]),
    "nunet_wgrad_reduce": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp]),
    "nunet_pack_weights": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "nunet_unpack_wgrad": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "nunet_bn_relu_fwd": (_i32, [C.POINTER(BnFwdDesc), _vp]),
    "nunet_bn_relu_bwd_reduce": (_i32, [C.POINTER(BnBwdDesc), _vp]),
    "nunet_bn_relu_bwd_apply": (_i32, [C.POINTER(BnBwdDesc), _vp]),
    "nunet_maxpool2x2_fwd": (_i32, [_i32] * 5 + [_vp, _i32, _vp, _i32, _vp]),
    "nunet_maxpool2x2_bwd": (_i32, [_i32] * 5 + [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp]),
    "nunet_upsample2x_fwd": (_i32, [_i32] * 5 + [_vp, _i32, _vp, _i32, _vp]),
    "nunet_upsample2x_bwd": (_i32, [_i32] * 5 + [_vp, _i32, _vp, _i32, _i32, _vp]),
    "nunet_head_fwd": (_i32, [_i32] * 6 + [_vp, _i32, _vp, _vp, _vp, _vp]),
    "nunet_head_bwd": (_i32, [_i32] * 6 + [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "nunet_head_bwd_bnr": (_i32, [_i32] * 6 + [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, C.POINTER(BnrDesc), _vp]),
    "nunet_bce_dice_ws_bytes": (C.c_size_t, [_i32]),
    "nunet_bce_dice_fwd": (_i32, [_vp, _vp, _i32, _i64, _vp, _sz, _vp, _vp]),
    "nunet_bce_dice_bwd": (_i32, [_vp, _vp, _i32, _i64, _vp, _sz, _vp, _vp, _vp]),
    "nunet_loss_step_ws_bytes": (C.c_size_t, [_i32, _i64, _i32, _i32]),
    "nunet_loss_step": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _sz, _vp, _vp, _vp, _f32, _vp]),
    "nunet_lovasz_ws_bytes": (C.c_size_t, [_i32, _i64]),
    "nunet_lovasz_hinge_fwd": (_i32, [_vp, _vp, _i32, _i64, _vp, _sz, _vp, _vp, _vp]),
    "nunet_lovasz_hinge_bwd": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "nunet_iou_counts": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp]),
    "nunet_sigmoid_u8": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "nunet_sgd_step": (_i32, [_vp, _vp, _vp, _i64, _vp, _f32, _f32, _i32, _i32, _f32, _vp]),
    "nunet_preprocess_u8": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp]),
    "nunet_nchw_to_nhwc": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "nunet_plan_create": (_vp, [C.POINTER(PlanCfg)]),
    "nunet_plan_destroy": (None, [_vp]),
    "nunet_plan_arena_bytes": (C.c_size_t, [_vp]),
    "nunet_plan_param_count": (_i64, [_vp]),
    "nunet_plan_bnbuf_count": (_i64, [_vp]),
    "nunet_plan_bn_layers": (_i32, [_vp]),
    "nunet_plan_num_heads": (_i32, [_vp]),
    "nunet_plan_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i32, _vp]),
    "nunet_plan_stage_u8": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _sz, _vp]),
    "nunet_plan_backward": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp, _i32, _vp]),
    "nunet_plan_backward_phase": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp, _i32, _i32, _vp]),
    "nunet_plan_grad_scratch": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "nunet_plan_bucket0_enable": (_i32, [_vp, _i32]),
    "nunet_plan_bucket0_wait": (_i32, [_vp, _vp]),
    "nunet_plan_update": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp, _f32, _f32, _i32, _f32, _vp, _vp]),
    "nunet_plan_set_inpass_update": (_i32, [_vp, _vp, _vp, _vp, _f32, _f32, _i32, _f32, _vp]),
    "nunet_plan_repack": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "nunet_plan_sgd": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp, _f32, _f32, _i32, _f32, _vp, _vp]),
    "nunet_plan_feature": (_i64, [_vp, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "nunet_plan_set_multistream": (_i32, [_vp, _i32]),
    "nunet_plan_set_schedule": (_i32, [_vp, _i32]),
    "nunet_plan_calibrate": (_i32, [_vp, _i32]),
    "nunet_plan_set_lane_priority": (_i32, [_vp, _i32]),
    "nunet_plan_reset_lanes": (_i32, [_vp]),
    "nunet_plan_set_lanes": (_i32, [_vp, C.POINTER(_vp), _i32]),
    "nunet_profile_begin": (_i32, []),
    "nunet_profile_end": (_i32, [C.POINTER(ProfEntry), _i32, C.POINTER(_i32)]),
    "nunet_debug_spin": (_i32, [_i32, _i32, _vp]),
    "nunet_debug_stamp": (_i32, [_vp, _vp]),
    "nunet_graph_begin": (_i32, [_vp]),
    "nunet_graph_end": (_i32, [_vp, C.POINTER(_vp)]),
    "nunet_graph_launch": (_i32, [_vp, _vp]),
    "nunet_graph_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "nunet_graph_destroy": (None, [_vp]),
    "nunet_seg_begin": (_i32, [_vp, _i32]),
    "nunet_seg_end": (_i32, [_vp, C.POINTER(_vp)]),
    "nunet_seg_launch": (_i32, [_vp, _vp]),
    "nunet_seg_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "nunet_seg_destroy": (None, [_vp]),
    "nunet_plan_stamps_read": (_i32, [_vp, _i32, C.POINTER(C.c_uint64), _i32, C.POINTER(_i32), C.c_char_p, _i32]),
}

_lib = None


class NunetError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libnunet.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise NunetError("building libnunet.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NunetError("libnunet.so is missing at %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU/PyTorch fallback for the HIP path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def profile_begin():
    check(lib().nunet_profile_begin(), "nunet_profile_begin")


def profile_end():
    """[{name, launches, ms, flops, bytes}] per kernel class; synchronises with the device."""
    arr = (ProfEntry * 32)()
    n = _i32(0)
    check(lib().nunet_profile_end(arr, 32, C.byref(n)), "nunet_profile_end")
    return [dict(name=arr[i].name.decode(), launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops,
                 bytes=arr[i].bytes) for i in range(n.value) if arr[i].launches]


def check(rc, what=""):
    if rc != 0:
        raise NunetError("%s failed (%d): %s" % (what or "libnunet call", rc, lib().nunet_last_error().decode()))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


LOSS_BCE_DICE, LOSS_LOVASZ_HINGE = 0, 1


def nbytes(t):
    """size in bytes a tensor hands to the library next to its pointer (every workspace / arena argument has one)"""
    return 0 if t is None else t.numel() * t.element_size()


def ptr(t, byte_offset=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + byte_offset)


def require_gpu_tensor(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise NunetError("%s must live on the MI355X (got %s); this package has no CPU path" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise NunetError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise NunetError("%s must be contiguous" % name)
