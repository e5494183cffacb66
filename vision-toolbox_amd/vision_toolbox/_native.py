"""ctypes binding of libvt_amd.so (C-ABI declared in include/vt_amd.h).

There is no fallback: if the shared library is missing or a call fails, the
caller gets an exception.  The reference has no native layer at all (it is
pure torch.nn, vision_toolbox/components.py:13-46); this module is the thin
boundary the north_star asks for.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("VT_AMD_LIB", _HERE.parent / "csrc" / "libvt_amd.so"))

VT_OK, VT_ERR_INVALID, VT_ERR_UNSUPPORTED, VT_ERR_HIP = 0, 1, 2, 3
VT_F32, VT_BF16, VT_I64 = 0, 1, 2  # (VT_I64: collectives only)
VT_MAX_TAPS = 36
VT_CONV_RELU, VT_CONV_STATS, VT_CONV_RESIDUAL, VT_CONV_AFFINE, VT_CONV_D2S, VT_CONV_NOSTORE = 1, 2, 4, 8, 16, 32
VT_STAT_REPLICAS = 16
# a statistics buffer is int64[VT_STAT_REPLICAS][2][C][2]: value = hi * 2^12 + lo / 2^33 (vt_amd.h)


def stat_floats(C_: int) -> int:
    """size of a statistics buffer in 4-byte units (VT_STAT_BYTES(C) / 4)"""
    return VT_STAT_REPLICAS * 2 * C_ * 4


def stats_buffer(C_: int, device="cuda"):
    """a zeroed statistics buffer for C channels"""
    import torch

    return torch.zeros(VT_STAT_REPLICAS, 2, C_, 2, dtype=torch.int64, device=device)


def stats_decode(buf):
    """statistics buffer as the kernels fill it -> float64 [2][C] (exact sum over the replicas)"""
    q = buf.sum(0)
    return q[..., 0].double() * 4096.0 + q[..., 1].double() / float(1 << 33)


def stats_encode(buf, which: int, values, replica: int = 0):
    """add `values` [C] (natural units) into replica `replica` of row `which` of a statistics buffer (tests)"""
    import torch

    v = values.double()
    hi = torch.trunc(v / 4096.0)
    buf[replica, which, :, 0] += hi.to(torch.int64)
    buf[replica, which, :, 1] += torch.round((v - hi * 4096.0) * float(1 << 33)).to(torch.int64)


VT_OP_MAX_PTR, VT_OP_MAX_INT, VT_OP_MAX_FLT, VT_MAX_BASES = 24, 110, 8, 16

(
    OP_MEMSET,
    OP_CONV_IGEMM,
    OP_CONV_WGRAD,
    OP_PACK_DGRAD,
    OP_BN_FINALIZE,
    OP_BN_EVAL_COEFFS,
    OP_BN_ACT_APPLY,
    OP_BN_BWD_REDUCE,
    OP_BN_BWD_FINALIZE,
    OP_BN_BWD_APPLY,
    OP_MAXPOOL_FWD,
    OP_MAXPOOL_BWD,
    OP_AVGPOOL_FWD,
    OP_AVGPOOL_BWD,
    OP_ESE_FWD,
    OP_ESE_BWD,
    OP_COLSUM,
    OP_XENT,
    OP_SGD,
    OP_COPY2D,
    OP_NCHW_TO_NHWC,
    OP_NHWC_TO_NCHW,
    OP_FORK,
    OP_JOIN,
    OP_RESAMPLE_FWD,
    OP_RESAMPLE_BWD,
    OP_FORK_MARK,
    OP_FORK_WAIT,
    OP_STEM_BWD_REDUCE,
    OP_STEM_BWD_COMBINE,
    OP_FIXED_TO_F32,
    OP_PW_STATS,
    OP_PW_APPLY,
    OP_PW_REDUCE,
    OP_PW_BWD,
    OP_STEM_BWD_S2,
    OP_ALLREDUCE,
    OP_STAT_SYNC,
    OP_XENT_EVAL,
    OP_CONV_DGRAD_BNRED,
    OP_BN_BWD_FUSED,
    OP_BN_FIN_APPLY,
    OP_BN_BWD_FIN_APPLY,
    OP_DWCONV_FWD,
    OP_DWCONV_DGRAD,
    OP_DWCONV_WGRAD,
    OP_PW_APPLY_FIN,
    OP_PW_BWD_FIN,
) = range(1, 49)
OP_SIDE_STREAM = 0x10000  # OR-ed into Op.kind: enqueue on the side stream

OP_NAMES = {
    OP_MEMSET: "memset",
    OP_CONV_IGEMM: "conv_igemm",
    OP_CONV_WGRAD: "conv_wgrad",
    OP_PACK_DGRAD: "pack_dgrad",
    OP_BN_FINALIZE: "bn_finalize",
    OP_BN_EVAL_COEFFS: "bn_eval_coeffs",
    OP_BN_ACT_APPLY: "bn_act_apply",
    OP_BN_BWD_REDUCE: "bn_bwd_reduce",
    OP_BN_BWD_FINALIZE: "bn_bwd_finalize",
    OP_BN_BWD_APPLY: "bn_bwd_apply",
    OP_MAXPOOL_FWD: "maxpool_fwd",
    OP_MAXPOOL_BWD: "maxpool_bwd",
    OP_AVGPOOL_FWD: "avgpool_fwd",
    OP_AVGPOOL_BWD: "avgpool_bwd",
    OP_ESE_FWD: "ese_fwd",
    OP_ESE_BWD: "ese_bwd",
    OP_COLSUM: "colsum",
    OP_XENT: "xent",
    OP_XENT_EVAL: "xent_eval",
    OP_CONV_DGRAD_BNRED: "conv_dgrad_bnred",
    OP_BN_BWD_FUSED: "bn_bwd_fused",
    OP_BN_FIN_APPLY: "bn_fin_apply",
    OP_BN_BWD_FIN_APPLY: "bn_bwd_fin_apply",
    OP_DWCONV_FWD: "dwconv_fwd",
    OP_DWCONV_DGRAD: "dwconv_dgrad",
    OP_DWCONV_WGRAD: "dwconv_wgrad",
    OP_PW_APPLY_FIN: "pw_apply_fin",
    OP_PW_BWD_FIN: "pw_bwd_fin",
    OP_SGD: "sgd",
    OP_COPY2D: "copy2d",
    OP_NCHW_TO_NHWC: "nchw_to_nhwc",
    OP_NHWC_TO_NCHW: "nhwc_to_nchw",
    OP_FORK: "fork",
    OP_JOIN: "join",
    OP_FORK_MARK: "fork_mark",
    OP_FORK_WAIT: "fork_wait",
    OP_STEM_BWD_REDUCE: "stem_bwd_reduce",
    OP_STEM_BWD_COMBINE: "stem_bwd_combine",
    OP_FIXED_TO_F32: "fixed_to_f32",
    OP_RESAMPLE_FWD: "resample_fwd",
    OP_RESAMPLE_BWD: "resample_bwd",
    OP_PW_STATS: "pw_stats",
    OP_PW_APPLY: "pw_apply",
    OP_PW_REDUCE: "pw_reduce",
    OP_PW_BWD: "pw_bwd",
    OP_STEM_BWD_S2: "stem_bwd_s2",
    OP_ALLREDUCE: "allreduce",
    OP_STAT_SYNC: "stat_sync",
}


class ConvDesc(C.Structure):
    """vt_conv_desc"""

    _fields_ = [
        ("dtype", C.c_int32),
        ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("sh", C.c_int32), ("sw", C.c_int32), ("h0", C.c_int32), ("w0", C.c_int32),
        ("Cout", C.c_int32), ("ldy", C.c_int32),
        ("oH", C.c_int32), ("oW", C.c_int32),
        ("oHs", C.c_int32), ("oWs", C.c_int32), ("oh0", C.c_int32), ("ow0", C.c_int32),
        ("ldw", C.c_int32), ("ldr", C.c_int32), ("flags", C.c_int32), ("ntaps", C.c_int32),
        ("dh", C.c_int8 * VT_MAX_TAPS),
        ("dw", C.c_int8 * VT_MAX_TAPS),
    ]  # fmt: skip


class PwDesc(C.Structure):
    """vt_pw_desc"""

    _fields_ = [
        ("dtype", C.c_int32), ("K", C.c_int32), ("ngroups", C.c_int32), ("relu", C.c_int32),
        ("M", C.c_int64),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("C", C.c_int32 * 2),
        ("w", C.c_void_p * 2),
        ("ldw", C.c_int32 * 2),
    ]  # fmt: skip


class BnFinFwd(C.Structure):
    """vt_bn_fin_fwd"""

    _fields_ = [("stats", C.c_void_p), ("count", C.c_double), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("momentum", C.c_float), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p)]


class BnFinBwd(C.Structure):
    """vt_bn_fin_bwd"""

    _fields_ = [("sums", C.c_void_p), ("count", C.c_double), ("pscale", C.c_double), ("train", C.c_int32),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class PackItem(C.Structure):
    """vt_pack_item"""

    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("ldw", C.c_int32), ("nsel", C.c_int32), ("Cout", C.c_int32),
                ("ntaps", C.c_int32), ("Cin", C.c_int32), ("sel", C.c_int8 * 36)]


class Ptr(C.Structure):
    """vt_ptr"""

    _fields_ = [("base", C.c_int32), ("pad", C.c_int32), ("offset", C.c_int64)]


class Op(C.Structure):
    """vt_op"""

    _fields_ = [
        ("kind", C.c_int32),
        ("tag", C.c_int32),
        ("ptr", Ptr * VT_OP_MAX_PTR),
        ("i", C.c_int32 * VT_OP_MAX_INT),
        ("f", C.c_double * VT_OP_MAX_FLT),
    ]


assert C.sizeof(ConvDesc) == 24 * 4 + 2 * VT_MAX_TAPS
assert C.sizeof(ConvDesc) // 4 + 1 <= VT_OP_MAX_INT

# every symbol include/vt_amd.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
SYMBOLS = {
    "vt_version": (_i32, []),
    "vt_last_error": (C.c_char_p, []),
    "vt_last_kernel_name": (C.c_char_p, []),
    "vt_launch_count": (_u64, []),
    "vt_set_knob": (_i32, [C.c_char_p, _i32]),
    "vt_memset": (_i32, [_vp, _i32, _u64, _vp]),
    "vt_conv_igemm": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vt_conv_dgrad_bnred": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "vt_conv_wgrad": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i32, _vp]),
    "vt_dwconv_fwd": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp] + [_i32] * 9 + [_vp]),
    "vt_dwconv_dgrad": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32] + [_i32] * 9 + [_vp]),
    "vt_dwconv_wgrad": (_i32, [_vp, _i32, _vp, _i32, _vp] + [_i32] * 9 + [_vp]),
    "vt_bn_act_bwd_fused": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _f64, _f64, _i32, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _i32, _vp]),
    "vt_bn_bwd_fused_timeouts": (_i32, [C.POINTER(C.c_uint32)]),
    "vt_bn_finalize_apply": (_i32, [_vp, _i32, _f64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                                    _vp, _i32, _i64, _i32, _i32, _vp]),
    "vt_bn_bwd_finalize_apply": (_i32, [_vp, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                        _i32, _i64, _i32, _i32, _vp]),
    "vt_pack_dgrad_filter_batch": (_i32, [_vp, _i32, _vp]),
    "vt_bn_eval_coeffs_batch": (_i32, [_vp, _i32, _vp]),
    "vt_pack_dgrad_filter": (_i32, [_vp, _i32, _i32, _vp, _i32, C.POINTER(_i32), _i32, _i32, _i32, _i32, _vp]),
    "vt_bn_finalize": (_i32, [_vp, _i32, _f64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vt_stat_fold": (_i32, [_vp, _i32, _vp]),
    "vt_stat_sync": (_i32, [_vp, _i32, _vp]),
    "vt_comm_unique_id": (_i32, [_vp]),
    "vt_comm_init": (_i32, [_vp, _i32, _i32]),
    "vt_comm_world": (_i32, []),
    "vt_comm_init_stat": (_i32, [_vp]),
    "vt_comm_has_stat": (_i32, []),
    "vt_comm_destroy": (_i32, []),
    "vt_allreduce_bucket": (_i32, [_vp, C.c_int64, _i32, _vp]),
    "vt_bn_eval_coeffs": (_i32, [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "vt_bn_act_apply": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "vt_bn_act_bwd_reduce": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "vt_bn_bwd_finalize": (_i32, [_vp, _i32, _f64, _f64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "vt_conv_wgrad_slabs": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    "vt_conv_wgrad_group": (_i32, [C.POINTER(ConvDesc), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i32, _vp]),
    "vt_fixed_to_f32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "vt_colsum_fixed": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "vt_stem_bn_bwd_scratch_bytes": (_i64, [_i32]),
    "vt_stem_bn_bwd_reduce": (_i32, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "vt_stem_bn_bwd_combine": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "vt_stem_bn_bwd_combine_y": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "vt_stem_bn_bwd_s2": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "vt_bn_act_bwd_apply": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "vt_pw_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "vt_pw_apply_supported": (_i32, [_i32, _i32, _i32]),
    "vt_pw_fwd_stats": (_i32, [C.POINTER(PwDesc), C.POINTER(_vp), _vp]),
    "vt_pw_fwd_apply": (_i32, [C.POINTER(PwDesc), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp), C.POINTER(_i32), _vp]),
    "vt_pw_bwd_reduce": (_i32, [C.POINTER(PwDesc), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp), _vp]),
    "vt_pw_bwd_apply": (_i32, [C.POINTER(PwDesc), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp), _vp, _i32, _vp, _i32,
                               C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp), C.POINTER(_i32), _vp]),
    "vt_pw_fwd_apply_finalize": (_i32, [C.POINTER(PwDesc), C.POINTER(BnFinFwd), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp),
                                        C.POINTER(_i32), _vp]),
    "vt_pw_bwd_apply_finalize": (_i32, [C.POINTER(PwDesc), _vp, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(BnFinBwd), C.POINTER(_vp),
                                        _vp, _i32, _vp, _i32, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_vp), C.POINTER(_i32), _vp]),
    "vt_bn_act_apply_pool": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                    _vp]),
    "vt_bn_act_bwd_reduce_pool": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                         _vp]),
    "vt_bn_act_bwd_apply_pool": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                        _vp]),
    "vt_maxpool3x3s2_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_maxpool3x3s2_bwd": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_global_avgpool_fwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_global_avgpool_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_resample2x_add_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_resample2x_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_ese_gate_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_ese_gate_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_colsum": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "vt_softmax_xent": (_i32, [_vp, _i32, _vp, _f32, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vt_softmax_xent_mix": (_i32, [_vp, _i32, _vp, _f32, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vt_softmax_xent_eval": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vt_mix_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vt_sgd_momentum": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _f32, _f32, _f32, _f32, _vp, _vp]),
    "vt_copy2d": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i64, _i32, _i32, _vp]),
    "vt_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_nhwc_to_nchw": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vt_run_ops": (_i32, [C.POINTER(Op), _i32, C.POINTER(_vp), _i32, _vp]),
    "vt_run_ops_streams": (_i32, [C.POINTER(Op), _i32, C.POINTER(_vp), _i32, _vp, _vp]),
    "vt_run_ops_streams_ex": (_i32, [C.POINTER(Op), _i32, C.POINTER(_vp), _i32, _vp, _vp, _i32]),
    "vt_stream_wait": (_i32, [_vp, _vp]),
    "vt_graph_create": (_i32, [C.POINTER(Op), _i32, C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "vt_graph_launch": (_i32, [_vp, _vp]),
    "vt_graph_destroy": (_i32, [_vp]),
    "vt_event_create": (_i32, [C.POINTER(_vp)]),
    "vt_event_record": (_i32, [_vp, _vp]),
    "vt_event_elapsed_ms": (_i32, [_vp, _vp, C.POINTER(_f32)]),
    "vt_event_destroy": (_i32, [_vp]),
}  # fmt: skip


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvt_amd error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {LIB_PATH.parent}`). There is no CPU/eager fallback for the HIP path."
            )
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().vt_last_error().decode(errors="replace")


def last_kernel_name() -> str:
    return lib().vt_last_kernel_name().decode(errors="replace")


def check(rc: int) -> None:
    if rc != VT_OK:
        raise NativeError(rc, last_error())


def set_knob(name: str, value: int) -> None:
    """set a dispatcher switch (vt_set_knob): the environment is only read once per process"""
    check(lib().vt_set_knob(name.encode(), int(value)))


def launch_count() -> int:
    return int(lib().vt_launch_count())


VT_RUN_LEAVE_SIDE_OPEN = 1


def run_ops(ops, n: int, bases, stream: int, side: int = 0, leave_side_open: bool = False) -> None:
    """ops: (Op * n) array; bases: list of device addresses (ints or None); `side` is the
    stream handle for ops flagged OP_SIDE_STREAM (0: run them in line).  `leave_side_open`: do not
    order `stream` behind the side stream on return (a list run in segments; the caller joins later)."""
    arr = (C.c_void_p * len(bases))(*[C.c_void_p(b) if b else None for b in bases])
    check(lib().vt_run_ops_streams_ex(ops, n, arr, len(bases), C.c_void_p(stream), C.c_void_p(side) if side else None,
                                      VT_RUN_LEAVE_SIDE_OPEN if (leave_side_open and side) else 0))


def stream_wait(waiter: int, signaller: int) -> None:
    """order stream `waiter` behind everything enqueued on `signaller` so far (no host synchronisation)."""
    check(lib().vt_stream_wait(C.c_void_p(waiter), C.c_void_p(signaller)))


class Graph:
    """A captured hipGraph of an op list (vt_graph_*)."""

    def __init__(self, ops, n: int, bases):
        arr = (C.c_void_p * len(bases))(*[C.c_void_p(b) if b else None for b in bases])
        h = C.c_void_p()
        check(lib().vt_graph_create(ops, n, arr, len(bases), C.byref(h)))
        self._h = h

    def launch(self, stream: int) -> None:
        check(lib().vt_graph_launch(self._h, C.c_void_p(stream)))

    def __del__(self):
        try:
            if self._h:
                lib().vt_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Event:
    """HIP event on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib().vt_event_create(C.byref(self._h)))

    def record(self, stream: int) -> None:
        check(lib().vt_event_record(self._h, C.c_void_p(stream)))

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        check(lib().vt_event_elapsed_ms(self._h, end._h, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            lib().vt_event_destroy(self._h)
        except Exception:
            pass
