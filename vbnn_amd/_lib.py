"""ctypes binding of the C ABI in include/vbnn_hip.h (vbnn_amd/lib/libvbnn_hip.so).

There is NO fallback: if the shared library is missing, or a call returns a non-zero status,
this module raises. The product path never routes through oracle/ or through PyTorch math.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBNN_HIP_LIB: another build of the same library (A/B builds; the name lua/vbnn_ffi.lua reads too)
LIB_PATH = os.environ.get("VBNN_HIP_LIB") or os.path.join(_HERE, "lib", "libvbnn_hip.so")
CSRC = os.path.join(_HERE, "csrc")

OK = 0
F32, BF16 = 0, 1
PACK_COPY, PACK_EXP, PACK_SQUARE, PACK_MUL, PACK_RELU, PACK_RELU_SQUARE = range(6)
KPAD = 64
STREAM_EPS, STREAM_ZETA, STREAM_INIT, STREAM_DATA, STREAM_HEINIT = 1, 2, 3, 4, 5

_vp, _i64, _u64, _u32, _f, _i = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_float, C.c_int


class VbnnError(RuntimeError):
    pass


class FwdArgs(C.Structure):
    _fields_ = [("w", _vp), ("w2", _vp), ("x", _vp), ("x2", _vp), ("ld_w", _i64), ("ld_x", _i64),
                ("N", _i64), ("I", _i64), ("O", _i64), ("bias", _vp),
                ("seed", _u64), ("layer", _u32), ("draw", _u32), ("row0", _i64),
                ("y", _vp), ("ld_y", _i64), ("r", _vp), ("ld_r", _i64), ("r_packed", _i), ("relu", _i),
                ("h", _vp), ("h2", _vp), ("ld_h", _i64), ("hT", _vp), ("h2T", _vp), ("ld_hT", _i64), ("rows_per_draw", _i64),
                ("draw_dev", _vp), ("head_w3", _vp), ("head_ld_w", _i64), ("head_C", _i64), ("head_slots", _vp)]


class DxArgs(C.Structure):
    _fields_ = [("wT", _vp), ("w2T", _vp), ("g", _vp), ("gv", _vp), ("ld_wT", _i64), ("ld_g", _i64),
                ("N", _i64), ("I", _i64), ("O", _i64), ("x", _vp), ("ld_x", _i64),
                ("gx", _vp), ("ld_gx", _i64), ("relu_mask", _i), ("r_prev", _vp), ("ld_r_prev", _i64), ("r_prev_packed", _i),
                ("g_prev", _vp), ("gv_prev", _vp), ("ld_gp", _i64),
                ("gT_prev", _vp), ("gvT_prev", _vp), ("ld_gpT", _i64),
                ("w", _vp), ("w2", _vp), ("ld_w", _i64)]


class PrepDesc(C.Structure):          # vbnn_prep_desc
    _fields_ = [("means", _vp), ("lvars", _vp), ("O", _i64), ("I", _i64), ("mu_s", _vp), ("var_s", _vp), ("ld_w", _i64),
                ("muT_s", _vp), ("varT_s", _vp), ("ld_wT", _i64), ("stats", _vp)]


class PackDesc(C.Structure):          # vbnn_pack_desc
    _fields_ = [("src", _vp), ("rows", _i64), ("cols", _i64), ("ld_src", _i64), ("dst", _vp), ("ld_dst", _i64),
                ("dstT", _vp), ("ld_dstT", _i64)]


class AdamCfg(C.Structure):           # vbnn_adam_cfg
    _fields_ = [("lr", _f), ("beta1", _f), ("beta2", _f), ("eps", _f), ("lambda_", _f), ("t", _i64)]


class UpdateDesc(C.Structure):        # vbnn_update_desc
    _fields_ = [("means", _vp), ("lvars", _vp), ("O", _i64), ("I", _i64), ("mu_s", _vp), ("var_s", _vp), ("ld_w", _i64),
                ("muT_s", _vp), ("varT_s", _vp), ("ld_wT", _i64), ("stats", _vp), ("grad_mu", _vp), ("grad_lv", _vp),
                ("m_mu", _vp), ("v_mu", _vp), ("m_lv", _vp), ("v_lv", _vp), ("mu", AdamCfg), ("lv", AdamCfg),
                ("bias", _vp), ("grad_bias", _vp), ("lr_bias", _f), ("B", _f), ("log14", _vp), ("kl_add", _f)]


class DwArgs(C.Structure):
    _fields_ = [("xT", _vp), ("x2T", _vp), ("gT", _vp), ("gvT", _vp), ("ld_n", _i64),
                ("N", _i64), ("I", _i64), ("O", _i64), ("scale", _f), ("accumulate", _i),
                ("gradWeight", _vp), ("gradSum", _vp), ("seed", _u64), ("layer", _u32), ("draw", _u32),
                ("lvars", _vp), ("grad_mu", _vp), ("grad_lv", _vp), ("means", _vp), ("stats", _vp),
                ("B", _f), ("S", _f), ("kl_scale", _f), ("gradBias", _vp),
                ("x", _vp), ("x2", _vp), ("g", _vp), ("gv", _vp), ("ld_x", _i64), ("ld_g", _i64),
                ("mu_s", _vp), ("var_s", _vp), ("ld_w", _i64), ("part", _i), ("draw_dev", _vp)]


class BoxInfo(C.Structure):           # vbnn_box_info
    _fields_ = [("mfma_clock_ghz", C.c_double), ("mfma_tflops", C.c_double), ("mfma_ms", C.c_double), ("hbm_TBps", C.c_double),
                ("hbm_ms", C.c_double), ("hbm_bytes", _i64), ("cus", _i), ("reserved", _i)]


class HeadArgs(C.Structure):          # vbnn_head_args
    _fields_ = [("h", _vp), ("ld_h", _i64), ("w3", _vp), ("ld_w", _i64), ("bias", _vp), ("target", _vp),
                ("N", _i64), ("H", _i64), ("C", _i64), ("rows_per_draw", _i64), ("inv_n", _f), ("accumulate", _i),
                ("logits", _vp), ("out", _vp), ("g_logits", _vp), ("loss_sum_dev", _vp), ("correct_dev", _vp),
                ("gradWeight", _vp), ("gradBias", _vp), ("gradBias_prev", _vp), ("relu_mask", _i), ("r_prev_packed", _i),
                ("r_prev", _vp), ("ld_r_prev", _i64), ("g_prev", _vp), ("gv_prev", _vp), ("ld_gp", _i64),
                ("gT_prev", _vp), ("gvT_prev", _vp), ("ld_gpT", _i64), ("logit_slots", _vp), ("n_slots", _i64)]


_SIGS = {
    "vbnn_abi_version": ([], _i),
    "vbnn_last_error": ([], C.c_char_p),
    "vbnn_debug_set": ([_i, _i], _i),
    "vbnn_kmajor_supported": ([_i64, _i64, _i64], _i),
    "vbnn_kmajor_supported_dw": ([_i64, _i64, _i64, _i], _i),
    "vbnn_ctx_kmajor_supported": ([_vp, _i64, _i64, _i64], _i),
    "vbnn_ctx_kmajor_supported_dw": ([_vp, _i64, _i64, _i64, _i], _i),
    "vbnn_ctx_create": ([_i, _vp, C.POINTER(_vp)], _i),
    "vbnn_ctx_create_cu_budget": ([_i, _i, C.POINTER(_vp)], _i),
    "vbnn_ctx_stream": ([_vp, C.POINTER(_vp), C.POINTER(_i)], _i),
    "vbnn_ctx_destroy": ([_vp], _i),
    "vbnn_ctx_set_stream": ([_vp, _vp], _i),
    "vbnn_sync": ([_vp], _i),
    "vbnn_buf_alloc": ([_vp, C.c_size_t, C.POINTER(_vp)], _i),
    "vbnn_buf_free": ([_vp, _vp], _i),
    "vbnn_buf_zero": ([_vp, _vp, C.c_size_t], _i),
    "vbnn_buf_upload": ([_vp, _vp, _vp, C.c_size_t], _i),
    "vbnn_buf_download": ([_vp, _vp, _vp, C.c_size_t], _i),
    "vbnn_fill_normal": ([_vp, _vp, _i64, _i64, _i64, _u64, _u32, _u32, _u32, _i64, _f], _i),
    "vbnn_fill_normal_hw": ([_vp, _vp, _i64, _i64, _i64, _u64, _u32, _u32, _u32, _i64, _f], _i),
    "vbnn_box_muller_forms": ([_vp, _vp, _vp, _vp, _vp, _i64], _i),
    "vbnn_compute_prior": ([_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i),
    "vbnn_wn_sample": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _u64, _u32, _u32], _i),
    "vbnn_pack": ([_vp, _i, _i, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64], _i),
    "vbnn_forward": ([_vp, _i, C.POINTER(FwdArgs)], _i),
    "vbnn_forward_head_slots": ([_vp, _i, _i64, _i64, _i64, _i64], _i),
    "vbnn_head_forward_slots": ([_vp, _vp, _i64, _vp, _vp, _i64, _i64, _f, _vp, _vp, _vp, _i, _vp, _vp, _i64], _i),
    "vbnn_grad_input": ([_vp, _i, C.POINTER(DxArgs)], _i),
    "vbnn_acc_grad_parameters": ([_vp, _i, C.POINTER(DwArgs)], _i),
    "vbnn_backward_pair": ([_vp, _i, C.POINTER(DxArgs), C.POINTER(DwArgs)], _i),
    "vbnn_head_forward_backward": ([_vp, _i, C.POINTER(HeadArgs)], _i),
    "vbnn_acc_grad_bias": ([_vp, _i, _vp, _i64, _i64, _i64, _f, _i, _vp], _i),
    "vbnn_prep_layer": ([_vp, _i, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp], _i),
    "vbnn_compute_mugrads": ([_vp, _vp, _vp, _f, _f, _vp, _vp, _i64], _i),
    "vbnn_compute_vargrads": ([_vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i64], _i),
    "vbnn_calc_lc": ([_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i64], _i),
    "vbnn_pack_input": ([_vp, _i, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64], _i),
    "vbnn_adam_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i64, _vp], _i),
    "vbnn_sgd_step": ([_vp, _vp, _vp, _i64, _f], _i),
    "vbnn_update": ([_vp, _i, _i, _vp, _vp], _i),
    "vbnn_comm_unique_id": ([_vp], _i),
    "vbnn_comm_create": ([_vp, _i, _i, _vp, C.POINTER(_vp)], _i),
    "vbnn_comm_destroy": ([_vp], _i),
    "vbnn_comm_info": ([_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _i),
    "vbnn_allreduce_grads": ([_vp, _vp, _i64], _i),
    "vbnn_comm_finish": ([_vp], _i),
    "vbnn_allreduce_grads_bf16": ([_vp, _vp, _i64], _i),
    "vbnn_cast_grads": ([_vp, _i, _vp, _vp, _i64], _i),
    "vbnn_comm_allgather_u64": ([_vp, _vp, _vp], _i),
    "vbnn_stats_combine": ([_vp, _i, _i, _vp, _vp], _i),
    "vbnn_transpose_packed": ([_vp, _i, _vp, _i64, _i64, _i64, _vp, _i64], _i),
    "vbnn_comm_reduce_scatter": ([_vp, _vp, _i64], _i),
    "vbnn_comm_all_gather": ([_vp, _vp, _i64], _i),
    "vbnn_p2p_reduce_scatter": ([_vp, C.c_size_t, _i64], _i),
    "vbnn_p2p_all_gather": ([_vp, C.c_size_t, _i64], _i),
    "vbnn_p2p_create": ([_vp, _i, _i, C.c_size_t, C.POINTER(_vp), C.POINTER(_vp), _vp], _i),
    "vbnn_p2p_connect": ([_vp, _vp], _i),
    "vbnn_p2p_allreduce": ([_vp, C.c_size_t, _i64], _i),
    "vbnn_p2p_finish": ([_vp], _i),
    "vbnn_p2p_status": ([_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_uint)], _i),
    "vbnn_p2p_set_timeout": ([_vp, C.c_double], _i),
    "vbnn_p2p_clear_status": ([_vp], _i),
    "vbnn_p2p_destroy": ([_vp], _i),
    "vbnn_p2p_set_grid": ([_vp, _i, _i], _i),
    "vbnn_p2p_standin": ([_vp, _i, C.c_double], _i),
    "vbnn_box_calibrate": ([_vp, C.POINTER(BoxInfo)], _i),
    "vbnn_sample": ([_vp, _vp, _u32], _i),
    "vbnn_capture_begin": ([_vp], _i),
    "vbnn_capture_end": ([_vp, C.POINTER(_vp)], _i),
    "vbnn_graph_launch": ([_vp], _i),
    "vbnn_graph_info": ([_vp, C.POINTER(_i), C.POINTER(_i)], _i),
    "vbnn_graph_destroy": ([_vp], _i),
    "vbnn_relu_forward": ([_vp, _vp, _vp, _i64], _i),
    "vbnn_relu_backward": ([_vp, _vp, _vp, _vp, _i64], _i),
    "vbnn_logsoftmax_nll": ([_vp, _vp, _i64, _vp, _i64, _i64, _f, _vp, _vp, _vp, _vp], _i),
    "vbnn_prepare": ([_vp, _i, _i, _vp, _vp], _i),
    "vbnn_head_forward": ([_vp, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _f, _vp, _vp, _vp, _i, _vp, _vp, _i64], _i),
    "vbnn_head_backward": ([_vp, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp,
                           _vp, _i64, _vp, _vp, _i64], _i),
    "vbnn_mse_forward": ([_vp, _vp, _i64, _vp, _i64, _i64, _i64, _f, _vp, _i64, _i, _vp], _i),
    "vbnn_mse_backward": ([_vp, _vp, _i64, _vp, _i64, _i64, _i64, _f, _vp, _i64], _i),
    "vbnn_nll_forward": ([_vp, _vp, _i64, _vp, _i64, _i64, _f, _vp, _vp], _i),
    "vbnn_nll_backward": ([_vp, _vp, _i64, _i64, _f, _vp], _i),
    "vbnn_logsoftmax_backward": ([_vp, _vp, _vp, _vp, _i64, _i64], _i),
}


def exported_symbols():
    """Every symbol include/vbnn_hip.h declares (kept in step by tests/test_abi.py)."""
    return sorted(_SIGS)


def build(verbose=False):
    """Compile libvbnn_hip.so for gfx950 with hipcc (vbnn_amd/csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VbnnError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)       # AttributeError if the library does not export it
            fn.argtypes = args
            fn.restype = res
        _lib = L
    return _lib


def check(status):
    if status != OK:
        msg = lib().vbnn_last_error()
        raise VbnnError(f"libvbnn_hip status {status}: {msg.decode() if msg else '?'}")


def pad_ld(k):
    return (int(k) + KPAD - 1) // KPAD * KPAD
