"""ctypes binding of librcgan_hip.so (the C ABI declared in include/rcgan_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCGAN_LIB_PATH") or os.path.join(_HERE, "librcgan_hip.so")   # override: kernel probes only
# the same sources built with -DRCGAN_HALF_FP16=1: its 16-bit activation dtype is IEEE half instead of bf16
LIB_PATH_F16 = os.environ.get("RCGAN_LIB_PATH_F16") or os.path.join(_HERE, "librcgan_hip_f16.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
CONV_IN_UPSAMPLE2X, CONV_IN_RELU, CONV_ACCUMULATE, CONV_FORCE_DIRECT, CONV_OUT_MEANPOOL2, CONV_RESID_UPSAMPLE2X = 1, 2, 4, 8, 16, 32
LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_NEG_MEAN, LOSS_CE_ONES, LOSS_CE_ZEROS = 0, 1, 2, 3, 4
QUERY_TR_READ = 0

RCGAN_EHIP, RCGAN_ERCCL = -4, -5
EUNSUPPORTED_SHAPE = -2
ERRORS = {-1: "RCGAN_EINVALID_ARG", -2: "RCGAN_EUNSUPPORTED_SHAPE", -3: "RCGAN_EWORKSPACE_TOO_SMALL",
          -4: "RCGAN_EHIP", -5: "RCGAN_ERCCL"}


def source_hash():
    """sha256 (16 hex digits) over the kernel sources and the ABI header: profiles record it so that a measurement taken on another
    build of the kernels is never attached to this one (bench.py roofline.traffic)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class RcganError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "RCGAN_E?"), code, msg))
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cin", C.c_int),
                ("cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("dtype", C.c_int), ("flags", C.c_int)]


class PrepareItem(C.Structure):
    _fields_ = [("desc", ConvDesc), ("w", C.c_void_p), ("sigma", C.c_void_p), ("prepared", C.c_void_p)]


class FragItem(C.Structure):
    _fields_ = [("item", C.c_int), ("ctn", C.c_int), ("ss", C.c_int), ("fwd", C.c_void_p), ("bwd", C.c_void_p)]


class HeadDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("d", C.c_int), ("v", C.c_int), ("e_dim", C.c_int),
                ("rows_a", C.c_int), ("kind_a", C.c_int), ("kind_b", C.c_int), ("weight", C.c_float),
                ("labels_a", C.c_void_p), ("wts_a", C.c_void_p), ("dwts_a", C.c_void_p),
                ("labels_b", C.c_void_p), ("wts_b", C.c_void_p), ("dwts_b", C.c_void_p),
                ("x", C.c_void_p), ("dx", C.c_void_p), ("x_dtype", C.c_int), ("hw", C.c_int), ("act", C.c_int),
                ("E_pre", C.c_void_p), ("defer_ws", C.c_void_p), ("defer_ws_bytes", C.c_size_t)]


class EmbedDesc(C.Structure):
    _fields_ = [("v", C.c_int), ("e_dim", C.c_int), ("d", C.c_int),
                ("table", C.c_void_p), ("w_e", C.c_void_p), ("sigma_e", C.c_void_p), ("b_e", C.c_void_p), ("E", C.c_void_p)]


class StepInputsDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("dtype", C.c_int), ("images", C.c_void_p), ("x", C.c_void_p), ("pooled", C.c_void_p),
                ("noise_lo", C.c_float), ("noise_hi", C.c_float), ("seed", C.c_uint64), ("rng_state", C.c_void_p),
                ("fill", C.c_void_p), ("fill_count", C.c_size_t),
                ("fakes", C.c_void_p), ("fake_slice", C.c_void_p), ("n_slices", C.c_int)]


class SnItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("sigma", C.c_void_p), ("save", C.c_void_p),
                ("k", C.c_int), ("c", C.c_int), ("update", C.c_int)]


class SnBwdItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dwbar", C.c_void_p), ("dw", C.c_void_p), ("save", C.c_void_p),
                ("k", C.c_int), ("c", C.c_int), ("accumulate", C.c_int)]


class SnAdam(C.Structure):
    _fields_ = [("w", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("count", C.c_size_t),
                ("hyper", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clip", C.c_float),
                ("grad_scale", C.c_float), ("n_ranges", C.c_int), ("ranges", C.POINTER(C.c_size_t))]


P, I, F, SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
DP = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/rcgan_hip.h declares
SIGNATURES = {
    "rcgan_create": (I, [C.POINTER(P), I, P]),
    "rcgan_destroy": (I, [P]),
    "rcgan_last_error": (C.c_char_p, [P]),
    "rcgan_version": (C.c_char_p, []),
    "rcgan_half_dtype": (I, []),
    "rcgan_crc32c": (C.c_uint, [C.c_uint, P, SZ]),
    "rcgan_set_stream": (I, [P, P]),
    "rcgan_side_begin": (I, [P]),
    "rcgan_side_end": (I, [P]),
    "rcgan_side_join": (I, [P]),
    "rcgan_stream_sync": (I, [P]),
    "rcgan_event_record": (I, [P, I]),
    "rcgan_event_elapsed_ms": (I, [P, I, I, C.POINTER(F)]),
    "rcgan_debug_stamps": (I, [P, P]),
    "rcgan_prof_begin": (I, [P, I]),
    "rcgan_prof_end": (I, [P, C.POINTER(I), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rcgan_prof_executed_flops": (I, [P, C.POINTER(C.c_double)]),
    "rcgan_prof_bn_in_launches": (I, [P, C.POINTER(I)]),
    "rcgan_graph_begin": (I, [P]),
    "rcgan_reserve_scratch": (I, [P, C.c_size_t]),
    "rcgan_scratch_bytes": (I, [P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rcgan_graph_end": (I, [P, C.POINTER(I)]),
    "rcgan_graph_abort": (I, [P]),
    "rcgan_graph_launch": (I, [P, I]),
    "rcgan_graph_destroy": (I, [P, I]),
    "rcgan_conv_prepared_bytes": (SZ, [DP]),
    "rcgan_conv_prepare": (I, [P, DP, P, P, P]),
    "rcgan_conv_prepare_batch": (I, [P, C.POINTER(PrepareItem), I]),
    "rcgan_conv_prepare_batch_embed": (I, [P, C.POINTER(PrepareItem), I, C.POINTER(EmbedDesc)]),
    "rcgan_conv_prepare_batch_riders": (I, [P, C.POINTER(PrepareItem), I, C.POINTER(EmbedDesc), C.POINTER(StepInputsDesc)]),
    "rcgan_conv_prepare_batch_frags": (I, [P, C.POINTER(PrepareItem), I, C.POINTER(EmbedDesc), C.POINTER(StepInputsDesc), C.POINTER(FragItem), I]),
    "rcgan_conv_workspace_bytes": (SZ, [DP]),
    "rcgan_conv2d_fwd": (I, [P, DP, P, P, P, P]),
    "rcgan_conv_bn_in_ok": (I, [DP]),
    "rcgan_conv2d_fwd_bn": (I, [P, DP, P, P, P, P, I, P, P, P, P, P, I]),
    "rcgan_conv2d_fwd_bn_residual": (I, [P, DP, P, P, P, P, P, I, P, P, P, P, P, I]),
    "rcgan_conv_fused_pool_ok": (I, [DP]),
    "rcgan_conv_resid_up_ok": (I, [DP]),
    "rcgan_conv_wgrad_pool_ok": (I, [DP]),
    "rcgan_conv2d_fwd_residual": (I, [P, DP, P, P, P, P, P]),
    "rcgan_conv2d_bwd_data": (I, [P, DP, P, P, P, P, P, SZ]),
    "rcgan_conv2d_bwd_data_residual": (I, [P, DP, P, P, P, P, P, P, SZ]),
    "rcgan_conv2d_bwd_weight": (I, [P, DP, P, P, P, P, I, P, SZ]),
    "rcgan_conv2d_bwd_weight_group": (I, [P, I, P, P, P, P, P, I, P, SZ]),
    "rcgan_deconv2d_fwd": (I, [P, DP, P, P, P, P]),
    "rcgan_deconv2d_bwd_data": (I, [P, DP, P, P, P]),
    "rcgan_deconv2d_bwd_data_cols": (I, [P, DP, P, P, P, I]),
    "rcgan_deconv2d_bwd_weight": (I, [P, DP, P, P, P, P, I, P, SZ]),
    "rcgan_deconv2d_bwd_weight_concat_bytes": (SZ, [DP, I]),
    "rcgan_deconv2d_bwd_weight_concat": (I, [P, DP, P, P, I, P, P, P, I, P, SZ]),
    "rcgan_linear_fwd": (I, [P, I, I, I, I, P, P, P, P, P]),
    "rcgan_linear_bwd_data": (I, [P, I, I, I, I, P, P, P, P, I]),
    "rcgan_linear_bwd_weight": (I, [P, I, I, I, I, P, P, P, P, I, P, SZ]),
    "rcgan_linear_workspace_bytes": (SZ, [I, I, I]),
    "rcgan_bn_workspace_bytes": (SZ, [I, I]),
    "rcgan_bn_stats": (I, [P, I, I, I, P, F, P, P, P, P, F, P, SZ]),
    "rcgan_bn_apply_fwd": (I, [P, I, I, I, I, I, P, P, P, P, P, P, I, P, P, SZ]),
    "rcgan_bn_fwd_segments": (I, [P, I, I, I, I, I, I, P, P, P, P, F, I, P, P, P, P, SZ]),
    "rcgan_bn_bwd": (I, [P, I, I, I, I, I, P, P, P, P, P, P, P, I, P, I, P, P, I, P, SZ]),
    "rcgan_bn_bwd2": (I, [P, I, I, I, I, I, P, P, P, P, P, P, P, P, I, P, I, P, P, I, P, SZ]),
    "rcgan_bn_infer": (I, [P, I, I, I, P, P, P, P, P, F, I, P]),
    "rcgan_bn_infer_bwd": (I, [P, I, I, I, P, P, P, P, F, I, P, I]),
    "rcgan_sn_save_floats": (SZ, [I, I]),
    "rcgan_sn_power_iter": (I, [P, C.POINTER(SnItem), I]),
    "rcgan_sn_bwd": (I, [P, C.POINTER(SnBwdItem), I]),
    "rcgan_sn_bwd_adam": (I, [P, C.POINTER(SnBwdItem), I, C.POINTER(SnAdam)]),
    "rcgan_act_fwd": (I, [P, SZ, I, I, P, P]),
    "rcgan_act_bwd": (I, [P, SZ, I, I, P, P, P, I]),
    "rcgan_add": (I, [P, SZ, I, P, P, P]),
    "rcgan_axpby": (I, [P, SZ, I, F, P, F, P]),
    "rcgan_cast": (I, [P, SZ, I, P, I, P]),
    "rcgan_meanpool2_fwd": (I, [P, I, I, I, I, I, P, P]),
    "rcgan_meanpool2_bwd": (I, [P, I, I, I, I, I, P, P, I]),
    "rcgan_upsample2_fwd": (I, [P, I, I, I, I, I, P, P]),
    "rcgan_upsample2_bwd": (I, [P, I, I, I, I, I, P, P, I]),
    "rcgan_pad_channels": (I, [P, SZ, I, I, I, I, P, P]),
    "rcgan_concat_channels_fwd": (I, [P, I, I, I, I, I, P, P, P]),
    "rcgan_concat_channels_bwd": (I, [P, I, I, I, I, I, P, P]),
    "rcgan_tile_rows_fwd": (I, [P, SZ, I, I, P, P]),
    "rcgan_tile_rows_bwd": (I, [P, SZ, I, I, P, P, I]),
    "rcgan_transpose_f32": (I, [P, I, I, P, P, I]),
    "rcgan_preprocess_cifar": (I, [P, I, P, P, I, P]),
    "rcgan_rng_fill": (I, [P, SZ, I, I, F, F, C.c_uint64, P, P]),
    "rcgan_act_meanhw_fwd": (I, [P, I, I, I, I, I, P, P]),
    "rcgan_act_meanhw_bwd": (I, [P, I, I, I, I, I, P, P, P]),
    "rcgan_gather_rows": (I, [P, I, I, P, P, P]),
    "rcgan_scatter_add_rows": (I, [P, I, I, I, P, P, P, I]),
    "rcgan_proj_logit_fwd": (I, [P, I, I, P, P, P, P]),
    "rcgan_proj_logit_bwd": (I, [P, I, I, P, P, P, P, P, P, I]),
    "rcgan_proj_logit_all_fwd": (I, [P, I, I, I, P, P, P, P]),
    "rcgan_proj_logit_all_bwd": (I, [P, I, I, I, P, P, P, P, P, P, I]),
    "rcgan_loss_fwd_bwd": (I, [P, I, I, I, P, P, F, P, P, P]),
    "rcgan_bce_onehot_fwd_bwd": (I, [P, I, I, P, P, F, P, P]),
    "rcgan_dtrunk": (I, [P, I, I, P, P, P, P, P]),
    "rcgan_dtrunk_pooled": (I, [P, I, I, P, P, P, P, P, P, P, P]),
    "rcgan_dtrunk_prepare": (I, [P, P, P]),
    "rcgan_conv_rf_ok": (I, [C.POINTER(ConvDesc)]),
    "rcgan_conv_rf_fragment_bytes": (SZ, [C.POINTER(ConvDesc)]),
    "rcgan_conv_rf_prepare": (I, [P, I, C.POINTER(ConvDesc), P, P]),
    "rcgan_fragments_prepare": (I, [P, P, P, I, C.POINTER(ConvDesc), P, P]),
    "rcgan_conv2d_rf": (I, [P, C.POINTER(ConvDesc), I, P, P, P, P, P, P]),
    "rcgan_dtrunk_fragment_bytes": (SZ, []),
    "rcgan_head_flush": (I, [P]),
    "rcgan_proj_head_fwd_bwd": (I, [P, C.POINTER(HeadDesc)] + [P] * 16 + [P, SZ]),
    "rcgan_recover_mse_fwd_bwd": (I, [P, I, I, I, I, P, P, P, P, P, P, P, SZ]),
    "rcgan_softmax_rows_fwd": (I, [P, I, I, P, P]),
    "rcgan_softmax_rows_bwd": (I, [P, I, I, P, P, P, I]),
    "rcgan_adam_tf": (I, [P, SZ, P, P, P, P, P, F, F, F, F, F]),
    "rcgan_adam_tf_host": (I, [P, SZ, P, P, P, P, F, F, F, F, F, F, F]),
    "rcgan_fill_f32": (I, [P, SZ, P, F]),
    "rcgan_copy_words": (I, [P, SZ, P, P]),
    "rcgan_conv_stats_ok": (I, [DP]),
    "rcgan_conv_stats_bytes": (SZ, [DP]),
    "rcgan_conv2d_fwd_stats": (I, [P, DP, P, P, P, P, P, P]),
    "rcgan_bn_stats_from_tiles": (I, [P, DP, I, F, P, P, P]),
    "rcgan_bn_apply_segments": (I, [P, I, I, I, I, I, I, P, P, P, P, P, P, I, P, P, SZ]),
    "rcgan_comm_unique_id": (I, [P]),
    "rcgan_comm_init": (I, [P, P, I, I]),
    "rcgan_comm_init_stub": (I, [P, I]),
    "rcgan_comm_destroy": (I, [P]),
    "rcgan_comm_world": (I, [P]),
    "rcgan_allreduce_sum": (I, [P, P, SZ]),
    "rcgan_allreduce_sum_buckets": (I, [P, I, P, P]),
    "rcgan_allreduce_sum_async": (I, [P, P, SZ]),
    "rcgan_allreduce_join": (I, [P]),
    "rcgan_comm_count": (I, [P, P]),
    "rcgan_allreduce_bf16_scratch_bytes": (SZ, [I, P]),
    "rcgan_allreduce_sum_bf16_buckets": (I, [P, I, P, P, P, SZ]),
    "rcgan_comm_stub_model": (I, [P, C.c_double, C.c_double]),
    "rcgan_comm_load_error": (C.c_char_p, []),
    "rcgan_set2_f32": (I, [P, P, F, F]),
    "rcgan_set_grad_scale": (I, [P, F, P]),
    "rcgan_grad_finite_check": (I, [P, SZ, P, P]),
    "rcgan_adam_tf_dyn": (I, [P, SZ, P, P, P, P, F, P, F, F, F, F, F, P]),
    "rcgan_loss_scale_update": (I, [P, P, P, P, F, F, F]),
    "rcgan_selftest": (I, [P]),
    "rcgan_query": (I, [P, I]),
}

_libs = {}


def load(half="bf16"):
    """Load the shared library (once per 16-bit format).  Loading needs no GPU; compute calls do.
    ``half``: "bf16" (librcgan_hip.so, also serves fp32) or "f16" (librcgan_hip_f16.so)."""
    if half in _libs:
        return _libs[half]
    path = LIB_PATH_F16 if half == "f16" else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing (%s): build it with robust-conditional-gan_amd/csrc/build.sh "
            "or __graft_entry__.build(); there is no CPU fallback" % (os.path.basename(path), path))
    # torch first: it bundles its own libamdhip64.so.7 and librcgan_hip.so must bind to the SAME HIP
    # runtime (one SONAME, first loader wins); two runtimes in a process cannot share streams or pointers
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    want = F16 if half == "f16" else BF16
    if lib.rcgan_half_dtype() != want:
        raise RuntimeError("%s was built for 16-bit dtype %d, expected %d" % (path, lib.rcgan_half_dtype(), want))
    _libs[half] = lib
    return lib
