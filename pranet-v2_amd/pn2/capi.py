"""ctypes binding of libpn2_hip.so (C ABI declared in include/pn2.h).

This is the reference-side binding a maintainer would add: plain pointers and sizes, no torch
types cross the boundary.  There is NO CPU fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN2_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libpn2_hip.so")      # PN2_LIB: another build of the same library (A/B of build flags)

F32, BF16 = 0, 1
F32F = 2          # PN2_F32F: fp32 storage like F32, conv contractions on the f32 matrix pipe (v_mfma_f32_16x16x4_f32) instead of the f64 one
# "fp32fast" (pn2.set_compute_dtype): the engine keeps working in F32 - same buffers, same element-wise kernels - and the conv GEMM / wgrad entry points below
# receive F32F instead of F32 at this boundary (the only place where the two differ)
F32_MMA = F32
_MMA_FUNCS = {"pn2_conv_gemm", "pn2_conv_gemm_ep", "pn2_conv_gemm_gated", "pn2_conv_gemm_affine", "pn2_conv_gemm_multi", "pn2_conv_gemm_tile", "pn2_conv_gemm_job_blocks",
              "pn2_conv_wgrad", "pn2_conv_wgrad_multi", "pn2_conv_wgrad_variant"}


def set_f32_mma(fast):
    global F32_MMA
    F32_MMA = F32F if fast else F32
CONV_STATS, CONV_ACCUM, CONV_BIAS = 1, 2, 4
CONV_AFFINE, CONV_RELU, CONV_RELU6 = 16, 32, 64


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "OH", "OW", "Cin_p", "ld_in", "Cout", "ld_out", "KH", "KW", "stride",
                                       "pad_h", "pad_w", "dil_h", "dil_w", "transposed", "Kp", "flags")]


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "OH", "OW", "Cin_p", "ld_x", "Cout_p", "ld_dy", "KH", "KW", "stride",
                                       "pad_h", "pad_w", "dil_h", "dil_w", "Rp", "Kp", "tune")]


class PackDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("Cout", "Cin", "KH", "KW", "Cout_p", "gw_out", "gwp_out", "Cin_p", "gw_in", "gwp_in",
                                       "Rp", "Kp", "transposed", "ld", "koff")]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("d", PackDesc)]


class WgradJob(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("slab", C.c_void_p), ("d", WgradDesc), ("nsplit", C.c_int), ("rot", C.c_int)]


class ReduceJob(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("gw", C.c_void_p), ("d", PackDesc), ("nsplit", C.c_int), ("accumulate", C.c_int)]


class ColsumInJob(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("partial", C.c_void_p), ("ld", C.c_int), ("M", C.c_int), ("C", C.c_int), ("rows", C.c_int), ("cvp", C.c_int), ("pad_", C.c_int)]


class ColsumJob(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("nblk", C.c_int), ("C", C.c_int), ("ld", C.c_int), ("accumulate", C.c_int)]


class TailMap(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dsrc", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("rh", C.c_float), ("rw", C.c_float),
                ("accumulate", C.c_int), ("pad_", C.c_int)]


class TailDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("P", C.c_int), ("align_corners", C.c_int), ("pad_", C.c_int),
                ("maps", TailMap * 16)]


class BnDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("Cp", C.c_int), ("C", C.c_int), ("gw", C.c_int), ("gwp", C.c_int),
                ("eps", C.c_float), ("momentum", C.c_float), ("ldp", C.c_int), ("tile_rows", C.c_int)]


BNB_STATS, BNB_MASK_RAW, BNB_MASK_Y, BNB_STORE_MASKED = 1, 2, 4, 8


class BnbTarget(C.Structure):
    _fields_ = [("out", C.c_void_p), ("ld_out", C.c_int), ("mode", C.c_int), ("raw", C.c_void_p), ("ld_raw", C.c_int),
                ("y", C.c_void_p), ("ld_y", C.c_int), ("par", C.c_void_p), ("ps", C.c_int), ("split", C.c_int),
                ("raw2", C.c_void_p), ("par2", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("ldp", C.c_int)]


class ConvEp(C.Structure):
    _fields_ = [("a", BnbTarget), ("b", BnbTarget), ("pool", C.c_void_p), ("ld_pool", C.c_int), ("pad_", C.c_int), ("c", BnbTarget)]


class BnSegs(C.Structure):
    _fields_ = [("nseg", C.c_int), ("c0", C.c_int * 4), ("nblk", C.c_int * 4), ("ldp", C.c_int * 4), ("p1", C.c_void_p * 4), ("p2", C.c_void_p * 4)]


class CopyJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ld_s", C.c_int), ("ld_d", C.c_int), ("M", C.c_int), ("C", C.c_int), ("accumulate", C.c_int), ("pad_", C.c_int)]


class ConvJob(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("wp", C.c_void_p), ("out", C.c_void_p), ("psum", C.c_void_p), ("psq", C.c_void_p), ("d", ConvDesc), ("pad_", C.c_int), ("ep", ConvEp)]


class BnFinJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("psum", "psq", "gamma", "beta", "running_mean", "running_var", "scale", "shift", "mean", "invstd")] + \
               [("d", BnDesc), ("nblk", C.c_int), ("cpb", C.c_int), ("pad_", C.c_int)]


class BnPrepJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gamma", "beta", "running_mean", "running_var", "scale", "shift")] + [("d", BnDesc), ("pad_", C.c_int)]


class AffineJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "y", "scale", "shift", "res", "add", "y2")] + \
               [(n, C.c_int) for n in ("ld_x", "ld_y", "ld_res", "ld_add", "ld_y2", "M", "C", "relu", "rows_per_blk", "cvp")]


class BnBFinJob(C.Structure):
    _fields_ = [("sg", BnSegs)] + [(n, C.c_void_p) for n in ("gamma", "invstd", "dgamma", "dbeta", "coef")] + \
               [("d", BnDesc), ("accumulate", C.c_int), ("cpb", C.c_int), ("pad_", C.c_int)]


class BnApplyJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dy", "y", "x", "mean", "invstd", "coef", "dx", "dres", "msc", "msh")] + \
               [(n, C.c_int) for n in ("ld_dy", "ld_y", "ld_x", "ld_dx", "ld_dres", "M", "Cp", "dres_accum", "r6", "rows_per_blk", "cvp", "pad_")]


class BnReduceJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dy", "y", "x", "mean", "invstd", "p1", "p2", "msc", "msh")] + \
               [(n, C.c_int) for n in ("ld_dy", "ld_y", "ld_x", "M", "Cp", "nblk", "rows_per_blk", "cvp", "r6", "pad_")]


P, I, LL, FL = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> argtypes ; every function returns int (0 = ok)
SIGNATURES = {
    "pn2_conv_tile_n": [I],
    "pn2_wgrad_tile_co": [I],
    "pn2_conv_tile_m": [I, I, I],
    "pn2_conv_stat_blocks": [I, I, I],
    "pn2_conv_gemm": [I, P, P, P, P, P, C.POINTER(ConvDesc), P],
    "pn2_conv_gemm_ep": [I, P, P, P, C.POINTER(ConvDesc), C.POINTER(ConvEp), P],
    "pn2_conv_gemm_tile": [I, C.POINTER(ConvDesc)],
    "pn2_conv_gemm_job_blocks": [I, C.POINTER(ConvJob), I, I],
    "pn2_conv_gemm_multi": [I, I, I, I, P, P, I, I, P],
    "pn2_bn_finalize_job_blocks": [C.POINTER(BnFinJob)],
    "pn2_bn_finalize_multi": [P, P, I, I, P],
    "pn2_affine_job_blocks": [I, C.POINTER(AffineJob)],
    "pn2_affine_multi": [I, P, P, I, I, P],
    "pn2_bn_bwd_finalize_job_blocks": [C.POINTER(BnBFinJob)],
    "pn2_bn_bwd_finalize_multi": [P, P, I, I, P],
    "pn2_bn_bwd_apply_job_blocks": [I, C.POINTER(BnApplyJob)],
    "pn2_bn_bwd_apply_multi": [I, P, P, I, I, P],
    "pn2_bn_bwd_reduce_job_blocks": [I, C.POINTER(BnReduceJob)],
    "pn2_bn_bwd_reduce_multi": [I, P, P, I, I, P],
    "pn2_conv_wgrad": [I, P, P, P, C.POINTER(WgradDesc), I, P],
    "pn2_conv_wgrad_variant": [I, C.POINTER(WgradDesc)],
    "pn2_conv_wgrad_blocks": [C.POINTER(WgradDesc), I],
    "pn2_conv_wgrad_multi": [I, I, P, P, I, I, P],
    "pn2_pack_weight": [I, P, P, C.POINTER(PackDesc), P],
    "pn2_conv_dgrad_small_cin": [I, P, I, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "pn2_conv_splitk_reduce": [I, P, I, I, I, P, I, P, P, P, I, P],
    "pn2_pack_patch_weight": [I, P, P, I, I, I, I, I, I, I, P],
    "pn2_depth_to_space": [I, P, I, P, I, I, I, I, I, I, I, I, I, P],
    "pn2_wgrad_reduce": [P, P, C.POINTER(PackDesc), I, I, P],
    "pn2_pack_blocks": [C.POINTER(PackDesc)],
    "pn2_pack_weights_multi": [I, P, P, I, I, P],
    "pn2_wgrad_reduce_blocks": [C.POINTER(PackDesc)],
    "pn2_wgrad_reduce_multi": [P, P, I, I, P],
    "pn2_bn_finalize": [P, P, I, C.POINTER(BnDesc), P, P, P, P, P, P, P, P, P],
    "pn2_bn_eval_prepare": [C.POINTER(BnDesc), P, P, P, P, P, P, P],
    "pn2_bn_eval_prepare_multi": [P, P, I, I, P],
    "pn2_affine_act_sum": [I, P, I, P, I, I, I, P, P, I, P, I, P, I, P],
    "pn2_affine_act_tee": [I, P, I, P, I, I, I, P, P, I, P, I, I, P],
    "pn2_bn_relu_maxpool_fwd": [I, P, I, P, P, P, I, P, I, I, I, I, I, I, P],
    "pn2_pool_bn_bwd_reduce": [I, P, I, P, P, I, I, I, I, I, I, I, P, P, P, P, P, P, I, P],
    "pn2_pool_bn_bwd_apply": [I, P, I, P, P, I, I, I, I, I, I, I, P, P, P, P, P, P, I, P],
    "pn2_affine_act": [I, P, I, I, P, I, I, I, P, P, P, I, I, P],
    "pn2_bn_bwd_reduce": [I, I, P, I, I, P, I, I, P, I, I, I, P, P, P, P, I, P, P, I, P],
    "pn2_bn_bwd_blocks": [I, I, I],
    "pn2_bn_bwd_finalize": [P, P, I, C.POINTER(BnDesc), P, P, P, P, I, P, P],
    "pn2_bn_bwd_finalize_seg": [C.POINTER(BnSegs), C.POINTER(BnDesc), P, P, P, P, I, P, P],
    "pn2_bn_bwd_apply": [I, I, P, I, I, P, I, I, P, I, I, I, P, P, P, P, I, P, I, I, P, P, I, P],
    "pn2_maxpool3x3s2_fwd": [I, P, I, P, I, P, I, I, I, I, I, I, P],
    "pn2_maxpool3x3s2_bwd": [I, P, I, P, P, I, I, I, I, I, I, I, P],
    "pn2_avgpool_fwd": [I, P, I, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "pn2_avgpool_bwd": [I, P, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "pn2_bilinear_fwd": [I, P, I, P, I, I, I, I, I, I, I, I, FL, FL, P],
    "pn2_bilinear_bwd": [I, P, I, P, I, I, I, I, I, I, I, I, FL, FL, I, P],
    "pn2_dsra_fuse_fwd": [P, P, P, P, I, I, I, P],
    "pn2_dsra_fuse_bwd": [P, P, P, P, P, P, P, I, I, I, P],
    "pn2_ra_gate_fwd": [I, P, I, P, P, I, I, I, P],
    "pn2_ra_gate_bwd": [I, P, I, P, P, I, P, I, I, P, I, I, P],
    "pn2_ra_gate_post_bwd": [I, P, I, P, P, I, P, I, P, I, I, P],
    "pn2_conv_gemm_gated": [I, P, P, P, P, P, P, P, P],
    "pn2_conv_gemm_affine": [I, P, P, P, P, P, P, I, P, P],
    "pn2_loss_weights": [P, P, I, I, I, I, P],
    "pn2_loss_weights_clear": [P, P, I, I, I, I, P, I, P],
    "pn2_loss_blocks": [I],
    "pn2_structure_loss_fwd": [P, LL, I, P, P, P, P, P, P, I, I, P],
    "pn2_structure_loss_bwd": [P, P, LL, I, P, P, P, P, FL, I, I, P],
    "pn2_structure_loss_bwd_dev": [P, P, LL, LL, I, P, P, P, P, P, FL, I, I, P],
    "pn2_dsra_tail_blocks": [I],
    "pn2_dsra_tail_fwd": [C.POINTER(TailDesc), P, P, P, P, P, P, P, P],
    "pn2_dsra_tail_scratch": [C.POINTER(TailDesc)],
    "pn2_dsra_tail_bwd": [C.POINTER(TailDesc), P, P, P, P, FL, P, LL, P],
    "pn2_dsra_tail_fused_ok": [C.POINTER(TailDesc)],
    "pn2_dsra_tail_fused_scratch": [C.POINTER(TailDesc)],
    "pn2_dsra_tail_fwd_bwd": [C.POINTER(TailDesc), P, P, P, P, P, P, P, P, FL, P, LL, P, P],
    "pn2_layernorm_fwd": [I, P, I, P, I, I, I, P, P, FL, P, P, P],
    "pn2_ln_slots": [I, I],
    "pn2_rows_blocks": [I, I],
    "pn2_layernorm_bwd": [I, P, I, P, I, I, I, P, P, P, P, I, I, P, P, I, P],
    "pn2_colsum_unit": [I, I],
    "pn2_colsum": [I, P, I, I, I, P, I, P],
    "pn2_colsum_finalize": [P, I, I, I, P, I, P],
    "pn2_colsum_job_blocks": [I, C.POINTER(ColsumInJob)],
    "pn2_colsum_multi": [I, P, P, I, I, P],
    "pn2_colsum_finalize_blocks": [I],
    "pn2_colsum_finalize_multi": [P, P, I, I, P],
    "pn2_dwconv3x3": [I, P, P, P, P, P, I, I, I, I, I, I, P],
    "pn2_dwconv3x3_colsum_blocks": [I, I, I, I, I],
    "pn2_dwconv3x3_colsum": [I, P, P, P, P, I, I, I, I, I, P, I, P],
    "pn2_gelu_bwd": [I, P, P, P, LL, P],
    "pn2_dwconv3x3_wgrad_blocks": [I, I, I, I, I],
    "pn2_dwconv3x3_wgrad": [I, P, P, P, I, I, I, I, I, P, P, P],
    "pn2_scale_samples": [I, P, P, P, P, I, LL, P],
    "pn2_attn_fwd": [I, P, I, P, I, P, I, P, I, I, I, I, I, FL, P],
    "pn2_attn_bwd_blocks": [I, I, I, I],
    "pn2_attn_bwd": [I, P, I, P, I, P, I, P, I, P, P, I, P, I, P, P, I, I, I, I, I, FL, P],
    "pn2_dwconv_blocks": [I, I, I, I, I, I, I],
    "pn2_dwconv": [I, P, P, P, I, I, I, I, I, I, I, P, P, P],
    "pn2_dwconv_wgrad": [I, P, P, P, I, I, I, I, I, P],
    "pn2_pairconv_blocks": [I, I, I, I, I],
    "pn2_pairconv3x3_fwd": [I, P, P, P, I, I, I, I, P, P, P],
    "pn2_pairconv3x3_dgrad": [I, P, P, P, I, I, I, I, I, P],
    "pn2_pairconv3x3_wgrad": [I, P, P, P, I, I, I, I, P],
    "pn2_gate_mul": [I, P, P, P, I, I, I, I, I, P],
    "pn2_gate_blocks": [I, I, I],
    "pn2_gate_bwd": [I, P, P, P, I, I, I, I, P],
    "pn2_global_pool": [I, P, P, P, P, I, I, I, P],
    "pn2_global_pool_bwd": [I, P, P, P, P, I, I, I, I, P],
    "pn2_chan_stats": [I, P, P, P, LL, I, P],
    "pn2_chan_stats_bwd": [I, P, P, P, LL, I, I, P],
    "pn2_upsample_nearest2x": [I, P, P, I, I, I, I, P],
    "pn2_upsample_nearest2x_bwd": [I, P, P, I, I, I, I, I, P],
    "pn2_gather_sum": [I, P, P, P, P, P, LL, I, P],
    "pn2_sigmoid": [I, P, I, I, P, LL, P],
    "pn2_sigmoid_bwd": [I, P, P, P, I, I, LL, I, P],
    "pn2_mutation_loss_blocks": [LL],
    "pn2_mutation_loss_width": [I],
    "pn2_mutation_loss_fwd": [P, P, P, P, I, LL, I, FL, FL, FL, P, P, P, P],
    "pn2_mutation_loss_bwd": [P, P, P, P, P, P, I, LL, I, FL, FL, FL, P, FL, P],
    "pn2_binary": [I, I, P, I, P, I, P, I, I, I, I, P],
    "pn2_mul_bwd": [I, P, I, P, I, P, I, P, I, I, P, I, I, I, I, P],
    "pn2_copy": [I, P, I, I, P, I, I, I, I, P],
    "pn2_copy_job_blocks": [I, C.POINTER(CopyJob)],
    "pn2_copy_multi": [I, P, P, I, I, P],
    "pn2_nchw_to_nhwc": [I, P, P, I, I, I, I, I, P],
    "pn2_bias_grad": [P, I, I, P, I, P],
    "pn2_resize_ksize": [I, I],
    "pn2_resize_coeffs": [I, I, P, P, P],
    "pn2_resize_u8_pass": [P, P, I, I, I, I, I, P, P, P, I, P],
    "pn2_u8_to_tensor": [P, P, I, I, I, P, P, P],
    "pn2_clamp_adam": [P, P, P, P, LL, FL, FL, FL, FL, FL, FL, P, FL, P],
    "pn2_adam_tick": [P, FL, FL, P],
    "pn2_eval_tail": [P, P, P, LL, P],
    "pn2_eval_hist": [P, P, LL, P, P],
    "pn2_eval_region_sums": [P, P, I, I, P, P],
    "pn2_eval_wfm_blocks": [I, I],
    "pn2_eval_wfm": [P, P, I, I, P, C.c_double, P, P, P, P],
}
# entry points that return a value rather than a status
_VALUE_FUNCS = {"pn2_copy_job_blocks", "pn2_conv_gemm_tile", "pn2_conv_gemm_job_blocks", "pn2_bn_finalize_job_blocks", "pn2_affine_job_blocks", "pn2_bn_bwd_finalize_job_blocks",
                "pn2_bn_bwd_apply_job_blocks", "pn2_bn_bwd_reduce_job_blocks", "pn2_dwconv3x3_colsum_blocks", "pn2_resize_ksize", "pn2_colsum_job_blocks", "pn2_colsum_finalize_blocks", "pn2_dwconv3x3_wgrad_blocks", "pn2_conv_tile_n", "pn2_wgrad_tile_co", "pn2_conv_stat_blocks", "pn2_conv_tile_m", "pn2_bn_bwd_blocks", "pn2_loss_blocks",
                "pn2_pack_blocks", "pn2_wgrad_reduce_blocks", "pn2_conv_wgrad_variant", "pn2_conv_wgrad_blocks",
                "pn2_dsra_tail_blocks", "pn2_dsra_tail_scratch", "pn2_dsra_tail_fused_ok", "pn2_dsra_tail_fused_scratch", "pn2_ln_slots", "pn2_rows_blocks", "pn2_colsum_unit", "pn2_attn_bwd_blocks", "pn2_mutation_loss_blocks", "pn2_mutation_loss_width",
                "pn2_dwconv_blocks", "pn2_pairconv_blocks", "pn2_gate_blocks", "pn2_eval_wfm_blocks"}

_lib = None
WORK = {}     # profiling annotation for the next launch (algorithmic flops / tag), consumed by pn2.profile.Recorder


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C pranet-v2_amd/csrc` (or __graft_entry__.build()); "
                           "pranet-v2_amd has no CPU / PyTorch fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export what pn2.h declares
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


class _Caller:
    """`call.pn2_xxx(...)` -> invokes the C entry point and raises RuntimeError on a non-zero status."""

    def __getattr__(self, name):
        fn = getattr(load(), name)
        if name in _MMA_FUNCS:
            raw = fn

            def fn(dt, *a):          # noqa: F811   (fp32fast: F32 -> F32F for the entry points that run MFMA contractions)
                return raw(F32_MMA if dt == F32 else dt, *a)
        if name in _VALUE_FUNCS:
            return fn

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                raise RuntimeError(f"{name} failed with status {rc}")
        setattr(self, name, checked)
        return checked


call = _Caller()
