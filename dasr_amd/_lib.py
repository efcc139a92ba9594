"""ctypes binding of libdasr_hip.so (include/dasr_hip.h).  Fails loudly when the library is missing:
there is no CPU / PyTorch fallback on the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libdasr_hip.so')

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class Tensor(C.Structure):
    _fields_ = [('p', c_vp), ('n_stride', c_i64), ('cb_stride', c_i64)]


class ConvParams(C.Structure):
    _fields_ = [('inp', Tensor), ('in_f32', c_i32), ('Hin', c_i32), ('Win', c_i32), ('ups', c_i32), ('cin', c_i32),
                ('w', c_vp), ('w_lo_off', c_i64), ('bias', c_vp),
                ('cout', c_i32), ('Hout', c_i32), ('Wout', c_i32), ('N', c_i32),
                ('kh', c_i32), ('stride', c_i32), ('pad', c_i32), ('prec', c_i32), ('mt', c_i32),
                ('act', c_i32), ('slope', c_f32),
                ('mask', Tensor), ('mask_f32', c_i32),
                ('alpha', c_f32), ('res1', Tensor), ('beta1', c_f32), ('res2', Tensor), ('beta2', c_f32),
                ('out_f32', Tensor), ('out_bf16', Tensor), ('gamma', c_f32), ('xcd_remap', c_i32),
                ('pad_x', c_i32), ('out_stride', c_i32), ('out_oy', c_i32), ('out_ox', c_i32), ('out_W', c_i32), ('slope_ptr', c_vp),
                ('in_stride', c_i32), ('in_oy', c_i32), ('in_ox', c_i32), ('in_W', c_i32), ('in_scale', c_f32), ('out16_f16', c_i32), ('in_wrap', c_i32), ('out16_lo', c_i32), ('res1_lo', c_i32), ('prelu_part', c_vp)]


class WgradPart(C.Structure):
    _fields_ = [('g', Tensor), ('g_f32', c_i32), ('inp', Tensor), ('in_f32', c_i32), ('ups', c_i32), ('n_ctiles', c_i32),
                ('g_planes', c_i32), ('in_planes', c_i32),
                ('Hin', c_i32), ('Win', c_i32), ('Hout', c_i32), ('Wout', c_i32), ('N', c_i32),
                ('kh', c_i32), ('stride', c_i32), ('pad', c_i32), ('want_bias', c_i32),
                ('ws_off', c_i64), ('ws_bias_off', c_i64), ('tap0', c_i32), ('g_scale', c_f32)]


class WgradReducePart(C.Structure):
    _fields_ = [('ws_off', c_i64), ('ws_bias_off', c_i64), ('nsplit', c_i32), ('ntaps', c_i32), ('oc0', c_i32), ('c0', c_i32),
                ('cout', c_i32), ('cin', c_i32), ('n_ctiles', c_i32), ('dst_w_off', c_i64), ('dst_b_off', c_i64),
                ('flip_io', c_i32), ('split_stride', c_i64), ('tap_stride', c_i64), ('bias_stride', c_i64),
                ('tap0', c_i32), ('ntaps_total', c_i32), ('bias_nsplit', c_i32), ('reserved_', c_i32)]


class PackSeg(C.Structure):
    _fields_ = [('src_off', c_i64), ('src_cout', c_i32), ('src_cin', c_i32), ('cin_start', c_i32), ('cin_len', c_i32),
                ('src_c0', c_i32), ('transpose', c_i32)]


class PackDesc(C.Structure):
    _fields_ = [('dst_off', c_i64), ('lo_off', c_i64), ('cout', c_i32), ('cin_pad', c_i32), ('ntaps', c_i32), ('mt', c_i32),
                ('nseg', c_i32), ('src_ntaps', c_i32), ('fmt', c_i32), ('tapmap', C.c_int8 * 32), ('tapmask', C.c_uint16 * 16), ('seg', PackSeg * 5)]


class CropDesc(C.Structure):
    _fields_ = [('src', c_vp), ('C', c_i32), ('H', c_i32), ('W', c_i32), ('vH', c_i32), ('vW', c_i32), ('y0', c_i32), ('x0', c_i32), ('flags', c_i32)]


class Op(C.Structure):
    _fields_ = [('op', c_i32), ('i', c_i32 * 8), ('f', c_f32 * 4), ('l', c_i64 * 4), ('p', c_vp * 4), ('t', Tensor * 5),
                ('conv', ConvParams), ('flops', C.c_double), ('bytes', C.c_double)]


OP_CONV, OP_WGRAD, OP_WGRAD_REDUCE, OP_PACK, OP_DOWNSUM, OP_AXPBY, OP_FILL, OP_L1LOSS, OP_NCHW2B, OP_B2NCHW = range(1, 11)
(OP_INORM_FWD, OP_INORM_BWD, OP_BCE, OP_DWT_FWD, OP_DWT_BWD, OP_LOWPASS, OP_MAXPOOL, OP_MAXPOOL_BWD, OP_L1DIFF, OP_AFFINE4,
 OP_BILINEAR, OP_LOGLOSS, OP_SIGMOID_BWD, OP_PRELU_GRAD, OP_LOWPASS_VALID, OP_ADD_FLAT, OP_SIGMOID_FWD, OP_EVENT_RECORD, OP_STREAM_WAIT,
 OP_SET_STREAM) = range(11, 31)
OP_CVT_F16, OP_DOWNSUM_F16, OP_PIXSHUF, OP_PIXUNSHUF = 31, 32, 33, 34
OP_LPIPS_S2D, OP_MAXPOOL3, OP_MAXPOOL3_BWD, OP_LPIPS_HEAD, OP_RAGAN = 35, 36, 37, 38, 39
OP_BNORM_FWD, OP_BNORM_BWD, OP_BNORM_RUNNING = 40, 41, 42
OP_DDM_SPREAD = 43
OP_INORM_JVP, OP_INORM_SECOND, OP_GRAD_PENALTY, OP_FILL_SCALED = 44, 45, 46, 47
OP_CONV_CHAIN = 48
OP_RDB_CHAIN = 49
OP_BNORM_JVP, OP_BNORM_SECOND = 50, 51   # --wgan with BatchNorm discriminators (round 6)
OP_PRELU_FINAL = 52

_SIGS = {
    'dasr_conv': [C.POINTER(ConvParams), c_vp],
    'dasr_conv_chain': [c_vp, C.POINTER(ConvParams), c_vp, c_i32, c_vp, c_vp, c_vp],
    'dasr_rdb_chain': [c_vp, C.POINTER(ConvParams), c_i32, c_vp, c_vp, c_vp],
    'dasr_conv_naive': [C.POINTER(ConvParams), c_vp, c_vp],
    'dasr_set_tuning': [c_i32, c_i32],
    'dasr_wgrad': [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
    'dasr_wgrad_set_mode': [c_i32],
    'dasr_wgrad_reduce': [c_vp, c_i32, c_vp, c_vp, c_f32, c_i32, c_vp],
    'dasr_pack_weights': [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp],
    'dasr_nchw_to_blocked': [c_vp, c_i32, c_i32, c_i32, c_i32, Tensor, Tensor, c_vp],
    'dasr_blocked_to_nchw': [Tensor, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
    'dasr_l1_loss': [Tensor, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, Tensor, c_i32, c_f32, c_vp],
    'dasr_pixel_shuffle_f16': [Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_vp],
    'dasr_pixel_unshuffle_f16': [Tensor, Tensor, c_f32, c_i32, c_i32, c_i32, c_i32, Tensor, c_vp],
    'dasr_cvt_f16': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, Tensor, c_vp],
    'dasr_cvt_split16': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, Tensor, c_i32, c_vp],
    'dasr_f16_residual': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, Tensor, c_vp],
    'dasr_downsum2x_f16': [Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_f32, c_f32, Tensor, Tensor, c_vp],
    'dasr_downsum2x': [Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_f32, Tensor, Tensor, c_vp],
    'dasr_axpby': [Tensor, c_f32, Tensor, c_f32, c_i32, c_i32, c_i32, c_i32, Tensor, Tensor, c_f32, Tensor, c_f32, c_vp, c_vp],
    'dasr_adam': [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp, c_vp, c_vp],
    'dasr_fill_f32': [c_vp, c_i64, c_f32, c_vp],
    'dasr_add_flat': [c_vp, c_vp, c_i64, c_vp],
    'dasr_inorm_lrelu_fwd': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, Tensor, c_vp, c_vp],
    'dasr_inorm_lrelu_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, Tensor, c_vp],
    'dasr_inorm_lrelu_jvp': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, Tensor, c_vp],
    'dasr_inorm_second': [Tensor, Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, Tensor, c_i32, c_vp],
    'dasr_grad_penalty': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp],
    'dasr_fill_scaled': [Tensor, c_i32, c_i32, c_i32, c_i32, c_vp, c_f32, c_vp],
    'dasr_bce_logits': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_f32, Tensor, c_vp],
    'dasr_gan_loss': [Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_f32, Tensor, c_vp],
    'dasr_dwt_fwd': [Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, Tensor, Tensor, c_vp],
    'dasr_dwt_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_vp],
    'dasr_lowpass': [Tensor, Tensor, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, Tensor, Tensor, c_i32, c_vp],
    'dasr_maxpool2': [Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_vp],
    'dasr_maxpool2_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_i32, c_vp],
    'dasr_l1_diff': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, Tensor, c_vp],
    'dasr_affine4': [Tensor, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, Tensor, c_i32, c_i32, c_vp],
    'dasr_bilinear_up': [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
    'dasr_logloss': [Tensor, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_f32, Tensor, c_i32, c_vp],
    'dasr_sigmoid_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_vp],
    'dasr_sigmoid_fwd': [Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_vp],
    'dasr_gather_crops': [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp],
    'dasr_event_create': [],
    'dasr_event_destroy': [c_vp],
    'dasr_prelu_grad': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_vp],
    'dasr_prelu_grad_f16': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_vp],
    'dasr_lowpass_valid': [Tensor, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_vp],
    'dasr_run_ops': [c_vp, c_i32, c_vp],
    'dasr_run_ops_mt': [c_vp, c_vp, c_vp, c_i32],
    'dasr_last_failed_op': [],
    'dasr_ddm_spread': [Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, Tensor, c_vp],
    'dasr_abi_version': [],
    'dasr_prof_filter': [C.c_char_p],
    'dasr_red_release': [],
    'dasr_probe_tr16': [c_vp],
    'dasr_rccl_unique_id': [c_vp],
    'dasr_rccl_init': [c_vp, c_i32, c_i32, c_vp],
    'dasr_allreduce': [c_vp, c_vp, c_i64, c_vp],
    'dasr_broadcast': [c_vp, c_vp, c_i64, c_i32, c_vp],
    'dasr_rccl_destroy': [c_vp],
    'dasr_bnorm_lrelu_fwd': [Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, Tensor, c_vp, c_vp],
    'dasr_prelu_final': [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_f32, c_vp],
    'dasr_bnorm_lrelu_jvp': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, Tensor, c_vp],
    'dasr_bnorm_second': [Tensor, Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, Tensor, c_i32, c_vp, c_f32, c_vp],
    'dasr_bnorm_lrelu_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, Tensor, c_vp, c_vp, c_f32, c_vp],
    'dasr_bnorm_running': [c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp],
    'dasr_ragan': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, Tensor, Tensor, c_vp],
    'dasr_lpips_s2d': [Tensor, c_i32, c_i32, c_i32, c_vp, c_vp, Tensor, c_i32, c_vp],
    'dasr_maxpool3s2': [Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_vp],
    'dasr_maxpool3s2_bwd': [Tensor, Tensor, c_i32, c_i32, c_i32, c_i32, Tensor, c_i32, c_i32, c_vp],
    'dasr_lpips_head': [Tensor, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_f32, c_f32, c_f32, c_vp, Tensor, c_i32, c_vp],
    'dasr_prof_begin': [c_i32],
    'dasr_prof_end': [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
}

# micro-benchmark probes: libdasr_bench.so (include/dasr_hip_bench.h), never part of the product library
_BENCH_SIGS = {
    'dasr_probe_mfma_data': [c_i32, c_i32, c_vp, c_vp],
    'dasr_probe_tile_sync': [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    'dasr_probe_mfma_peak': [c_i32, c_vp, c_vp],
    'dasr_probe_spin': [c_i32, c_i32, c_vp],
    'dasr_probe_store': [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp],
}
BENCH_LIB_PATH = os.path.join(HERE, 'libdasr_bench.so')

ABI_VERSION = 20
_lib = None
_bench = None


def bench_lib():
    """the probe library (bench.py, scripts/micro_*.py); raises when it has not been built (`python -m dasr_amd.build --bench`)"""
    global _bench
    if _bench is None:
        if not os.path.exists(BENCH_LIB_PATH):
            raise DasrHipError('libdasr_bench.so is missing (%s): run `python -m dasr_amd.build --bench`' % BENCH_LIB_PATH)
        import torch  # noqa: F401  (same HIP runtime as torch, see lib())
        L = C.CDLL(BENCH_LIB_PATH)
        for name, args in _BENCH_SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_i32
        _bench = L
    return _bench


class DasrHipError(RuntimeError):
    pass


def exported_symbols():
    """Names every entry point include/dasr_hip.h declares (used by the CPU symbol test)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DasrHipError('libdasr_hip.so is missing (%s): run `python -m dasr_amd.build` or __graft_entry__.build(); '
                               'the DASR MI355X path has no fallback.' % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7, but NEEDED as "libamdhip64.so"); if this library is
        # loaded before torch, /opt/rocm's runtime comes in with it and torch then maps a SECOND HIP runtime -- torch's streams and
        # allocations are foreign to ours and the first launch fails (hipErrorNoDevice).  With torch loaded, our NEEDED
        # libamdhip64.so.7 resolves by SONAME to the runtime torch uses.  (A C consumer without torch gets /opt/rocm's, alone.)
        import torch  # noqa: F401
        L = C.CDLL(os.environ.get('DASR_HIP_LIB') or LIB_PATH)  # DASR_HIP_LIB: instrumented build for scripts/ (same ABI)
        for name, args in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.argtypes = args
            fn.restype = c_vp if name == 'dasr_event_create' else c_i32
        if L.dasr_abi_version() != ABI_VERSION:
            raise DasrHipError('libdasr_hip.so ABI %d != binding ABI %d; rebuild' % (L.dasr_abi_version(), ABI_VERSION))
        for kv in [x for x in os.environ.get('DASR_TUNE', '').split(',') if x]:   # A/B of kernel variants without code changes: DASR_TUNE="1=14,2=12"
            k, v = kv.split('=')
            if L.dasr_set_tuning(int(k), int(v)) != 0:
                raise DasrHipError('DASR_TUNE: bad tuning key/value %r' % kv)
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        raise DasrHipError('%s failed with code %d' % (what or 'dasr call', rc))
