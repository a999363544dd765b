"""ctypes binding of libsemseg_hip.so (C ABI declared in include/semseg_hip.h).

The product path has NO CPU or eager-torch fallback: if the shared library is missing, or a kernel
returns a non-zero code, this raises.  The library is loaded after `import torch` so that it binds
to the HIP runtime torch already loaded (same soname libamdhip64.so.7)."""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEMSEG_NATIVE_LIB: another build of the same library (in-box A/B of kernel changes, tools/gpu_ab_lib.sh)
LIB_PATH = os.environ.get('SEMSEG_NATIVE_LIB') or os.path.join(_HERE, 'libsemseg_hip.so')
_lib = None

c_int, c_f, c_sz, vp = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p


class SgdTensor(ctypes.Structure):
    _fields_ = [('param', vp), ('grad', vp), ('momentum_buf', vp), ('numel', ctypes.c_int64),
                ('weight_decay', c_f), ('first_step', c_int)]


class SgdTensor2(ctypes.Structure):
    _fields_ = [('param', vp), ('grad', vp), ('momentum_buf', vp), ('numel', ctypes.c_int64), ('weight_decay', c_f),
                ('first_step', c_int), ('slabs', vp), ('splits', c_int), ('absmax_slots', vp)]


class SlabTensor(ctypes.Structure):
    _fields_ = [('slabs', vp), ('out', vp), ('numel', ctypes.c_int64), ('splits', c_int)]


class WgradProblem(ctypes.Structure):
    _fields_ = [('xs', vp), ('dys', vp), ('slabs', vp), ('slabs_bytes', c_sz)] + \
               [(k, c_int) for k in ('N', 'H', 'W', 'C', 'K', 'R', 'S', 'stride', 'pad', 'dil', 'splits')]


class WPrepTensor(ctypes.Structure):
    _fields_ = [('w', vp), ('krsc', vp), ('crsk', vp), ('K', c_int), ('T', c_int), ('C', c_int), ('wino', vp), ('wino_t', vp)]


# name -> (restype, argtypes); mirrors include/semseg_hip.h one to one
SIGNATURES = {
    'semseg_abi_version': (c_int, []),
    'semseg_conv2d_workspace_bytes': (c_sz, [c_int] * 10),
    'semseg_conv2d_fwd': (c_int, [vp, c_int, vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_dgrad': (c_int, [vp, c_int, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_wgrad': (c_int, [vp, c_int, vp, c_int, vp, vp] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_weight_krsc_to_crsk': (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    'semseg_split3_bytes': (c_sz, [c_int, c_int]),
    'semseg_split3': (c_int, [vp, c_int, vp, c_int, c_int, vp]),
    'semseg_conv2d_s3_workspace_bytes': (c_sz, [c_int] * 10),
    'semseg_conv2d_fwd_s3': (c_int, [vp, vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_dgrad_s3': (c_int, [vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_wgrad_s3': (c_int, [vp, vp, vp] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_s3_set_plan': (c_int, [c_int] * 13),
    'semseg_split_h2_bytes': (c_sz, [c_int, c_int]),
    'semseg_split_h2': (c_int, [vp, c_int, vp, c_int, c_int, vp]),
    'semseg_split_h2_bounds': (c_int, [vp, c_int, vp, c_int, c_int, ctypes.POINTER(vp), c_int, vp]),
    'semseg_bound_sum': (c_int, [ctypes.POINTER(vp), c_int, vp, vp]),
    'semseg_absmax': (c_int, [vp, c_int, c_int, c_int, vp, vp, c_sz, vp]),
    'semseg_conv2d_h2_workspace_bytes': (c_sz, [c_int] * 10),
    'semseg_conv2d_fwd_h2': (c_int, [vp, vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_fwd_stats_bytes': (c_sz, [c_int]),
    'semseg_conv2d_fwd_stats_h2': (c_int, [vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp, c_sz, vp, ctypes.POINTER(c_int), vp]),
    'semseg_conv2d_dgrad_h2': (c_int, [vp, vp, vp, c_int] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_wgrad_h2': (c_int, [vp, vp, vp] + [c_int] * 10 + [vp, c_sz, vp]),
    'semseg_conv2d_wgrad_slabs_bytes': (c_sz, [c_int] * 10),
    'semseg_conv2d_wgrad_slabs_h2': (c_int, [vp, vp, vp, c_sz, ctypes.POINTER(c_int)] + [c_int] * 10 + [vp]),
    'semseg_reduce_slabs_multi': (c_int, [ctypes.POINTER(SlabTensor), c_int, vp]),
    'semseg_conv2d_wgrad_tile_h2': (c_int, [c_int] * 10),
    'semseg_conv2d_wgrad_multi_h2': (c_int, [ctypes.POINTER(WgradProblem), c_int, vp]),
    'semseg_conv2d_wgrad_member_plan': (c_int, [c_int]),
    'semseg_conv2d_h2_set_plan': (c_int, [c_int] * 13),
    'semseg_bias_grad': (c_int, [vp, c_int, vp, c_int, c_int, vp, c_sz, vp]),
    'semseg_bn_workspace_bytes': (c_sz, [c_int, c_int]),
    'semseg_bn_stats': (c_int, [vp, c_int, c_int, vp, vp, c_sz, vp]),
    'semseg_bn_finalize': (c_int, [vp, c_int, vp, vp, vp, vp, vp, c_f, c_f, vp, vp, vp, vp, vp]),
    'semseg_bn_eval_coeffs': (c_int, [vp, vp, vp, vp, c_f, c_int, vp, vp, vp, vp, vp]),
    'semseg_bn_apply': (c_int, [vp, vp, vp, vp, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'semseg_bn_bwd_reduce': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, c_sz, vp]),
    'semseg_bn_bwd_apply': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int, vp]),
    'semseg_bn_mm_workspace_bytes': (c_sz, [c_int, c_int]),
    'semseg_bn_stats_mm': (c_int, [vp, c_int, c_int, vp, vp, vp, c_sz, vp]),
    'semseg_bn_stats_mm_partial': (c_int, [vp, c_int, c_int, vp, c_sz, vp]),
    'semseg_bn_finalize_mm': (c_int, [vp, vp, c_int, vp, vp, vp, vp, vp, c_f, c_f, c_int, vp, vp, vp, vp, vp, vp, vp,
                                      c_int, vp]),
    'semseg_bn_apply_h2': (c_int, [vp, vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int, vp, vp, vp]),
    'semseg_bn_apply_h2_gate': (c_int, [vp, vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int, vp, vp, vp, vp]),
    'semseg_bn_bwd_reduce_mm': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, vp, c_sz,
                                        vp]),
    'semseg_bn_bwd_bound': (c_int, [vp, vp, vp, vp, vp, vp, vp, c_int, c_int, vp, c_int, vp]),
    'semseg_bn_bwd_apply_h2': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int,
                                       vp, vp, vp, vp]),
    'semseg_bn_fwd_stats_fused': (c_int, [vp, c_int, c_int, vp, vp, vp, vp, vp, vp, vp, c_f, c_f, c_int, vp, vp, vp, vp, vp,
                                          vp, vp, c_sz, vp]),
    'semseg_bn_bwd_reduce_fused': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, c_int,
                                           vp, vp, vp, vp, vp, c_sz, vp]),
    'semseg_bn_fwd_stats_fused_peer': (c_int, [vp, c_int, c_int, vp, vp, vp, vp, vp, vp, vp, c_f, c_f, c_int, vp, vp, vp, vp, vp,
                                               vp, vp, c_sz, vp] + [vp, vp]),
    'semseg_bn_fwd_stats_fused_bound': (c_int, [vp, c_int, c_int, vp, vp, vp, vp, vp, vp, vp, c_f, c_f, c_int, vp, vp, vp, vp, vp,
                                                vp, vp, c_sz, vp] + [vp]),
    'semseg_bn_fwd_finish_fused': (c_int, [vp, c_sz, c_int, c_int, c_int, vp, vp, vp, vp, vp, vp, vp, c_f, c_f, c_int, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp]),
    'semseg_bn_bwd_reduce_fused_peer': (c_int, [vp, c_int, vp, c_int, vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, c_int,
                                           vp, vp, vp, vp, vp, c_sz, vp] + [vp]),
    'semseg_bn_bwd_reduce_fused_sum2': (c_int, [vp, c_int, vp, c_int, vp, c_int, vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, c_int,
                                                vp, vp, vp, vp, vp, c_sz, vp, vp]),
    'semseg_bn_bwd_apply_h2_sum2': (c_int, [vp, c_int, vp, c_int, vp, c_int, vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int,
                                            vp, vp, vp, vp]),
    'semseg_weights_prepare_h2': (c_int, [ctypes.POINTER(WPrepTensor), c_int, vp]),
    'semseg_add_act': (c_int, [vp, c_int, vp, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'semseg_relu_bwd': (c_int, [vp, c_int, vp, c_int, vp, c_int, c_int, c_int, vp]),
    'semseg_clamp_max': (c_int, [vp, c_int, c_f, vp, c_int, c_int, c_int, vp]),
    'semseg_clamp_max_bwd': (c_int, [vp, c_int, vp, c_int, c_f, vp, c_int, c_int, c_int, vp]),
    'semseg_dropout_mask': (c_int, [vp, c_int, c_f, vp, vp]),
    'semseg_upsample_softmax': (c_int, [vp, c_int, vp, c_int, c_int, c_f] + [c_int] * 6 + [vp]),
    'semseg_copy2d': (c_int, [vp, c_int, vp, c_int, c_int, c_int, c_int, vp]),
    'semseg_scale_nc': (c_int, [vp, vp, vp, c_int, c_int, c_int, vp]),
    'semseg_nchw_to_nhwc': (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    'semseg_nhwc_to_nchw': (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    'semseg_maxpool3x3s2_fwd': (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp]),
    'semseg_maxpool3x3s2_bwd': (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp]),
    'semseg_adaptive_avgpool_fwd': (c_int, [vp, c_int, vp] + [c_int] * 6 + [vp]),
    'semseg_adaptive_avgpool_bwd': (c_int, [vp, vp, c_int, c_int] + [c_int] * 6 + [vp]),
    'semseg_adaptive_avgpool_multi_workspace_bytes': (c_sz, [c_int, c_int, c_int, ctypes.POINTER(c_int), c_int]),
    'semseg_adaptive_avgpool_multi_fwd': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int),
                                                  ctypes.POINTER(vp), vp, c_sz, vp]),
    'semseg_adaptive_avgpool_multi_bwd': (c_int, [ctypes.POINTER(vp), ctypes.POINTER(c_int), c_int, vp, c_int, c_int, c_int,
                                                  c_int, c_int, vp]),
    'semseg_bilinear_fwd': (c_int, [vp, c_int, vp, c_int, c_int, c_int] + [c_int] * 6 + [vp]),
    'semseg_bilinear_bwd': (c_int, [vp, c_int, vp, c_int, c_int] + [c_int] * 6 + [vp]),
    'semseg_log_softmax_fwd': (c_int, [vp, vp, c_int, c_int, vp]),
    'semseg_log_softmax_bwd': (c_int, [vp, vp, vp, c_int, c_int, vp]),
    'semseg_softmax_fwd': (c_int, [vp, vp, c_int, c_int, vp]),
    'semseg_nll_acc_fwd': (c_int, [vp, vp, c_int, c_int, c_int, vp, vp, c_sz, vp]),
    'semseg_nll_bwd': (c_int, [vp, vp, vp, c_int, vp, c_int, c_int, vp]),
    'semseg_argmax_metrics': (c_int, [vp, c_int, vp, c_int, c_int, vp, vp, vp]),
    'semseg_label_metrics': (c_int, [vp, vp, c_int, c_int, vp, vp]),
    'semseg_winograd_tiles': (c_int, [c_int, c_int, c_int, c_int]),
    'semseg_winograd_input_h2': (c_int, [vp, c_int, ctypes.POINTER(vp), c_int, vp, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_winograd_input_planes_h2': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_winograd_gemm_h2': (c_int, [vp, vp, vp, c_int, c_int, c_int, vp]),
    'semseg_winograd_gemm_output_h2': (c_int, [vp, vp, vp, c_int] + [c_int] * 7 + [vp]),
    'semseg_winograd_output': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_winograd_dm_h2': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_winograd_wgrad_workspace_bytes': (c_sz, [c_int, c_int, c_int]),
    'semseg_winograd_wgrad_gemm_h2': (c_int, [vp, vp, vp, c_int, c_int, c_int, vp, c_sz, vp]),
    'semseg_winograd_dg': (c_int, [vp, vp, c_int, c_int, vp]),
    'semseg_depthwise3x3_workspace_bytes': (c_sz, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'semseg_depthwise3x3_fwd': (c_int, [vp, c_int, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_depthwise3x3_dgrad': (c_int, [vp, c_int, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'semseg_depthwise3x3_wgrad': (c_int, [vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_sz,
                                          vp]),
    'semseg_grouped3x3_workspace_bytes': (c_sz, [c_int] * 9),
    'semseg_grouped3x3_fwd': (c_int, [vp, c_int, vp, vp, c_int] + [c_int] * 9 + [vp]),
    'semseg_grouped3x3_dgrad': (c_int, [vp, c_int, vp, vp, c_int] + [c_int] * 9 + [vp]),
    'semseg_grouped3x3_wgrad': (c_int, [vp, c_int, vp, c_int, vp] + [c_int] * 9 + [vp, c_sz, vp]),
    'semseg_input_resample_ksize': (c_int, [c_int, c_int]),
    'semseg_input_resample_coeffs': (c_int, [c_int, c_int, vp, vp]),
    'semseg_input_nearest_table': (c_int, [c_int, c_int, vp]),
    'semseg_input_resample_h_u8': (c_int, [vp, c_int, c_int, c_int, vp, vp, c_int, vp, c_int, vp]),
    'semseg_input_resample_v_normalize': (c_int, [vp, c_int, c_int, vp, vp, c_int, c_int, vp, vp, c_int, vp]),
    'semseg_input_label_gather': (c_int, [vp, c_int, vp, vp, c_int, c_int, vp, c_int, vp]),
    'semseg_sgd_step': (c_int, [ctypes.POINTER(SgdTensor), c_int, vp, c_f, c_f, vp]),
    'semseg_sgd_step_fused': (c_int, [ctypes.POINTER(SgdTensor2), c_int, vp, c_f, c_f, vp]),
    'semseg_weights_absmax_slots': (vp, [vp, c_int, c_int, c_int]),
    'semseg_weights_prepare_h2_after_sgd': (c_int, [ctypes.POINTER(WPrepTensor), c_int, c_int, vp]),
    'semseg_comm_available': (c_int, []),
    'semseg_comm_version': (c_int, []),
    'semseg_comm_unique_id': (c_int, [vp]),
    'semseg_comm_init': (c_int, [c_int, c_int, vp, ctypes.POINTER(vp)]),
    'semseg_comm_count': (c_int, [vp, ctypes.POINTER(c_int)]),
    'semseg_comm_allreduce_sum_f32': (c_int, [vp, vp, c_sz, vp]),
    'semseg_comm_allreduce_sum_f64': (c_int, [vp, vp, c_sz, vp]),
    'semseg_comm_allreduce_sum_f64_multi': (c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(c_sz), c_int, vp]),
    'semseg_comm_destroy': (c_int, [vp]),
    'semseg_peer_max_world': (c_int, []),
    'semseg_peer_create': (c_int, [c_int, c_int, c_int, ctypes.c_double, ctypes.POINTER(vp)]),
    'semseg_peer_set_timeout': (c_int, [vp, ctypes.c_double]),
    'semseg_peer_handle': (c_int, [vp, vp]),
    'semseg_peer_attach': (c_int, [vp, c_int, vp]),
    'semseg_peer_attach_local': (c_int, [vp, c_int, vp]),
    'semseg_peer_allreduce_sum_f64': (c_int, [vp, vp, c_sz, vp]),
    'semseg_peer_status': (c_int, [vp]),
    'semseg_peer_destroy': (c_int, [vp]),
    'semseg_bn_peer_channel_capacity': (c_int, []),
    'semseg_batch_begin': (c_int, [c_int, vp]),
    'semseg_batch_branch': (c_int, [c_int]),
    'semseg_batch_next_op': (c_int, []),
    'semseg_batch_next_unit': (c_int, []),
    'semseg_batch_flush': (c_int, []),
    'semseg_batch_end': (c_int, []),
    'semseg_batch_abort': (c_int, []),
    'semseg_batch_active': (c_int, []),
    'semseg_batch_stats': (c_int, [ctypes.POINTER(ctypes.c_longlong)]),
    'semseg_batch_plan': (c_int, [c_int, c_int]),
    'semseg_probe_timestamp': (c_int, [vp, vp]),
    'semseg_probe_mfma_f16': (c_int, [vp, c_int, c_int, vp, vp]),
    'semseg_probe_copy': (c_int, [vp, vp, c_sz, vp]),
    'semseg_probe_empty': (c_int, [vp]),
    'semseg_probe_gather': (c_int, [vp, ctypes.c_uint, ctypes.c_uint, c_int, c_int, c_int, c_int, vp, vp]),
}


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises NativeLibraryMissing if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                '%s not built: run `python semantic-segmentation-pytorch_amd/build_native.py` '
                '(there is no CPU/eager fallback for the HIP hot path)' % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class NativeError(RuntimeError):
    """a C-ABI entry point returned a non-zero code; `.code` is that code (SEMSEG_E* < 0, else a hipError_t)"""

    def __init__(self, what, code):
        self.what, self.code = what, int(code)
        super().__init__('%s failed with code %d (%s)' % (
            what, code, {-1: 'SEMSEG_EINVAL', -2: 'SEMSEG_EWORKSPACE'}.get(code, 'hipError_t')))


def is_capture_error(exc):
    """True if `exc` says that an operation was refused BECAUSE a stream capture is under way (hipErrorStreamCapture* = 900 ... 908
    from an entry point, or torch's own 'operation not permitted when stream is capturing' family) -- the one kind of failure
    after which running the same step eagerly is the right answer.  Never an out-of-memory error."""
    if isinstance(exc, torch.cuda.OutOfMemoryError):
        return False
    code = getattr(exc, 'code', None)
    if isinstance(code, int) and 900 <= code <= 908:
        return True
    msg = str(exc).lower()
    return 'captur' in msg and 'out of memory' not in msg


RECORDING = [False]        # ops.BranchesFn: a side-by-side scope (semseg_batch_begin) is open -- every checked C-ABI call is one ordinal


def next_unit():
    """ops: the current branch of an open side-by-side scope enters its next conv -> BN unit (semseg_batch_next_unit)"""
    if RECORDING[0]:
        _lib.semseg_batch_next_unit()


def check(rc, what):
    if rc != 0:
        raise NativeError(what, rc)
    if RECORDING[0]:
        _lib.semseg_batch_next_op()
