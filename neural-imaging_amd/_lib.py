"""
ctypes binding of libnimg.so (the C ABI declared in include/nimg.h).

The library is built in-tree by neural-imaging_amd/csrc/build.sh (hipcc --offload-arch=gfx950).  There is NO
fallback: if the library is missing, or an entry point returns an error code, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NIMG_LIBPATH') or os.path.join(_HERE, 'libnimg.so')     # override: A/B of kernel builds

P = c_void_p  # every device pointer / stream is passed as an opaque pointer

# name -> (restype, argtypes); must list every symbol of include/nimg.h (tests/test_abi.py checks this)
PROTOTYPES = {
    'nimg_abi_version': (c_int, []),
    'nimg_djpeg_fwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_djpeg_bwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_djpeg_dq_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'nimg_djpeg_bwd_dq': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'nimg_jpeg_qtable': (c_int, [c_int, c_int, P]),
    'nimg_conv2d_fwd': (c_int, [P, c_int, P, c_int, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv_flip_weights': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_conv2d_wgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'nimg_conv2d_wgrad': (c_int, [P, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'nimg_bias_grad_workspace_bytes': (c_size_t, [c_long, c_int]),
    'nimg_bias_grad': (c_int, [P, P, c_long, c_int, c_int, P, c_size_t, P]),
    'nimg_bias_grad_ex': (c_int, [P, P, c_long, c_int, c_int, P, c_size_t, c_int, P]),
    'nimg_convt2x2_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_maxpool2_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_maxpool2_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_maxpool2_fwd_bf16': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_maxpool2_bwd_bf16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv2d_pool_fwd': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv2d_pool_fwd_bf16': (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                          P]),
    'nimg_maxpool2_unpool': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_d2s_clip_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    'nimg_d2s_clip_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_lrelu_bwd': (c_int, [P, P, P, c_long, c_float, P]),
    'nimg_add': (c_int, [P, P, P, c_long, P]),
    'nimg_add_n': (c_int, [P, c_int, P, c_long, P]),
    'nimg_avgpool_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_avgpool_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_mse255_workspace_bytes': (c_size_t, []),
    'nimg_mse255': (c_int, [P, P, P, P, c_long, c_float, c_int, P, c_size_t, P]),
    'nimg_mse255_sum_s2d3': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    'nimg_fan_head_fwd': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_fan_head_bwd': (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    'nimg_adam_step': (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_int, c_float, P, P]),
    'nimg_adam_step_dev': (c_int, [P, P, P, P, c_long, P, c_float, c_float, c_float, c_float, P, P]),
    'nimg_conv3_rows_bf16': (c_int, [P, c_int, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'nimg_conv3_rows_d2s_bf16': (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, P]),
    'nimg_conv2d_wgrad_bf16_deferred': (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_int, c_int, P, c_size_t, c_int, P, P]),
    'nimg_conv2d_wgrad_bf16_chained': (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_int, c_int, P, c_size_t, c_int, P, P, P]),
    'nimg_reduce_entry_bytes': (c_size_t, []),
    'nimg_reduce_batch_max': (c_int, []),
    'nimg_reduce_slabs_batch': (c_int, [P, c_int, P]),
    'nimg_conv3_rows_c4_bf16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_nan_flag': (c_int, [P, c_long, P, P]),
    'nimg_int_words': (c_int, [P, P, c_long, c_int, c_int, P]),
    'nimg_float_fill': (c_int, [P, c_long, c_float, P]),
    'nimg_bind_tickets': (c_int, [P, P, c_size_t]),
    'nimg_stream_create_cu_mask': (c_int, [c_int, P]),
    'nimg_stream_destroy': (c_int, [P]),
    'nimg_head_fused_ok': (c_int, [c_int, c_int]),
    'nimg_head_fwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    'nimg_head_wgrad_workspace_bytes': (c_size_t, [c_int, c_int]),
    'nimg_head_wgrad': (c_int, [P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_float, c_int, P, c_size_t, P]),
    'nimg_head_dgrad': (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_float, P]),
    'nimg_head_dact': (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_float, P]),
    'nimg_fan_dense_fwd': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    'nimg_fan_dense_bwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    'nimg_constrained_kernel_fwd': (c_int, [P, P, c_int, c_int, c_float, P]),
    'nimg_constrained_kernel_bwd': (c_int, [P, P, P, c_int, c_int, c_float, P]),
    'nimg_fold_pad': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_gaussian_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_gaussian_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'nimg_dwfilter_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_dwfilter_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_sharpen_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    'nimg_sharpen_bwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'nimg_affine': (c_int, [P, P, c_long, c_float, c_float, P]),
    'nimg_lrelu_fwd': (c_int, [P, P, c_long, c_float, P]),
    'nimg_zero_insert2': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_s2d2_affine_bf16': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    'nimg_s2d_conv_weights': (c_int, [P, P, c_int, c_int, c_int, P]),
    'nimg_s2d_conv_weights_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_d2s2_scale': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_latent_workspace_bytes': (c_size_t, [c_int]),
    'nimg_latent_fwd': (c_int, [P, P, P, c_int, c_float, c_float, c_int, P, P, c_long, c_long, P, c_size_t, c_int, P]),
    'nimg_latent_entropy_finalize': (c_int, [c_int, c_long, P, P, P]),
    'nimg_latent_bwd': (c_int, [P, P, P, P, c_float, P, c_int, c_float, c_float, c_int, P, P, c_int, c_long, P,
                                c_size_t, P]),
    'nimg_l2_loss_workspace_bytes': (c_size_t, []),
    'nimg_l2_loss': (c_int, [P, P, P, P, c_long, c_float, c_int, P, c_size_t, P]),
    'nimg_conv_weights_bf16_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'nimg_conv_weights_bf16': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_conv2d_fwd_bf16': (c_int, [P, c_int, P, c_int, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv2d_wgrad_bf16_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'nimg_conv2d_wgrad_bf16': (c_int, [P, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'nimg_conv2d_wgrad_pooled_bf16': (c_int, [P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t,
                                              P]),
    'nimg_conv2d_dgrad_fewin_pooled_bf16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_ssim_workspace_bytes': (c_size_t, [c_int]),
    'nimg_ssim': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    'nimg_tanh_fwd': (c_int, [P, P, c_long, P]),
    'nimg_tanh_bwd': (c_int, [P, P, P, c_long, P]),
    'nimg_clip01': (c_int, [P, P, c_long, P]),
    'nimg_isp_residual_fwd': (c_int, [P, P, P, P, c_long, c_int, P]),
    'nimg_isp_residual_workspace_bytes': (c_long, []),
    'nimg_isp_residual_bwd': (c_int, [P, P, P, P, P, P, c_long, c_int, P]),
    'nimg_activation_fwd': (c_int, [P, P, c_long, c_int, c_float, P]),
    'nimg_activation_bwd': (c_int, [P, P, P, c_long, c_int, c_float, P]),
    'nimg_sigmoid_fwd': (c_int, [P, P, c_long, P]),
    'nimg_sigmoid_bwd': (c_int, [P, P, P, c_long, P]),
    'nimg_gamma_ste_fwd': (c_int, [P, P, c_long, c_float, c_float, c_float, P]),
    'nimg_gamma_ste_bwd': (c_int, [P, P, P, c_long, c_float, c_float, c_float, P]),
    'nimg_mae255_workspace_bytes': (c_size_t, []),
    'nimg_mae255': (c_int, [P, P, P, P, c_long, c_float, c_int, P, c_size_t, P]),
    'nimg_ssim_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'nimg_ssim_loss': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_float, c_int, P, c_size_t, P]),
    'nimg_patch_stats': (c_int, [P, c_int, c_int, c_int, P, P, c_int, c_int, c_int, P, P, P]),
    'nimg_patch_select': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    'nimg_patch_gather': (c_int, [P, P, c_int, c_int, c_int, P, P, c_int, c_int, P, P, P]),
    'nimg_mask_scale': (c_int, [P, P, P, c_long, c_float, P]),
    'nimg_conv5_dgrad_sparse_image_bytes': (c_size_t, [c_int, c_int]),
    'nimg_conv5_dgrad_sparse_weights': (c_int, [P, P, c_int, c_int, P]),
    'nimg_conv5_dgrad_sparse': (c_int, [P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_float, c_int, P]),
    'nimg_conv2d_fwd_bf16_res': (c_int, [P, c_int, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_float, c_int, P]),
    'nimg_conv2d_fwd_bf16_unpool': (c_int, [P, P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_float, c_int, P]),
    'nimg_conv2d_wgrad_bf16_unpool': (c_int, [P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'nimg_cconv3': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_conv5c3_bf16': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'nimg_cconv3_dgrad_border': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'nimg_conv1_pool_fwd_c4': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P]),
    'nimg_conv1_wgrad_c4_workspace_bytes': (c_size_t, []),
    'nimg_conv1_wgrad_c4': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'nimg_conv1_dgrad_pooled': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_confusion_accumulate': (c_int, [P, P, P, P, c_int, c_int, P]),
    'nimg_ssim_planes_workspace_bytes': (c_size_t, [c_int, c_int]),
    'nimg_ssim_planes': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, P, P, P, c_int, P, c_size_t, P]),
    'nimg_msssim_combine': (c_int, [P, P, c_int, c_int, P, P, P]),
    'nimg_ssim_maps_grad': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_float, c_int, P]),
    'nimg_pad2d': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_conv2d_fwd_bf16_ex': (c_int, [P, c_int, P, c_int, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'nimg_conv2d_wgrad_bf16_ex': (c_int, [P, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    'nimg_conv2d_pool_fwd_bf16_ex': (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                             c_int, P]),
    'nimg_conv2d_dgrad_unpool_out_bf16': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv2d_fwd_pool_also_bf16': (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'nimg_conv2d_wgrad_pooled_bf16_ex': (c_int, [P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P,
                                                 c_size_t, c_int, P]),
    'nimg_conv2d_dgrad_fewin_pooled_bf16_ex': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_maxpool2_unpool_ex': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'nimg_conv_weights_bf16_batch': (c_int, [P, c_int, P]),
    'nimg_convt2x2_fwd_bf16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_convt2x2_fwd_bf16_ex': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_conv2d_fwd_smallc_bf16': (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_float, P]),
    'nimg_conv2d_fwd_smallc_bf16_ex': (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_float, c_int, P]),
    'nimg_conv2d_dgrad_fewin_bf16': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'nimg_awgn_fwd': (c_int, [P, P, P, P, c_long, c_float, P]),
    'nimg_awgn_bwd': (c_int, [P, P, P, P, P, c_long, c_float, P]),
    'nimg_gamma_fwd': (c_int, [P, P, c_long, c_float, P]),
    'nimg_gamma_bwd': (c_int, [P, P, P, c_long, c_float, P]),
    'nimg_median_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_median_bwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'nimg_sparse_axis_apply': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
}

ERRORS = {-1: 'NIMG_ERR_ARG (invalid argument / unsupported configuration)',
          -2: 'NIMG_ERR_LAUNCH (HIP kernel launch failed)',
          -3: 'NIMG_ERR_WORKSPACE (workspace too small)'}

_lib = None


ABI_VERSION = 5         # include/nimg.h NIMG_ABI_VERSION
TICKET_BYTES = 64 * 1024        # include/nimg.h NIMG_TICKET_BYTES


def load():
    """Load libnimg.so (once) and attach prototypes.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError('libnimg.so not found at {} - build it with neural-imaging_amd/csrc/build.sh '
                           '(or __graft_entry__.build()); there is no CPU fallback'.format(LIB_PATH))
    # torch first: its wheel carries its own copy of the HIP runtime, and libnimg.so must bind to THAT instance (the streams and
    # device pointers it is handed come from it).  Loaded before torch, libnimg.so pulls in /opt/rocm's libamdhip64 as a second
    # runtime and every launch on a torch stream fails (seen as NIMG_ERR_LAUNCH from `python __graft_entry__.py smoke`, whose
    # build() loaded the library before anything had imported torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI drift; let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.nimg_abi_version() != ABI_VERSION:
        raise RuntimeError('libnimg.so at {} has ABI version {}, this binding was written against {} - rebuild it '
                           '(neural-imaging_amd/csrc/build.sh)'.format(LIB_PATH, lib.nimg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError('{} failed: {}'.format(name, ERRORS.get(rc, rc)))
    return rc
