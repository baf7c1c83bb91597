"""ctypes binding of libtecogan_hip.so (the C ABI in include/tecogan_hip.h).

The library is built in-tree by `csrc/build.sh` (or __graft_entry__.build()).
Loading failures are loud: there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TECOGAN_HIP_LIB: measurement builds of the same sources (tools/build_lab_libs.sh); never a fallback
LIB_PATH = os.environ.get('TECOGAN_HIP_LIB') or os.path.join(_HERE, 'libtecogan_hip.so')

TG_OK = 0
ABI_MAJOR = 2            # include/tecogan_hip.h TG_ABI_MAJOR
ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_TANH24 = 0, 1, 2, 3
UP_NONE, UP_BICUBIC, UP_BILINEAR = 0, 1, 2

P, I, I64, F, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class FrnetCfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ('in_nc', 'out_nc', 'nf', 'nb', 'scale', 'up_mode', 'n', 'h', 'w', 'fnet_only')]


class WinoLayer(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x2', C.c_void_p), ('u_packed', C.c_void_p), ('bias', C.c_void_p),
                ('res', C.c_void_p), ('y', C.c_void_p), ('x_nstride', C.c_int64), ('x2_nstride', C.c_int64),
                ('res_nstride', C.c_int64), ('y_nstride', C.c_int64), ('c1', C.c_int), ('cin', C.c_int),
                ('act', C.c_int)]


class ChainLayer(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x2', C.c_void_p), ('w_packed', C.c_void_p), ('bias', C.c_void_p),
                ('res', C.c_void_p), ('relu_mask', C.c_void_p), ('y', C.c_void_p),
                ('x_nstride', C.c_int64), ('x2_nstride', C.c_int64), ('res_nstride', C.c_int64),
                ('mask_nstride', C.c_int64), ('y_nstride', C.c_int64),
                ('c1', C.c_int), ('cin', C.c_int), ('cout', C.c_int), ('act', C.c_int)]


class PackedLayer(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p)]


class PackItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('out', C.c_void_p), ('cin', C.c_int), ('cout', C.c_int), ('transposed', C.c_int),
                ('i_total', C.c_int), ('i_off', C.c_int)]


class WresConvT(C.Structure):
    _fields_ = [('u_packed', C.c_void_p), ('bias', C.c_void_p), ('y', C.c_void_p), ('act', C.c_int)]


class LayerWeights(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p), ('u', C.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/tecogan_hip.h declares
SIGNATURES = {
    'tg_version': (I, []),
    'tg_last_error_string': (C.c_char_p, []),
    'tg_build_info': (C.c_char_p, []),
    'tg_conv3x3_pick_ocb': (I, [I]),
    'tg_conv3x3_packed_floats': (SZ, [I, I, I]),
    'tg_conv3x3_pack': (I, [P, P, I, I, I, I, P]),
    'tg_conv3x3_fwd': (I, [P, I64, I, P, I64, P, I, P, P, I64, P, I64, I, I, I, I, I, I, P]),
    'tg_conv3x3_fwd_phased': (I, [P, I64, P, I, P, P, I64, I, I, I, I, I, I, I, I, I, I, P]),
    'tg_conv3x3_fwd_phased_masked': (I, [P, I64, P, I, P, P, I64, P, I64, I, I, I, I, I, I, I, I, I, I, P]),
    'tg_conv3x3s2_supported': (I, [I, I, I, I, I]),
    'tg_conv3x3s2_fwd': (I, [P, I64, P, P, P, I64, P, I64, I, I, I, I, I, I, P]),
    'tg_conv3x3_fwd_masked': (I, [P, I64, I, P, I64, P, I, P, P, I64, P, I64, P, I64, I, I, I, I, I, I, P]),
    'tg_conv3x3_pick_ksplit': (I, [I, I, I, I, I]),
    'tg_conv3x3_splitk_fwd': (I, [P, I64, I, P, I64, P, I, P, P, I, I, I, I, I, I, I, P, I, P]),
    'tg_convt3x3s2_fwd': (I, [P, I64, P, P, P, I64, I, I, I, I, I, I, P]),
    'tg_convt_pack_wz': (I, [P, P, I, I, P]),
    'tg_convt3x3s2_z_fwd': (I, [P, I64, P, P, P, I, P, I64, I, I, I, I, I, I, P]),
    'tg_convt3x3s2_z_fwd_form': (I, [P, I64, P, P, P, I, P, I64, I, I, I, I, I, I, I, P]),
    'tg_convout_tail': (I, [P, I64, I, P, P, I, I, P, I64, P, I, I, I, P]),
    'tg_convout_tail_form': (I, [P, I64, I, P, P, I, I, P, I64, P, I, I, I, I, P]),
    'tg_conv3x3_small_fwd': (I, [P, I64, P, P, P, I, I, P, I64, I, I, I, I, I, I, P]),
    'tg_conv3x3_fewin_fwd': (I, [P, I64, P, P, I64, P, I64, I, I, I, I, I, P]),
    'tg_conv3x3_small_fwd_res': (I, [P, I64, P, P, P, I64, P, I64, I, I, I, I, I, I, P]),
    'tg_conv3x3_small_fwd_u8': (I, [P, I64, P, P, P, I, I, P, I64, P, I, I, I, I, I, I, P]),
    'tg_conv3x3_small_can_fuse_u8': (I, [P, I64, P, I64, I, I, I, I]),
    'tg_flowup_warp_s2d_fwd': (I, [P, I, I, P, P, I64, P, I, I, I, I, I, I, P]),
    'tg_copy_ceiling': (I, [P, P, I64, I, I, P]),
    'tg_backward_warp_fwd': (I, [P, P, P, I, I, I, I, P]),
    'tg_space_to_depth': (I, [P, P, I64, I, I, I, I, I, P]),
    'tg_upsample_fwd': (I, [P, P, I, I, I, I, I, F, P]),
    'tg_maxpool2_fwd': (I, [P, P, I, I, I, P]),
    'tg_quantize_u8_hwc': (I, [P, P, I, I, I, P]),
    'tg_dequantize_u8_hwc': (I, [P, P, I, I, I, I, P]),
    'tg_psnr_sse_u8': (I, [P, P, P, I, I, I, I, P]),
    'tg_luma_u8': (I, [P, P, I64, P]),
    'tg_wgrad3x3_workspace_floats': (SZ, [I, I, I, I, I]),
    'tg_wgrad3x3': (I, [P, I64, P, I64, P, P, I, I, I, I, I, I, I, I, P]),
    'tg_wgrad3x3_multi': (I, [P, P, I, I64, I64, P, P, I, I, I, I, I, I, I, I, P]),
    'tg_wgrad3x3_multi_bias': (I, [P, P, I, I64, I64, P, P, P, I, I, I, I, I, I, I, I, P]),
    'tg_wgrad3x3_multi_phased': (I, [P, P, I, I64, I64, P, P, I, I, I, I, I, I, I, I, I, P]),
    'tg_conv3x3_phased_pick_ksplit': (I, [I, I, I, I, I, I]),
    'tg_conv3x3_fwd_phased_splitk': (I, [P, I64, P, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P]),
    'tg_wgrad3x3_convt_workspace_floats': (SZ, [I, I, I, I, I]),
    'tg_wgrad3x3_convt_multi': (I, [P, P, I, P, P, P, I, I, I, I, I, I, P]),
    'tg_bias_grad_multi': (I, [P, I, P, I, I, I, I, P]),
    'tg_act_bwd': (I, [P, P, P, I64, I, P]),
    'tg_bias_grad': (I, [P, P, I, I, I, I, P]),
    'tg_maxpool2_bwd': (I, [P, P, P, I, I, I, P]),
    'tg_upsample_bwd': (I, [P, P, I, I, I, I, I, F, P]),
    'tg_backward_warp_bwd': (I, [P, P, P, P, P, I, I, I, I, P]),
    'tg_backward_warp_bwd_acc': (I, [P, P, P, P, P, I, I, I, I, P]),
    'tg_conv3x3_wino_resident_ct_floats': (SZ, []),
    'tg_conv3x3_wino_resident_ct_pack': (I, [P, P, P]),
    'tg_conv3x3_wino_resident_ct': (I, [P, I, I, I, I, P, I, P, P]),
    'tg_conv4x4s2_supported': (I, [I, I, I, I, I]),
    'tg_conv4x4s2_packed_floats': (SZ, [I, I]),
    'tg_conv4x4s2_pack': (I, [P, P, P, I, I, P]),
    'tg_conv4x4s2_workspace_floats': (SZ, [I, I, I, I, I, I]),
    'tg_conv4x4s2_fwd': (I, [P, P, P, P, I, I, I, I, I, P]),
    'tg_conv4x4s2_dgrad': (I, [P, P, P, I, P, P, I, I, I, I, I, P]),
    'tg_backward_warp_s2d_fwd': (I, [P, P, P, I, I, I, I, I, P]),
    'tg_backward_warp_s2d_bwd': (I, [P, P, P, P, I, P, I, I, I, I, I, P]),
    'tg_depth_to_space': (I, [P, P, I, I, I, I, I, P]),
    'tg_depth_to_space_act_bwd_supported': (I, [P, P, P, I, I]),
    'tg_depth_to_space_act_bwd': (I, [P, P, I, P, I, I, I, I, I, P]),
    'tg_charbonnier': (I, [P, P, I64, F, F, P, F, P, P]),
    'tg_channel_norm': (I, [P, P, P, P, I, I, I64, P]),
    'tg_cosine_loss': (I, [P, P, I, I, I64, F, F, P, F, P, P]),
    'tg_pixel_loss': (I, [P, P, I64, I, F, P, F, P, P]),
    'tg_bce_logits': (I, [P, I64, F, F, P, F, P, P]),
    'tg_lsgan_loss': (I, [P, I64, F, F, P, F, P, P]),
    'tg_adam_step': (I, [P, P, P, P, I64, F, F, F, F, F, I, P]),
    'tg_adam_step_guarded': (I, [P, P, P, P, I64, F, F, F, F, F, I, P, P]),
    'tg_fault_to_slot': (I, [P, P, P]),
    'tg_axpy': (I, [P, P, F, I64, P]),
    'tg_div_scalar': (I, [P, P, F, I64, P]),
    'tg_bn_lrelu_train_fwd': (I, [P, P, P, P, P, F, F, F, P, P, P, I, I, I, P]),
    'tg_bn_lrelu_train_bwd': (I, [P, P, P, P, P, P, F, P, P, P, I, P, I, I, I, P]),
    'tg_bn_local_stats': (I, [P, P, I, I, I, P]),
    'tg_bn_merge_stats': (I, [P, I, F, F, F, P, P, P, P, I, P]),
    'tg_bn_lrelu_apply': (I, [P, P, P, P, P, F, P, I, I, I, P]),
    'tg_bn_lrelu_bwd_reduce': (I, [P, P, P, P, P, F, P, I, I, I, P]),
    'tg_bn_lrelu_bwd_apply': (I, [P, P, P, P, P, P, P, F, F, P, I, I, I, P]),
    'tg_linear1_fwd': (I, [P, P, P, P, I, I, P]),
    'tg_linear1_bwd': (I, [P, P, P, P, P, P, I, I, I, P]),
    'tg_downsample_bd': (I, [P, P, P, I, I, I, I, I, I, P]),
    'tg_conv3x3_wino_packed_floats': (I64, [I, I]),
    'tg_conv3x3_wino_chain_flag_ints': (I64, [I, I, I, I]),
    'tg_conv3x3_wino_chain': (I, [C.POINTER(WinoLayer), I, I, I, I, I, P, I, P]),
    'tg_conv3x3_wino_resident_supported': (I, [I, I, I, I]),
    'tg_conv3x3_wino_resident_ws_bytes': (I64, [I, I]),
    'tg_conv3x3_wino_resident': (I, [C.POINTER(WinoLayer), I, I, I, I, P, I, P]),
    'tg_conv3x3_chain_flag_ints': (I64, [I, I, I, I]),
    'tg_conv3x3_chain_supported': (I, [I, I, I, I]),
    'tg_conv3x3_pack16_floats': (SZ, []),
    'tg_conv3x3_pack16': (I, [P, P, I, I, I, P]),
    'tg_conv3x3_chain_packed_floats': (SZ, [I, I]),
    'tg_conv3x3_chain_pack': (I, [P, I, I, P]),
    'tg_conv3x3_chain': (I, [C.POINTER(ChainLayer), I, I, I, I, I, P, P, C.c_uint32, I, P]),
    'tg_srnet_body_fwd': (I, [C.POINTER(PackedLayer), I, I, P, I, P, I, P, I, I, I, I, P, P, C.c_uint32, I, P]),
    'tg_srnet_body_bwd': (I, [C.POINTER(PackedLayer), I, I, P, P, P, I, I, I, I, I, P, P, C.c_uint32, I, P]),
    'tg_wgrad3x3_body_workspace_floats': (SZ, [I, I, I, I, I, I]),
    'tg_wgrad3x3_body': (I, [P, P, I, I64, I, P, P, I, I, I, I, I, P]),
    'tg_wgrad3x3_body_bias': (I, [P, P, I, I64, I, P, P, P, I, I, I, I, I, P]),
    'tg_bias_grad_body': (I, [P, I, I64, I, P, I, I, I, P]),
    'tg_conv3x3_prefers_wino': (I, [I, I, I, I, I]),
    'tg_pack_conv3x3_wino': (I, [P, P, I, I, I, P]),
    'tg_conv3x3_wino_fwd': (I, [P, I64, I, P, I64, P, P, P, I64, P, I64, P, I64, I, I, I, I, I, I, P]),
    'tg_time_gather': (I, [P, P, P, I, I, I, I64, P]),
    'tg_transpose01': (I, [P, P, I, I, I64, P]),
    'tg_stack_time': (I, [P, I, P, I, I64, P]),
    'tg_index_gather': (I, [P, P, P, I64, I64, I, P]),
    'tg_pingpong_grad': (I, [P, P, I, I, I64, P]),
    'tg_d_assemble_fwd': (I, [P, I, P, P, I, P, I, I, I, I, I, I, I, P]),
    'tg_d_assemble_bwd': (I, [P, P, I, P, I, I, I, I, I, I, I, P]),
    'tg_gather_clips_u8': (I, [P, P, P, P, I, I, I, I, P]),
    'tg_comm_get_unique_id': (I, [P]),
    'tg_comm_init_rank': (I, [P, I, I, C.POINTER(C.c_void_p)]),
    'tg_comm_destroy': (I, [P]),
    'tg_comm_world': (I, [P]),
    'tg_comm_rank': (I, [P]),
    'tg_comm_query': (I, [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'tg_comm_library_origin': (C.c_char_p, []),
    'tg_allreduce_sum_f32': (I, [P, P, I64, P]),
    'tg_allgather_f32': (I, [P, P, P, I64, P]),
    'tg_frnet_workspace_floats': (SZ, [C.POINTER(FrnetCfg)]),
    'tg_frnet_plan_create': (I, [C.POINTER(FrnetCfg), C.POINTER(LayerWeights), I, P,
                                 C.POINTER(C.c_void_p)]),
    'tg_frnet_plan_destroy': (None, [P]),
    'tg_frnet_plan_chain_status': (I, [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'tg_frnet_plan_set_chain_poll_limit': (I, [P, I]),
    'tg_frnet_plan_set_chain_rearm': (I, [P, I]),
    'tg_frnet_plan_chain_rearms': (I, [P, P, P]),
    'tg_frnet_step': (I, [P, P, P, P, P, P, P]),
    'tg_frnet_step_phase': (I, [P, I, I, P, P, P, P, P, P]),
    'tg_frnet_plan_launches': (I, [P]),
    'tg_frnet_plan_flow': (P, [P, I]),
    'tg_frnet_step_srnet': (I, [P, P, P, P, P, P, P]),
    'tg_frnet_plan_kinds': (I, []),
    'tg_frnet_kind_name': (C.c_char_p, [I]),
    'tg_frnet_plan_kind_stats': (I, [P, I, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    'tg_frnet_replay': (I, [P, P, P, P, P, C.c_uint, I, P]),
    'tg_frnet_step_masked': (I, [P, P, P, P, P, P, C.c_uint, P]),
}

_lib = None


class TecoganHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise TecoganHipError(
                f'{LIB_PATH} not found: build it with '
                f'`bash {os.path.join(_HERE, "csrc", "build.sh")}` '
                '(hipcc --offload-arch=gfx950).  There is no CPU/ATen fallback.')
        # torch bundles its own HIP runtime; it must be the one already mapped
        # when our library (linked against libamdhip64) is opened, otherwise two
        # runtimes coexist and ours sees no device.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        handle.tg_version.restype = I
        if handle.tg_version() // 100 != ABI_MAJOR:
            raise TecoganHipError(f'{LIB_PATH}: ABI {handle.tg_version()} but this binding is written for '
                                  f'major {ABI_MAJOR} (include/tecogan_hip.h TG_ABI_MAJOR): rebuild the library')
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != TG_OK:
        msg = lib().tg_last_error_string().decode('utf-8', 'replace')
        raise TecoganHipError(f'{what} failed (code {rc}): {msg}')
