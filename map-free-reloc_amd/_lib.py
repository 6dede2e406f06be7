"""ctypes loader for csrc/libmfr_hip.so (C-ABI: include/mfr_hip.h).

torch is imported first so that its bundled HIP runtime (libamdhip64.so.7) is the one the
library binds to -- device pointers and streams handed over from torch tensors then belong to
the same runtime.  Loading fails loudly; nothing in this package falls back to a CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libmfr_hip.so")

_lib = None

_vp, _i, _d, _u64, _sz = C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/mfr_hip.h declaration by declaration
SIGNATURES = {
    "mfr_abi_version": (_i, []),
    "mfr_target_arch": (C.c_char_p, []),
    "mfr_f16x2_guard_bind": (_i, [_vp]),
    "mfr_test_f64_ops": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "mfr_test_sample": (_i, [_u64, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mfr_pnp_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_pnp_solve_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _d, _d, _u64, _vp,
                                 _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_depth_min": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mfr_pnp_lift": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mfr_pnp_ransac": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _d, _d, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp]),
    "mfr_emat_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_magsac_lut": (_i, [_vp, _i]),
    "mfr_emat_solve_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _d, _d, _i, _u64, _vp, _i, _vp, _i, _d, _vp, _sz,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_procrustes_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_procrustes_solve_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _d, _d, _i, _u64, _vp, _vp, _sz,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_procrustes_icp_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_procrustes_icp_refine": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _d, _d, _d, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_sp_scoremap": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mfr_sp_nms_candidates": (_i, [_vp, _i, _i, _i, _i, C.c_float, _i, _vp, _vp, _i, _vp, _vp]),
    "mfr_sp_select_topk": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mfr_sp_sample_descriptors": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "mfr_gemm_f16x2_pack_bytes": (_sz, [_i, _i]),
    "mfr_gemm_f16x2_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_gemm_f16x2": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mfr_gemm_f16x2_pack_batched": (_i, [_vp, _i, _i, C.c_longlong, _i, _i, C.c_float, _vp, _vp]),
    "mfr_gemm_f16x2_batched": (_i, [_vp, _i, C.c_longlong, _vp, _vp, _vp, _i, C.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "mfr_conv_igemm_k": (_i, [_i, _i, _i]),
    "mfr_conv_igemm_f16x2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mfr_conv_igemm_f16x2_upadd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mfr_gemm_f16x2_windows": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mfr_gemm_bf16x3_windows": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mfr_mlp_ln_f16x2": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "mfr_mlp_ln_bf16x3": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "mfr_gemm_f16x2_ln": (_i, [_vp, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _i, _i, _vp]),
    "mfr_gemm_bf16x3_ln": (_i, [_vp, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _i, _i, _i, _i, _i, _vp]),
    "mfr_gemm_bf16x3_pack_bytes": (_sz, [_i, _i]),
    "mfr_gemm_bf16x3_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_gemm_bf16x3": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mfr_sg_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "mfr_sg_attention_variant": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "mfr_sg_match_workspace_bytes": (_sz, [_i, _i]),
    "mfr_sg_sinkhorn_match": (_i, [_vp, _i, _i, _vp, _vp, C.c_float, _i, C.c_float, _vp, _vp, _i, _vp, _sz,
                                   _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mfr_sg_sinkhorn_match_variant": (_i, [_vp, _i, _i, _vp, _vp, C.c_float, _i, C.c_float, _vp, _vp, _i, _vp, _sz,
                                           _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "mfr_loftr_linear_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_loftr_linear_attention": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp]),
    "mfr_loftr_coarse_match_workspace_bytes": (_sz, [_i, _i, _i]),
    "mfr_loftr_coarse_match": (_i, [_vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "mfr_loftr_coarse_match_variant": (_i, [_vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, _i, _vp, _sz, _vp, _vp, _vp, _vp, _i, _vp]),
    "mfr_loftr_fine_attention": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mfr_loftr_gather_windows": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mfr_loftr_fine_match": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "mfr_conv3x3_c1_relu": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mfr_bias_relu_nchw": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mfr_conv1x1_nchw": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mfr_nchw_to_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, C.c_longlong, _i, _vp]),
    "mfr_bias_pool2_relu_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mfr_wino_bf16x3_filter_bytes": (_sz, [_i, _i]),
    "mfr_wino_bf16x3_filter_transform": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_conv3x3_wino_bf16x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_sp_conv1ab_f16x2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mfr_wino_f16x2_filter_bytes": (_sz, [_i, _i]),
    "mfr_wino_f16x2_filter_transform": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_conv3x3_wino_f16x2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_conv3x3_direct_f16x2_filter_bytes": (_sz, [_i, _i]),
    "mfr_conv3x3_direct_f16x2_filter_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_conv3x3_direct_f16x2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_conv3x3s2_direct_f16x2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_conv3x3_direct_f16x2_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mfr_wino_filter_bytes": (_sz, [_i, _i]),
    "mfr_wino_filter_transform": (_i, [_vp, _i, _i, _vp, _vp]),
    "mfr_conv3x3_wino": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_conv3x3_wino_variant": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, C.c_longlong, _i, C.c_float, _vp, _i, _vp]),
    "mfr_conv_gemm_bf16": (_i, [_vp, C.c_longlong, _vp, _vp, C.c_longlong, _vp, _i, _i, _i, _vp, _vp, C.c_longlong, _i,
                                _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mfr_conv_pack_nhwc_halo": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mfr_conv_unpack_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mfr_conv_pack_cm_halo": (_i, [_vp, _i, C.c_longlong, _i, _i, _i, C.c_longlong, _i, _i, C.c_longlong, _vp, _vp]),
    "mfr_upsample2x_add": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mfr_upsample_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mfr_corr_warp_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_corr_warp_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfr_kabsch_fwd": (_i, [_vp, _i, _vp, _vp]),
    "mfr_kabsch_bwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mfr_rootsift": (_i, [_vp, _i, _vp, _vp, _vp]),
    "mfr_desc_ratio_match": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _d, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mfr_scale_workspace_bytes": (_sz, [_i, _i]),
    "mfr_scale_from_depth_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                        _d, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
}


class MfrLibraryError(RuntimeError):
    pass


def load(require_gpu=False):
    """Load libmfr_hip.so and bind every symbol of the C-ABI.  Raises MfrLibraryError if the
    library (or, with require_gpu, a GPU) is missing -- never falls back."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (binds the HIP runtime first)
        if not os.path.exists(SO_PATH):
            raise MfrLibraryError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MfrLibraryError(f"libmfr_hip.so does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_gpu:
        import torch
        if not torch.cuda.is_available():
            raise MfrLibraryError("no HIP device visible: the HIP path cannot run (there is no CPU fallback)")
    return _lib


def check(rc, what):
    if rc != 0:
        raise MfrLibraryError(f"{what} failed with code {rc} (see include/mfr_hip.h MFR_E_*)")


def ptr(t):
    """device (or host) pointer of a contiguous torch tensor, or None"""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI needs contiguous buffers"
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
