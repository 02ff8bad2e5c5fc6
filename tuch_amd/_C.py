"""ctypes binding of the C ABI in include/tuch_amd.h (libtuch_amd.so).

There is no fallback: if the library is missing or a call fails, this raises.
torch only supplies device memory and the current stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported first so the HIP runtime it ships is the one in the process)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TUCH_AMD_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get('TUCH_AMD_LIB') or os.path.join(_HERE, 'libtuch_amd.so')
_lib = None

# name -> (restype, argtypes); mirrors include/tuch_amd.h
_SIGNATURES = {
    'tuch_last_error': (c_char_p, []),
    'tuch_abi_version': (c_int, []),
    'tuch_batch_pairwise_dist': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tuch_batch_pairwise_dist_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p]),
    'tuch_solid_angles_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'tuch_solid_angles': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tuch_gather_triangles': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tuch_winding_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'tuch_winding_numbers': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                     c_void_p, c_size_t, c_void_p]),
    'tuch_geomask_words': (c_int, [c_int]),
    'tuch_geomask_bits_bytes': (c_size_t, [c_int]),
    'tuch_pack_geomask': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'tuch_v2v_workspace_bytes': (c_size_t, [c_int, c_int]),
    'tuch_v2v_min_masked': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    'tuch_v2v_min_indexed_workspace_bytes': (c_size_t, [c_int, c_int]),
    'tuch_v2v_min_indexed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    'tuch_v2v_min_indexed_mfma_workspace_bytes': (c_size_t, [c_int, c_int]),
    'tuch_v2v_min_indexed_mfma': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    'tuch_contact_terms_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p]),
    'tuch_contact_terms_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p]),
    'tuch_valid_mean_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tuch_valid_mean_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tuch_smpl_backward_split_adam': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                              c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_size_t, c_void_p, c_void_p]),
    'tuch_contact_terms_bwd_fixed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                             c_void_p, c_void_p, c_void_p]),
    'tuch_smplify_small_terms': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_float, c_float, c_float] + [c_void_p] * 5),
    'tuch_smplify_stage1_terms': (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_float, c_float, c_float] + [c_void_p] * 7),
    'tuch_adam_step': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                c_void_p]),
    'tuch_smplify_objective': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    'tuch_smplify_objective_bwd': (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    'tuch_smplify_tail_bwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_float] + [c_void_p] * 6),
    'tuch_contact_model_create': (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p,
                                          c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_void_p, c_void_p,
                                          c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'tuch_contact_model_destroy': (None, [c_void_p]),
    'tuch_contact_model_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'tuch_contact_model_get_option': (c_int, [c_void_p, c_char_p, POINTER(c_int)]),
    'tuch_contact_model_canary_hits': (c_int, [c_void_p, POINTER(c_int), c_int]),
    'tuch_contact_model_canary_selftest': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_contact_model_mask_bits': (c_void_p, [c_void_p]),
    'tuch_contact_model_faces': (c_void_p, [c_void_p]),
    'tuch_contact_model_tickets': (c_void_p, [c_void_p]),
    'tuch_contact_model_tree_mask_bits': (c_void_p, [c_void_p]),
    'tuch_contact_model_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                        POINTER(c_int), POINTER(c_int)]),
    'tuch_contact_model_strips': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_void_p, c_void_p]),
    'tuch_cluster_tree_build': (c_int, [c_int, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    'tuch_cluster_tree_free': (None, [c_void_p]),
    'tuch_cluster_tree_info': (c_int, [c_void_p] + [POINTER(c_int)] * 6),
    'tuch_cluster_tree_export': (c_int, [c_void_p] * 10),
    'tuch_hd_points_fwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    'tuch_hd_points_bwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    'tuch_estimate_translation': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    'tuch_rotmat_to_angle_axis': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tuch_winding_tree_work': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    'tuch_hd_model_create': (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'tuch_hd_model_create_k': (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'tuch_hd_model_destroy': (None, [c_void_p]),
    'tuch_hd_model_info': (c_int, [c_void_p, POINTER(c_int), c_void_p]),
    'tuch_hd_contact_saved_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_hd_contact_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_hd_contact_fwd': (c_int, [c_void_p] * 6 + [c_int, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                    c_void_p]),
    'tuch_hd_contact_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_hd_contact_selection': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'tuch_hd_contact_details': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'tuch_ray_work': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    'tuch_contact_model_tree_order': (c_int, [c_void_p, c_void_p, c_void_p]),
    'tuch_v2v_model_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_v2v_hint_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_v2v_min_model': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_v2v_min_model_shared': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                          c_void_p]),
    'tuch_v2v_min_model_shared_zero': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                               c_void_p, c_size_t, c_void_p]),
    'tuch_exterior_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_v2v_min_model_can_cap': (c_int, [c_void_p]),
    'tuch_v2v_min_model_capped': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                          c_void_p, c_size_t, c_void_p, c_float, c_void_p]),
    'tuch_v2v_min_model_fix': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    'tuch_exterior_flags': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_winding_points_workspace_bytes': (c_size_t, [c_void_p, c_int, c_int]),
    'tuch_winding_points': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    'tuch_contact_terms_ragged_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                              c_void_p]),
    'tuch_contact_terms_ragged_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                              c_void_p, c_void_p]),
    'tuch_region_pair_min': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'tuch_smpl_model_create': (c_int, [POINTER(c_void_p), c_int] + [c_void_p] * 9),
    'tuch_smpl_model_destroy': (None, [c_void_p]),
    'tuch_smpl_model_info': (c_int, [c_void_p, POINTER(c_int)]),
    'tuch_smpl_forward_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_smpl_backward_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'tuch_smpl_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    'tuch_smpl_backward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_smplify_stage2_fused_scratch_floats': (c_size_t, [c_int]),
    'tuch_smplify_stage2_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                           c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    'tuch_set_deterministic': (None, [c_int]),
    'tuch_get_deterministic': (c_int, []),
    'tuch_region_pair_keys': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'tuch_smplify_stage2_finish': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                           c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    'tuch_smplify_stage2_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                        c_float, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'tuch_smpl_forward_split': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    'tuch_smpl_backward_split': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t,
                                         c_void_p]),
    'tuch_smpl_backward_split_add': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                             c_size_t, c_void_p, c_void_p]),
    'tuch_fixed_to_float': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    'tuch_region_pair_min_bwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libtuch_amd.so is not built (%s). Run `python -m tuch_amd._build` (needs hipcc); '
                'there is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the ABI and the binding drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class TuchError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise TuchError('libtuch_amd error %d: %s' % (rc, lib().tuch_last_error().decode()))


def ptr(t) -> c_void_p:
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise TuchError('tuch_amd kernels need tensors on a HIP device, got %s' % t.device)
    if not t.is_contiguous():
        raise TuchError('tuch_amd kernels need contiguous tensors')
    return c_void_p(t.data_ptr())


def row_ptr(t) -> c_void_p:
    """Device pointer of a 2-D tensor whose rows are contiguous (the row stride goes to the kernel separately)."""
    if not t.is_cuda:
        raise TuchError('tuch_amd kernels need tensors on a HIP device, got %s' % t.device)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise TuchError('tuch_amd kernels need contiguous rows')
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream() -> c_void_p:
    """The current stream of the current device as a hipStream_t.  (torch.cuda.current_stream() builds a Python Stream
    object through several device-index helpers: ~9 us a call, nine calls per eager step.)"""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
