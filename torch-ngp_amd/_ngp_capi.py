"""ctypes binding of libngp_hip.so (the C ABI declared in include/ngp_hip.h).

This is the only place the product touches native code.  There is NO fallback: if the shared library
has not been built (``python __graft_entry__.py`` / ``make -C torch-ngp_amd/csrc``) importing any of
the four operator packages raises ImportError, and calling an op without a visible gfx950 device
raises RuntimeError from the HIP runtime.

The helpers below turn ``torch.Tensor`` arguments into raw device pointers, pass the current HIP
stream (so the kernels order with the surrounding PyTorch work and can be captured into HIP graphs)
and convert a non-zero return code into the ``RuntimeError`` the reference's ``TORCH_CHECK`` /
``std::runtime_error`` would have raised (e.g. gridencoder.cu:15-18, 381).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NGP_HIP_LIBRARY: developer override for A/B runs of compile-time variants (tools/build_variant.sh); default = the in-tree build
LIB_PATH = os.environ.get('NGP_HIP_LIBRARY') or os.path.join(_HERE, 'libngp_hip.so')

NGP_F32, NGP_F16 = 0, 1
NGP_FF_INPUT_PLANAR, NGP_FF_DX_PLANAR, NGP_FF_LAYERED, NGP_FF_SINGLE_WAVE, NGP_FF_DEFER_REDUCE, NGP_FF_RECOMPUTE = 1, 2, 4, 8, 16, 32
NGP_MARCH_RESET_COUNTER, NGP_MARCH_ZERO_TAIL, NGP_MARCH_NOISE_FROM_SEED, NGP_MARCH_SCAN_LAUNCH = 1, 2, 4, 8
NGP_OPT_PHASE_CHECK, NGP_OPT_PHASE_UPDATE, NGP_OPT_PHASE_COMMIT, NGP_OPT_PHASE_FLIP = 1, 2, 4, 8
ABI_VERSION = 10

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py, or "
        f"make -C {os.path.join(_HERE, 'csrc')}).  There is no CPU fallback for these operators.")

lib = ctypes.CDLL(LIB_PATH)

_vp, _u32, _f32, _i32, _sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_size_t

_SIGNATURES = {
    'ngp_grid_encode_forward': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _i32, _u32, _i32, _vp],
    'ngp_grid_encode_backward': [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _vp],
    'ngp_grad_total_variation': [_vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _i32, _vp],
    'ngp_grid_corner_indices': [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _vp],
    'ngp_grid_level_table': [_u32, _f32, _u32, _vp, _vp],
    'ngp_sh_encode_forward': [_vp, _vp, _u32, _u32, _u32, _vp, _i32, _vp],
    'ngp_sh_encode_backward': [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _i32, _vp],
    'ngp_freq_encode_forward': [_vp, _u32, _u32, _u32, _u32, _vp, _vp],
    'ngp_freq_encode_backward': [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp],
    'ngp_near_far_from_aabb': [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    'ngp_sph_from_ray': [_vp, _vp, _f32, _u32, _vp, _vp],
    'ngp_morton3D': [_vp, _u32, _vp, _vp],
    'ngp_morton3D_invert': [_vp, _u32, _vp, _vp],
    'ngp_packbits': [_vp, _u32, _f32, _vp, _vp],
    'ngp_packbits_ex': [_vp, _u32, _f32, _vp, _vp, _vp],
    'ngp_march_rays_train': [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'ngp_composite_rays_train_forward': [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp],
    'ngp_composite_rays_train_backward': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp],
    'ngp_march_rays': [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'ngp_march_rays_ex': [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    'ngp_composite_rays': [_u32, _u32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'ngp_compact_rays': [_vp, _u32, _vp, _vp, _vp, _vp],
    'ngp_coarse_occupancy': [_vp, _u32, _u32, _vp, _vp],
    'ngp_cull_rays': [_vp, _vp, _vp, _vp, _u32, _f32, _u32, _u32, _vp, _vp, _vp],
    'ngp_march_rays_dev': [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    'ngp_composite_rays_dev': [_vp, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'ngp_compact_rays_dev': [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp],
    'ngp_render_iterations_dev': [_vp, _u32, _u32, _vp],
    'ngp_ffmlp_forward': [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    'ngp_ffmlp_inference': [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    'ngp_ffmlp_backward': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _vp],
    'ngp_grid_encode_forward_ex': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _i32, _u32, _i32, _f32, _vp],
    'ngp_grid_encode_forward_sched': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _i32, _u32, _i32, _f32, _vp, _vp],
    'ngp_grid_encode_forward_sel': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _u32, _i32, _f32, _vp, _vp],
    'ngp_grid_encode_backward_ex': [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _f32, _vp],
    'ngp_grid_encode_backward_ws': [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _f32, _vp, _vp,
                                    _sz, _vp],
    'ngp_grid_encode_backward_checked': [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _f32, _vp,
                                         _vp, _sz, _vp, _vp],
    # ... + const ngp_slab_sets_t* (SlabSets below, passed with ctypes.byref; None: plain checked backward)
    'ngp_grid_encode_backward_checked_slabs': [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _f32,
                                               _vp, _vp, _sz, _vp, _vp, _vp],
    'ngp_density_grid_update': [_vp, _vp, _u32, _f32, _f32, _vp, _u32, _vp, _f32, _vp, _vp, _vp, _vp],
    'ngp_ffmlp_forward_ex': [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _vp],
    'ngp_ffmlp_inference_ex': [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _vp],
    'ngp_ffmlp_backward_ex': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _u32, _vp],
    'ngp_ffmlp_backward_ws': [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _u32, _vp, _sz, _vp],
    'ngp_network_forward': [_vp, _vp, _u32, _u32, _vp, _vp, _u32, _u32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    'ngp_network_forward_rows': [_vp, _vp, _u32, _u32, _vp, _vp, _u32, _u32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp],
    'ngp_march_rays_dev_rows': [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp],
    'ngp_pipeline_mid_forward': [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp],
    'ngp_pipeline_rgb_forward': [_vp, _vp, _u32, _vp],
    'ngp_pipeline_rgb_backward': [_vp, _vp, _vp, _u32, _vp],
    'ngp_pipeline_mid_backward': [_vp, _vp, _vp, _vp, _u32, _f32, _vp],
    'ngp_pipeline_mse_loss': [_vp, _vp, _u32, _vp, _vp, _vp, _vp],
    'ngp_rays_from_pixels': [_vp, _u32, _f32, _f32, _f32, _f32, _u32, _vp, _u32, _u32, _vp, _vp, _vp],
    'ngp_march_rays_train_ex': [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    'ngp_march_rays_train_aabb': [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32,
                                  _vp],
    'ngp_composite_rays_train_forward_ex': [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    'ngp_composite_rays_train_backward_ex': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _i32, _f32, _vp, _vp, _vp],
    'ngp_composite_train_loss_backward': [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _sz, _vp],
    'ngp_network_backward_color': [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _f32, _vp, _vp, _u32, _vp],
    'ngp_ffmlp_reduce_slabs_pair': [_vp, _u32, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    'ngp_optim_adam_step': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp],
    'ngp_optim_adam_step_ex': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _u32, _vp],
    'ngp_optim_adam_small_commit': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _i32, _vp, _vp,
                                    ctypes.c_uint64, _vp],
    'ngp_optim_ema_update': [_i32, _vp, _vp, _vp, _f32, _vp],
    'ngp_optim_poison_shards': [_vp, _u32, ctypes.c_uint64, _vp, _vp],
    'ngp_optim_shard_verdict': [_vp, _vp, _vp, _u32, ctypes.c_uint64, _vp],
    'ngp_linear_stack_pack': [_vp, _u32, _u32, _u32, _u32, _i32, _vp, _vp],
    'ngp_linear_stack_unpack_grad': [_vp, _u32, _u32, _u32, _u32, _i32, _vp, _vp],
    'ngp_pad_2d_fp16': [_vp, _u32, _u32, _u32, _vp, _u32, _u32, _vp],
    'ngp_allocate_splitk': [_sz],
    'ngp_free_splitk': [],
}
for _name, _args in _SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int
lib.ngp_last_error.restype = ctypes.c_char_p
lib.ngp_target_arch.restype = ctypes.c_char_p
lib.ngp_abi_version.restype = ctypes.c_int
lib.ngp_ffmlp_backward_slab_count.argtypes = [_u32, _u32, _u32, _u32]
lib.ngp_ffmlp_backward_slab_count.restype = _u32
lib.ngp_march_rays_train_workspace_bytes.argtypes = [_u32]
lib.ngp_march_rays_train_workspace_bytes.restype = _sz
lib.ngp_compact_rays_workspace_bytes.argtypes = [_u32]
lib.ngp_ffmlp_backward_workspace_bytes.argtypes = [_u32, _u32, _u32, _u32]
lib.ngp_ffmlp_backward_workspace_bytes.restype = _sz
lib.ngp_grid_backward_workspace_bytes.argtypes = [_vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _i32]
lib.ngp_grid_backward_workspace_bytes.restype = _sz
lib.ngp_compact_rays_workspace_bytes.restype = _sz
lib.ngp_grid_table_adam_prefix.argtypes = [_vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _i32]
lib.ngp_grid_table_adam_prefix.restype = _u32
lib.ngp_coarse_occupancy_bytes.argtypes = [_u32, _u32]
lib.ngp_coarse_occupancy_bytes.restype = _sz
lib.ngp_grid_forward_work_lists.argtypes = [_u32, _u32, _vp, _vp, _vp, _vp]
lib.ngp_grid_forward_work_lists.restype = _u32
lib.ngp_density_grid_update_workspace_bytes.argtypes = [_u32]
lib.ngp_linear_stack_flat_size.argtypes = [_u32, _u32, _u32, _u32, _i32]
lib.ngp_linear_stack_flat_size.restype = _u32
lib.ngp_density_grid_update_workspace_bytes.restype = _sz

if lib.ngp_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH}: ABI version {lib.ngp_abi_version()} != expected {ABI_VERSION}; rebuild the extension")

EXPORTED = sorted(list(_SIGNATURES) + ['ngp_last_error', 'ngp_target_arch', 'ngp_abi_version',
                                       'ngp_march_rays_train_workspace_bytes', 'ngp_compact_rays_workspace_bytes',
                                       'ngp_grid_backward_workspace_bytes', 'ngp_ffmlp_backward_workspace_bytes',
                                       'ngp_ffmlp_backward_slab_count', 'ngp_density_grid_update_workspace_bytes', 'ngp_grid_forward_work_lists', 'ngp_coarse_occupancy_bytes',
                                       'ngp_grid_table_adam_prefix', 'ngp_linear_stack_flat_size'])


def check(rc):
    """Raise RuntimeError with the library's message when a call failed."""
    if rc != 0:
        raise RuntimeError(lib.ngp_last_error().decode('utf-8', 'replace'))


class SlabSets(ctypes.Structure):
    """ngp_slab_sets_t (include/ngp_hip.h): two sets of deferred FFMLP weight-gradient slabs for ngp_grid_encode_backward_checked_slabs"""
    _fields_ = [('slabs_a', ctypes.c_void_p), ('n_slabs_a', ctypes.c_uint32), ('n_params_a', ctypes.c_uint32), ('grad_weights_a', ctypes.c_void_p),
                ('slabs_b', ctypes.c_void_p), ('n_slabs_b', ctypes.c_uint32), ('n_params_b', ctypes.c_uint32), ('grad_weights_b', ctypes.c_void_p),
                ('ray_err', ctypes.c_void_p), ('n_rays', ctypes.c_uint32), ('loss', ctypes.c_void_p), ('overwrite_table', ctypes.c_uint32),
                ('table_adam', ctypes.c_void_p)]


class RenderLoop(ctypes.Structure):
    """ngp_render_loop_t (include/ngp_hip.h): the argument block of ngp_render_iterations_dev"""
    _fields_ = ([('state', ctypes.c_void_p), ('alive', ctypes.c_void_p * 2)] +
                [(n, ctypes.c_void_p) for n in ('rays_t', 'rays_o', 'rays_d', 'nears', 'fars', 'grid', 'noises', 'xyzs', 'dirs', 'deltas', 'enc', 'sigmas',
                                                'rgbs', 'embeddings', 'offsets', 'level_cost_host', 'w_sigma', 'w_color', 'weights_sum', 'depth', 'image',
                                                'compact_workspace', 'rows_used')] +
                [(n, ctypes.c_uint32) for n in ('lanes', 'rows', 'n_total', 'n_step_cap', 'max_steps', 'cascade', 'grid_size', 'L', 'H', 'gridtype',
                                                'interp', 'num_layers_sigma', 'num_layers_color')] +
                [('align_corners', ctypes.c_int32)] + [(n, ctypes.c_float) for n in ('bound', 'dt_gamma', 'T_thresh', 'S', 'density_scale')])


class TableAdam(ctypes.Structure):
    """ngp_table_adam_t (include/ngp_hip.h): the two buffer sets of a table whose Adam sweep rides in the grid backward's accumulate launch"""
    _fields_ = [('param', ctypes.c_void_p * 2), ('exp_avg', ctypes.c_void_p * 2), ('exp_avg_sq', ctypes.c_void_p * 2), ('param_fp16', ctypes.c_void_p * 2),
                ('state', ctypes.c_void_p), ('lr', ctypes.c_float), ('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('eps', ctypes.c_float)]


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # same wording as CHECK_CUDA (ROCm devices are 'cuda' in torch)


def require_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def require_int32(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def float_code(t, name):
    """dtype code of a floating tensor (fp64 is not provided by this library)."""
    if t.dtype == torch.float16:
        return NGP_F16
    if t.dtype == torch.float32:
        return NGP_F32
    if t.dtype == torch.float64:
        raise RuntimeError(f"{name}: float64 is not supported by the MI355X kernels (use float32 or float16)")
    raise RuntimeError(f"{name} must be a floating tensor")


def dense(t, name):
    require_device(t, name)
    require_contiguous(t, name)
    return t


def host_offsets(offsets):
    """HOST copy (ctypes int32 array) of a grid encoder's `offsets` tensor, cached on the tensor object; None when it would have to be
    read back during stream capture (the caller then uses the workspace-free path)"""
    cached = getattr(offsets, '_ngp_host', None)
    if cached is not None and cached[0] == (offsets.data_ptr(), offsets.numel(), offsets._version):
        return cached[1]
    if offsets.is_cuda and torch.cuda.is_current_stream_capturing():
        return None
    vals = [int(v) for v in offsets.detach().cpu().tolist()]
    arr = (ctypes.c_int32 * len(vals))(*vals)
    offsets._ngp_host = ((offsets.data_ptr(), offsets.numel(), offsets._version), arr)
    return arr


def grid_backward_workspace(offsets, B, D, C, L, S, H, gridtype, align_corners, code):
    """(offsets_host, workspace tensor or None, bytes) for ngp_grid_encode_backward_ws"""
    arr = host_offsets(offsets)
    if arr is None:
        return None, None, 0
    n = int(lib.ngp_grid_backward_workspace_bytes(ctypes.cast(arr, ctypes.c_void_p), B, D, C, L, float(S), H, gridtype, int(bool(align_corners)),
                                                  code))
    if n == 0:
        return arr, None, 0
    return arr, torch.empty(n, dtype=torch.uint8, device=offsets.device), n


_LEVEL_COSTS = {}


def ray_level_costs(L, S, H, step_unit, log2_hashmap_size=19, D=3):
    """relative cost per level of gathering ray-ordered samples `step_unit` apart (in the encoder's unit cube) -- for
    ngp_grid_encode_forward_sched.  Fitted to k_grid_forward_fast on MI355X (profiles/r05_grid_forward_levels.txt, one level per XCD, ~8 us of
    launch + ramp subtracted): a DENSE level ((resolution + 1)^D entries fit the table: one lane per point, bound by its instruction stream)
    costs ~10 us whatever its resolution; a HASHED level costs the more the more often consecutive samples change cell -- 17 us (resolution
    81) .. 43.6 us (resolution >= ~700 at a step of 1/590: saturated), i.e. 0.32 + 0.65 * min(1, resolution * step / 1.2) of the saturated
    cost, against 0.24 for a dense level.  (Round 3's kernel: 0.41 + 0.59 * min(...) for every level.)  Returns a c_void_p to a cached host
    float array (kept alive here)."""
    key = (int(L), float(S), int(H), round(float(step_unit), 9), int(log2_hashmap_size), int(D))
    hit = _LEVEL_COSTS.get(key)
    if hit is None:
        scale = (ctypes.c_float * L)()
        res = (ctypes.c_uint32 * L)()
        check(lib.ngp_grid_level_table(L, float(S), H, ctypes.cast(scale, ctypes.c_void_p), ctypes.cast(res, ctypes.c_void_p)))
        dense = [(int(res[l]) + 1) ** int(D) <= (1 << int(log2_hashmap_size)) for l in range(L)]
        arr = (ctypes.c_float * L)(*[0.24 if dense[l] else 0.32 + 0.65 * min(1.0, float(res[l]) * float(step_unit) / 1.2) for l in range(L)])
        hit = (arr, ctypes.cast(arr, ctypes.c_void_p))
        _LEVEL_COSTS[key] = hit
    return hit[1]
