"""Ray generation of the data path: `get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1)` with the argument
meaning and the result dictionary of the reference (nerf/utils.py:53-137: 'rays_o', 'rays_d' [B,N,3], 'inds' [B,N] when N > 0,
'inds_coarse' with an error map).  Which pixels are drawn is decided with the same torch calls as there (uniform with replacement,
patch corners, or a multinomial draw on the 128 x 128 error map refined by a uniform offset); turning pixels into rays -- pinhole
direction, normalisation, camera-to-world rotation, origin broadcast: eight elementwise/bmm launches in the reference -- is one HIP
kernel (ngp_rays_from_pixels) that writes the [B,N,3] outputs directly, optionally into caller-provided buffers (`out=`), e.g. the static
input buffers of graph.GraphedTrainStep.

SURVEY.md 8(f).4.  Fails loudly without the HIP library: there is no CPU fallback."""
import torch

import _ngp_capi as capi


def custom_meshgrid(*args):
    return torch.meshgrid(*args, indexing='ij')


def _draw_pixels(B, H, W, N, error_map, patch_size, device):
    """-> (inds [B,N] or [N] int64, inds_coarse or None); the three sampling modes of the reference, same distributions"""
    if patch_size > 1:  # square patches from random top-left corners (the error map is ignored, as in the reference)
        n_patches = N // (patch_size ** 2)
        top = torch.randint(0, H - patch_size, size=[n_patches], device=device)
        left = torch.randint(0, W - patch_size, size=[n_patches], device=device)
        dy, dx = custom_meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device))
        rows = top[:, None] + dy.reshape(1, -1)
        cols = left[:, None] + dx.reshape(1, -1)
        return (rows * W + cols).reshape(-1), None
    if error_map is None:
        return torch.randint(0, H * W, size=[N], device=device), None
    coarse = torch.multinomial(error_map.to(device), N, replacement=False)  # [B,N] cells of the 128 x 128 map
    cell_h, cell_w = H / 128, W / 128
    rows = ((coarse // 128) * cell_h + torch.rand(B, N, device=device) * cell_h).long().clamp(max=H - 1)
    cols = ((coarse % 128) * cell_w + torch.rand(B, N, device=device) * cell_w).long().clamp(max=W - 1)
    return rows * W + cols, coarse


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1, out=None):
    """poses [B,4,4] cam2world (CUDA, fp32), intrinsics (fx, fy, cx, cy).  `out=(rays_o, rays_d)`: optional contiguous fp32 [B,n,3]
    tensors to fill (extension)."""
    if not poses.is_cuda:
        raise RuntimeError('get_rays: poses must be a CUDA tensor (the MI355X build has no CPU path)')
    poses = poses.float().contiguous()
    B = poses.shape[0]
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    results = {}
    if N > 0:
        N = min(N, H * W)
        inds, coarse = _draw_pixels(B, H, W, N, error_map, patch_size, poses.device)
        n = inds.shape[-1]
        shared = inds.dim() == 1
        inds = inds.contiguous()
        results['inds'] = inds.expand(B, n) if shared else inds
        if coarse is not None:
            results['inds_coarse'] = coarse
        inds_ptr, stride = inds.data_ptr(), (0 if shared else n)
    else:
        n, inds_ptr, stride = H * W, None, 0
    if out is None:
        rays_o = torch.empty(B, n, 3, device=poses.device, dtype=torch.float32)
        rays_d = torch.empty(B, n, 3, device=poses.device, dtype=torch.float32)
    else:
        rays_o, rays_d = out
        for t in (rays_o, rays_d):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == B * n * 3):
                raise RuntimeError('get_rays: out tensors must be contiguous float32 CUDA tensors of B*N*3 elements')
    capi.check(capi.lib.ngp_rays_from_pixels(poses.data_ptr(), B, fx, fy, cx, cy, W, inds_ptr, stride, n, rays_o.data_ptr(), rays_d.data_ptr(),
                                             capi.stream()))
    results['rays_o'] = rays_o.view(B, n, 3)
    results['rays_d'] = rays_d.view(B, n, 3)
    return results
