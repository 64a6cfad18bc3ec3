"""Host-side mirror of the two reference callers of the hot path: `nerf.renderer.NeRFRenderer` (cuda_ray path)
and `nerf.network_ff.NeRFNetwork`.  The reference's own files run unchanged against the operator packages
(tests/test_dropin_reference.py); these mirrors exist because the reference checkout is not available on the
GPU box and pulls in packages (trimesh, cv2, ...) that are not part of the hot path."""
