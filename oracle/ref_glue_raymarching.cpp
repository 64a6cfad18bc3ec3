/* ref_glue_raymarching.cpp -- TEST INFRASTRUCTURE.  Appended by oracle/Makefile to the (piped, never stored) text of the reference's
 * raymarching/src/raymarching.cu: plain-pointer C entry points over the reference's own launchers (raymarching.h:7-18); all floating
 * tensors fp32 (the reference's wrappers force it, raymarching.py custom_fwd(cast_inputs=float32)). */
#define ORC_EXPORT extern "C" __attribute__((visibility("default")))
#define F32(p) at::Tensor((void*)(p), at::ScalarType::Float)
#define I32(p) at::Tensor((void*)(p), at::ScalarType::Int)
#define U8(p) at::Tensor((void*)(p), at::ScalarType::Byte)
#define ORC_TRY(name, call) try { call; } catch (const std::exception& e) { fprintf(stderr, name ": %s\n", e.what()); return 1; } return 0

ORC_EXPORT int ref_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears, float* fars) {
    ORC_TRY("near_far_from_aabb", near_far_from_aabb(F32(rays_o), F32(rays_d), F32(aabb), N, min_near, F32(nears), F32(fars)));
}
ORC_EXPORT int ref_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    ORC_TRY("sph_from_ray", sph_from_ray(F32(rays_o), F32(rays_d), radius, N, F32(coords)));
}
ORC_EXPORT int ref_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) { ORC_TRY("morton3D", morton3D(I32(coords), N, I32(indices))); }
ORC_EXPORT int ref_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) { ORC_TRY("morton3D_invert", morton3D_invert(I32(indices), N, I32(coords))); }
ORC_EXPORT int ref_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    ORC_TRY("packbits", packbits(F32(grid), N, density_thresh, U8(bitfield)));
}
ORC_EXPORT int ref_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                                    uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                    float* deltas, int32_t* rays, int32_t* counter, const float* noises) {
    ORC_TRY("march_rays_train", march_rays_train(F32(rays_o), F32(rays_d), U8(grid), bound, dt_gamma, max_steps, N, C, H, M, F32(nears), F32(fars),
                                                 F32(xyzs), F32(dirs), F32(deltas), I32(rays), I32(counter), F32(noises)));
}
ORC_EXPORT int ref_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                                float T_thresh, float* weights_sum, float* depth, float* image) {
    ORC_TRY("composite_rays_train_forward", composite_rays_train_forward(F32(sigmas), F32(rgbs), F32(deltas), I32(rays), M, N, T_thresh,
                                                                         F32(weights_sum), F32(depth), F32(image)));
}
ORC_EXPORT int ref_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                                 const float* deltas, const int32_t* rays, const float* weights_sum, const float* image, uint32_t M,
                                                 uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs) {
    ORC_TRY("composite_rays_train_backward", composite_rays_train_backward(F32(grad_weights_sum), F32(grad_image), F32(sigmas), F32(rgbs), F32(deltas),
                                                                           I32(rays), F32(weights_sum), F32(image), M, N, T_thresh, F32(grad_sigmas),
                                                                           F32(grad_rgbs)));
}
ORC_EXPORT int ref_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d,
                              float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                              const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises) {
    ORC_TRY("march_rays", march_rays(n_alive, n_step, I32(rays_alive), F32(rays_t), F32(rays_o), F32(rays_d), bound, dt_gamma, max_steps, C, H, U8(grid),
                                     F32(nears), F32(fars), F32(xyzs), F32(dirs), F32(deltas), F32(noises)));
}
ORC_EXPORT int ref_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t, const float* sigmas,
                                  const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image) {
    ORC_TRY("composite_rays", composite_rays(n_alive, n_step, T_thresh, I32(rays_alive), F32(rays_t), F32(sigmas), F32(rgbs), F32(deltas),
                                             F32(weights_sum), F32(depth), F32(image)));
}
/* the integer / cascade helpers on their own (raymarching.cu:42-81) */
ORC_EXPORT void ref_mip_from_pos(const float* xyz, uint32_t n, float max_cascade, int32_t* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = mip_from_pos(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], max_cascade);
}
ORC_EXPORT void ref_mip_from_dt(const float* dt, uint32_t n, float H, float max_cascade, int32_t* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = mip_from_dt(dt[i], H, max_cascade);
}
ORC_EXPORT void ref_morton_pair(const uint32_t* xyz, uint32_t n, uint32_t* code, uint32_t* back) {
    for (uint32_t i = 0; i < n; i++) {
        code[i] = __morton3D(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
        back[3 * (size_t)i] = __morton3D_invert(code[i]);
        back[3 * (size_t)i + 1] = __morton3D_invert(code[i] >> 1);
        back[3 * (size_t)i + 2] = __morton3D_invert(code[i] >> 2);
    }
}
