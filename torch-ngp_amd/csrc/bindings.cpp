// Compiled Python bindings over the C ABI of libngp_hip.so -- the modules the reference imports FIRST:
//   `import _gridencoder as _backend`  (gridencoder/grid.py:9-12, table gridencoder/src/bindings.cpp:5-9)
//   `import _shencoder  as _backend`   (shencoder/sphere_harmonics.py:8-11, shencoder/src/bindings.cpp:5-8)
//   `import _raymarching as _backend`  (raymarching/raymarching.py:9-12, raymarching/src/bindings.cpp:5-19)
//   `import _ffmlp as _backend`        (ffmlp/ffmlp.py:9-12, ffmlp/src/bindings.cpp:5-11)
//   `import _freqencoder as _backend`  (freqencoder/freq.py:9-12, freqencoder/src/bindings.cpp:5-8)
// Same callable names, positional arguments and error behaviour (TORCH_CHECK -> RuntimeError) as those tables; at::Tensor arguments are
// turned into raw device pointers, the launch goes to PyTorch's CURRENT HIP stream (so it orders with the surrounding PyTorch work and is
// captured by HIP graphs), a non-zero return code becomes std::runtime_error(ngp_last_error()).  Plain host C++ (g++): no device code
// here, every kernel lives behind include/ngp_hip.h.  The ctypes `_backend` objects (torch-ngp_amd/*/backend.py) are the same calls
// from Python; this file removes their per-call marshalling cost from the drop-in path.
// One translation unit, five PYBIND11_MODULEs; __graft_entry__.build() links it once and installs it under the five module names.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/ngp_hip.h"

namespace {

#define CHECK_CUDA(x) TORCH_CHECK(x.device().is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be a contiguous tensor")
#define CHECK_IS_INT(x) TORCH_CHECK(x.scalar_type() == at::ScalarType::Int, #x " must be an int tensor")
#define CHECK_IS_FLOATING(x)                                                                                                                 \
    TORCH_CHECK(x.scalar_type() == at::ScalarType::Float || x.scalar_type() == at::ScalarType::Half || x.scalar_type() == at::ScalarType::Double, \
                #x " must be a floating tensor")
#define CHECK_IS_HALF(x) TORCH_CHECK(x.scalar_type() == at::ScalarType::Half, #x " must be a Half tensor")
#define CHECK_DENSE(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)
#define CHECK_F32(x)   \
    CHECK_DENSE(x);    \
    TORCH_CHECK(x.scalar_type() == at::ScalarType::Float, #x " must be a float32 tensor (the reference wrappers cast with custom_fwd(cast_inputs=float32))")
#define CHECK_I32(x)   \
    CHECK_DENSE(x);    \
    CHECK_IS_INT(x)

using at::Tensor;
using OptTensor = at::optional<at::Tensor>;

inline ngp_stream_t stream() { return reinterpret_cast<ngp_stream_t>(c10::hip::getCurrentHIPStream().stream()); }
inline void* ptr(const Tensor& t) { return t.data_ptr(); }
inline void* ptr(const OptTensor& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }
inline void check(int rc) {
    if (rc != NGP_OK) throw std::runtime_error(ngp_last_error());
}
inline int float_code(const Tensor& t, const char* name) {
    if (t.scalar_type() == at::ScalarType::Half) return NGP_F16;
    if (t.scalar_type() == at::ScalarType::Float) return NGP_F32;
    TORCH_CHECK(t.scalar_type() != at::ScalarType::Double, name, ": float64 is not supported by the MI355X kernels (use float32 or float16)");
    TORCH_CHECK(false, name, " must be a floating tensor");
    return -1;
}
inline Tensor scratch(size_t bytes, const Tensor& like) {
    return at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
}

// HOST copy of a grid encoder's `offsets` (steers the plan of the record-sort backward): read back once per tensor, never during stream
// capture (the workspace-free atomic path serves that call instead).  An entry is valid only for the SAME, still living tensor object
// (weak reference to its TensorImpl) at the same version: an address handed out again by the caching allocator is a miss.
const int32_t* host_offsets(const Tensor& offsets) {
    struct Entry {
        c10::weak_intrusive_ptr<c10::TensorImpl> owner;
        uint32_t version;
        std::vector<int32_t> v;
    };
    static std::mutex mu;
    static std::unordered_map<const c10::TensorImpl*, Entry> cache;
    std::lock_guard<std::mutex> lock(mu);
    const c10::TensorImpl* impl = offsets.unsafeGetTensorImpl();
    auto it = cache.find(impl);
    if (it != cache.end()) {
        auto alive = it->second.owner.lock();
        if (alive && alive.get() == impl && it->second.version == offsets._version() && (int64_t)it->second.v.size() == offsets.numel())
            return it->second.v.data();
        cache.erase(it);
    }
    if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return nullptr;
    Tensor h = offsets.to(at::kCPU).contiguous();
    if (cache.size() > 64) {   // drop entries whose tensors are gone
        for (auto e = cache.begin(); e != cache.end();) e = e->second.owner.expired() ? cache.erase(e) : std::next(e);
        if (cache.size() > 64) cache.clear();
    }
    Entry e{c10::weak_intrusive_ptr<c10::TensorImpl>(offsets.getIntrusivePtr()), offsets._version(),
            std::vector<int32_t>(h.data_ptr<int32_t>(), h.data_ptr<int32_t>() + h.numel())};
    return cache.insert_or_assign(impl, std::move(e)).first->second.v.data();
}

// ---- gridencoder (gridencoder.h:12-15) ----
void grid_common(const Tensor& inputs, const Tensor& embeddings, const Tensor& offsets) {
    CHECK_DENSE(inputs);
    CHECK_DENSE(embeddings);
    CHECK_DENSE(offsets);
    CHECK_IS_INT(offsets);
    // the reference reads inputs through data_ptr<float>() whatever the dispatch type (gridencoder.cu:469)
    TORCH_CHECK(inputs.scalar_type() == at::ScalarType::Float, "expected scalar type Float for inputs but found ", inputs.scalar_type());
}

void grid_encode_forward(const Tensor inputs, const Tensor embeddings, const Tensor offsets, Tensor outputs, const uint32_t B, const uint32_t D,
                         const uint32_t C, const uint32_t L, const float S, const uint32_t H, OptTensor dy_dx, const uint32_t gridtype,
                         const bool align_corners, const uint32_t interp) {
    grid_common(inputs, embeddings, offsets);
    CHECK_DENSE(outputs);
    check(ngp_grid_encode_forward((const float*)ptr(inputs), ptr(embeddings), (const int32_t*)ptr(offsets), ptr(outputs), B, D, C, L, S, H, ptr(dy_dx),
                                  gridtype, align_corners ? 1 : 0, interp, float_code(embeddings, "embeddings"), stream()));
}

void grid_encode_backward(const Tensor grad, const Tensor inputs, const Tensor embeddings, const Tensor offsets, Tensor grad_embeddings,
                          const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L, const float S, const uint32_t H, OptTensor dy_dx,
                          OptTensor grad_inputs, const uint32_t gridtype, const bool align_corners, const uint32_t interp) {
    grid_common(inputs, embeddings, offsets);
    CHECK_DENSE(grad);
    CHECK_DENSE(grad_embeddings);
    const int code = float_code(grad, "grad");
    // large fp16 batches: the atomic-free record sort needs scratch memory the reference signature has no argument for
    const int32_t* host = host_offsets(offsets);
    size_t bytes = host ? ngp_grid_backward_workspace_bytes(host, B, D, C, L, S, H, gridtype, align_corners ? 1 : 0, code) : 0;
    Tensor ws;
    if (bytes) ws = scratch(bytes, grad);
    check(ngp_grid_encode_backward_ws(ptr(grad), (const float*)ptr(inputs), ptr(embeddings), (const int32_t*)ptr(offsets), ptr(grad_embeddings), B, D, C,
                                      L, S, H, ptr(dy_dx), ptr(grad_inputs), gridtype, align_corners ? 1 : 0, interp, code, 0.0f, host,
                                      bytes ? ws.data_ptr() : nullptr, bytes, stream()));
}

void grad_total_variation(const Tensor inputs, const Tensor embeddings, Tensor grad, const Tensor offsets, const float weight, const uint32_t B,
                          const uint32_t D, const uint32_t C, const uint32_t L, const float S, const uint32_t H, const uint32_t gridtype,
                          const bool align_corners) {
    CHECK_DENSE(inputs);
    CHECK_DENSE(embeddings);
    CHECK_DENSE(grad);
    CHECK_DENSE(offsets);
    TORCH_CHECK(inputs.scalar_type() == embeddings.scalar_type() && grad.scalar_type() == embeddings.scalar_type(),
                "grad_total_variation: inputs, embeddings and grad must share one dtype");
    check(ngp_grad_total_variation(ptr(inputs), ptr(embeddings), ptr(grad), (const int32_t*)ptr(offsets), weight, B, D, C, L, S, H, gridtype,
                                   align_corners ? 1 : 0, float_code(embeddings, "embeddings"), stream()));
}

void grid_corner_indices(const Tensor inputs, const Tensor offsets, Tensor indices, const uint32_t B, const uint32_t D, const uint32_t L, const float S,
                         const uint32_t H, const uint32_t gridtype, const bool align_corners) {
    check(ngp_grid_corner_indices((const float*)ptr(inputs), (const int32_t*)ptr(offsets), (uint32_t*)ptr(indices), B, D, L, S, H, gridtype,
                                  align_corners ? 1 : 0, stream()));
}

// ---- shencoder (shencoder.h:9-10) ----
void sh_encode_forward(Tensor inputs, Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t C, OptTensor dy_dx) {
    CHECK_DENSE(inputs);
    CHECK_DENSE(outputs);
    CHECK_IS_FLOATING(inputs);
    check(ngp_sh_encode_forward(ptr(inputs), ptr(outputs), B, D, C, ptr(dy_dx), float_code(inputs, "inputs"), stream()));
}

void sh_encode_backward(Tensor grad, Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t C, Tensor dy_dx, Tensor grad_inputs) {
    CHECK_DENSE(grad);
    CHECK_DENSE(inputs);
    CHECK_DENSE(dy_dx);
    CHECK_DENSE(grad_inputs);
    CHECK_IS_FLOATING(grad);
    check(ngp_sh_encode_backward(ptr(grad), ptr(inputs), B, D, C, ptr(dy_dx), ptr(grad_inputs), float_code(grad, "grad"), stream()));
}

// ---- freqencoder (freqencoder.h:9-13): fp32 only, as the reference (data_ptr<float>() on every tensor) ----
void freq_encode_forward(Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C, Tensor outputs) {
    CHECK_DENSE(inputs);
    CHECK_DENSE(outputs);
    TORCH_CHECK(inputs.scalar_type() == at::ScalarType::Float && outputs.scalar_type() == at::ScalarType::Float, "expected scalar type Float");
    check(ngp_freq_encode_forward((const float*)ptr(inputs), B, D, deg, C, (float*)ptr(outputs), stream()));
}

void freq_encode_backward(Tensor grad, Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C, Tensor grad_inputs) {
    CHECK_DENSE(grad);
    CHECK_DENSE(outputs);
    CHECK_DENSE(grad_inputs);
    TORCH_CHECK(grad.scalar_type() == at::ScalarType::Float && outputs.scalar_type() == at::ScalarType::Float &&
                    grad_inputs.scalar_type() == at::ScalarType::Float,
                "expected scalar type Float");
    check(ngp_freq_encode_backward((const float*)ptr(grad), (const float*)ptr(outputs), B, D, deg, C, (float*)ptr(grad_inputs), stream()));
}

// ---- raymarching (raymarching.h:7-18) ----
// The reference dispatches these on the tensor dtype (AT_DISPATCH_FLOATING_TYPES_AND_HALF, raymarching.cu:486); its own wrappers always hand
// over fp32 (custom_fwd(cast_inputs=float32)).  For direct callers with fp16 tensors: the same fp32 kernels run on fp32 copies and every
// floating argument is copied back (outputs are caller-allocated arguments): fp32 arithmetic rounded once to fp16.
struct F32View {
    Tensor orig, f32;
    bool is_output;
    F32View(const Tensor& t, bool out) : orig(t), f32(t.scalar_type() == at::ScalarType::Half ? t.to(at::kFloat) : t), is_output(out) {}
    F32View(const F32View&) = delete;
    F32View& operator=(const F32View&) = delete;
    float* p() const { return (float*)f32.data_ptr(); }
    // copy an OUTPUT back into the caller's fp16 tensor -- called explicitly after the kernel was issued (not from a destructor,
    // where an exception terminates the process, and never for inputs: that was a wasted round trip and bumped the version counter of
    // tensors a caller may have saved for backward)
    void finish() {
        if (is_output && orig.scalar_type() == at::ScalarType::Half) orig.copy_(f32);
    }
};
#define F32ANY(x, OUT)    \
    CHECK_DENSE(x);  \
    CHECK_IS_FLOATING(x); \
    F32View x##_v(x, OUT); \
    TORCH_CHECK(x##_v.f32.scalar_type() == at::ScalarType::Float, #x " must be a float32 tensor (the reference wrappers cast with custom_fwd(cast_inputs=float32))")
#define F32ARG(x) F32ANY(x, false)   // input
#define F32OUT(x) F32ANY(x, true)    // output or in/out: finished with F32DONE after the call
#define F32DONE(x) x##_v.finish()

void near_far_from_aabb(const Tensor rays_o, const Tensor rays_d, const Tensor aabb, const uint32_t N, const float min_near, Tensor nears, Tensor fars) {
    F32ARG(rays_o); F32ARG(rays_d); F32ARG(aabb); F32OUT(nears); F32OUT(fars);
    check(ngp_near_far_from_aabb(rays_o_v.p(), rays_d_v.p(), aabb_v.p(), N, min_near, nears_v.p(), fars_v.p(), stream()));
    F32DONE(nears); F32DONE(fars);
}

void sph_from_ray(const Tensor rays_o, const Tensor rays_d, const float radius, const uint32_t N, Tensor coords) {
    F32ARG(rays_o); F32ARG(rays_d); F32OUT(coords);
    check(ngp_sph_from_ray(rays_o_v.p(), rays_d_v.p(), radius, N, coords_v.p(), stream()));
    F32DONE(coords);
}

void morton3D(const Tensor coords, const uint32_t N, Tensor indices) {
    CHECK_I32(coords); CHECK_I32(indices);
    check(ngp_morton3D((const int32_t*)ptr(coords), N, (int32_t*)ptr(indices), stream()));
}

void morton3D_invert(const Tensor indices, const uint32_t N, Tensor coords) {
    CHECK_I32(indices); CHECK_I32(coords);
    check(ngp_morton3D_invert((const int32_t*)ptr(indices), N, (int32_t*)ptr(coords), stream()));
}

void packbits(const Tensor grid, const uint32_t N, const float density_thresh, Tensor bitfield) {
    F32ARG(grid);
    CHECK_DENSE(bitfield);
    TORCH_CHECK(bitfield.scalar_type() == at::ScalarType::Byte, "bitfield must be a uint8 tensor");
    check(ngp_packbits(grid_v.p(), N, density_thresh, (uint8_t*)ptr(bitfield), stream()));
}

void packbits_capped(const Tensor grid, const uint32_t N, const float density_thresh, const Tensor thresh_cap, Tensor bitfield) {
    check(ngp_packbits_ex((const float*)ptr(grid), N, density_thresh, (const float*)ptr(thresh_cap), (uint8_t*)ptr(bitfield), stream()));
}

void density_grid_update(const Tensor sigmas, const Tensor cells, const uint32_t n, const float density_scale, const float decay, Tensor density_grid,
                         const uint32_t n_cells, Tensor scratch, const float density_thresh, Tensor mean_out, Tensor bitfield, Tensor workspace) {
    CHECK_F32(sigmas); CHECK_F32(density_grid); CHECK_F32(scratch); CHECK_F32(mean_out);
    CHECK_DENSE(cells);
    TORCH_CHECK(cells.scalar_type() == at::ScalarType::Long, "cells must be an int64 tensor");
    check(ngp_density_grid_update((const float*)ptr(sigmas), (const int64_t*)ptr(cells), n, density_scale, decay, (float*)ptr(density_grid), n_cells,
                                  (float*)ptr(scratch), density_thresh, (float*)ptr(mean_out), (uint8_t*)ptr(bitfield), ptr(workspace), stream()));
}

size_t density_grid_update_workspace_bytes(const uint32_t n_cells) { return ngp_density_grid_update_workspace_bytes(n_cells); }

void march_rays_train(const Tensor rays_o, const Tensor rays_d, const Tensor grid, const float bound, const float dt_gamma, const uint32_t max_steps,
                      const uint32_t N, const uint32_t C, const uint32_t H, const uint32_t M, const Tensor nears, const Tensor fars, Tensor xyzs,
                      Tensor dirs, Tensor deltas, Tensor rays, Tensor counter, Tensor noises) {
    F32ARG(rays_o); F32ARG(rays_d); F32ARG(nears); F32ARG(fars); F32OUT(xyzs); F32OUT(dirs); F32OUT(deltas); F32ARG(noises);
    CHECK_I32(rays); CHECK_I32(counter);
    CHECK_DENSE(grid);
    Tensor ws = scratch(ngp_march_rays_train_workspace_bytes(N), rays);
    check(ngp_march_rays_train(rays_o_v.p(), rays_d_v.p(), (const uint8_t*)ptr(grid), bound, dt_gamma, max_steps, N, C, H, M, nears_v.p(), fars_v.p(),
                               xyzs_v.p(), dirs_v.p(), deltas_v.p(), (int32_t*)ptr(rays), (int32_t*)ptr(counter), noises_v.p(), ws.data_ptr(), stream()));
    F32DONE(xyzs); F32DONE(dirs); F32DONE(deltas);
}

void composite_rays_train_forward(const Tensor sigmas, const Tensor rgbs, const Tensor deltas, const Tensor rays, const uint32_t M, const uint32_t N,
                                  const float T_thresh, Tensor weights_sum, Tensor depth, Tensor image) {
    F32ARG(sigmas); F32ARG(rgbs); F32ARG(deltas); F32OUT(weights_sum); F32OUT(depth); F32OUT(image);
    CHECK_I32(rays);
    check(ngp_composite_rays_train_forward(sigmas_v.p(), rgbs_v.p(), deltas_v.p(), (const int32_t*)ptr(rays), M, N, T_thresh, weights_sum_v.p(),
                                           depth_v.p(), image_v.p(), stream()));
    F32DONE(weights_sum); F32DONE(depth); F32DONE(image);
}

void composite_rays_train_backward(const Tensor grad_weights_sum, const Tensor grad_image, const Tensor sigmas, const Tensor rgbs, const Tensor deltas,
                                   const Tensor rays, const Tensor weights_sum, const Tensor image, const uint32_t M, const uint32_t N,
                                   const float T_thresh, Tensor grad_sigmas, Tensor grad_rgbs) {
    F32ARG(grad_weights_sum); F32ARG(grad_image); F32ARG(sigmas); F32ARG(rgbs); F32ARG(deltas); F32ARG(weights_sum); F32ARG(image);
    F32OUT(grad_sigmas); F32OUT(grad_rgbs);
    CHECK_I32(rays);
    check(ngp_composite_rays_train_backward(grad_weights_sum_v.p(), grad_image_v.p(), sigmas_v.p(), rgbs_v.p(), deltas_v.p(), (const int32_t*)ptr(rays),
                                            weights_sum_v.p(), image_v.p(), M, N, T_thresh, grad_sigmas_v.p(), grad_rgbs_v.p(), stream()));
    F32DONE(grad_sigmas); F32DONE(grad_rgbs);
}

void march_rays(const uint32_t n_alive, const uint32_t n_step, const Tensor rays_alive, const Tensor rays_t, const Tensor rays_o, const Tensor rays_d,
                const float bound, const float dt_gamma, const uint32_t max_steps, const uint32_t C, const uint32_t H, const Tensor grid,
                const Tensor nears, const Tensor fars, Tensor xyzs, Tensor dirs, Tensor deltas, Tensor noises) {
    F32ARG(rays_t); F32ARG(rays_o); F32ARG(rays_d); F32ARG(nears); F32ARG(fars); F32OUT(xyzs); F32OUT(dirs); F32OUT(deltas); F32ARG(noises);
    CHECK_I32(rays_alive);
    CHECK_DENSE(grid);
    check(ngp_march_rays(n_alive, n_step, (const int32_t*)ptr(rays_alive), rays_t_v.p(), rays_o_v.p(), rays_d_v.p(), bound, dt_gamma, max_steps, C, H,
                         (const uint8_t*)ptr(grid), nears_v.p(), fars_v.p(), xyzs_v.p(), dirs_v.p(), deltas_v.p(), noises_v.p(), stream()));
    F32DONE(xyzs); F32DONE(dirs); F32DONE(deltas);
}

void march_rays_ex(const uint32_t n_alive, const uint32_t n_step, const Tensor rays_alive, const Tensor rays_t, const Tensor rays_o, const Tensor rays_d,
                   const float bound, const float dt_gamma, const uint32_t max_steps, const uint32_t C, const uint32_t H, const Tensor grid,
                   const Tensor nears, const Tensor fars, Tensor xyzs, Tensor dirs, Tensor deltas, OptTensor noises, const uint32_t zero_rows) {
    CHECK_F32(rays_t); CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(nears); CHECK_F32(fars); CHECK_F32(xyzs); CHECK_F32(dirs); CHECK_F32(deltas);
    CHECK_I32(rays_alive);
    CHECK_DENSE(grid);
    check(ngp_march_rays_ex(n_alive, n_step, (const int32_t*)ptr(rays_alive), (const float*)ptr(rays_t), (const float*)ptr(rays_o), (const float*)ptr(rays_d),
                            bound, dt_gamma, max_steps, C, H, (const uint8_t*)ptr(grid), (const float*)ptr(nears), (const float*)ptr(fars),
                            (float*)ptr(xyzs), (float*)ptr(dirs), (float*)ptr(deltas), (const float*)ptr(noises), zero_rows, stream()));
}

void composite_rays(const uint32_t n_alive, const uint32_t n_step, const float T_thresh, Tensor rays_alive, Tensor rays_t, Tensor sigmas, Tensor rgbs,
                    Tensor deltas, Tensor weights_sum, Tensor depth, Tensor image) {
    F32OUT(rays_t); F32ARG(sigmas); F32ARG(rgbs); F32ARG(deltas); F32OUT(weights_sum); F32OUT(depth); F32OUT(image);
    CHECK_I32(rays_alive);
    check(ngp_composite_rays(n_alive, n_step, T_thresh, (int32_t*)ptr(rays_alive), rays_t_v.p(), sigmas_v.p(), rgbs_v.p(), deltas_v.p(), weights_sum_v.p(),
                             depth_v.p(), image_v.p(), stream()));
    F32DONE(rays_t); F32DONE(weights_sum); F32DONE(depth); F32DONE(image);
}

void compact_rays(const Tensor rays_alive, const uint32_t n_alive, Tensor out_alive, Tensor out_count) {
    CHECK_I32(rays_alive); CHECK_I32(out_alive); CHECK_I32(out_count);
    Tensor ws = scratch(ngp_compact_rays_workspace_bytes(n_alive), rays_alive);
    check(ngp_compact_rays((const int32_t*)ptr(rays_alive), n_alive, (int32_t*)ptr(out_alive), (int32_t*)ptr(out_count), ws.data_ptr(), stream()));
}

// empty-ray culling of the inference loop (extension): a dilated (H/4)^3 occupancy per cascade, and the rays whose [near, far] segment
// provably meets no occupied voxel get -1 in the initial alive list
void coarse_occupancy(const Tensor grid, const uint32_t C, const uint32_t H, Tensor coarse) {
    CHECK_DENSE(grid); CHECK_DENSE(coarse);
    TORCH_CHECK((size_t)coarse.numel() * coarse.element_size() >= ngp_coarse_occupancy_bytes(C, H), "coarse_occupancy: `coarse` is too small");
    check(ngp_coarse_occupancy((const uint8_t*)ptr(grid), C, H, (uint8_t*)ptr(coarse), stream()));
}

void cull_rays(const Tensor rays_o, const Tensor rays_d, const Tensor nears, const Tensor fars, const uint32_t N, const float bound, const uint32_t C,
               const uint32_t H, const Tensor coarse, Tensor rays_alive) {
    CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(nears); CHECK_F32(fars); CHECK_DENSE(coarse); CHECK_I32(rays_alive);
    check(ngp_cull_rays((const float*)ptr(rays_o), (const float*)ptr(rays_d), (const float*)ptr(nears), (const float*)ptr(fars), N, bound, C, H,
                        (const uint8_t*)ptr(coarse), (int32_t*)ptr(rays_alive), stream()));
}

void march_rays_dev(const Tensor state, const uint32_t alive_bound, const uint32_t n_total, const uint32_t n_step_cap, const Tensor rays_alive, const Tensor rays_t, const Tensor rays_o,
                    const Tensor rays_d, const float bound, const float dt_gamma, const uint32_t max_steps, const uint32_t C, const uint32_t H,
                    const Tensor grid, const Tensor nears, const Tensor fars, Tensor xyzs, Tensor dirs, Tensor deltas, OptTensor noises, const uint32_t rows) {
    CHECK_F32(rays_t); CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(nears); CHECK_F32(fars); CHECK_F32(xyzs); CHECK_F32(dirs); CHECK_F32(deltas);
    CHECK_I32(rays_alive); CHECK_I32(state);
    CHECK_DENSE(grid);
    check(ngp_march_rays_dev((const int32_t*)ptr(state), alive_bound, n_total, n_step_cap, (const int32_t*)ptr(rays_alive), (const float*)ptr(rays_t),
                             (const float*)ptr(rays_o), (const float*)ptr(rays_d), bound, dt_gamma, max_steps, C, H, (const uint8_t*)ptr(grid),
                             (const float*)ptr(nears), (const float*)ptr(fars), (float*)ptr(xyzs), (float*)ptr(dirs), (float*)ptr(deltas),
                             (const float*)ptr(noises), rows, stream()));
}

void composite_rays_dev(const Tensor state, const uint32_t alive_bound, const uint32_t n_total, const uint32_t n_step_cap, const float T_thresh, Tensor rays_alive, Tensor rays_t,
                        const Tensor sigmas, const Tensor rgbs, const Tensor deltas, Tensor weights_sum, Tensor depth, Tensor image) {
    CHECK_F32(rays_t); CHECK_F32(sigmas); CHECK_F32(rgbs); CHECK_F32(deltas); CHECK_F32(weights_sum); CHECK_F32(depth); CHECK_F32(image);
    CHECK_I32(rays_alive); CHECK_I32(state);
    check(ngp_composite_rays_dev((const int32_t*)ptr(state), alive_bound, n_total, n_step_cap, T_thresh, (int32_t*)ptr(rays_alive), (float*)ptr(rays_t),
                                 (const float*)ptr(sigmas), (const float*)ptr(rgbs), (const float*)ptr(deltas), (float*)ptr(weights_sum),
                                 (float*)ptr(depth), (float*)ptr(image), stream()));
}

void compact_rays_dev(const Tensor state, const uint32_t alive_bound, const uint32_t n_total, const uint32_t n_step_cap, const uint32_t max_steps, const Tensor rays_alive,
                      Tensor out_alive, Tensor out_state, Tensor workspace) {
    CHECK_I32(rays_alive); CHECK_I32(out_alive); CHECK_I32(out_state); CHECK_I32(state);
    check(ngp_compact_rays_dev((const int32_t*)ptr(state), alive_bound, n_total, n_step_cap, max_steps, (const int32_t*)ptr(rays_alive), (int32_t*)ptr(out_alive),
                               (int32_t*)ptr(out_state), ptr(workspace), stream()));
}

// ---- ffmlp (ffmlp.h:8-14): Half only (CHECK_IS_HALF, ffmlp.cu:636-642) ----
#define HALF_ARG(x) \
    CHECK_DENSE(x); \
    CHECK_IS_HALF(x)

void ffmlp_forward(const Tensor inputs, const Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim, const uint32_t hidden_dim,
                   const uint32_t num_layers, const uint32_t activation, const uint32_t output_activation, Tensor forward_buffer, Tensor outputs) {
    HALF_ARG(inputs); HALF_ARG(weights); HALF_ARG(forward_buffer); HALF_ARG(outputs);
    check(ngp_ffmlp_forward(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, ptr(forward_buffer),
                            ptr(outputs), stream()));
}

void ffmlp_inference(const Tensor inputs, const Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim,
                     const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation, const uint32_t output_activation,
                     Tensor inference_buffer, Tensor outputs) {
    HALF_ARG(inputs); HALF_ARG(weights); HALF_ARG(outputs);
    check(ngp_ffmlp_inference(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                              ptr(inference_buffer), ptr(outputs), stream()));
}

void ffmlp_backward(const Tensor grad, const Tensor inputs, const Tensor weights, const Tensor forward_buffer, const uint32_t B, const uint32_t input_dim,
                    const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation,
                    const uint32_t output_activation, const bool calc_grad_inputs, Tensor backward_buffer, Tensor grad_inputs, Tensor grad_weights) {
    HALF_ARG(grad); HALF_ARG(inputs); HALF_ARG(weights); HALF_ARG(forward_buffer); HALF_ARG(backward_buffer); HALF_ARG(grad_inputs); HALF_ARG(grad_weights);
    // shapes outside the register-resident kernels split the weight-gradient reduction over sample chunks: scratch the reference signature lacks
    const size_t bytes = ngp_ffmlp_backward_workspace_bytes(B, input_dim, hidden_dim, num_layers);
    Tensor ws;
    if (bytes) ws = scratch(bytes, grad);
    check(ngp_ffmlp_backward_ws(ptr(grad), ptr(inputs), ptr(weights), ptr(forward_buffer), B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                output_activation, calc_grad_inputs ? 1 : 0, ptr(backward_buffer), ptr(grad_inputs), ptr(grad_weights), 0,
                                bytes ? ws.data_ptr() : nullptr, bytes, stream()));
}

void allocate_splitk(size_t size) { check(ngp_allocate_splitk(size)); }
void free_splitk() { check(ngp_free_splitk()); }

}  // namespace

PYBIND11_MODULE(_gridencoder, m) {
    m.def("grid_encode_forward", &grid_encode_forward, "grid_encode_forward (HIP, gfx950)");
    m.def("grid_encode_backward", &grid_encode_backward, "grid_encode_backward (HIP, gfx950)");
    m.def("grad_total_variation", &grad_total_variation, "grad_total_variation (HIP, gfx950)");
    m.def("grid_corner_indices", &grid_corner_indices, "diagnostic extension: corner table indices per point and level");
}

PYBIND11_MODULE(_shencoder, m) {
    m.def("sh_encode_forward", &sh_encode_forward, "SH encode forward (HIP, gfx950)");
    m.def("sh_encode_backward", &sh_encode_backward, "SH encode backward (HIP, gfx950)");
}

PYBIND11_MODULE(_freqencoder, m) {
    m.def("freq_encode_forward", &freq_encode_forward, "freq encode forward (HIP, gfx950)");
    m.def("freq_encode_backward", &freq_encode_backward, "freq encode backward (HIP, gfx950)");
}

PYBIND11_MODULE(_raymarching, m) {
    // utils
    m.def("packbits", &packbits, "packbits (HIP, gfx950)");
    m.def("near_far_from_aabb", &near_far_from_aabb, "near_far_from_aabb (HIP, gfx950)");
    m.def("sph_from_ray", &sph_from_ray, "sph_from_ray (HIP, gfx950)");
    m.def("morton3D", &morton3D, "morton3D (HIP, gfx950)");
    m.def("morton3D_invert", &morton3D_invert, "morton3D_invert (HIP, gfx950)");
    // train
    m.def("march_rays_train", &march_rays_train, "march_rays_train (HIP, gfx950)");
    m.def("composite_rays_train_forward", &composite_rays_train_forward, "composite_rays_train_forward (HIP, gfx950)");
    m.def("composite_rays_train_backward", &composite_rays_train_backward, "composite_rays_train_backward (HIP, gfx950)");
    // infer
    m.def("march_rays", &march_rays, "march rays (HIP, gfx950)");
    m.def("composite_rays", &composite_rays, "composite rays (HIP, gfx950)");
    // extensions used by the mirror's on-device inference loop and occupancy refresh (include/ngp_hip.h)
    m.def("packbits_capped", &packbits_capped, "packbits against min(density_thresh, device scalar)");
    m.def("density_grid_update", &density_grid_update, "occupancy refresh, apply half: scatter / EMA-max / mean / packbits in three launches");
    m.def("density_grid_update_workspace_bytes", &density_grid_update_workspace_bytes, "bytes of its (zero-initialised) workspace");
    m.def("march_rays_ex", &march_rays_ex, "march_rays that zeroes the rows it does not fill");
    m.def("compact_rays", &compact_rays, "order-preserving compaction of rays_alive");
    m.def("coarse_occupancy", &coarse_occupancy, "dilated (H/4)^3 occupancy per cascade (empty-ray culling)");
    m.def("cull_rays", &cull_rays, "rays_alive[n] = n, or -1 for a ray that provably meets no occupied voxel");
    m.def("march_rays_dev", &march_rays_dev, "march_rays with the alive count on the device");
    m.def("composite_rays_dev", &composite_rays_dev, "composite_rays with the alive count on the device");
    m.def("compact_rays_dev", &compact_rays_dev, "compaction with the alive count on the device");
}

PYBIND11_MODULE(_ffmlp, m) {
    m.def("ffmlp_forward", &ffmlp_forward, "ffmlp_forward (HIP, gfx950)");
    m.def("ffmlp_inference", &ffmlp_inference, "ffmlp_inference (HIP, gfx950)");
    m.def("ffmlp_backward", &ffmlp_backward, "ffmlp_backward (HIP, gfx950)");
    m.def("allocate_splitk", &allocate_splitk, "allocate_splitk (no-op: no side streams on this backend)");
    m.def("free_splitk", &free_splitk, "free_splitk (no-op)");
}
