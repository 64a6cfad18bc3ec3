// Glue kernels of the fused sample pipeline: what nerf/network_ff.py:51-123 does between the native
// ops with ~20 small PyTorch kernels (slice, trunc_exp, SH, cat, zero pad, casts, sigmoid and their autograd twins),
// restated as four streaming kernels with the same rounding points as the autocast path:
//   mid forward  : sigma = exp(float(h[:,0]))                       (activation.py:5-17, trunc_exp forward)
//                  color_in = [ half(SH_4(dir)) | h[:,1:16] | 0 ]    (network_ff.py:64-68)
//   rgb forward  : rgb = float(half(sigmoid(float(out[:, :3]))))     (network_ff.py:72: torch.sigmoid on the fp16 output)
//   rgb backward : g_out[:, :3] = half(g_rgb * y * (1 - y)), g_out[:, 3:] = 0
//   mid backward : g_h[:,0] = half(g_sigma * exp(clamp(float(h[:,0]), -15, 15))) (trunc_exp backward), g_h[:,1:16] = g_color_in[:,16:31]
// All are pure streams (HBM-bound, 60-130 B per sample); one lane handles one sample row with 16-byte accesses.
#include "common.h"
#include "sh_poly.inc"

namespace ngp {

constexpr int PL_THREADS = 256;

__global__ __launch_bounds__(PL_THREADS) void k_mid_forward(const half_t* __restrict__ h, const float* __restrict__ dirs,
                                                            float* __restrict__ sigma, half_t* __restrict__ color_in, uint32_t M,
                                                            uint32_t M_valid, float density_scale) {
    const uint32_t b = blockIdx.x * PL_THREADS + threadIdx.x;
    if (b >= M) return;
    const half8_t h0 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16);
    const half8_t h1 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16 + 8);
    sigma[b] = density_scale * expf((float)h0[0]);  // renderer.py:296: sigmas = density_scale * sigmas
    float sh[16];
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (b < M_valid) { x = dirs[(size_t)b * 3]; y = dirs[(size_t)b * 3 + 1]; z = dirs[(size_t)b * 3 + 2]; }
#define SH_OUT(i, v) sh[i] = (v)
    SH_BAND_0_VALUES;
    SH_BAND_1_VALUES;
    SH_BAND_2_VALUES;
    SH_BAND_3_VALUES;
#undef SH_OUT
    half8_t o[4];
#pragma unroll
    for (int i = 0; i < 8; i++) { o[0][i] = to_half_rne(sh[i]); o[1][i] = to_half_rne(sh[8 + i]); }
#pragma unroll
    for (int i = 0; i < 7; i++) o[2][i] = h0[i + 1];
    o[2][7] = h1[0];
#pragma unroll
    for (int i = 0; i < 7; i++) o[3][i] = h1[i + 1];
    o[3][7] = (half_t)0.0f;
    half8_t* dst = reinterpret_cast<half8_t*>(color_in + (size_t)b * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) dst[q] = o[q];
}

__global__ __launch_bounds__(PL_THREADS) void k_rgb_forward(const half_t* __restrict__ out16, float* __restrict__ rgb, uint32_t M) {
    const uint32_t b = blockIdx.x * PL_THREADS + threadIdx.x;
    if (b >= M) return;
    const half4_t o = *reinterpret_cast<const half4_t*>(out16 + (size_t)b * 16);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float s = 1.0f / (1.0f + expf(-(float)o[c]));
        rgb[(size_t)b * 3 + c] = (float)to_half_rne(s);
    }
}

__global__ __launch_bounds__(PL_THREADS) void k_rgb_backward(const float* __restrict__ grad_rgb, const float* __restrict__ rgb,
                                                             half_t* __restrict__ grad_out16, uint32_t M) {
    const uint32_t b = blockIdx.x * PL_THREADS + threadIdx.x;
    if (b >= M) return;
    half8_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) { lo[i] = (half_t)0.0f; hi[i] = (half_t)0.0f; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float y = rgb[(size_t)b * 3 + c];
        lo[c] = to_half_rne(grad_rgb[(size_t)b * 3 + c] * (y * (1.0f - y)));
    }
    half8_t* dst = reinterpret_cast<half8_t*>(grad_out16 + (size_t)b * 16);
    dst[0] = lo;
    dst[1] = hi;
}

__global__ __launch_bounds__(PL_THREADS) void k_mid_backward(const float* __restrict__ grad_sigma, const half_t* __restrict__ h,
                                                             const half_t* __restrict__ grad_color_in, half_t* __restrict__ grad_h,
                                                             uint32_t M, float density_scale) {
    const uint32_t b = blockIdx.x * PL_THREADS + threadIdx.x;
    if (b >= M) return;
    const float x = (float)h[(size_t)b * 16];
    const float gs = (density_scale * grad_sigma[b]) * expf(fminf(15.0f, fmaxf(-15.0f, x)));
    const half8_t g2 = *reinterpret_cast<const half8_t*>(grad_color_in + (size_t)b * 32 + 16);
    const half8_t g3 = *reinterpret_cast<const half8_t*>(grad_color_in + (size_t)b * 32 + 24);
    half8_t lo, hi;
    lo[0] = to_half_rne(gs);
#pragma unroll
    for (int i = 0; i < 7; i++) lo[i + 1] = g2[i];
    hi[0] = g2[7];
#pragma unroll
    for (int i = 0; i < 7; i++) hi[i + 1] = g3[i];
    half8_t* dst = reinterpret_cast<half8_t*>(grad_h + (size_t)b * 16);
    dst[0] = lo;
    dst[1] = hi;
}

// nerf/utils.py:516,557 (criterion = MSELoss(reduction='none'), .mean(-1), .mean()) and its backward times the loss scale, in ONE launch:
//   loss = mean((image - target)^2),  grad_image = (2/n * (image - target)) * loss_scale        (same operation order as
// torch's mse_loss backward followed by the GradScaler multiply).  One workgroup: n is a ray batch (a few 10^4 values); fixed-order tree
// reduction -> deterministic.
constexpr int LOSS_THREADS = 1024;
__global__ __launch_bounds__(LOSS_THREADS) void k_mse_loss(const float* __restrict__ image, const float* __restrict__ target, uint32_t n,
                                                           const float* __restrict__ loss_scale, float* __restrict__ loss,
                                                           float* __restrict__ grad_image) {
    __shared__ float part[LOSS_THREADS / 64];
    const float scale = loss_scale ? loss_scale[0] : 1.0f;
    const float norm = 2.0f / (float)n;
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += LOSS_THREADS) {
        const float diff = image[i] - target[i];
        acc = __builtin_fmaf(diff, diff, acc);
        grad_image[i] = (norm * diff) * scale;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < LOSS_THREADS / 64 ? part[threadIdx.x] : 0.0f;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) loss[0] = v / (float)n;
    }
}

// nerf/utils.py:53-137 (get_rays), the arithmetic part: pixel index -> camera-space direction through the pinhole intrinsics ->
// normalise -> rotate by the camera-to-world pose; the ray origin is the camera position.  Pixel p of a W-wide image sits at
// (column p % W + 0.5, row p / W + 0.5) (the reference's transposed meshgrid, flattened row-major).  One lane per ray.
__global__ __launch_bounds__(PL_THREADS) void k_rays_from_pixels(const float* __restrict__ poses, uint32_t B, float fx, float fy, float cx,
                                                                 float cy, uint32_t W, const long long* __restrict__ inds,
                                                                 uint32_t inds_batch_stride, uint32_t N, float* __restrict__ rays_o,
                                                                 float* __restrict__ rays_d) {
    const uint32_t t = blockIdx.x * PL_THREADS + threadIdx.x;
    if (t >= B * N) return;
    const uint32_t b = t / N, n = t - b * N;
    const long long pix = inds ? inds[(size_t)b * inds_batch_stride + n] : (long long)n;
    const float i = (float)(uint32_t)(pix % W) + 0.5f, j = (float)(uint32_t)(pix / W) + 0.5f;
    const float xs = (i - cx) / fx, ys = (j - cy) / fy;
    const float len = sqrtf((xs * xs + ys * ys) + 1.0f);
    const float dx = xs / len, dy = ys / len, dz = 1.0f / len;
    const float* P = poses + (size_t)b * 16;  // row-major 4x4, rotation in the upper-left 3x3, position in the last column
    float* o = rays_o + (size_t)t * 3;
    float* d = rays_d + (size_t)t * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d[k] = (dx * P[4 * k] + dy * P[4 * k + 1]) + dz * P[4 * k + 2];
        o[k] = P[4 * k + 3];
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" int ngp_pipeline_mid_forward(const void* h16, const float* dirs, float* sigma, void* color_in, uint32_t M, uint32_t M_valid,
                                        float density_scale, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(h16 && dirs && sigma && color_in, NGP_ERR_INVALID, "pipeline_mid_forward: NULL tensor");
    hipLaunchKernelGGL(k_mid_forward, dim3(cdiv(M, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), (const half_t*)h16, dirs, sigma,
                       (half_t*)color_in, M, M_valid, density_scale);
    return check_launch("pipeline_mid_forward");
}

extern "C" int ngp_pipeline_rgb_forward(const void* out16, float* rgb, uint32_t M, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(out16 && rgb, NGP_ERR_INVALID, "pipeline_rgb_forward: NULL tensor");
    hipLaunchKernelGGL(k_rgb_forward, dim3(cdiv(M, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), (const half_t*)out16, rgb, M);
    return check_launch("pipeline_rgb_forward");
}

extern "C" int ngp_pipeline_rgb_backward(const float* grad_rgb, const float* rgb, void* grad_out16, uint32_t M, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(grad_rgb && rgb && grad_out16, NGP_ERR_INVALID, "pipeline_rgb_backward: NULL tensor");
    hipLaunchKernelGGL(k_rgb_backward, dim3(cdiv(M, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), grad_rgb, rgb, (half_t*)grad_out16, M);
    return check_launch("pipeline_rgb_backward");
}

extern "C" int ngp_pipeline_mid_backward(const float* grad_sigma, const void* h16, const void* grad_color_in, void* grad_h16, uint32_t M,
                                         float density_scale, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(grad_sigma && h16 && grad_color_in && grad_h16, NGP_ERR_INVALID, "pipeline_mid_backward: NULL tensor");
    hipLaunchKernelGGL(k_mid_backward, dim3(cdiv(M, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), grad_sigma, (const half_t*)h16,
                       (const half_t*)grad_color_in, (half_t*)grad_h16, M, density_scale);
    return check_launch("pipeline_mid_backward");
}

extern "C" int ngp_pipeline_mse_loss(const float* image, const float* target, uint32_t n, const float* loss_scale, float* loss,
                                     float* grad_image, ngp_stream_t stream) {
    NGP_REQUIRE(loss, NGP_ERR_INVALID, "pipeline_mse_loss: NULL tensor");
    NGP_REQUIRE(n == 0 || (image && target && grad_image), NGP_ERR_INVALID, "pipeline_mse_loss: NULL tensor");
    NGP_REQUIRE(n > 0, NGP_ERR_INVALID, "pipeline_mse_loss: empty batch (the mean of no values is undefined)");
    hipLaunchKernelGGL(k_mse_loss, dim3(1), dim3(LOSS_THREADS), 0, as_stream(stream), image, target, n, loss_scale, loss, grad_image);
    return check_launch("pipeline_mse_loss");
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// nn.Linear stack <-> the fused MLP's flat weight vector (nerf/network.py: BASELINE config 5 evaluates its bias-free Linear / ReLU stacks on
// the fused-MLP kernels).  PyTorch assembled the vector with pad / eye / cat (and autograd took it apart again): ~8 launches per stack and
// direction, three stacks per step.  One launch each way: flat = half(W_0 [hidden, n_in] padded to in_pad columns) | [identity, if
// `identity`] | half(W_1 .. W_{depth-2}) [hidden, hidden] | half(W_{depth-1} [n_out, hidden]) padded to 16 rows; the gradient goes back as
// fp32 slices (the padding's and the identity's gradient are dropped).
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace ngp {
struct LinearStack {
    const float* w[NGP_LINEAR_STACK_MAX];
    float* g[NGP_LINEAR_STACK_MAX];
    uint32_t depth, n_in, in_pad, hidden, n_out, identity;
    __host__ __device__ uint32_t total() const { return hidden * in_pad + (identity ? hidden * hidden : 0u) + (depth - 2u) * hidden * hidden + 16u * hidden; }
};

// flat index -> (layer, row, column, live): live = the element is a weight (not padding, not the identity block)
__device__ __forceinline__ bool linear_stack_locate(const LinearStack& st, uint32_t i, uint32_t& layer, uint32_t& src, float& constant) {
    constant = 0.0f;
    uint32_t seg = st.hidden * st.in_pad;
    if (i < seg) {
        const uint32_t r = i / st.in_pad, c = i - r * st.in_pad;
        layer = 0u;
        src = r * st.n_in + c;
        return c < st.n_in;
    }
    i -= seg;
    seg = st.hidden * st.hidden;
    if (st.identity) {
        if (i < seg) {
            const uint32_t r = i / st.hidden, c = i - r * st.hidden;
            constant = r == c ? 1.0f : 0.0f;
            return false;
        }
        i -= seg;
    }
    const uint32_t mid = st.depth - 2u;
    if (i < mid * seg) {
        layer = 1u + i / seg;
        src = i - (layer - 1u) * seg;
        return true;
    }
    i -= mid * seg;
    layer = st.depth - 1u;
    src = i;   // [16, hidden] rows beyond n_out are padding
    return i < st.n_out * st.hidden;
}

__global__ __launch_bounds__(PL_THREADS) void k_linear_stack_pack(LinearStack st, half_t* __restrict__ flat) {
    const uint32_t i = blockIdx.x * PL_THREADS + threadIdx.x;
    if (i >= st.total()) return;
    uint32_t layer = 0u, src = 0u;
    float constant;
    const bool live = linear_stack_locate(st, i, layer, src, constant);
    float v = constant;
#pragma unroll
    for (uint32_t l = 0; l < NGP_LINEAR_STACK_MAX; l++)   // (a dynamic index into the by-value pointer array would go through scratch memory)
        if (live && l == layer) v = st.w[l][src];
    flat[i] = (half_t)v;
}

__global__ __launch_bounds__(PL_THREADS) void k_linear_stack_unpack(LinearStack st, const half_t* __restrict__ grad_flat) {
    const uint32_t i = blockIdx.x * PL_THREADS + threadIdx.x;
    if (i >= st.total()) return;
    uint32_t layer = 0u, src = 0u;
    float constant;
    if (!linear_stack_locate(st, i, layer, src, constant)) return;
    const float v = (float)grad_flat[i];
#pragma unroll
    for (uint32_t l = 0; l < NGP_LINEAR_STACK_MAX; l++)
        if (l == layer) st.g[l][src] = v;
}

// dst [dst_rows, dst_cols] = src [src_rows, src_cols] (row stride src_stride elements) in the top-left corner, zeros elsewhere: the batch / column
// padding the fused-MLP kernels want, one launch (PyTorch's constant_pad_nd is a fill and a copy)
__global__ __launch_bounds__(PL_THREADS) void k_pad_2d(const half_t* __restrict__ src, uint32_t src_rows, uint32_t src_cols, uint32_t src_stride,
                                                        half_t* __restrict__ dst, uint32_t dst_rows, uint32_t dst_cols) {
    const uint32_t i = blockIdx.x * PL_THREADS + threadIdx.x;
    if (i >= dst_rows * dst_cols) return;
    const uint32_t r = i / dst_cols, c = i - r * dst_cols;
    dst[i] = (r < src_rows && c < src_cols) ? src[(size_t)r * src_stride + c] : (half_t)0.0f;
}
}  // namespace ngp

static int linear_stack_args(const float* const* weights, float* const* grads, uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out,
                             int identity, LinearStack& st, const char* who) {
    NGP_REQUIRE(depth >= 2 && depth <= NGP_LINEAR_STACK_MAX, NGP_ERR_INVALID, "%s: depth %u outside 2 .. %d", who, depth, NGP_LINEAR_STACK_MAX);
    NGP_REQUIRE(n_in > 0 && hidden > 0 && n_out > 0 && n_out <= 16, NGP_ERR_INVALID, "%s: n_in / hidden must be positive, n_out in 1 .. 16", who);
    st = LinearStack{};
    for (uint32_t l = 0; l < depth; l++) {
        NGP_REQUIRE((weights == nullptr || weights[l]) && (grads == nullptr || grads[l]), NGP_ERR_INVALID, "%s: NULL tensor for layer %u", who, l);
        st.w[l] = weights ? weights[l] : nullptr;
        st.g[l] = grads ? grads[l] : nullptr;
    }
    st.depth = depth; st.n_in = n_in; st.in_pad = (n_in + 15u) / 16u * 16u; st.hidden = hidden; st.n_out = n_out; st.identity = identity ? 1u : 0u;
    return NGP_OK;
}

extern "C" uint32_t ngp_linear_stack_flat_size(uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity) {
    LinearStack st{};
    st.depth = depth < 2 ? 2 : depth; st.n_in = n_in; st.in_pad = (n_in + 15u) / 16u * 16u; st.hidden = hidden; st.n_out = n_out; st.identity = identity ? 1u : 0u;
    return st.total();
}

extern "C" int ngp_linear_stack_pack(const float* const* weights, uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity,
                                     void* flat_fp16, ngp_stream_t stream) {
    LinearStack st;
    NGP_REQUIRE(weights && flat_fp16, NGP_ERR_INVALID, "linear_stack_pack: NULL tensor");
    if (int rc = linear_stack_args(weights, nullptr, depth, n_in, hidden, n_out, identity, st, "linear_stack_pack")) return rc;
    hipLaunchKernelGGL(k_linear_stack_pack, dim3(cdiv(st.total(), PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), st, (half_t*)flat_fp16);
    return check_launch("linear_stack_pack");
}

extern "C" int ngp_linear_stack_unpack_grad(const void* grad_flat_fp16, uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity,
                                            float* const* grads, ngp_stream_t stream) {
    LinearStack st;
    NGP_REQUIRE(grads && grad_flat_fp16, NGP_ERR_INVALID, "linear_stack_unpack_grad: NULL tensor");
    if (int rc = linear_stack_args(nullptr, grads, depth, n_in, hidden, n_out, identity, st, "linear_stack_unpack_grad")) return rc;
    hipLaunchKernelGGL(k_linear_stack_unpack, dim3(cdiv(st.total(), PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), st, (const half_t*)grad_flat_fp16);
    return check_launch("linear_stack_unpack_grad");
}

extern "C" int ngp_pad_2d_fp16(const void* src, uint32_t src_rows, uint32_t src_cols, uint32_t src_row_stride, void* dst, uint32_t dst_rows,
                               uint32_t dst_cols, ngp_stream_t stream) {
    if (dst_rows == 0 || dst_cols == 0) return NGP_OK;
    NGP_REQUIRE(dst && (src || src_rows == 0 || src_cols == 0), NGP_ERR_INVALID, "pad_2d_fp16: NULL tensor");
    NGP_REQUIRE(src_rows <= dst_rows && src_cols <= dst_cols && src_row_stride >= src_cols, NGP_ERR_INVALID,
                "pad_2d_fp16: source [%u, %u] (row stride %u) does not fit the destination [%u, %u]", src_rows, src_cols, src_row_stride, dst_rows, dst_cols);
    NGP_REQUIRE((uint64_t)dst_rows * dst_cols <= 0xffffffffull, NGP_ERR_INVALID, "pad_2d_fp16: destination too large");
    hipLaunchKernelGGL(k_pad_2d, dim3(cdiv(dst_rows * dst_cols, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), (const half_t*)src, src_rows,
                       src_cols, src_row_stride, (half_t*)dst, dst_rows, dst_cols);
    return check_launch("pad_2d_fp16");
}

extern "C" int ngp_rays_from_pixels(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t W, const int64_t* inds,
                                    uint32_t inds_batch_stride, uint32_t N, float* rays_o, float* rays_d, ngp_stream_t stream) {
    if (B == 0 || N == 0) return NGP_OK;
    NGP_REQUIRE(poses && rays_o && rays_d, NGP_ERR_INVALID, "rays_from_pixels: NULL tensor");
    NGP_REQUIRE(W > 0 && fx != 0.0f && fy != 0.0f, NGP_ERR_INVALID, "rays_from_pixels: W, fx and fy must be non-zero");
    NGP_REQUIRE((uint64_t)B * N <= 0xffffffffull, NGP_ERR_INVALID, "rays_from_pixels: B * N must fit 32 bits");
    hipLaunchKernelGGL(k_rays_from_pixels, dim3(cdiv(B * N, PL_THREADS)), dim3(PL_THREADS), 0, as_stream(stream), poses, B, fx, fy, cx, cy, W,
                       (const long long*)inds, inds_batch_stride, N, rays_o, rays_d);
    return check_launch("rays_from_pixels");
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The eval loop of NeRFRenderer.run_cuda (renderer.py:341-367), n iterations per call: march -> encode -> network -> composite -> compact,
// ping-pong alive lists / device state, issued from here -- ONE call where the Python loop made five per iteration plus three allocations
// (round 5 measured ~95 us of host time per iteration against ~30 us of GPU work in the tail of an opaque frame).  Same kernels, same
// arguments, same order as the per-stage calls: the image is the same bit for bit.
// ---------------------------------------------------------------------------------------------------------------------------------------
extern "C" int ngp_render_iterations_dev(const ngp_render_loop_t* a, uint32_t n_iter, uint32_t first_cur, ngp_stream_t stream) {
    NGP_REQUIRE(a, NGP_ERR_INVALID, "render_iterations: NULL argument block");
    NGP_REQUIRE(a->state && a->alive[0] && a->alive[1] && a->rays_t && a->rays_o && a->rays_d && a->nears && a->fars && a->grid, NGP_ERR_INVALID,
                "render_iterations: NULL ray state");
    NGP_REQUIRE(a->xyzs && a->dirs && a->deltas && a->enc && a->sigmas && a->rgbs && a->embeddings && a->offsets && a->w_sigma && a->w_color,
                NGP_ERR_INVALID, "render_iterations: NULL sample buffer / network tensor");
    NGP_REQUIRE(a->weights_sum && a->depth && a->image && a->compact_workspace, NGP_ERR_INVALID, "render_iterations: NULL accumulator");
    NGP_REQUIRE(a->rows % 128u == 0u && a->rows > 0u, NGP_ERR_INVALID, "render_iterations: rows must be a positive multiple of 128 (the fused network's tile)");
    NGP_REQUIRE(first_cur < 2u, NGP_ERR_INVALID, "render_iterations: first_cur is 0 or 1");
    for (uint32_t i = 0; i < n_iter; i++) {
        const uint32_t cur = (first_cur + i) & 1u, nxt = cur ^ 1u;
        const int32_t* st = a->state + 2 * cur;
        // rows_used (optional device word): the march publishes the rows that can carry a sample; the encoder and the network, launched for
        // `rows` (the caller's stale bound), stop there -- so a batch of iterations may be issued without a read-back in between
        int rc = ngp_march_rays_dev_rows(st, a->lanes, a->n_total, a->n_step_cap, a->alive[cur], a->rays_t, a->rays_o, a->rays_d, a->bound, a->dt_gamma,
                                         a->max_steps, a->cascade, a->grid_size, a->grid, a->nears, a->fars, a->xyzs, a->dirs, a->deltas,
                                         i == 0 ? a->noises : nullptr, a->rows, a->rows_used, stream);
        if (rc) return rc;
        rc = ngp_grid_encode_forward_sel(a->xyzs, a->embeddings, nullptr, nullptr, a->rows_used, a->offsets, a->enc, a->rows, 3, 2, a->L, a->S, a->H,
                                         a->gridtype, a->align_corners, a->interp, NGP_F16, a->bound, a->level_cost_host, stream);
        if (rc) return rc;
        rc = ngp_network_forward_rows(a->enc, a->dirs, a->rows, a->rows, a->w_sigma, a->w_color, a->num_layers_sigma, a->num_layers_color,
                                      a->density_scale, 0, nullptr, nullptr, a->sigmas, nullptr, nullptr, a->rgbs, NGP_FF_INPUT_PLANAR, a->rows_used, stream);
        if (rc) return rc;
        rc = ngp_composite_rays_dev(st, a->lanes, a->n_total, a->n_step_cap, a->T_thresh, a->alive[cur], a->rays_t, a->sigmas, a->rgbs, a->deltas,
                                    a->weights_sum, a->depth, a->image, stream);
        if (rc) return rc;
        rc = ngp_compact_rays_dev(st, a->lanes, a->n_total, a->n_step_cap, a->max_steps, a->alive[cur], a->alive[nxt], a->state + 2 * nxt,
                                  a->compact_workspace, stream);
        if (rc) return rc;
    }
    return NGP_OK;
}
