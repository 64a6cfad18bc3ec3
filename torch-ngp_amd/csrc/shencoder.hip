// Real spherical-harmonics direction encoder for gfx950 (MI355X).
//
// Behaviour restated from shencoder/src/shencoder.cu of the reference:
//   forward   :27-355  (kernel_sh: C bands -> C*C polynomial values, optional analytic d/dx,d/dy,d/dz)
//   backward  :358-382 (kernel_sh_backward: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch])
// The polynomials come from tools/gen_sh.py (sh_poly.inc): the reference's Cartesian forms restated
// symbolically; the derivative tables are produced by symbolic differentiation, not transcribed.
//
// MI355X design: the op is a pure stream (12 B in, 4*C*C B out per direction) with ~100 flops per
// point, i.e. HBM-bound.  One lane evaluates one direction in registers; the C*C results of a wave are
// staged through LDS and written back as full 64-lane-coalesced rows instead of 64 strided 4-byte
// stores per component.
#include "common.h"
#include "sh_poly.inc"

namespace ngp {

constexpr int SH_THREADS = 256;

template <typename T, int BANDS, bool WITH_GRAD>
__global__ __launch_bounds__(SH_THREADS) void k_sh_forward(const T* __restrict__ inputs, T* __restrict__ outputs, uint32_t B,
                                                           T* __restrict__ dy_dx) {
    constexpr int N = BANDS * BANDS;
    __shared__ float stage[SH_THREADS / 64][64 * N + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t wave_base = (blockIdx.x * (SH_THREADS / 64) + wid) * 64;  // first point of this wave
    if (wave_base >= B) return;
    const uint32_t b = wave_base + lane;
    const bool valid = b < B;
    float x = 0.f, y = 0.f, z = 0.f;
    if (valid) {
        x = (float)inputs[(size_t)b * 3];
        y = (float)inputs[(size_t)b * 3 + 1];
        z = (float)inputs[(size_t)b * 3 + 2];
    }
    float* row = stage[wid];
    const uint32_t n_valid = min(64u, B - wave_base);

#define SH_OUT(i, v) row[lane * N + (i)] = (v)
    SH_BAND_0_VALUES;
    if constexpr (BANDS > 1) { SH_BAND_1_VALUES; }
    if constexpr (BANDS > 2) { SH_BAND_2_VALUES; }
    if constexpr (BANDS > 3) { SH_BAND_3_VALUES; }
    if constexpr (BANDS > 4) { SH_BAND_4_VALUES; }
    if constexpr (BANDS > 5) { SH_BAND_5_VALUES; }
    if constexpr (BANDS > 6) { SH_BAND_6_VALUES; }
    if constexpr (BANDS > 7) { SH_BAND_7_VALUES; }
#undef SH_OUT
    // the wave's outputs are one contiguous [n_valid * N] span of the output tensor
    {
        T* dst = outputs + (size_t)wave_base * N;
        const uint32_t total = n_valid * N;
        for (uint32_t i = lane; i < total; i += 64) dst[i] = (T)row[i];
    }
    if constexpr (WITH_GRAD) {
        // dy_dx is [B, 3, N]: three passes through the same staging rows
        T* dst = dy_dx + (size_t)wave_base * 3 * N;
#define SH_DX(i, v) row[lane * N + (i)] = (v)
#define SH_DY(i, v) ((void)0)
#define SH_DZ(i, v) ((void)0)
#define SH_ALL_GRADS                                   \
    SH_BAND_0_GRADS;                                   \
    if constexpr (BANDS > 1) { SH_BAND_1_GRADS; }      \
    if constexpr (BANDS > 2) { SH_BAND_2_GRADS; }      \
    if constexpr (BANDS > 3) { SH_BAND_3_GRADS; }      \
    if constexpr (BANDS > 4) { SH_BAND_4_GRADS; }      \
    if constexpr (BANDS > 5) { SH_BAND_5_GRADS; }      \
    if constexpr (BANDS > 6) { SH_BAND_6_GRADS; }      \
    if constexpr (BANDS > 7) { SH_BAND_7_GRADS; }
        SH_ALL_GRADS
        for (uint32_t i = lane; i < n_valid * N; i += 64) dst[(size_t)(i / N) * 3 * N + (i % N)] = (T)row[i];
#undef SH_DX
#undef SH_DY
#define SH_DX(i, v) ((void)0)
#define SH_DY(i, v) row[lane * N + (i)] = (v)
        SH_ALL_GRADS
        for (uint32_t i = lane; i < n_valid * N; i += 64) dst[(size_t)(i / N) * 3 * N + N + (i % N)] = (T)row[i];
#undef SH_DY
#undef SH_DZ
#define SH_DY(i, v) ((void)0)
#define SH_DZ(i, v) row[lane * N + (i)] = (v)
        SH_ALL_GRADS
        for (uint32_t i = lane; i < n_valid * N; i += 64) dst[(size_t)(i / N) * 3 * N + 2 * N + (i % N)] = (T)row[i];
#undef SH_DX
#undef SH_DY
#undef SH_DZ
#undef SH_ALL_GRADS
    }
}

// shencoder.cu:358-382
template <typename T>
__global__ void k_sh_backward(const T* __restrict__ grad, uint32_t B, uint32_t N, const T* __restrict__ dy_dx,
                              T* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const T* g = grad + (size_t)b * N;
    const T* dd = dy_dx + ((size_t)b * 3 + d) * N;
    float r = (float)grad_inputs[t];
    for (uint32_t i = 0; i < N; i++) r = __builtin_fmaf((float)g[i], (float)dd[i], r);
    grad_inputs[t] = (T)r;
}

template <typename T, int BANDS>
static int launch_sh(const void* inputs, void* outputs, uint32_t B, void* dy_dx, hipStream_t st) {
    dim3 grid(cdiv(B, SH_THREADS));
    if (dy_dx)
        hipLaunchKernelGGL((k_sh_forward<T, BANDS, true>), grid, dim3(SH_THREADS), 0, st, (const T*)inputs, (T*)outputs, B, (T*)dy_dx);
    else
        hipLaunchKernelGGL((k_sh_forward<T, BANDS, false>), grid, dim3(SH_THREADS), 0, st, (const T*)inputs, (T*)outputs, B, (T*)nullptr);
    return check_launch("sh_encode_forward");
}

template <typename T>
static int dispatch_sh(uint32_t C, const void* inputs, void* outputs, uint32_t B, void* dy_dx, hipStream_t st) {
    switch (C) {
        case 1: return launch_sh<T, 1>(inputs, outputs, B, dy_dx, st);
        case 2: return launch_sh<T, 2>(inputs, outputs, B, dy_dx, st);
        case 3: return launch_sh<T, 3>(inputs, outputs, B, dy_dx, st);
        case 4: return launch_sh<T, 4>(inputs, outputs, B, dy_dx, st);
        case 5: return launch_sh<T, 5>(inputs, outputs, B, dy_dx, st);
        case 6: return launch_sh<T, 6>(inputs, outputs, B, dy_dx, st);
        case 7: return launch_sh<T, 7>(inputs, outputs, B, dy_dx, st);
        default: return launch_sh<T, 8>(inputs, outputs, B, dy_dx, st);
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" int ngp_sh_encode_forward(const void* inputs, void* outputs, uint32_t B, uint32_t D, uint32_t C, void* dy_dx, int dtype,
                                     ngp_stream_t stream) {
    NGP_REQUIRE(D == 3, NGP_ERR_INVALID, "sh_encode_forward: SH encoder only support input dim == 3 (got %u)", D);
    NGP_REQUIRE(C >= 1 && C <= 8, NGP_ERR_INVALID, "sh_encode_forward: SH encoder only supports degree in [1, 8] (got %u)", C);
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_ERR_INVALID, "sh_encode_forward: inputs must be float32 or float16");
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && outputs, NGP_ERR_INVALID, "sh_encode_forward: NULL tensor");
    return dtype == NGP_F16 ? dispatch_sh<half_t>(C, inputs, outputs, B, dy_dx, as_stream(stream))
                            : dispatch_sh<float>(C, inputs, outputs, B, dy_dx, as_stream(stream));
}

extern "C" int ngp_sh_encode_backward(const void* grad, const void* inputs, uint32_t B, uint32_t D, uint32_t C, const void* dy_dx,
                                      void* grad_inputs, int dtype, ngp_stream_t stream) {
    (void)inputs;
    NGP_REQUIRE(D == 3, NGP_ERR_INVALID, "sh_encode_backward: SH encoder only support input dim == 3 (got %u)", D);
    NGP_REQUIRE(C >= 1 && C <= 8, NGP_ERR_INVALID, "sh_encode_backward: SH encoder only supports degree in [1, 8] (got %u)", C);
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_ERR_INVALID, "sh_encode_backward: grad must be float32 or float16");
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && dy_dx && grad_inputs, NGP_ERR_INVALID, "sh_encode_backward: NULL tensor");
    hipStream_t st = as_stream(stream);
    if (dtype == NGP_F16)
        hipLaunchKernelGGL((k_sh_backward<half_t>), dim3(cdiv(B * 3, 256)), dim3(256), 0, st, (const half_t*)grad, B, C * C,
                           (const half_t*)dy_dx, (half_t*)grad_inputs);
    else
        hipLaunchKernelGGL((k_sh_backward<float>), dim3(cdiv(B * 3, 256)), dim3(256), 0, st, (const float*)grad, B, C * C,
                           (const float*)dy_dx, (float*)grad_inputs);
    return check_launch("sh_encode_backward");
}
