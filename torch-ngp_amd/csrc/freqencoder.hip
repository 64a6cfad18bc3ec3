// Frequency (positional) encoder for gfx950 (MI355X) -- SURVEY.md 8(f).3.
//
// Behaviour restated from freqencoder/src/freqencoder.cu of the reference:
//   forward   :30-61   outputs[b] = [ x_0..x_{D-1} | for f in 0..deg-1: sin(2^f x_d) (d = 0..D-1), cos(2^f x_d) (d = 0..D-1) ]
//                      with cos evaluated as sin(. + pi/2) in fp32, C = D + 2 * D * deg
//   backward  :64-94   grad_x_d = g[d] + sum_f 2^f * (g_sin[f,d] * cos[f,d] - g_cos[f,d] * sin[f,d]) from the STORED outputs
// fp32 only (the reference reads data_ptr<float>() on both sides, its wrapper casts with custom_fwd(cast_inputs=float32)).
//
// MI355X design: a pure stream, 4*D B in / 4*C B out per point (C = 27 for D = 3, deg = 4).  One lane produces one OUTPUT
// element so that a wave writes 256 contiguous bytes; the D inputs of a point are re-read from L1.  The sine uses the accurate
// sinf (the reference's __sinf fast intrinsic is the looser of the two; results agree within its error).
#include "common.h"
#include <math.h>

namespace ngp {

constexpr int FQ_THREADS = 256;

__global__ __launch_bounds__(FQ_THREADS) void k_freq_forward(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                                                             float* __restrict__ outputs) {
    const uint64_t total = (uint64_t)B * C;
    for (uint64_t t = (uint64_t)blockIdx.x * FQ_THREADS + threadIdx.x; t < total; t += (uint64_t)gridDim.x * FQ_THREADS) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
        const float* x = inputs + (size_t)b * D;
        float v;
        if (c < D) {
            v = x[c];
        } else {
            const uint32_t col = c / D - 1u, d = c % D;
            const float phase = (col & 1u) ? 1.5707963267948966f : 0.0f;  // (col % 2) * (PI / 2), PI = 3.141592653589793f
            v = sinf(scalbnf(x[d], (int)(col >> 1)) + phase);
        }
        outputs[t] = v;
    }
}

__global__ __launch_bounds__(FQ_THREADS) void k_freq_backward(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B,
                                                              uint32_t D, uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint64_t total = (uint64_t)B * D;
    for (uint64_t t = (uint64_t)blockIdx.x * FQ_THREADS + threadIdx.x; t < total; t += (uint64_t)gridDim.x * FQ_THREADS) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
        const float* g = grad + (size_t)b * C;
        const float* o = outputs + (size_t)b * C;
        float r = g[d];
        for (uint32_t f = 0; f < deg; f++) {
            const uint32_t s = D + 2u * f * D + d, c = s + D;  // sin and cos slots of frequency f
            r += scalbnf(1.0f, (int)f) * (g[s] * o[c] - g[c] * o[s]);
        }
        grad_inputs[t] = r;
    }
}

}  // namespace ngp

using namespace ngp;

static int check_freq(const char* fn, uint32_t D, uint32_t deg, uint32_t C) {
    NGP_REQUIRE(D >= 1, NGP_ERR_INVALID, "%s: input dim must be positive", fn);
    NGP_REQUIRE(C == D + 2u * D * deg, NGP_ERR_INVALID, "%s: output_dim must be input_dim + 2 * input_dim * degree (got %u for D=%u, degree=%u)",
                fn, C, D, deg);
    return NGP_OK;
}

extern "C" int ngp_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                                       ngp_stream_t stream) {
    int rc = check_freq("freq_encode_forward", D, deg, C);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && outputs, NGP_ERR_INVALID, "freq_encode_forward: NULL tensor");
    uint64_t blocks = cdiv64((uint64_t)B * C, FQ_THREADS);
    if (blocks > 65536u) blocks = 65536u;
    hipLaunchKernelGGL(k_freq_forward, dim3((uint32_t)blocks), dim3(FQ_THREADS), 0, as_stream(stream), inputs, B, D, C, outputs);
    return check_launch("freq_encode_forward");
}

extern "C" int ngp_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                        float* grad_inputs, ngp_stream_t stream) {
    int rc = check_freq("freq_encode_backward", D, deg, C);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && outputs && grad_inputs, NGP_ERR_INVALID, "freq_encode_backward: NULL tensor");
    uint64_t blocks = cdiv64((uint64_t)B * D, FQ_THREADS);
    if (blocks > 65536u) blocks = 65536u;
    hipLaunchKernelGGL(k_freq_backward, dim3((uint32_t)blocks), dim3(FQ_THREADS), 0, as_stream(stream), grad, outputs, B, D, deg, C, grad_inputs);
    return check_launch("freq_encode_backward");
}
