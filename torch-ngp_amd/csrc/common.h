// Shared host/device helpers for the gfx950 kernels of libngp_hip.so.
// gfx950 only: wave64, no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdarg>

#include "../../include/ngp_hip.h"

namespace ngp {

using half_t = _Float16;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// ---- error plumbing: every extern "C" entry returns 0 / non-zero and records a message ----
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return NGP_ERR_LAUNCH;
    }
    return NGP_OK;
}

#define NGP_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            ::ngp::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline uint64_t cdiv64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

inline hipStream_t as_stream(ngp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- debug build (make DEBUG_BOUNDS=1 -> libngp_hip_dbg.so): device-side range traps on the indices the fast paths trust ----
// (encoder work lists and table indices, record-sort slots / descriptors / counters, accumulate record and entry indices, marcher sample
// rows).  A violated bound aborts the kernel (s_trap), which the host sees as a failed synchronisation; the product build compiles the
// checks away.
#ifdef NGP_DEBUG_BOUNDS
#define NGP_BOUNDS(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define NGP_BOUNDS(cond) do { } while (0)
#endif

// ---- device helpers ----
// fp32 -> fp16, round-to-nearest-even of the fp32 VALUE.  Without the barrier the compiler folds `(half)(a * b)` into v_fma_mixlo_f16, which
// rounds the exact product once: a different result on exact ties than "compute in fp32, then cast" -- what the reference's autocast path
// does (a fp32 tensor, then .to(half)) and what two kernels sharing this arithmetic must agree on bit for bit.
__device__ __forceinline__ _Float16 to_half_rne(float v) {
    asm volatile("" : "+v"(v));
    return (_Float16)v;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// wave64 inclusive prefix sum of a uint32 (DPP-free, 6 shuffles)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- wave64 DPP scans (shared by the compositor and the carried loss sum: the ORDER of the additions is part of the result's bits) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float identity, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, src), CTRL, ROW_MASK,
                                                                 0xF, false));
}
__device__ __forceinline__ float wave_incl_sum(float v, int) {
    v += dpp_or<0x111, 0xF>(0.0f, v);  // row_shr:1
    v += dpp_or<0x112, 0xF>(0.0f, v);  // row_shr:2
    v += dpp_or<0x114, 0xF>(0.0f, v);  // row_shr:4
    v += dpp_or<0x118, 0xF>(0.0f, v);  // row_shr:8
    v += dpp_or<0x142, 0xA>(0.0f, v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or<0x143, 0xC>(0.0f, v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ float wave_total(float v) { return lane63(wave_incl_sum(v, 0)); }   // wave-uniform sum

// The training loss VALUE from the per-ray squared errors the compositor left (raymarching.hip: k_composite_train_loss_bwd): threads 0..255
// of the calling block add err[t], err[t + 256], ... in order, a DPP scan per wave, the four wave sums in wave order, / (3 N) -- ONE routine
// for the compositor's last workgroup and for the block that carries the sum in a later launch: the same bits.  Called by all threads of a
// block of >= 256 threads; part: 4 floats of LDS.  The errors were written in an earlier launch or as write-through stores (agent scope).
__device__ __forceinline__ void loss_sum_block(const float* err, uint32_t N, float* loss, float* part) {
    const uint32_t t = threadIdx.x;
    float acc = 0.0f;
    if (t < 256u) {
        for (uint32_t i = t; i < N; i += 256u) acc += __hip_atomic_load(&err[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = wave_total(acc);
        if ((t & 63u) == 0u) part[t >> 6] = acc;
    }
    __syncthreads();
    if (t == 0u) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; w++) v += part[w];
        loss[0] = v / (float)(3u * N);
    }
}

// ---- Adam + loss-scale arithmetic shared by optim.hip (k_adam, k_adam_small_commit) and the slice accumulate of the grid backward that
// carries the table's Adam sweep in its flush (gridencoder.hip): ONE statement of the update, so the fused and the separate form produce the
// same bits (the library is built with -ffp-contract=off).
// state[0] = loss scale, [1] = growth tracker, [2] = found_inf, [3] = Adam step count t, [4] = lr multiplier, [5] = table parity (0 / 1: which of
// the two buffer sets of a speculatively updated table is current, ngp_table_adam_t)
struct AdamConsts {
    float inv_scale, bc1, bc2_sqrt, lr_mult;
    bool skip;  // found_inf already raised, or the loss scale is unusable (underflowed to 0 / denormal: GradScaler's 1 / scale = inf)
};
__device__ __forceinline__ AdamConsts adam_consts(const float* state, float beta1, float beta2, float grad_mult) {
    AdamConsts a;
    const bool scale_dead = !__builtin_isfinite(1.0f / state[0]);
    a.skip = state[2] != 0.0f || scale_dead;
    a.inv_scale = scale_dead ? 0.0f : grad_mult / state[0];
    const float t = state[3] + 1.0f;  // this step's count (the commit stores it)
    a.bc1 = 1.0f - powf(beta1, t);
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, t));
    a.lr_mult = state[4];
    return a;
}
// one element: g = the (already unscaled) gradient; PyTorch's Adam (amsgrad = False, weight_decay = 0)
__device__ __forceinline__ void adam_element(float g, float& pm, float& pv, float& pp, float beta1, float beta2, float eps, float step_size,
                                             float bc2_sqrt) {
    pm = beta1 * pm + (1.0f - beta1) * g;
    pv = beta2 * pv + (1.0f - beta2) * g * g;
    pp -= step_size * pm / (sqrtf(pv) / bc2_sqrt + eps);
}
// the table's Adam sweep inside the grid backward (device-side copy of ngp_table_adam_t)
struct TableAdam {
    float* p[2];
    float* m[2];
    float* v[2];
    _Float16* p16[2];
    const float* state;
    float lr, beta1, beta2, eps;
};
// which copy of a double-buffered fp16 table a reader takes (device-side copy of the (embeddings_alt, parity) pair of the *_sel entry points)
struct TableSel {
    const _Float16* alt;
    const float* parity;
    const uint32_t* rows;   // optional device word: only the first min(B, *rows) points carry work (the eval loop's emitted-row count)
};

// ---- fixed-order sum of per-workgroup fp32 weight-gradient slabs (the FFMLP backward's deferred reduction), TWO sets per launch ----
// One block of RS_PARAMS x RS_GROUPS threads sums RS_PARAMS parameters of one set: group g adds slabs g, g + RS_GROUPS, ..., the groups'
// partial sums are added in group order and rounded once to fp16 -- the same order for every launch shape, so the same bits whether the
// blocks are a launch of their own (ffmlp.hip: k_ffmlp_reduce_slabs_pair) or ride in another kernel's grid (gridencoder.hip: the slice
// accumulate of the training step, which would otherwise be followed by a 7 us launch of ~270 small blocks).
constexpr int RS_PARAMS = 64, RS_GROUPS = 16;
struct SlabSets {
    const float* slabs[2];
    uint32_t n_slabs[2], n_params[2];
    _Float16* grad_weights[2];
    uint32_t blocks[2];  // blocks serving each set: cdiv(n_params, RS_PARAMS), or 0 for a set with nothing to do
    // optional third job, one more block (index blocks[0] + blocks[1]): loss[0] = sum(ray_err[0 .. n_rays)) / (3 n_rays), loss_sum_block
    const float* ray_err;
    uint32_t n_rays;
    float* loss;
    __host__ __device__ uint32_t total_blocks() const { return blocks[0] + blocks[1] + (loss ? 1u : 0u); }
};
// block `b` of a grid row that carries the jobs (all threads of a RS_PARAMS x RS_GROUPS block; part: RS_GROUPS x RS_PARAMS floats of LDS)
__device__ __forceinline__ void slab_reduce_block(const SlabSets& s, uint32_t b, float (*part)[RS_PARAMS], float* found_inf);
__device__ __forceinline__ void carried_block(const SlabSets& s, uint32_t b, float (*part)[RS_PARAMS], float* found_inf) {
    if (b < s.blocks[0] + s.blocks[1]) slab_reduce_block(s, b, part, found_inf);
    else if (s.loss && b == s.blocks[0] + s.blocks[1]) loss_sum_block(s.ray_err, s.n_rays, s.loss, &part[0][0]);
}
// block `b` of blocks[0] + blocks[1]; part: RS_GROUPS x RS_PARAMS floats of LDS; found_inf (optional) is set to 1 when a resulting gradient is
// not finite (a set with n_slabs = 0 holds gradients its backward stored directly: only swept).  Called by ALL threads of the block.
__device__ __forceinline__ void slab_reduce_block(const SlabSets& s, uint32_t b, float (*part)[RS_PARAMS], float* found_inf) {
    const int set = b >= s.blocks[0] ? 1 : 0;
    const float* __restrict__ slabs = s.slabs[set];
    const uint32_t n_slabs = s.n_slabs[set], n_params = s.n_params[set];
    _Float16* __restrict__ grad_weights = s.grad_weights[set];
    const uint32_t li = threadIdx.x & (RS_PARAMS - 1), g = threadIdx.x / RS_PARAMS;
    const uint32_t i = (b - (set ? s.blocks[0] : 0u)) * RS_PARAMS + li;
    float acc = 0.0f;
    if (i < n_params)
        for (uint32_t k = g; k < n_slabs; k += RS_GROUPS) acc += slabs[(size_t)k * n_params + i];
    part[g][li] = acc;
    __syncthreads();
    bool nonfinite = false;
    if (g == 0 && i < n_params) {
        _Float16 r;
        if (n_slabs) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < RS_GROUPS; q++) t += part[q][li];
            r = (_Float16)t;
            grad_weights[i] = r;
        } else {
            r = grad_weights[i];
        }
        nonfinite = !__builtin_isfinite((float)r);
    }
    if (found_inf && __any(nonfinite) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;
}

}  // namespace ngp
