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

}  // namespace ngp
