/* ngp_cuda_on_host.h -- TEST INFRASTRUCTURE.  A host execution model for the reference's CUDA translation units.
 *
 * oracle/_ref is built from the reference's OWN kernel sources where they lie (/root/reference/.../*.cu, never copied into this
 * repository): the Makefile pipes each .cu through one sed expression that rewrites the launch syntax
 *        kernel<T...><<<grid, block>>>(args);   ->   orc_launch(grid, block, [&]() { kernel<T...>(args); });
 * (the only construct of those files that is not C++) straight into g++, with this directory first on the include path.  The headers
 * here stand in for <cuda.h>, <cuda_fp16.h>, <cuda_runtime.h>, <ATen/cuda/CUDAContext.h> and <torch/torch.h> and provide exactly
 * what the four kernel files use:
 *   - __global__/__device__/__host__/__restrict__ (erased), blockIdx/threadIdx/blockDim/gridDim (thread-local), dim3;
 *   - orc_launch: runs the kernel body once per (block, thread) of the launch, blocks and threads in increasing order -- a legal
 *     CUDA schedule (the kernels use neither shared memory nor barriers), and the one under which the reference's
 *     atomicAdd-allocated outputs (march_rays_train's sample slots) come out in ray order;
 *   - atomicAdd for float / int / uint32_t / at::Half / __half2 as the plain sequential read-modify-write;
 *   - the device math spellings (__expf, rsqrtf, fminf/fmaxf on mixed int/float arguments, min/max);
 *   - a minimal at::Tensor (pointer + scalar type), at::optional, at::Half = the REAL c10::Half of the installed PyTorch (so the
 *     fp16 rounding points of `scalar_t = at::Half` instantiations are the genuine ones), TORCH_CHECK and an
 *     AT_DISPATCH_FLOATING_TYPES_AND_HALF that instantiates float, double and Half like the original.
 * Floating-point contraction is a compiler decision in both worlds (nvcc fuses a*b+c by default): the Makefile builds every file
 * twice, -ffp-contract=off and -ffp-contract=fast -mfma, and the pinning tests say which results are contraction-independent.
 */
#ifndef NGP_CUDA_ON_HOST_H
#define NGP_CUDA_ON_HOST_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>

#include <c10/util/Half.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct orc_uint3 { unsigned x, y, z; };
extern thread_local orc_uint3 blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;
#ifdef ORC_DEFINE_BUILTINS
thread_local orc_uint3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0};
thread_local dim3 blockDim, gridDim;
#endif

template <class F>
static inline void orc_launch(dim3 grid, dim3 block, F body) {
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
                for (unsigned tz = 0; tz < block.z; tz++)
                    for (unsigned ty = 0; ty < block.y; ty++)
                        for (unsigned tx = 0; tx < block.x; tx++) {
                            blockIdx = {bx, by, bz};
                            threadIdx = {tx, ty, tz};
                            body();
                        }
}

/* ---- half types of <cuda_fp16.h> over the real c10::Half ---- */
struct __half {
    c10::Half h;
    __half() = default;
    __half(float f) : h(f) {}
    __half(c10::Half v) : h(v) {}
    operator float() const { return static_cast<float>(h); }
};
struct __half2 { __half x, y; };
static_assert(sizeof(__half) == 2 && sizeof(__half2) == 4, "layout of the fp16 stand-ins");

/* ---- atomics: sequential read-modify-write, returning the old value ---- */
template <class T> static inline T orc_atomic_add(T* p, T v) { T old = *p; *p = old + v; return old; }
static inline float atomicAdd(float* p, float v) { return orc_atomic_add(p, v); }
static inline double atomicAdd(double* p, double v) { return orc_atomic_add(p, v); }
static inline int atomicAdd(int* p, int v) { return orc_atomic_add(p, v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return orc_atomic_add(p, v); }
static inline __half2 atomicAdd(__half2* p, __half2 v) {  /* atomicAdd(__half2*): two fp16 adds, each rounded once */
    __half2 old = *p;
    p->x = __half(static_cast<float>(old.x) + static_cast<float>(v.x));
    p->y = __half(static_cast<float>(old.y) + static_cast<float>(v.y));
    return old;
}

/* ---- device math spellings ---- */
/* __expf / __powf (fast-math intrinsics of the compositing kernels) -> libm's expf / powf: at least as accurate as the device
 * intrinsic; results agree to fp32 tolerance, not bit for bit.  (glibc declares but does not export these names: macros.) */
#define __expf(x) expf(x)
#define __powf(a, b) powf(a, b)
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
using std::max;
using std::min;
/* CUDA's global min/max/fminf/fmaxf accept mixed arithmetic arguments (e.g. fmaxf(0, exponent), min(uint32_t, int)) */
template <class A, class B, class = std::enable_if_t<!std::is_same<A, B>::value && std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
static inline std::common_type_t<A, B> min(A a, B b) { using T = std::common_type_t<A, B>; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B, class = std::enable_if_t<!std::is_same<A, B>::value && std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
static inline std::common_type_t<A, B> max(A a, B b) { using T = std::common_type_t<A, B>; return (T)a > (T)b ? (T)a : (T)b; }

/* ---- the slice of ATen the launchers touch ---- */
namespace at {
using Half = c10::Half;
enum class ScalarType { Byte, Int, Long, Half, Float, Double };
struct Device { bool is_cuda() const { return true; } };
struct Tensor {
    void* ptr = nullptr;
    ScalarType type = ScalarType::Float;
    Tensor() = default;
    Tensor(void* p, ScalarType t) : ptr(p), type(t) {}
    template <class T> T* data_ptr() const { return static_cast<T*>(ptr); }
    ScalarType scalar_type() const { return type; }
    Device device() const { return Device(); }
    bool is_contiguous() const { return true; }
};
template <class T> using optional = std::optional<T>;
namespace cuda { static inline void* getCurrentCUDAStream() { return nullptr; } }
}  // namespace at
namespace torch { using Tensor = at::Tensor; }

#define TORCH_CHECK(cond, ...) do { if (!(cond)) throw std::runtime_error("TORCH_CHECK failed: " #cond); } while (0)
#define AT_DISPATCH_FLOATING_TYPES_AND_HALF(TYPE, NAME, ...)                                          \
    do {                                                                                              \
        switch (TYPE) {                                                                               \
            case at::ScalarType::Float: { using scalar_t = float; __VA_ARGS__(); break; }             \
            case at::ScalarType::Double: { using scalar_t = double; __VA_ARGS__(); break; }           \
            case at::ScalarType::Half: { using scalar_t = at::Half; __VA_ARGS__(); break; }           \
            default: throw std::runtime_error(std::string(NAME) + ": unsupported scalar type");       \
        }                                                                                             \
    } while (0)
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...)                                                   \
    do {                                                                                              \
        switch (TYPE) {                                                                               \
            case at::ScalarType::Float: { using scalar_t = float; __VA_ARGS__(); break; }             \
            case at::ScalarType::Double: { using scalar_t = double; __VA_ARGS__(); break; }           \
            default: throw std::runtime_error(std::string(NAME) + ": unsupported scalar type");       \
        }                                                                                             \
    } while (0)

#endif
