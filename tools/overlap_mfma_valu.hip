// Do the matrix pipe and the vector pipe of a gfx950 SIMD overlap?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/overlap_mfma_valu.hip -o tools/overlap_mfma_valu && tools/overlap_mfma_valu
// One loop iteration = 8 x v_mfma_f32_32x32x16_f16 (two accumulators alternating: the network kernels' hidden layer) and / or NV packed
// 16-bit vector instructions on eight other registers (nothing depends on the MFMA results), in exact program order (asm volatile):
//   mode 0: MFMAs only          mode 1: vector only          mode 2: 8 MFMAs, then NV vector          mode 3: (1 MFMA, NV / 8 vector) x 8
// for 1 .. 4 waves per SIMD.  Reported: shader cycles per iteration and SIMD (wall time x 2.4 GHz / iterations), i.e. 256 = the MFMAs alone.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VOP(r) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(c))

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8_t a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    float16_t acc0, acc1;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    unsigned v[8], c = threadIdx.x;
    for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 7 + j;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) { MFMA(acc0); MFMA(acc1); }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < NV; q++) VOP(v[q & 7]);
        }
        if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q & 1) MFMA(acc1); else MFMA(acc0);
#pragma unroll
                for (int e = 0; e < NV / 8; e++) VOP(v[(q * (NV / 8) + e) & 7]);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    unsigned x = 0;
    for (int j = 0; j < 8; j++) x ^= v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)x;
}

template <int MODE, int NV>
static void run(int waves_per_simd, float* out) {
    const int iters = 4000, blocks = 256 * waves_per_simd;   // a 256-thread workgroup = one wave per SIMD of its CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles the SIMD spent per iteration of ALL its waves: wall / iterations, at 2.4 GHz
    printf("mode %d  vector ops/iteration %3d  waves/SIMD %d : %7.1f us, %6.1f cycles per iteration and SIMD (per wave-iteration %6.1f)\n", MODE, NV,
           waves_per_simd, ms * 1e3, ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 4; w++) {
        run<0, 32>(w, out);
        run<1, 32>(w, out);
        run<2, 32>(w, out);
        run<3, 32>(w, out);
        run<1, 64>(w, out);
        run<2, 64>(w, out);
        run<3, 64>(w, out);
    }
    return 0;
}
