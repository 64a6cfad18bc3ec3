// Hardware-layout probe for gfx950 (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o tools/probe_gfx950
//  1. verifies the operand/result lane layout of v_mfma_f32_32x32x16_f16 that ffmlp.hip relies on
//     (A: row = lane&31, B: column = lane&31, both: 8 consecutive k slots selected by lane>>5;
//      C/D: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5));
//  2. dumps what ds_read_b64_tr_b16 returns for a lane-linear address pattern (input to a later optimisation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

__global__ void k_mfma(const _Float16* A, const _Float16* B, float* C, int kmap) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    half8_t a, b;
    for (int j = 0; j < 8; j++) {
        const int k = kmap == 0 ? 8 * h + j : (j < 4 ? 4 * h + j : 8 + 4 * h + (j - 4));
        a[j] = A[i * 16 + k];   // A is 32x16 row-major
        b[j] = B[k * 32 + i];   // B is 16x32 row-major
    }
    float16_t c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[lane * 16 + r] = c[r];
}

__global__ void k_trread(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 8;  // lane-linear 8-byte chunks
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}

int main() {
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    srand(1);
    for (auto& v : A) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : B) v = (_Float16)(float)(rand() % 5 - 2);
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> C(64 * 16);
    for (int kmap = 0; kmap < 2; kmap++) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, kmap);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        int bad_assumed = 0, bad_transposed = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 16; r++) {
                const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float ref = 0, reft = 0;
                for (int k = 0; k < 16; k++) { ref += (float)A[row * 16 + k] * (float)B[k * 32 + col]; reft += (float)A[col * 16 + k] * (float)B[k * 32 + row]; }
                if (C[lane * 16 + r] != ref) bad_assumed++;
                if (C[lane * 16 + r] != reft) bad_transposed++;
            }
        printf("MFMA 32x32x16 f16, k-map %d: mismatches with assumed C layout = %d, with transposed layout = %d\n", kmap, bad_assumed, bad_transposed);
    }
    unsigned short* dO; hipMalloc(&dO, 64 * 4 * 2);
    hipLaunchKernelGGL(k_trread, dim3(1), dim3(64), 0, 0, dO);
    std::vector<unsigned short> O(256);
    hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, lane l reads 8 bytes at lds + 8*l; returned element indices per lane:\n");
    for (int l = 0; l < 64; l++) printf("  lane %2d: %4d %4d %4d %4d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
    return 0;
}
