#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xF, 0xF, false);  // row_bcast:15
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);  // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);  // wave_shl:1
    int d = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0x4, 0xF, false);  // row_bcast:15 into row 2 only
    int e = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xF, 0xF, false);  // row_bcast:31
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d; out[256 + threadIdx.x] = e;
}
int main() {
    int* d; hipMalloc(&d, 320 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"row_bcast15", "wave_shr1", "wave_shl1", "row_bcast15 mask4", "row_bcast31"};
    for (int r = 0; r < 5; r++) { printf("%-18s", names[r]); for (int i = 0; i < 64; i++) printf("%d ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
