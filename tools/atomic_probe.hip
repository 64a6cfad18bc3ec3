// Atomic-throughput probe for gfx950: how fast are packed-fp16 / fp32 global atomics on a 24 MB table with
// hash-grid-like random addresses, by memory scope (sc1 = device scope goes to the memory side; no sc bits = executed
// in the issuing XCD's L2), and are per-XCD private copies + L2-scope atomics correct?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hash3(uint32_t i) { uint32_t x = i * 2654435761u; x ^= x >> 15; x *= 805459861u; x ^= x >> 13; return x; }

template <int MODE>
__global__ void k_atomics(uint32_t* table, uint32_t n_entries, uint32_t n_ops, uint32_t copies_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    const uint32_t idx = hash3(i) % n_entries;
    half2_t v = {(_Float16)1.0f, (_Float16)0.5f};
    if (MODE == 0) {  // builtin (what unsafeAtomicAdd(half2) uses)
        (void)__builtin_amdgcn_flat_atomic_fadd_v2f16(reinterpret_cast<half2_t*>(table + idx), v);
    } else if (MODE == 1) {  // fp32 agent scope
        __hip_atomic_fetch_add(reinterpret_cast<float*>(table + idx), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {  // fp32 workgroup scope
        __hip_atomic_fetch_add(reinterpret_cast<float*>(table + idx), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 3) {  // pk_f16, no scope bits, into the XCD-private copy
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        uint32_t* p = table + (size_t)xcc * copies_stride + idx;
        uint32_t data = __builtin_bit_cast(uint32_t, v);
        asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(p), "v"(data) : "memory");
    } else if (MODE == 4) {  // pk_f16 with sc1 (device scope) single copy
        uint32_t* p = table + idx;
        uint32_t data = __builtin_bit_cast(uint32_t, v);
        asm volatile("global_atomic_pk_add_f16 %0, %1, off sc1" ::"v"(p), "v"(data) : "memory");
    } else if (MODE == 5) {  // pk_f16 no scope bits single copy (INCORRECT across XCDs; rate only)
        uint32_t* p = table + idx;
        uint32_t data = __builtin_bit_cast(uint32_t, v);
        asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(p), "v"(data) : "memory");
    } else if (MODE == 6) {  // fp32 no scope bits, XCD-private
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        float* p = reinterpret_cast<float*>(table + (size_t)xcc * copies_stride + idx);
        float one = 1.0f;
        asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(one) : "memory");
    }
}

int main() {
    const uint32_t n_entries = 6119864, n_ops = 1u << 25;
    uint32_t* table;
    const size_t bytes = (size_t)n_entries * 4 * 8;
    hipMalloc(&table, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[] = {"builtin flat_atomic_fadd_v2f16", "fp32 fetch_add agent", "fp32 fetch_add workgroup", "asm pk_add_f16 no-sc XCD-private copies",
                           "asm pk_add_f16 sc1", "asm pk_add_f16 no-sc single copy", "asm add_f32 no-sc XCD-private copies"};
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(table, 0, bytes);
            hipDeviceSynchronize();
            hipEventRecord(a);
            dim3 g((n_ops + 255) / 256), t(256);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_atomics<0>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                case 1: hipLaunchKernelGGL(k_atomics<1>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                case 2: hipLaunchKernelGGL(k_atomics<2>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                case 3: hipLaunchKernelGGL(k_atomics<3>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                case 4: hipLaunchKernelGGL(k_atomics<4>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                case 5: hipLaunchKernelGGL(k_atomics<5>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
                default: hipLaunchKernelGGL(k_atomics<6>, g, t, 0, 0, table, n_entries, n_ops, n_entries); break;
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        // correctness: total of all (copies of) entry sums must equal n_ops (first component / fp32 value)
        std::vector<uint32_t> h((size_t)n_entries * 8);
        hipMemcpy(h.data(), table, bytes, hipMemcpyDeviceToHost);
        double total = 0;
        for (size_t i = 0; i < h.size(); i++) {
            if (mode == 1 || mode == 2 || mode == 6) total += __builtin_bit_cast(float, h[i]);
            else { half2_t v = __builtin_bit_cast(half2_t, h[i]); total += (float)v.x; }
        }
        printf("%-44s %8.3f ms  %7.2f G atomics/s   sum=%.0f (expected %u)%s\n", names[mode], best, n_ops / best / 1e6, total, n_ops,
               total == (double)n_ops ? "" : "  <-- MISMATCH");
    }
    return 0;
}
