// Scatter-throughput probe #2 for gfx950: what bounds the hash-grid backward scatter?
//   op      : pk_add_f16 atomic | add_f32 atomic | plain 4-B store | plain 8-B store | 4-B gather load | LDS ds_add_f32
//   table   : 256 KiB, 2 MiB, 24 MiB (fits L2 / one XCD L2 / only MALL)
//   pattern : random word | 16 consecutive lanes share one 64-B line | fully coalesced
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hash3(uint32_t i) { uint32_t x = i * 2654435761u; x ^= x >> 15; x *= 805459861u; x ^= x >> 13; return x; }

template <int OP, int PAT>
__global__ void k_scatter(uint32_t* table, uint32_t n_words, uint32_t n_ops, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    uint32_t idx;
    if (PAT == 0) idx = hash3(i) % n_words;
    else if (PAT == 1) idx = ((hash3(i >> 4) % (n_words >> 4)) << 4) + (i & 15);
    else idx = i % n_words;
    if (OP == 0) {
        half2_t v = {(_Float16)1.0f, (_Float16)0.5f};
        (void)__builtin_amdgcn_flat_atomic_fadd_v2f16(reinterpret_cast<half2_t*>(table + idx), v);
    } else if (OP == 1) {
        unsafeAtomicAdd(reinterpret_cast<float*>(table + idx), 1.0f);
    } else if (OP == 2) {
        table[idx] = i;
    } else if (OP == 3) {
        reinterpret_cast<uint2*>(table)[idx >> 1] = make_uint2(i, i);
    } else if (OP == 4) {
        uint32_t v = table[idx];
        if (v == 0xdeadbeefu) sink[0] = v;
    }
}

__global__ void k_lds_atomic(uint32_t* sink, uint32_t words, uint32_t per_thread) {
    extern __shared__ float lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; k++) {
        s = hash3(s + k);
        atomicAdd(&lds[s % words], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0 && lds[0] == -1.f) sink[0] = 1;
}

template <int OP, int PAT>
static float run(uint32_t* table, uint32_t n_words, uint32_t n_ops, uint32_t* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_scatter<OP, PAT>), dim3((n_ops + 255) / 256), dim3(256), 0, 0, table, n_words, n_ops, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t n_ops = 1u << 25;
    uint32_t *table, *sink;
    hipMalloc(&table, 64u << 20); hipMalloc(&sink, 64);
    hipMemset(table, 0, 64u << 20);
    const char* ops[] = {"pk_add_f16", "add_f32", "store4", "store8", "load4"};
    const char* pats[] = {"random", "16-lane-line", "coalesced"};
    const uint32_t sizes[] = {64u << 10, 512u << 10, 6u << 20};  // words: 256 KiB, 2 MiB, 24 MiB
    for (int op = 0; op < 5; op++)
        for (int pat = 0; pat < 3; pat++)
            for (int s = 0; s < 3; s++) {
                float ms = 0;
#define R(O, P) if (op == O && pat == P) ms = run<O, P>(table, sizes[s], n_ops, sink);
                R(0,0) R(0,1) R(0,2) R(1,0) R(1,1) R(1,2) R(2,0) R(2,1) R(2,2) R(3,0) R(3,1) R(3,2) R(4,0) R(4,1) R(4,2)
                printf("%-10s %-13s table %6u KiB : %8.3f ms  %8.2f Gop/s\n", ops[op], pats[pat], sizes[s] / 256, ms, n_ops / ms / 1e6);
            }
    // LDS atomics: 1024 blocks x 256 threads x 256 ops on 32K words (128 KiB)
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_atomic), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        for (int threads = 256; threads <= 1024; threads *= 2) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_lds_atomic, dim3(1024), dim3(threads), 128 << 10, 0, sink, 32768u, 256u);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("ds_add_f32 random 128 KiB, %4d threads/block: %8.3f ms  %8.2f Gop/s\n", threads, best, 1024.0 * threads * 256 / best / 1e6);
        }
    }
    return 0;
}
