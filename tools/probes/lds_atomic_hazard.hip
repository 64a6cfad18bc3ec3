// Probe (gfx950): two hypotheses that were EXCLUDED while chasing the irreproducible build of k_grid_backward_accumulate (EXPERIMENTS.md round 6;
// the cause turned out to be the 64-bit shift of tools/probes/vgpr_last_probe.hip).  Both came back clean on MI355X, in every combination:
//  1. a VALU write of the DATA registers of a 64-bit LDS atomic in the very next instruction (the compiler's hazard recogniser only covers LDS
//     data wider than 64 bits).  The kernel's instruction sequence in isolation:
//         v_lshlrev_b64 d, s0, a ; ds_add_u64 addr, d ; v_lshlrev_b64 d, s1, b ; ds_add_u64 addr, d offset:8
//     FLAGS: 1 = separate data pairs, 2 = 12-byte global loads in flight during the sequence (the kernel's record prefetch), 4 = the sequence
//     under a partial EXEC mask, 8 = other LDS traffic of the wave in front, 16 / 32 = the register numbers of the failing / the shipped build;
//  2. a VALU write of v32 while LDS reads into v33 / v34 are still outstanding (k_probe_return).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomic_hazard.hip -o lds_hazard && ./lds_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

__host__ __device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int FLAGS>
__global__ __launch_bounds__(1024) void k_probe(unsigned long long* __restrict__ out, const uint32_t* __restrict__ gbuf, uint32_t iters,
                                                 uint32_t n_slots) {
    extern __shared__ unsigned long long acc[];   // [n_slots][2], then 1024 words of per-lane scratch
    volatile uint32_t* side = reinterpret_cast<volatile uint32_t*>(acc + 2 * n_slots);
    for (uint32_t i = threadIdx.x; i < 2 * n_slots; i += blockDim.x) acc[i] = 0ull;
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    uint32_t nxt[3] = {0u, 0u, 0u};
    if (FLAGS & 2) __builtin_memcpy(nxt, gbuf + 3u * ((blockIdx.x * 1024u + tid) % 4096u), 12);
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t h = mix(tid * 131u + it * 7919u + blockIdx.x * 104729u);
        uint32_t extra = 0u, bump = 0u;
        if (FLAGS & 2) {   // the values loaded one trip ago feed this trip; the next load is in flight while the atomics issue
            extra = nxt[0] ^ nxt[1] ^ nxt[2];
            __builtin_memcpy(nxt, gbuf + 3u * ((h >> 3) % 4096u), 12);
        }
        if (FLAGS & 8) {
            side[tid] = h | 1u;
            bump += side[tid ^ 1u] == 0u ? 1u : 0u;   // (never: every stored word is odd)
        }
        if (FLAGS & 2) bump += extra == 0x5a5a5a5au ? 0u : 1u;   // (never: the buffer holds that byte pattern)
        const unsigned long long a = (unsigned long long)((h & 0xffffu) + 1u + bump), b = (unsigned long long)((h >> 16) + 1u);
        const uint32_t s0 = (h & 1u) ? 21u : 0u, s1 = (h & 2u) ? 21u : 0u;
        const uint32_t slot = (h >> 4) % n_slots;
        const uint32_t addr = slot * 16u;   // (the dynamic array starts at LDS address 0: no static LDS in this kernel)
        const bool on = !(FLAGS & 4) || ((h >> 9) & 3u) != 0u;
        if (on) {
            if (FLAGS & 16) {
                // the failing build's registers: address v6 (bank 2), data pair v[32:33] (banks 0, 1), shift amounts v13 / v12, second source v[34:35]
                asm volatile("v_mov_b32 v6, %0\n\tv_mov_b32 v12, %1\n\tv_mov_b32 v13, %3\n\t"
                             "v_mov_b32 v32, %2\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, %4\n\tv_mov_b32 v35, 0\n\t"
                             "v_lshlrev_b64 v[32:33], v12, v[32:33]\n\tv_nop\n\tv_nop\n\t"
                             "ds_add_u64 v6, v[32:33]\n\tv_lshlrev_b64 v[32:33], v13, v[34:35]\n\tds_add_u64 v6, v[32:33] offset:8"
                             :: "v"(addr), "v"(s0), "v"((uint32_t)a), "v"(s1), "v"((uint32_t)b) : "memory", "v6", "v12", "v13", "v32", "v33", "v34", "v35");
            } else if (FLAGS & 32) {
                // the shipped build's registers: address v6 (bank 2), data pair v[30:31] (banks 2, 3), second source v[32:33], shift amount v11
                asm volatile("v_mov_b32 v6, %0\n\tv_mov_b32 v12, %1\n\tv_mov_b32 v11, %3\n\t"
                             "v_mov_b32 v30, %2\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, %4\n\tv_mov_b32 v33, 0\n\t"
                             "v_lshlrev_b64 v[30:31], v12, v[30:31]\n\tv_nop\n\tv_nop\n\t"
                             "ds_add_u64 v6, v[30:31]\n\tv_lshlrev_b64 v[30:31], v11, v[32:33]\n\tds_add_u64 v6, v[30:31] offset:8"
                             :: "v"(addr), "v"(s0), "v"((uint32_t)a), "v"(s1), "v"((uint32_t)b) : "memory", "v6", "v11", "v12", "v30", "v31", "v32", "v33");
            } else if (FLAGS & 1) {
                unsigned long long d0, d1;
                asm volatile("v_lshlrev_b64 %0, %3, %4\n\tv_lshlrev_b64 %1, %5, %6\n\tds_add_u64 %2, %0\n\tds_add_u64 %2, %1 offset:8\n\ts_nop 1"
                             : "=&v"(d0), "=&v"(d1) : "v"(addr), "v"(s0), "v"(a), "v"(s1), "v"(b) : "memory");
            } else {
                unsigned long long d;
                asm volatile("v_lshlrev_b64 %0, %2, %3\n\tds_add_u64 %1, %0\n\tv_lshlrev_b64 %0, %4, %5\n\tds_add_u64 %1, %0 offset:8"
                             : "=&v"(d) : "v"(addr), "v"(s0), "v"(a), "v"(s1), "v"(b) : "memory");
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * n_slots; i += blockDim.x) out[(size_t)blockIdx.x * 2 * n_slots + i] = acc[i];
}

template <int FLAGS>
static int run(uint32_t blocks, uint32_t iters, uint32_t n_slots, const uint32_t* gbuf) {
    unsigned long long* out;
    const size_t n = (size_t)blocks * 2 * n_slots;
    CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMemset(out, 0xff, n * 8));
    hipLaunchKernelGGL((k_probe<FLAGS>), dim3(blocks), dim3(1024), 2 * n_slots * 8 + 4096, 0, out, gbuf, iters, n_slots);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> got(n), want(n, 0ull);
    CHECK(hipMemcpy(got.data(), out, n * 8, hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < blocks; b++)
        for (uint32_t tid = 0; tid < 1024; tid++)
            for (uint32_t it = 0; it < iters; it++) {
                const uint32_t h = mix(tid * 131u + it * 7919u + b * 104729u);
                if ((FLAGS & 4) && ((h >> 9) & 3u) == 0u) continue;
                const uint32_t slot = (h >> 4) % n_slots;
                want[((size_t)b * n_slots + slot) * 2] += (unsigned long long)((h & 0xffffu) + 1u) << ((h & 1u) ? 21 : 0);
                want[((size_t)b * n_slots + slot) * 2 + 1] += (unsigned long long)((h >> 16) + 1u) << ((h & 2u) ? 21 : 0);
            }
    size_t bad0 = 0, bad1 = 0;
    for (size_t i = 0; i < n; i += 2) { bad0 += got[i] != want[i]; bad1 += got[i + 1] != want[i + 1]; }
    printf("flags %2d (%s%s%s%s) blocks %4u slots %4u: wrong words: first atomic %zu, second atomic %zu of %zu each\n", FLAGS,
           (FLAGS & 16) ? "regs-of-failing-build " : (FLAGS & 32) ? "regs-of-shipped-build " : (FLAGS & 1) ? "own-regs " : "overwritten ", (FLAGS & 2) ? "+vmem " : "", (FLAGS & 4) ? "+partial-exec " : "", (FLAGS & 8) ? "+lds-traffic" : "",
           blocks, n_slots, bad0, bad1, n / 2);
    (void)hipFree(out);
    return (int)(bad0 + bad1);
}


// Second question: a VALU write of v32 while LDS reads into v33 / v34 are still outstanding (the failing build converts channel 0's addend
// into v32 BEFORE its s_waitcnt lgkmcnt(0); the cured builds happen to wait first).  Does v32 survive?
__global__ __launch_bounds__(1024) void k_probe_return(unsigned long long* __restrict__ mismatches, uint32_t iters, uint32_t n_slots) {
    extern __shared__ unsigned long long acc[];
    uint32_t* table = reinterpret_cast<uint32_t*>(acc + 2 * n_slots);   // [2048] words read back by the ds_reads
    for (uint32_t i = threadIdx.x; i < 2 * n_slots; i += blockDim.x) acc[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) table[i] = 0xabcd0000u + i;
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    unsigned long long bad = 0ull;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t h = mix(tid * 131u + it * 7919u + blockIdx.x * 104729u);
        const float x = (float)(int)((h & 0xfffffu) - 0x80000) * 0.03125f;   // |x| < 16384, a multiple of 2^-5
        const float m = (h & 0x100000u) ? 8.0f : 256.0f;
        const uint32_t hw = h;
        const uint32_t ra = (2 * n_slots) * 8u + ((tid * 2u) & 2047u) * 4u;    // table[2 tid], table[2 tid + 1]
        const uint32_t aa = ((h >> 4) % n_slots) * 16u;
        unsigned long long one = 1ull;
        uint32_t lo, hi, r33, r34;
        asm volatile("ds_add_u64 %[aa], %[one]\n\tds_add_u64 %[aa], %[one] offset:8\n\t"      // LDS work queued in front of the reads
                     "ds_read_b32 v34, %[ra]\n\tds_read_b32 v33, %[ra] offset:4\n\t"
                     "v_mul_f32 v13, %[m], %[x]\n\t"
                     "v_cvt_i32_f32 v32, v13\n\t"
                     "v_cvt_f32_f16_sdwa v13, %[hw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_mov_b32 %[r33], v33\n\tv_mov_b32 %[r34], v34\n\t"
                     "v_ashrrev_i32 v33, 31, v32\n\t"
                     "v_mov_b32 %[lo], v32\n\tv_mov_b32 %[hi], v33"
                     : [lo] "=v"(lo), [hi] "=v"(hi), [r33] "=v"(r33), [r34] "=v"(r34)
                     : [aa] "v"(aa), [one] "v"(one), [ra] "v"(ra), [m] "v"(m), [x] "v"(x), [hw] "v"(hw)
                     : "memory", "v13", "v32", "v33", "v34");
        const int want = (int)(m * x);
        bad += (int)lo != want || (int)hi != (want >> 31) || r34 != 0xabcd0000u + ((tid * 2u) & 2047u) || r33 != 0xabcd0000u + ((tid * 2u) & 2047u) + 1u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

static int run_return(uint32_t blocks, uint32_t iters, uint32_t n_slots) {
    unsigned long long* d;
    CHECK(hipMalloc(&d, 8));
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_probe_return, dim3(blocks), dim3(1024), 2 * n_slots * 8 + 8192, 0, d, iters, n_slots);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("VALU write of v32 under outstanding LDS reads into v33/v34: blocks %4u slots %4u: %llu mismatches of %llu\n", blocks, n_slots, h,
           (unsigned long long)blocks * 1024ull * iters);
    (void)hipFree(d);
    return (int)h;
}

int main() {
    for (uint32_t n_slots : {32u, 4096u}) for (uint32_t blocks : {512u, 4096u}) run_return(blocks, 256, n_slots);
    uint32_t* gbuf;
    CHECK(hipMalloc(&gbuf, 3 * 4096 * 4));
    CHECK(hipMemset(gbuf, 0x5a, 3 * 4096 * 4));
    for (uint32_t n_slots : {32u, 4096u}) {
        for (uint32_t blocks : {512u, 4096u}) {
            run<0>(blocks, 96, n_slots, gbuf);
            run<2>(blocks, 96, n_slots, gbuf);
            run<4>(blocks, 96, n_slots, gbuf);
            run<8>(blocks, 96, n_slots, gbuf);
            run<14>(blocks, 96, n_slots, gbuf);
            run<15>(blocks, 96, n_slots, gbuf);
            run<16>(blocks, 96, n_slots, gbuf);
            run<30>(blocks, 96, n_slots, gbuf);
            run<32>(blocks, 96, n_slots, gbuf);
            run<46>(blocks, 96, n_slots, gbuf);
        }
    }
    return 0;
}
