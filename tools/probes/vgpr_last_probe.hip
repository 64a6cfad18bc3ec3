// Probe (gfx950): the LAST register of a wave's VGPR allocation as a 32-bit source of a 64-bit VALU instruction.
// Background (EXPERIMENTS.md round 6): k_grid_backward_accumulate built with exactly 40 VGPRs, v39 = shift amount of
//   v_lshlrev_b64 v[32:33], v39, v[32:33]
// gave wrong sums (60-200 table entries per launch); the same code bytes with a 48-register allocation (descriptor patched), or with v36 in
// place of v39 (two instructions patched), are right.  This program isolates it: the operand register holds the right value (read back with
// v_mov_b32), the instruction's result is wrong.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/vgpr_last_probe.hip -o vgpr_last && ./vgpr_last
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

struct Result { unsigned long long wrong, wrong_reg, sample_got, sample_want; uint32_t sample_amount, sample_set; };

// OP: 0 v_lshlrev_b64, 1 v_lshrrev_b64, 2 v_ashrrev_i64, 3 v_mad_u64_u32 (src0 = the register, src1 = 3, src2 = the pair), 4 v_lshlrev_b32 (control),
//     5 v_ldexp_f64 (exponent), 6 v_cvt_f64_u32, 7 v_cvt_f64_f32
// LAST: the register that holds the 32-bit operand; TOP: highest register the kernel touches (allocation = TOP + 1 rounded up to 8)
#define BODY(OPSTR, REG, TOPREG)                                                                                                              \
    asm volatile("v_mov_b32 v32, %[lo]\n\tv_mov_b32 v33, %[hi]\n\tv_mov_b32 " TOPREG ", 0\n\tv_mov_b32 " REG ", %[amt]\n\tv_nop\n\tv_nop\n\t" \
                 OPSTR "\n\tv_mov_b32 %[rlo], v32\n\tv_mov_b32 %[rhi], v33\n\tv_mov_b32 %[rreg], " REG                                        \
                 : [rlo] "=v"(rlo), [rhi] "=v"(rhi), [rreg] "=v"(rreg) : [amt] "v"(amount), [lo] "v"(lo), [hi] "v"(hi)                        \
                 : "s8", "s9", "vcc", "v32", "v33", REG, TOPREG)

template <int OP, int LAST, int TOP>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_last(Result* __restrict__ res, uint32_t iters) {
    unsigned long long wrong = 0ull, wrong_reg = 0ull;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t h = (threadIdx.x * 2654435761u) ^ (it * 40503u) ^ (blockIdx.x * 7919u);
        const uint32_t amount = (h >> 3) % 23u;
        const uint32_t lo = h | 1u, hi = (uint32_t)((int32_t)(h << 7) >> 31) & 0x7fffffffu;
        uint32_t rlo, rhi, rreg;
#define CASE(L, T, RS, TS)                                                                                                                     \
        if (LAST == L && TOP == T) {                                                                                                            \
            if (OP == 0) BODY("v_lshlrev_b64 v[32:33], " RS ", v[32:33]", RS, TS);                                                              \
            if (OP == 1) BODY("v_lshrrev_b64 v[32:33], " RS ", v[32:33]", RS, TS);                                                              \
            if (OP == 2) BODY("v_ashrrev_i64 v[32:33], " RS ", v[32:33]", RS, TS);                                                              \
            if (OP == 3) BODY("v_mad_u64_u32 v[32:33], s[8:9], " RS ", 3, v[32:33]", RS, TS);                                                   \
            if (OP == 4) BODY("v_lshlrev_b32 v32, " RS ", v32", RS, TS);                                                                        \
            if (OP == 5) BODY("v_ldexp_f64 v[32:33], v[32:33], " RS, RS, TS);                                                                   \
            if (OP == 6) BODY("v_cvt_f64_u32 v[32:33], " RS, RS, TS);                                                                           \
            if (OP == 7) BODY("v_cvt_f64_f32 v[32:33], " RS, RS, TS);                                                                           \
        }
        CASE(39, 39, "v39", "v39")
        CASE(39, 47, "v39", "v47")
        CASE(38, 39, "v38", "v39")
        CASE(37, 39, "v37", "v39")
        CASE(47, 47, "v47", "v47")
        CASE(63, 63, "v63", "v63")
        const unsigned long long x = ((unsigned long long)hi << 32) | lo;
        unsigned long long want = OP == 0 ? x << amount : OP == 1 ? x >> amount : OP == 2 ? (unsigned long long)((long long)x >> amount)
                                : OP == 3 ? x + 3ull * amount : (((unsigned long long)hi << 32) | (uint32_t)(lo << amount));
        if (OP == 5) want = __builtin_bit_cast(unsigned long long, __builtin_ldexp(__builtin_bit_cast(double, x), (int)amount));
        if (OP == 6) want = __builtin_bit_cast(unsigned long long, (double)amount);
        if (OP == 7) want = __builtin_bit_cast(unsigned long long, (double)__builtin_bit_cast(float, amount));
        const unsigned long long got = ((unsigned long long)rhi << 32) | rlo;
        if (got != want) {
            wrong++;
            if (atomicCAS(&res->sample_set, 0u, 1u) == 0u) { res->sample_got = got; res->sample_want = want; res->sample_amount = amount; }
        }
        wrong_reg += rreg != amount;
    }
    if (wrong) atomicAdd(&res->wrong, wrong);
    if (wrong_reg) atomicAdd(&res->wrong_reg, wrong_reg);
}

template <int OP, int LAST, int TOP>
static int run(const char* op) {
    Result* d;
    CHECK(hipMalloc(&d, sizeof(Result)));
    CHECK(hipMemset(d, 0, sizeof(Result)));
    hipLaunchKernelGGL((k_last<OP, LAST, TOP>), dim3(1024), dim3(1024), 0, 0, d, 32u);
    CHECK(hipDeviceSynchronize());
    Result h;
    CHECK(hipMemcpy(&h, d, sizeof(Result), hipMemcpyDeviceToHost));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_last<OP, LAST, TOP>)));
    printf("%-14s 32-bit operand in v%d, kernel allocates %2d VGPRs: wrong results %9llu of %llu (operand register read back wrong: %llu)", op, LAST, fa.numRegs,
           h.wrong, 1024ull * 1024ull * 32ull, h.wrong_reg);
    if (h.wrong) printf("   e.g. operand %u: got %016llx, want %016llx", h.sample_amount, h.sample_got, h.sample_want);
    printf("\n");
    (void)hipFree(d);
    return 0;
}


// In-range accesses that END at the last register (no over-fetch expected): the 64-bit shift's data pair in v[38:39] with the amount elsewhere,
// and a 16-byte global store of v[36:39] (register tuples are 64-bit aligned on this chip: a 12-byte tuple cannot end at the last register).
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tail_ok(unsigned long long* __restrict__ bad, uint32_t* __restrict__ buf, uint32_t iters) {
    unsigned long long wrong = 0ull;
    uint32_t* mine = buf + 4u * (blockIdx.x * 1024u + threadIdx.x);
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t h = (threadIdx.x * 2654435761u) ^ (it * 40503u) ^ (blockIdx.x * 7919u);
        const uint32_t amount = (h >> 3) % 23u, lo = h | 1u, hi = (h >> 9) & 0xffu;
        uint32_t rlo, rhi;
        asm volatile("v_mov_b32 v38, %[lo]\n\tv_mov_b32 v39, %[hi]\n\tv_mov_b32 v32, %[amt]\n\tv_nop\n\tv_nop\n\t"
                     "v_lshlrev_b64 v[38:39], v32, v[38:39]\n\tv_mov_b32 %[rlo], v38\n\tv_mov_b32 %[rhi], v39\n\t"
                     "v_mov_b32 v36, %[lo]\n\tv_mov_b32 v37, %[lo]\n\tv_mov_b32 v38, %[hi]\n\tv_mov_b32 v39, %[amt]\n\tv_nop\n\t"
                     "global_store_dwordx4 %[p], v[36:39], off\n\ts_waitcnt vmcnt(0)"
                     : [rlo] "=&v"(rlo), [rhi] "=&v"(rhi) : [amt] "v"(amount), [lo] "v"(lo), [hi] "v"(hi), [p] "v"(mine)
                     : "memory", "v32", "v36", "v37", "v38", "v39");
        const unsigned long long want = ((((unsigned long long)hi << 32) | lo) << amount), got = ((unsigned long long)rhi << 32) | rlo;
        wrong += got != want;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const volatile uint32_t* v = mine;
        wrong += (v[0] != lo || v[1] != lo || v[2] != hi || v[3] != amount) ? 1ull << 32 : 0ull;
    }
    if (wrong) atomicAdd(bad, wrong);
}

static int run_tail_ok() {
    unsigned long long* bad;
    uint32_t* buf;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&buf, 4ull * 1024 * 1024 * 4));
    CHECK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_tail_ok, dim3(1024), dim3(1024), 0, 0, bad, buf, 32u);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_tail_ok)));
    printf("in range, ending at the last register (%d VGPRs): v_lshlrev_b64 v[38:39], v32, v[38:39]: %llu wrong; global_store_dwordx4 v[36:39]: %llu wrong of %llu\n",
           fa.numRegs, h & 0xffffffffull, h >> 32, 1024ull * 1024ull * 32ull);
    (void)hipFree(bad);
    (void)hipFree(buf);
    return 0;
}

int main() {
    run<0, 39, 39>("v_lshlrev_b64"); run<0, 39, 47>("v_lshlrev_b64"); run<0, 38, 39>("v_lshlrev_b64"); run<0, 37, 39>("v_lshlrev_b64");
    run<0, 47, 47>("v_lshlrev_b64"); run<0, 63, 63>("v_lshlrev_b64");
    run<1, 39, 39>("v_lshrrev_b64"); run<1, 39, 47>("v_lshrrev_b64");
    run<2, 39, 39>("v_ashrrev_i64"); run<2, 39, 47>("v_ashrrev_i64");
    run<3, 39, 39>("v_mad_u64_u32"); run<3, 39, 47>("v_mad_u64_u32");
    run<4, 39, 39>("v_lshlrev_b32"); run<4, 39, 47>("v_lshlrev_b32");
    run<5, 39, 39>("v_ldexp_f64"); run<5, 39, 47>("v_ldexp_f64");
    run<6, 39, 39>("v_cvt_f64_u32"); run<6, 39, 47>("v_cvt_f64_u32");
    run<7, 39, 39>("v_cvt_f64_f32"); run<7, 39, 47>("v_cvt_f64_f32");
    run_tail_ok();
    return 0;
}
