// Multiresolution hash / tiled grid encoder for gfx950 (MI355X).
//
// Behaviour restated from the reference kernels (paths relative to the reference checkout):
//   forward   gridencoder/src/gridencoder.cu:87-245   (kernel_grid)
//   backward  gridencoder/src/gridencoder.cu:248-340  (kernel_grid_backward)
//   dL/dx     gridencoder/src/gridencoder.cu:343-369  (kernel_input_backward)
//   TV grad   gridencoder/src/gridencoder.cu:506-610  (kernel_grad_tv)
//
// MI355X design:
//   * forward: XCD-aware level placement (a level's table lives in ONE XCD's L2), the two first-coordinate corners of a point
//     on neighbouring lanes (same cache line -> one request), fp32 weights/accumulation, one rounding to the table dtype;
//   * backward: a global float atomic is a fabric operation here (~20 G requests/s chip-wide, one per (wave instruction, 64-byte
//     line), tools/atomic_probe2.hip).  fp16 C = 2 tables (the instant-ngp configuration) therefore take an atomic-free path: every
//     contribution becomes an 8-byte record, a workgroup counting-sorts its records by table slice in LDS and streams them into its
//     own chunk, and one workgroup per slice sums them EXACTLY in a 64-bit fixed-point LDS accumulator (integer LDS atomics are
//     14-22x faster than float ones) and rounds once: 2.2 ms -> 0.40 ms (best atomic kernel) -> 0.16 ms, bit-reproducible.  Other
//     dtypes / shapes / small batches use the atomic kernel (corner pairs on adjacent lanes, parity slots, wave-wide DPP run merge);
//   * the per-level scale/resolution table is computed on the host with a reproducible recipe (ngp_grid_level_table) and
//     passed by value, so cell indices are bit-identical to the oracle; all per-level quantities are wave-uniform (SGPRs).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace ngp {

struct GridLevels {
    float scale[NGP_MAX_LEVELS];
    uint32_t res[NGP_MAX_LEVELS];
};

__constant__ const uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                          2097192037u, 1434869437u, 2165219737u};

// Wave-uniform description of how a level is indexed (gridencoder.cu:66-84, get_grid_index).
template <int D>
struct LevelIndexer {
    uint32_t stride[D];  // dense strides of the dims that take part (0 for the others)
    uint32_t size;       // hashmap_size
    uint32_t mask;       // size-1 if size is a power of two else 0
    bool hashed;
    bool need_mod;       // false when a dense index is provably < size

    __host__ __device__ __forceinline__ void init(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                                  uint32_t resolution) {
        uint32_t s = 1;
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s <= hashmap_size) {
                stride[d] = s;
                s *= align_corners ? resolution : (resolution + 1u);
            } else {
                stride[d] = 0;
            }
        }
        hashed = (gridtype == 0u) && (s > hashmap_size);
        // without align_corners every corner coordinate is <= resolution and the strides are powers of (resolution + 1):
        // a dense index over all D dims is < (resolution+1)^D <= size.  With align_corners the stride base is
        // `resolution` while a corner can sit AT `resolution`, so the index can wrap (gridencoder.cu:66-84).
        need_mod = hashed || (s > hashmap_size) || align_corners;
        size = hashmap_size;
        mask = ((hashmap_size & (hashmap_size - 1u)) == 0u) ? hashmap_size - 1u : 0u;
    }

    // The same index from per-dimension terms: term(d, c) for the lower vertex coordinate c, step(d) to get the upper one
    // ((c + 1) * k == c * k + k in uint32 arithmetic), combine() over one term per dimension.  A cell's 2^D corners then cost D
    // multiplications instead of D * 2^D (v_mul_lo_u32 is a quarter-rate instruction).
    __device__ __forceinline__ uint32_t term(int d, uint32_t c) const { return hashed ? c * kPrimes[d] : c * stride[d]; }
    __device__ __forceinline__ uint32_t step(int d) const { return hashed ? kPrimes[d] : stride[d]; }
    __device__ __forceinline__ uint32_t combine(const uint32_t (&t)[D]) const {
        uint32_t idx = 0;
        if (hashed) {
#pragma unroll
            for (int d = 0; d < D; d++) idx ^= t[d];
        } else {
#pragma unroll
            for (int d = 0; d < D; d++) idx += t[d];
        }
        if (!need_mod) return idx;
        return mask ? (idx & mask) : (idx % size);
    }

    __device__ __forceinline__ uint32_t operator()(const uint32_t (&pg)[D]) const {
        uint32_t idx = 0;
        if (hashed) {
#pragma unroll
            for (int d = 0; d < D; d++) idx ^= pg[d] * kPrimes[d];
        } else {
#pragma unroll
            for (int d = 0; d < D; d++) idx += pg[d] * stride[d];
        }
        if (!need_mod) return idx;
        return mask ? (idx & mask) : (idx % size);
    }
};

// gridencoder.cu:146-159: position inside the level.  Returns false when the point is outside [0,1]^D.
// Input mapping of the fused path: the module maps [-bound, bound] -> [0, 1] as (x + bound) * (1 / (2 bound)) in fp32
// (grid.py:149 through PyTorch's scalar-division kernel); InputMap{shift = bound, scale = 1/(2 bound)} reproduces those two
// roundings inside the kernel, scale == 0 means the inputs already are unit coordinates (the reference op contract).
struct InputMap {
    float shift, scale;
};

template <int D>
__device__ __forceinline__ bool locate(const float* __restrict__ x, float scale, bool align_corners, uint32_t interp,
                                       float (&frac)[D], float (&deriv)[D], uint32_t (&cell)[D], InputMap im = InputMap{0.0f, 0.0f}) {
    float xv[D];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; d++) {
        xv[d] = x[d];
        if (im.scale != 0.0f) xv[d] = (xv[d] + im.shift) * im.scale;
        inside = inside && !(xv[d] < 0.0f || xv[d] > 1.0f);
    }
    if (!inside) return false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        float p = __builtin_fmaf(xv[d], scale, align_corners ? 0.0f : 0.5f);
        float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1u) {
            deriv[d] = 6.0f * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            deriv[d] = 1.0f;
        }
        frac[d] = p;
    }
    return true;
}

template <typename T, int C>
struct Vec;  // C consecutive table features
template <int C>
struct Vec<float, C> {
    float v[C];
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int c = 0; c < C; c++) v[c] = p[c];
    }
};
template <>
struct Vec<float, 2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2_t t = *reinterpret_cast<const float2_t*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct Vec<float, 4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4_t t = *reinterpret_cast<const float4_t*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};
template <>
struct Vec<float, 8> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        float4_t a = *reinterpret_cast<const float4_t*>(p);
        float4_t b = *reinterpret_cast<const float4_t*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <>
struct Vec<half_t, 1> {
    float v[1];
    __device__ __forceinline__ void load(const half_t* p) { v[0] = (float)p[0]; }
};
template <>
struct Vec<half_t, 2> {
    float v[2];
    __device__ __forceinline__ void load(const half_t* p) {
        half2_t t = *reinterpret_cast<const half2_t*>(p);
        v[0] = (float)t.x; v[1] = (float)t.y;
    }
};
template <>
struct Vec<half_t, 4> {
    float v[4];
    __device__ __forceinline__ void load(const half_t* p) {
        half4_t t = *reinterpret_cast<const half4_t*>(p);
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
};
template <>
struct Vec<half_t, 8> {
    float v[8];
    __device__ __forceinline__ void load(const half_t* p) {
        half8_t t = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = (float)t[c];
    }
};

template <typename T, int C>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[C]) {
    if constexpr (sizeof(T) == 2 && C == 2) {
        half2_t t = {(half_t)v[0], (half_t)v[1]};
        *reinterpret_cast<half2_t*>(p) = t;
    } else if constexpr (sizeof(T) == 2 && C == 4) {
        half4_t t = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4_t*>(p) = t;
    } else if constexpr (sizeof(T) == 4 && C == 2) {
        float2_t t = {v[0], v[1]};
        *reinterpret_cast<float2_t*>(p) = t;
    } else if constexpr (sizeof(T) == 4 && C == 4) {
        float4_t t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4_t*>(p) = t;
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) p[c] = (T)v[c];
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
constexpr int FWD_THREADS = 256;

template <typename T, int D, int C, bool WITH_DYDX>
__global__ __launch_bounds__(FWD_THREADS) void k_grid_forward(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                              const int32_t* __restrict__ offsets, T* __restrict__ outputs,
                                                              uint32_t B, uint32_t L, GridLevels lv, T* __restrict__ dy_dx,
                                                              uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t level = blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, align_corners, hashmap_size, lv.res[level]);
    const T* __restrict__ table = grid + (size_t)off0 * C;

    for (uint32_t b = blockIdx.x * FWD_THREADS + threadIdx.x; b < B; b += gridDim.x * FWD_THREADS) {
        float frac[D], deriv[D];
        uint32_t cell[D];
        T* out = outputs + ((size_t)level * B + b) * C;
        T* dyo = WITH_DYDX ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] = 0.0f;
        if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, frac, deriv, cell)) {
            store_vec<T, C>(out, acc);
            if (WITH_DYDX) {
#pragma unroll
                for (int i = 0; i < D * C; i++) dyo[i] = (T)0.0f;
            }
            continue;
        }
        // gather all 2^D corners first (independent loads), then combine
        Vec<T, C> corner[1 << D];
#pragma unroll
        for (int k = 0; k < (1 << D); k++) {
            uint32_t pg[D];
#pragma unroll
            for (int d = 0; d < D; d++) pg[d] = cell[d] + ((k >> d) & 1);
            corner[k].load(table + (size_t)indexer(pg) * C);
        }
#pragma unroll
        for (int k = 0; k < (1 << D); k++) {
            float w = 1.0f;
#pragma unroll
            for (int d = 0; d < D; d++) w *= ((k >> d) & 1) ? frac[d] : (1.0f - frac[d]);
#pragma unroll
            for (int c = 0; c < C; c++) acc[c] = __builtin_fmaf(w, corner[k].v[c], acc[c]);
        }
        store_vec<T, C>(out, acc);

        if (WITH_DYDX) {
            // gridencoder.cu:201-244: d out / d x_g = scale * sum_{other corners} w_other * (v_right - v_left) * deriv_g
#pragma unroll
            for (int g = 0; g < D; g++) {
                float ga[C];
#pragma unroll
                for (int c = 0; c < C; c++) ga[c] = 0.0f;
#pragma unroll
                for (int k = 0; k < (1 << D); k++) {
                    if ((k >> g) & 1) continue;  // enumerate the "left" corners
                    float w = scale;
#pragma unroll
                    for (int d = 0; d < D; d++)
                        if (d != g) w *= ((k >> d) & 1) ? frac[d] : (1.0f - frac[d]);
#pragma unroll
                    for (int c = 0; c < C; c++)
                        ga[c] = __builtin_fmaf(w * deriv[g], corner[k | (1 << g)].v[c] - corner[k].v[c], ga[c]);
                }
#pragma unroll
                for (int c = 0; c < C; c++) dyo[g * C + c] = (T)ga[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, production variant (no dy_dx): XCD-aware level placement + corner pairs on neighbouring lanes
//
// * Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with a private 4 MiB L2.
//   With a plain (tile, level) grid every XCD streams every level's table through its own L2 (8 x 23 MiB of
//   fills per call).  Here block b serves level (b % 8) + 8 * k only, k advancing with b / 8: a level's table is
//   fetched into ONE L2, and an XCD works through its levels one after the other, so the live table (<= 2 MiB
//   fp16) always fits next to the point stream.
// * The vector memory pipeline pays per distinct cache line of a wave instruction (random 4-byte gathers run at
//   ~0.4 lanes/clk/CU, 16 lanes on a line >4x faster -- tools/atomic_probe2.hip).  Lanes 2p and 2p+1 fetch the two
//   corners of point p that differ in the first coordinate: neighbours in memory on dense levels and, because the
//   first hash prime is 1, also on hashed levels whenever x and x+1 share an aligned 16-entry block.  The two
//   half sums are combined with one DPP quad permute per channel.
// ------------------------------------------------------------------------------------------------
constexpr int FWD_PTS_PER_WAVE = 32;
// Work list per XCD.  Whole levels go to XCD (level mod 8) as described above; that alone leaves the XCDs unevenly loaded when the levels
// differ in cost -- on ray-ordered samples a level costs the more the finer it is (consecutive samples stop sharing cells and cache lines):
// measured 19 .. 46 us per level on one XCD, XCD 0 (levels 0, 8) done after 47 us, XCD 7 (levels 7, 15) after 70 us, and the launch lasts
// as long as the slowest XCD (profiles/r03_grid_forward_levels.txt).  With per-level costs from the caller the tail of the most loaded
// XCDs' last level is handed, tile range by tile range, to the least loaded ones.
constexpr int FWD_MAX_SEG = 8;
struct FwdSchedule {
    uint16_t level[8][FWD_MAX_SEG];
    uint32_t tile0[8][FWD_MAX_SEG];
    uint32_t end[8][FWD_MAX_SEG];   // slots of this XCD consumed after the segment (cumulative); unused segments repeat the last value
};

__device__ __forceinline__ float quad_swap1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <typename T, int D, int C>
__device__ __forceinline__ void forward_pair_block(const float* __restrict__ inputs, const T* __restrict__ table, T* __restrict__ olevel, float scale,
                                                   const LevelIndexer<D>& indexer, uint32_t hashmap_size, bool align_corners, uint32_t interp,
                                                   InputMap im, uint32_t b_begin, uint32_t b_end) {
    constexpr int NJ = 1 << (D - 1);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pl = lane >> 1;
    const uint32_t xb = lane & 1;
    for (uint32_t base = b_begin + wid * FWD_PTS_PER_WAVE; base < b_end; base += (FWD_THREADS / 64) * FWD_PTS_PER_WAVE) {
        const uint32_t b = base + pl;
        const bool in_range = b < b_end;
        float frac[D], deriv[D];
        uint32_t cell[D];
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] = 0.0f;
        if (in_range && locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, frac, deriv, cell, im)) {
            Vec<T, C> corner[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                uint32_t pg[D];
                pg[0] = cell[0] + xb;
#pragma unroll
                for (int d = 1; d < D; d++) pg[d] = cell[d] + ((j >> (d - 1)) & 1);
                NGP_BOUNDS(indexer(pg) < hashmap_size);
                corner[j].load(table + (size_t)indexer(pg) * C);
            }
            const float w0 = xb ? frac[0] : 1.0f - frac[0];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                float w = w0;
#pragma unroll
                for (int d = 1; d < D; d++) w *= ((j >> (d - 1)) & 1) ? frac[d] : (1.0f - frac[d]);
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = __builtin_fmaf(w, corner[j].v[c], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] += quad_swap1(acc[c]);  // all lanes execute (no cross-lane read under divergence)
        if (in_range && xb == 0) store_vec<T, C>(olevel + (size_t)b * C, acc);
    }
}

template <typename T, int D, int C>
__global__ __launch_bounds__(FWD_THREADS) void k_grid_forward_pair(const float* __restrict__ inputs, const T* __restrict__ grid_a,
                                                                   const int32_t* __restrict__ offsets, T* __restrict__ outputs,
                                                                   uint32_t B, uint32_t L, GridLevels lv, uint32_t gridtype,
                                                                   bool align_corners, uint32_t interp, FwdSchedule sched,
                                                                   uint32_t points_per_block, InputMap im, TableSel sel) {
    // double-buffered table (ngp_grid_encode_forward_sel): which copy is current is a device word, read once per workgroup (scalar load)
    const T* __restrict__ grid = sel.parity && sel.parity[0] != 0.0f ? reinterpret_cast<const T*>(sel.alt) : grid_a;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    uint32_t level = 0xffffu, tile = 0u, begin = 0u;
#pragma unroll
    for (int sg = 0; sg < FWD_MAX_SEG; sg++) {   // this XCD's work list: runs of consecutive tiles of one level, in order
        const uint32_t e = sched.end[xcd][sg];
        if (level == 0xffffu && slot < e) {
            level = sched.level[xcd][sg];
            tile = sched.tile0[xcd][sg] + (slot - begin);
        }
        begin = e;
    }
    NGP_BOUNDS(level < L || level == 0xffffu);
    if (level >= L) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, align_corners, hashmap_size, lv.res[level]);
    const T* __restrict__ table = grid + (size_t)off0 * C;
    T* __restrict__ olevel = outputs + (size_t)level * B * C;
    NGP_BOUNDS((uint64_t)tile * points_per_block < (uint64_t)B);  // every listed tile holds at least one point

    const uint32_t b_begin = tile * points_per_block;
    const uint32_t B_work = sel.rows ? min(B, sel.rows[0]) : B;   // (device-side row count: the rows behind it carry nothing)
    if (b_begin >= B_work) return;
    const uint32_t b_end = min(B_work, b_begin + points_per_block);
    forward_pair_block<T, D, C>(inputs, table, olevel, scale, indexer, hashmap_size, align_corners, interp, im, b_begin, b_end);
}

// ------------------------------------------------------------------------------------------------
// forward, round 5: the instant-ngp configuration (fp16 table, D = 3, C = 2, no align_corners) with the level's INDEX MODE resolved once per
// workgroup instead of per corner.
//
// What round 4's counters said about k_grid_forward_pair (profiles/r03_grid_forward_pmc.json, profiles/r03_grid_forward_levels.txt): 15.9 M
// VALU wave-instructions per launch = 26 us of pure issue in a 56-63 us kernel; a COARSE level alone on its XCD takes 19 us of which 13 us are
// VALU issue -- the coarse levels are bound by their instruction stream, not by the table reads (their few cells are L1-resident: staging
// them in LDS would change nothing -- see EXPERIMENTS.md round 5), the fine hashed levels by L2 -> L1 line traffic (4 lines per point).
// The generic kernel above spends its instructions on generality: both the stride- and the prime-products of every coordinate (v_mul_lo_u32
// is quarter rate), a scalar branch tree per corner for {hashed, dense} x {mask, modulo}, and the position work of a point twice (two lanes
// per point).  Here:
//   * DENSE levels (index = x + y s1 + z s2, provably < size: levels 0-4 of the lego table): ONE LANE PER POINT -- 64 points per wave
//     instruction stream instead of 32.  The two corners that differ in x are neighbours in memory, so the four (y, z) combinations are four
//     8-byte loads (dword-aligned global_load_dwordx2) that bring both: half the position / index instructions per point, the same number of
//     cache lines per point.  Summation order kept: (x-low chain over j) + (x-high chain over j), i.e. bit-identical with the pair kernel.
//   * HASHED levels with a power-of-two table (levels 5-15): the pair layout stays (4 lines per point is what the hash allows, and a lane
//     pair shares them), but the index is two multiplies, two adds and per corner two XORs + one AND; no mode branches.
//   * both: the position of the NEXT wave-step is requested before this step's corners are waited for.
// Any other level shape (tiled grids, align_corners, non-power-of-two hashed tables) runs the generic body.  Scheduling (per-XCD work
// lists) is unchanged.
// ------------------------------------------------------------------------------------------------
// position of one point for the fast paths: the load (issued early) and the arithmetic (locate's, expression for expression) are separate
struct FwdPos3 {
    float x[3];
    // unconditional (the index is clamped to the last point; a lane without a point discards what it read): a conditional request would make
    // the number of outstanding loads unknown to the compiler, which then waits for ALL of them -- the prefetched position included
    __device__ __forceinline__ void load(const float* __restrict__ inputs, uint32_t b, uint32_t last) {
        float t[3];
        __builtin_memcpy(t, inputs + (size_t)min(b, last) * 3, sizeof(t));  // 4-byte aligned: global_load_dwordx3
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2];
    }
};

#ifndef NGP_FWD_FAST
#define NGP_FWD_FAST 1
#endif
// (measured and dropped, EXPERIMENTS.md round 5: non-temporal table gathers on the fine hashed levels -- no effect: 58.0 / 58.1 vs 57.5 us)
template <bool SMOOTH /* interp == 1 (smoothstep) */, bool MAPPED /* inputs are world coordinates: InputMap */>
__global__ __launch_bounds__(FWD_THREADS) void k_grid_forward_fast(const float* __restrict__ inputs, const half_t* __restrict__ grid_a,
                                                                   const int32_t* __restrict__ offsets, half_t* __restrict__ outputs, uint32_t B,
                                                                   uint32_t L, GridLevels lv, uint32_t gridtype, uint32_t interp, FwdSchedule sched,
                                                                   uint32_t points_per_block, InputMap im, TableSel sel) {
    constexpr int D = 3, C = 2;
    // double-buffered table (ngp_grid_encode_forward_sel): which copy is current is a device word, read once per workgroup (scalar load)
    const half_t* __restrict__ grid = sel.parity && sel.parity[0] != 0.0f ? sel.alt : grid_a;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    uint32_t level = 0xffffu, tile = 0u, begin = 0u;
#pragma unroll
    for (int sg = 0; sg < FWD_MAX_SEG; sg++) {   // this XCD's work list: runs of consecutive tiles of one level, in order
        const uint32_t e = sched.end[xcd][sg];
        if (level == 0xffffu && slot < e) {
            level = sched.level[xcd][sg];
            tile = sched.tile0[xcd][sg] + (slot - begin);
        }
        begin = e;
    }
    NGP_BOUNDS(level < L || level == 0xffffu);
    if (level >= L) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, false, hashmap_size, lv.res[level]);
    const half_t* __restrict__ table = grid + (size_t)off0 * C;
    half_t* __restrict__ olevel = outputs + (size_t)level * B * C;
    NGP_BOUNDS((uint64_t)tile * points_per_block < (uint64_t)B);  // every listed tile holds at least one point
    const uint32_t b_begin = tile * points_per_block;
    const uint32_t B_work = sel.rows ? min(B, sel.rows[0]) : B;   // (device-side row count: the rows behind it carry nothing)
    if (b_begin >= B_work) return;
    const uint32_t b_end = min(B_work, b_begin + points_per_block);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool dense = !indexer.hashed && !indexer.need_mod && indexer.stride[0] == 1u && indexer.stride[1] != 0u && indexer.stride[2] != 0u;
    const bool hashed_pow2 = indexer.hashed && indexer.mask != 0u;
    constexpr bool mapped = MAPPED;   // (== im.scale != 0: resolved by the host, like the interpolation mode -- no per-point selects)

    // locate<3>() on registers: same expressions, same order (fmaf, floor, smoothstep), `ok` = the point lies inside [0, 1]^3
    auto place = [&](const FwdPos3& p, float (&frac)[3], uint32_t (&cell)[3]) -> bool {
        float xv[3];
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            xv[d] = p.x[d];
            if (mapped) xv[d] = (xv[d] + im.shift) * im.scale;
            ok = ok && !(xv[d] < 0.0f || xv[d] > 1.0f);
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float q = __builtin_fmaf(xv[d], scale, 0.5f);
            const float fl = floorf(q);
            cell[d] = (uint32_t)fl;
            q -= (float)cell[d];
            if (SMOOTH) q = q * q * (3.0f - 2.0f * q);
            frac[d] = q;
        }
        return ok;
    };

    if (NGP_FWD_FAST && dense) {
        // ---- one lane per point, x-pairs by 8-byte loads ----
        const uint32_t s1 = indexer.stride[1], s2 = indexer.stride[2];
        constexpr uint32_t STEP = FWD_THREADS;  // points per workgroup step
        uint32_t b = b_begin + (uint32_t)threadIdx.x;
        FwdPos3 cur, nxt;
        cur.load(inputs, b, B - 1u);
        for (; b - (uint32_t)threadIdx.x < b_end; b += STEP) {   // (workgroup-uniform trip count)
            const bool in_range = b < b_end;
            float frac[3];
            uint32_t cell[3];
            const bool ok = place(cur, frac, cell) && in_range;
            // a lane without a point reads entry 0 and discards it: the loads stay unconditional, so the NEXT step's position can be requested
            // right behind them (the vector-memory counter is in order: a request issued BEFORE the corners would have to land before them)
            const uint32_t i00 = ok ? cell[0] + cell[1] * s1 + cell[2] * s2 : 0u;
            NGP_BOUNDS(i00 + s1 + s2 + 1u < hashmap_size);
            uint2 q[4];
            const uint32_t ofs[4] = {0u, s1, s2, s1 + s2};
#pragma unroll
            for (int j = 0; j < 4; j++) __builtin_memcpy(&q[j], table + (size_t)(i00 + ofs[j]) * 2, sizeof(uint2));  // entries x, x + 1 (4-byte aligned)
            nxt.load(inputs, b + STEP, B - 1u);                     // in flight during this step's arithmetic and the next step's index work
            float accl[2] = {0.0f, 0.0f}, acch[2] = {0.0f, 0.0f};
            const float wl0 = 1.0f - frac[0], wh0 = frac[0];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float wy = (j & 1) ? frac[1] : 1.0f - frac[1], wz = (j & 2) ? frac[2] : 1.0f - frac[2];
                const float wl = wl0 * wy * wz, wh = wh0 * wy * wz;
                const half2_t lo = __builtin_bit_cast(half2_t, q[j].x), hi = __builtin_bit_cast(half2_t, q[j].y);
                accl[0] = __builtin_fmaf(wl, (float)lo.x, accl[0]);
                accl[1] = __builtin_fmaf(wl, (float)lo.y, accl[1]);
                acch[0] = __builtin_fmaf(wh, (float)hi.x, acch[0]);
                acch[1] = __builtin_fmaf(wh, (float)hi.y, acch[1]);
            }
            float out[2] = {accl[0] + acch[0], accl[1] + acch[1]};
            if (!ok) out[0] = out[1] = 0.0f;
            if (in_range) store_vec<half_t, 2>(olevel + (size_t)b * 2, out);
            cur = nxt;
        }
        return;
    }
    if (NGP_FWD_FAST && hashed_pow2) {
        // ---- corner pairs on neighbouring lanes, index = (x ^ y p1 ^ z p2) & mask ----
        const uint32_t mask = indexer.mask;
        const int pl = lane >> 1;
        const uint32_t xb = lane & 1;
        constexpr uint32_t STEP = (FWD_THREADS / 64) * FWD_PTS_PER_WAVE;
        uint32_t b = b_begin + (uint32_t)wid * FWD_PTS_PER_WAVE + (uint32_t)pl;
        FwdPos3 cur, nxt;
        cur.load(inputs, b, B - 1u);
        for (uint32_t base = b_begin + (uint32_t)wid * FWD_PTS_PER_WAVE; base < b_end; base += STEP, b += STEP) {
            const bool in_range = b < b_end;
            float frac[3];
            uint32_t cell[3];
            const bool ok = place(cur, frac, cell) && in_range;
            const uint32_t tx = cell[0] + xb;                      // first prime is 1
            const uint32_t ty0 = cell[1] * kPrimes[1], ty1 = ty0 + kPrimes[1];
            const uint32_t tz0 = cell[2] * kPrimes[2], tz1 = tz0 + kPrimes[2];
            // (a lane without a point reads entry 0 and discards it: unconditional loads, the next position requested right behind them)
            const uint32_t live = ok ? mask : 0u;
            const uint32_t idx[4] = {(tx ^ ty0 ^ tz0) & live, (tx ^ ty1 ^ tz0) & live, (tx ^ ty0 ^ tz1) & live, (tx ^ ty1 ^ tz1) & live};
            uint32_t q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = *reinterpret_cast<const uint32_t*>(table + (size_t)idx[j] * 2);
            nxt.load(inputs, b + STEP, B - 1u);
            float acc[2] = {0.0f, 0.0f};
            const float w0 = xb ? frac[0] : 1.0f - frac[0];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float wy = (j & 1) ? frac[1] : 1.0f - frac[1], wz = (j & 2) ? frac[2] : 1.0f - frac[2];
                const float w = w0 * wy * wz;
                const half2_t v = __builtin_bit_cast(half2_t, q[j]);
                acc[0] = __builtin_fmaf(w, (float)v.x, acc[0]);
                acc[1] = __builtin_fmaf(w, (float)v.y, acc[1]);
            }
            if (!ok) acc[0] = acc[1] = 0.0f;
            acc[0] += quad_swap1(acc[0]);  // all lanes execute (no cross-lane read under divergence)
            acc[1] += quad_swap1(acc[1]);
            if (in_range && xb == 0) store_vec<half_t, 2>(olevel + (size_t)b * 2, acc);
            cur = nxt;
        }
        return;
    }
    forward_pair_block<half_t, D, C>(inputs, table, olevel, scale, indexer, hashmap_size, false, interp, im, b_begin, b_end);
}

// corner-index diagnostic (same locate/indexer code path as the forward kernel)
template <int D>
__global__ __launch_bounds__(FWD_THREADS) void k_grid_corner_indices(const float* __restrict__ inputs,
                                                                     const int32_t* __restrict__ offsets,
                                                                     uint32_t* __restrict__ out, uint32_t B, GridLevels lv,
                                                                     uint32_t gridtype, bool align_corners) {
    const uint32_t level = blockIdx.y;
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - (uint32_t)offsets[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, align_corners, hashmap_size, lv.res[level]);
    for (uint32_t b = blockIdx.x * FWD_THREADS + threadIdx.x; b < B; b += gridDim.x * FWD_THREADS) {
        float frac[D], deriv[D];
        uint32_t cell[D];
        uint32_t* o = out + ((size_t)level * B + b) * (1 << D);
        const bool ok = locate<D>(inputs + (size_t)b * D, lv.scale[level], align_corners, 0u, frac, deriv, cell);
#pragma unroll
        for (int k = 0; k < (1 << D); k++) {
            uint32_t pg[D];
#pragma unroll
            for (int d = 0; d < D; d++) pg[d] = cell[d] + ((k >> d) & 1);
            o[k] = ok ? indexer(pg) : 0xFFFFFFFFu;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward (scatter-add into grad_embeddings)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_feat(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_pk(half_t* p, float a, float b) {
    half2_t v = {(half_t)a, (half_t)b};
    (void)__builtin_amdgcn_flat_atomic_fadd_v2f16(reinterpret_cast<half2_t*>(p), v);  // global_atomic_pk_add_f16
}
// fp16 table with odd C (C == 1): emulate with a 32-bit CAS on the containing word (the reference
// routes this case to a slow scalar half atomicAdd; under autocast the wrapper never produces it).
__device__ __forceinline__ void atomic_add_half1(half_t* p, float a) {
    uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    uint32_t* word = reinterpret_cast<uint32_t*>(addr & ~uintptr_t(3));
    const bool hi = (addr & 2) != 0;
    uint32_t old = *word, assumed;
    do {
        assumed = old;
        uint16_t bits = hi ? (uint16_t)(assumed >> 16) : (uint16_t)(assumed & 0xFFFFu);
        half_t h = __builtin_bit_cast(half_t, bits);
        h = (half_t)((float)h + a);
        uint16_t nb = __builtin_bit_cast(uint16_t, h);
        uint32_t nw = hi ? ((assumed & 0x0000FFFFu) | ((uint32_t)nb << 16)) : ((assumed & 0xFFFF0000u) | nb);
        old = atomicCAS(word, assumed, nw);
    } while (old != assumed);
}

template <typename T, int C>
__device__ __forceinline__ void scatter_add(T* dst, const float (&g)[C], float w) {
    if constexpr (sizeof(T) == 2) {
        if constexpr (C % 2 == 0) {
#pragma unroll
            for (int c = 0; c < C; c += 2) atomic_add_pk(reinterpret_cast<half_t*>(dst) + c, w * g[c], w * g[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) atomic_add_half1(reinterpret_cast<half_t*>(dst) + c, w * g[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) atomic_add_feat(reinterpret_cast<float*>(dst) + c, w * g[c]);
    }
}

constexpr int BWD_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// backward: scatter-add of w * grad into grad_embeddings               (gridencoder.cu:248-340)
//
// What bounds a scatter on gfx950 (tools/atomic_probe2.hip, measured): a global float atomic costs one
// memory-side REQUEST per (wave instruction, distinct 64-byte line) at ~20 G requests/s chip-wide, whatever
// the scope bits and however small the table is; lanes of ONE instruction that fall into one line ride along
// for free (16 lanes on a line: 320 G atomics/s).  So the kernel is organised to minimise (instruction, line)
// pairs, not lane operations:
//   * the lanes of a point (2 for fp16 C=2, 4 for fp32 C=2, ... up to 16) cover both corners that differ in the FIRST
//     coordinate and all their channels.  Those two entries are neighbours in memory -- on dense levels by construction
//     and on hashed levels too, because the first hash prime is 1: (x ^ A) and ((x+1) ^ A) share an aligned
//     16-entry block whenever x and x+1 do -- so every atomic instruction touches at most one line per point;
//   * corners are assigned to lane classes / instruction slots by the ABSOLUTE PARITY of the vertex coordinates, so a vertex
//     shared by neighbouring cells sits in the same slot for every point that touches it.  Consecutive samples of a ray that
//     share a vertex form a run of equal table addresses in that slot: runs are summed in fp32 with a segmented wave scan and
//     only the last lane of a run issues the atomic (no same-address serialisation -- an extra same-address lane costs about as
//     much as a request -- and one rounding to the table dtype per run); vertices of different points that fall into one cache
//     line are issued by the same instruction and ride in one request;
//   * samples behind an early-terminated ray carry an exactly-zero gradient and are skipped.
// Measured ladder on the lego batch (277 k samples x 16 levels): one corner per instruction, one point per lane 2205 us ->
// corner pairs on adjacent lanes 1467 us -> + same-cell run merge 532 us -> parity slots + same-vertex run merge 396 us.
// fp16 tables with even C use global_atomic_pk_add_f16 (what the reference's half2 atomicAdd does), everything
// else global_atomic_add_f32.  The summation order of colliding atomics is not defined (as in the reference).
// ------------------------------------------------------------------------------------------------
template <typename T, int C>
struct BwdLanes {
    // Lane layout inside a point: [first-coordinate corner xb][channel group cl].  A lane carries CPL channels of one corner
    // (2 with the packed fp16 atomic, else 1), so the LPP lanes of a point cover 2 * C consecutive table values = one contiguous
    // span of 8..64 bytes: ONE atomic request per (point, remaining-corner j) whatever the dtype and C.
    static constexpr int CPL = (sizeof(T) == 2 && C % 2 == 0) ? 2 : 1;
    static constexpr int CL = C / CPL;    // lanes per corner
    static constexpr int LPP = 2 * CL;    // lanes per point (a power of two <= 16)
    static constexpr int PTS = 64 / LPP;  // points per wave
};

struct LevelList {  // the levels a launch covers (blockIdx.y indexes this list)
    uint8_t level[NGP_MAX_LEVELS];
};

// One wave-step of the backward: point slot `pl` of this wave handles sample b.  Produces, per remaining-corner slot j, the table
// address, the (run-merged) fp32 contribution of this lane's CPL channels, and whether this lane issues it.
// the global loads of one wave-step (sample position, this lane's gradient channels), issued together and ahead of the arithmetic
template <typename T, int D, int C>
struct BwdSample {
    float x[D];
    float g[BwdLanes<T, C>::CPL];
    bool live;
    __device__ __forceinline__ void load(const float* __restrict__ inputs, const T* __restrict__ glevel, uint32_t b, bool in_range, int c0) {
        constexpr int CPL = BwdLanes<T, C>::CPL;
        live = in_range;
#pragma unroll
        for (int d = 0; d < D; d++) x[d] = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; c++) g[c] = 0.0f;
        if (in_range) {
#pragma unroll
            for (int d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
            Vec<T, CPL> gv;
            gv.load(glevel + (size_t)b * C + c0);
#pragma unroll
            for (int c = 0; c < CPL; c++) g[c] = gv.v[c];
        }
    }
};

// DPP row shifts (VALU, no LDS traffic): lane i of each 16-lane row reads lane i -/+ N of the SAME row; a lane without a source reads 0
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x110 + N, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x100 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t src) {  // lane i reads lane i-1 across the whole wave; lane 0 reads 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t src) {  // lane i reads lane i+1; lane 63 reads 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t row_bcast15(uint32_t src) {  // every lane of rows 1..3 reads lane 15 of the row below (row 0: unused)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x142, 0xE, 0xF, false);
}
template <int N>
__device__ __forceinline__ float row_shr_f(float src) {
    return __builtin_bit_cast(float, row_shr<N>(__builtin_bit_cast(uint32_t, src)));
}

// MERGE: 0 = every sample issues its own contribution; 1 = runs of equal table address over the whole wave (cross-lane reads through
// ds_bpermute: ~15 clk of the CU's LDS pipe each, measured -- affordable only where the alternative is a fabric atomic); 2 = runs
// inside a 16-lane row (8 samples of the fp16 C = 2 layout) with DPP row shifts: pure VALU.
template <typename T, int D, int C, int MERGE>
__device__ __forceinline__ void corner_runs(const BwdSample<T, D, C>& smp, float scale, bool align_corners, uint32_t interp,
                                            const LevelIndexer<D>& indexer, InputMap im, int pl, uint32_t xb,
                                            uint32_t (&addr)[1 << (D - 1)], float (&v)[1 << (D - 1)][BwdLanes<T, C>::CPL],
                                            bool (&issue)[1 << (D - 1)]) {
    constexpr int NJ = 1 << (D - 1);
    constexpr int CPL = BwdLanes<T, C>::CPL, LPP = BwdLanes<T, C>::LPP, PTS = BwdLanes<T, C>::PTS;
    float frac[D], deriv[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; d++) { frac[d] = 0.0f; cell[d] = 0u; }
    float g[CPL];
    bool live = smp.live;
    if (live) live = locate<D>(smp.x, scale, align_corners, interp, frac, deriv, cell, im);
    {
        bool nz = false;
#pragma unroll
        for (int c = 0; c < CPL; c++) { g[c] = smp.g[c]; nz = nz || (g[c] != 0.0f); }
        live = live && nz;  // per lane: a lane whose own channels carry an exactly-zero gradient has nothing to add
    }
    // This lane's NJ corners, chosen by ABSOLUTE PARITY of the vertex coordinates: lane class xb owns the vertex whose first
    // coordinate has parity xb, slot s owns the parities (s bit d-1) of the remaining coordinates.  A vertex shared by
    // neighbouring cells therefore sits in the same lane class and the same slot for every point that touches it: consecutive
    // samples of a ray that share a VERTEX (not only a cell) form a run of equal addresses in one slot and are merged below,
    // and vertices of different points that share a cache line are issued by the same instruction (one request).
    uint32_t lower[D], upper[D];  // index terms of the cell's lower / upper vertex coordinate per dimension
#pragma unroll
    for (int d = 0; d < D; d++) {
        lower[d] = indexer.term(d, cell[d]);
        upper[d] = lower[d] + indexer.step(d);
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        uint32_t t[D];
        const uint32_t bit0 = (xb ^ cell[0]) & 1u;
        t[0] = bit0 ? upper[0] : lower[0];
        float w = live ? (bit0 ? frac[0] : 1.0f - frac[0]) : 0.0f;
#pragma unroll
        for (int d = 1; d < D; d++) {
            const uint32_t bit = (((uint32_t)j >> (d - 1)) ^ cell[d]) & 1u;
            t[d] = bit ? upper[d] : lower[d];
            w *= bit ? frac[d] : (1.0f - frac[d]);
        }
        addr[j] = indexer.combine(t);
#pragma unroll
        for (int c = 0; c < CPL; c++) v[j][c] = w * g[c];
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) issue[j] = live;
    if constexpr (MERGE == 2 || MERGE == 3) {
        // Segmented scan with DPP only (VALU; no LDS crossbar).  MERGE == 2: runs inside a 16-lane row (8 samples).  MERGE == 3: runs
        // over the whole wave -- the links to the neighbouring sample cross rows with wave_shr/shl:1 (twice: two lanes per sample), the
        // scan runs inside the rows, and the rows are then chained by three carry rounds (row_bcast:15 hands the last sample of a row
        // to the next row; the class-0 lane first moves into lane 15 with row_shr:1).
        static_assert((MERGE != 2 && MERGE != 3) || LPP == 2, "the DPP merges are written for two lanes per sample");
        auto prev2 = [](uint32_t x) {  // value of the lane two below (0 for a sample without predecessor)
            if constexpr (MERGE == 2) return row_shr<2>(x);
            else return wave_shr1(wave_shr1(x));
        };
        auto next2 = [](uint32_t x) {
            if constexpr (MERGE == 2) return row_shl<2>(x);
            else return wave_shl1(wave_shl1(x));
        };
        const uint32_t prev_live = prev2((uint32_t)live);
        const uint32_t row = (uint32_t)(threadIdx.x & 63) >> 4;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t prev_addr1 = prev2(addr[j] + 1u);
            const bool same = live & (prev_live != 0u) & (prev_addr1 == addr[j] + 1u);
            // the record sort merges only when a fair share of the wave continues a run (fine levels: almost never);
            // in front of a fabric atomic every merged lane pays
            if (__popcll(__ballot(same)) >= (MERGE == 2 ? 12 : 1)) {
                uint32_t closed = same ? 0u : 1u;  // the scan of this lane has reached the head of its run
                // inside the row, distances 1, 2, 4 samples; a lane without a source reads 0: adds nothing and stays open
#define NGP_ROW_SCAN_STEP(N)                                                                                  \
                {                                                                                                 \
                    const uint32_t c_n = row_shr<N>(closed);                                                     \
                    float t[CPL];                                                                                 \
                    _Pragma("unroll") for (int c = 0; c < CPL; c++) t[c] = row_shr_f<N>(v[j][c]);               \
                    _Pragma("unroll") for (int c = 0; c < CPL; c++) v[j][c] += closed ? 0.0f : t[c];            \
                    closed = closed ? 1u : c_n;                                                                   \
                }
                NGP_ROW_SCAN_STEP(2)
                NGP_ROW_SCAN_STEP(4)
                NGP_ROW_SCAN_STEP(8)
#undef NGP_ROW_SCAN_STEP
                if constexpr (MERGE == 3) {
#pragma unroll
                    for (uint32_t r = 1; r < 4; r++) {  // carry of row r-1 (its last sample, already carrying its own) into row r
                        const uint32_t c1 = row_bcast15(closed), c0 = row_bcast15(row_shr<1>(closed));
                        const uint32_t cc = xb ? c1 : c0;
                        const bool take = (row == r) && !closed;
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            const float t1 = __builtin_bit_cast(float, row_bcast15(__builtin_bit_cast(uint32_t, v[j][c])));
                            const float t0 = __builtin_bit_cast(float, row_bcast15(row_shr<1>(__builtin_bit_cast(uint32_t, v[j][c]))));
                            v[j][c] += take ? (xb ? t1 : t0) : 0.0f;
                        }
                        closed = take ? cc : closed;
                    }
                }
                const uint32_t next_same = next2((uint32_t)same);
                issue[j] = live && !next_same;  // the last lane of a run holds the run total
            }
        }
    }
    if constexpr (MERGE == 1) {
        // (every shuffle is executed by all lanes: no short-circuit in front of a cross-lane read)
        const int prev_live = __shfl_up((int)live, LPP, 64);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            // same destination as the previous point slot?  (equal table address: equal vertex, or a hash collision -- either way
            // the contributions go to the same place)
            const uint32_t prev_addr = __shfl_up(addr[j], LPP, 64);
            const bool same = live & (pl > 0) & (prev_live != 0) & (prev_addr == addr[j]);
            if (__any(same)) {
                bool reached = !same;  // the scan of this lane has reached the head of its run
#pragma unroll
                for (int o = 1; o < PTS; o <<= 1) {
                    const int r_o = __shfl_up((int)reached, LPP * o, 64);
                    const bool take = !reached && pl >= o;
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        const float t = __shfl_up(v[j][c], LPP * o, 64);
                        if (take) v[j][c] += t;
                    }
                    if (take) reached = r_o != 0;
                }
                const int next_same = __shfl_down((int)same, LPP, 64);
                issue[j] = live && (pl == PTS - 1 || !next_same);  // the last lane of a run holds the run total
            }
        }
    }
}

// the atomic path of one workgroup: `points_per_block` consecutive samples (from block_x * points_per_block) of one level
template <typename T, int D, int C, int MERGE, int THREADS>
__device__ __forceinline__ void backward_atomic_block(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                      const int32_t* __restrict__ offsets, T* __restrict__ grad_grid, uint32_t B,
                                                      uint32_t level, const GridLevels& lv, uint32_t gridtype, bool align_corners,
                                                      uint32_t interp, uint32_t points_per_block, InputMap im, uint32_t block_x) {
    constexpr int NJ = 1 << (D - 1);  // corners per lane (all combinations of the coordinates 1..D-1)
    constexpr int CPL = BwdLanes<T, C>::CPL, CL = BwdLanes<T, C>::CL, LPP = BwdLanes<T, C>::LPP, PTS = BwdLanes<T, C>::PTS;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, align_corners, hashmap_size, lv.res[level]);
    T* __restrict__ gtable = grad_grid + (size_t)off0 * C;
    const T* __restrict__ glevel = grad + (size_t)level * B * C;

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pl = lane / LPP;                        // point slot inside the wave
    const uint32_t xb = (uint32_t)(lane / CL) & 1u;   // which first-coordinate corner class this lane owns
    const int c0 = (lane % CL) * CPL;                 // first channel this lane owns
    const uint32_t b_begin = block_x * points_per_block;
    const uint32_t b_end = min(B, b_begin + points_per_block);

    for (uint32_t base = b_begin + wid * PTS; base < b_end; base += (THREADS / 64) * PTS) {
        const uint32_t b = base + pl;
        uint32_t addr[NJ];
        float v[NJ][CPL];
        bool issue[NJ];
        BwdSample<T, D, C> smp;
        smp.load(inputs, glevel, b, b < b_end, c0);
        corner_runs<T, D, C, MERGE>(smp, scale, align_corners, interp, indexer, im, pl, xb, addr, v, issue);
#pragma unroll
        for (int j = 0; j < NJ; j++)
            if (issue[j]) scatter_add<T, CPL>(gtable + (size_t)addr[j] * C + c0, v[j], 1.0f);
    }
}

template <typename T, int D, int C, int MERGE>
__global__ __launch_bounds__(BWD_THREADS) void k_grid_backward(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                               const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                               uint32_t B, LevelList ll, GridLevels lv, uint32_t gridtype,
                                                               bool align_corners, uint32_t interp, uint32_t points_per_block, InputMap im) {
    backward_atomic_block<T, D, C, MERGE, BWD_THREADS>(grad, inputs, offsets, grad_grid, B, ll.level[blockIdx.y], lv, gridtype, align_corners,
                                                       interp, points_per_block, im, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// backward, BINNED levels (fp16 tables with C = 2: the instant-ngp configuration) -- no memory-side atomics at all.
//
// A global atomic is a fabric operation on this chip (~20 G requests/s, above); a fine hashed level needs ~4 of them per sample
// and nothing merges them.  LDS is no faster for FLOAT atomics (ds_add_f32 / ds_pk_add_f16: 0.33 lanes/clk/CU, serialised),
// but INTEGER LDS atomics run at ~7 lanes/clk/CU (tools/lds_atomic_probe.hip).  And every contribution is an fp16 number, i.e. an
// integer multiple of 2^-24 below 2^16: a 64-bit fixed-point accumulator holds the EXACT sum of any number of them.  So:
//   pass 1 (k_grid_backward_bin): the same corner/weight/run-merge arithmetic, but each issued contribution becomes an 8-byte
//       record {table index, fp16x2 value}.  A workgroup counting-sorts the <= 4096 records of its 512 samples by table SLICE
//       (4096 entries) in LDS and streams them, sorted, into ITS OWN chunk of the workspace (fully coalesced, no reservation, no
//       overflow), plus one descriptor {begin, count} per slice;
//   pass 2 (k_grid_backward_accumulate): one workgroup per (level, slice) walks the descriptors of all chunks, adds the matching
//       runs into a 64 KiB int64 LDS accumulator with ds_add_u64, rounds each sum ONCE and adds it to the gradient table with plain
//       loads/stores (it owns the slice).
// The result is the exact sum rounded once -- more accurate than the reference's fp16 atomics and bit-reproducible run to run
// (integer addition commutes); non-finite contributions poison their entry with NaN (the loss scaler skips the step either way).
// ------------------------------------------------------------------------------------------------
// (tuning knobs of the record sort, compile-time: -DNGP_BIN_THREADS=... -DNGP_BIN_RESIDENT=... -DNGP_BIN_MERGE_MIN=...)
#ifndef NGP_BIN_THREADS
#define NGP_BIN_THREADS 512
#endif
constexpr int BIN_THREADS = NGP_BIN_THREADS;
constexpr int BIN_PPB = BIN_THREADS;                           // samples per workgroup item = per chunk (one lane per sample)
#ifndef NGP_BIN_SLICE_BITS
#define NGP_BIN_SLICE_BITS 12
#endif
constexpr int BIN_SLICE_BITS = NGP_BIN_SLICE_BITS;             // 4096 table entries per slice / bin
constexpr int BIN_SLICE = 1 << BIN_SLICE_BITS;
constexpr int BIN_MAX_BINS = 512;                              // bins per level at most (every wave scans them, 8 per lane)
#ifndef NGP_BIN_RESIDENT
#define NGP_BIN_RESIDENT 3
#endif
constexpr int BIN_RESIDENT = NGP_BIN_RESIDENT;                 // persistent sort workgroups per CU
#ifndef NGP_ACC_THREADS
#define NGP_ACC_THREADS 1024
#endif
#ifndef NGP_ACC_RUNS_AHEAD
#define NGP_ACC_RUNS_AHEAD 4
#endif
constexpr int ACC_THREADS = NGP_ACC_THREADS;
// Dense levels: samples cluster where the scene is, so contiguous slices would be very unevenly loaded (and a 4913-entry level would
// have two of them).  Their entries are dealt round-robin to BIN_DENSE_BINS workgroups instead: bin = index mod 128, slot = index / 128.
constexpr int BIN_DENSE_BITS = 7;
constexpr int BIN_DENSE_BINS = 1 << BIN_DENSE_BITS;

constexpr int BIN_LC_WORDS = 12;
struct BinPlan {
    // per binned level li (blockIdx.y of the accumulate, item index of the sort): constants in 32-bit words, so that a wave-uniform index
    // becomes a few scalar loads from the kernel-argument segment.  Made from the caller's HOST copy of the offsets.
    //   [0] level  [1] n_bins  [2] flags (1: interleaved bins, 4: hashed, 8: index needs the modulo)  [3] table entries  [4] scale (float bits)
    //   [5] mask (entries - 1 if a power of two, else 0)  [7] first descriptor of the level ([bin][chunk])  [8..10] dense strides
    uint32_t lc[NGP_MAX_LEVELS][BIN_LC_WORDS];
    uint32_t n_chunks;                   // chunks (BIN_PPB samples) per level
    uint32_t n_levels;                   // binned levels
    // interleaved = 0: bin = index >> 12 (contiguous slices); 1: bin = index & 127 (dense levels, see above)
    __host__ __device__ __forceinline__ uint32_t level(uint32_t li) const { return lc[li][0]; }
    __host__ __device__ __forceinline__ uint32_t n_bins(uint32_t li) const { return lc[li][1]; }
    __host__ __device__ __forceinline__ bool interleaved(uint32_t li) const { return (lc[li][2] & 1u) != 0u; }
    __host__ __device__ __forceinline__ uint32_t desc_base(uint32_t li) const { return lc[li][7]; }
};

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {
    const half2_t h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void atomic_add_packed(half_t* p, uint32_t packed) {
    (void)__builtin_amdgcn_flat_atomic_fadd_v2f16(reinterpret_cast<half2_t*>(p), __builtin_bit_cast(half2_t, packed));
}

// Pass 1 and the atomic levels share ONE launch: the record sort is VALU-bound, the atomic levels wait on the LDS crossbar (run merge)
// and on the fabric, so workgroups of the two kinds are interleaved (evenly, by a Bresenham split of the block index) and overlap on
// every CU instead of running back to back.
struct AtomicPart {
    LevelList levels;
    uint32_t n_blocks;          // workgroups of the atomic kind in this launch (0: none)
    uint32_t blocks_per_level;  // = cdiv(B, points_per_block)
    uint32_t points_per_block;
};

// DPP carries between the 16-lane rows (profiles/r01_dpp_probe.txt): row_bcast:15 hands lane 15 of a row to every lane of the NEXT row,
// row_bcast:31 hands lane 31 to every lane of rows 2 and 3; the row mask keeps the other rows at `old` = 0.
__device__ __forceinline__ uint32_t bcast15_rows13(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x142, 0xA, 0xF, false);
}
__device__ __forceinline__ uint32_t bcast31_rows23(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, 0x143, 0xC, 0xF, false);
}

// Record sort, round 4: ONE LANE PER SAMPLE, count pass + place pass.
//
// A wave64 VALU instruction occupies its SIMD for 4 cycles; the round-3 sort (a sample on two lanes, each repeating the position work)
// issued 52 M of them = 85 us of the chip's 1024 SIMDs: it WAS its instruction stream (EXPERIMENTS.md).  Now a lane computes the position
// inside the level, the per-dimension index terms and weights ONCE and emits all 2^D corner records of its sample.
//   * Corner slot s holds the vertex whose coordinates have the ABSOLUTE parities of the bits of s, so a vertex shared by the cells of
//     consecutive samples of a ray sits in the same slot on neighbouring lanes: runs of equal table address in a slot are summed in fp32
//     over all 64 samples of the wave (segmented Hillis-Steele scan: four DPP row shifts, two broadcast carries -- pure VALU, a fixed
//     tree: bit-reproducible) and only the LAST lane of a run issues a record.  A slot whose wave has fewer than NGP_BIN_MERGE_MIN
//     continuing lanes skips the scan (fine levels: consecutive samples share no vertex).
//   * COUNT pass: addresses, run structure (lane masks, kept in scalar registers), one non-returning LDS add per record into the wave's
//     OWN row of bin counters.  After ONE barrier every wave reads all rows, scans the bins for itself and knows where its own records
//     of every bin go (chunk offset of the bin + records of the waves before it): no shared cursor, no second barrier.  PLACE pass:
//     values, run sums, packing; a returning LDS add on the wave's own cursor row gives the final slot in the staging area.  What a lane
//     carries across the barrier are 2 D index terms and 2 D weights -- not addresses, values and ranks of 2^D records: <= 64 VGPRs, four
//     workgroups per CU.  Second barrier, coalesced copy of the sorted chunk (16 bytes per lane).
//   * Workgroups are persistent over a contiguous range of (chunk, level) items, level fastest: positions are loaded once per chunk,
//     the gradient of the next item is in flight while the current one is sorted, per-level constants are scalar loads from the plan.
#ifdef NGP_BIN_PHASE_PROBE  // timing probe only: shader-clock cycles per phase of the sort's item loop, summed over wave 0 of every workgroup
__device__ unsigned long long g_bin_probe[16];
#define NGP_PROBE_T(i) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_probe; t_probe = t_now; }
#else
#define NGP_PROBE_T(i)
#endif
#ifndef NGP_BIN_MERGE_MIN
#define NGP_BIN_MERGE_MIN 8
#endif

// record key bits (16-bit keys): channel 0 / 1 of the value is 1/64 of the contribution (a run sum beyond the fp16 range); bits 0..11 = entry
// inside the slice
constexpr uint32_t BIN_KEY_SCALED0 = 0x8000u, BIN_KEY_SCALED1 = 0x4000u, BIN_KEY_SCALED = BIN_KEY_SCALED0 | BIN_KEY_SCALED1;
// Records travel in PAIR UNITS of 12 bytes: {key of record 0 | key of record 1 << 16, value 0, value 1}.  Sorted by bin, a record needs only
// its index INSIDE the slice (12 bits) and the two scale flags: 6 bytes per record instead of 8 on the way to memory and back (the record
// stream is what the two kernels move: 157 -> 118 MB each way per step).  A bin with an odd number of records ends in a half-used unit whose
// second half is {key 0, value +0}.
constexpr uint32_t BIN_UNIT_BYTES = 12;

// Wave-uniform 64-bit lane masks, pinned to scalar registers where they are made.  and / andn2 / shifts / bit counts have 64-bit scalar
// forms; only COMPARES are written on the 32-bit halves (there is no scalar u64 less-than: a 64-bit compare would be done by the vector
// unit and drag the mask into vector registers).
struct LaneMask {
    unsigned long long v;
    __device__ __forceinline__ static LaneMask of(unsigned long long b) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        return LaneMask{((unsigned long long)hi << 32) | lo};
    }
    __device__ __forceinline__ bool any() const { return ((uint32_t)v | (uint32_t)(v >> 32)) != 0u; }
    __device__ __forceinline__ bool any_of(uint32_t each_half) const { return (((uint32_t)v | (uint32_t)(v >> 32)) & each_half) != 0u; }
    __device__ __forceinline__ int count() const { return __builtin_popcountll(v); }
};

// Segmented inclusive scan of two floats inside the 16-lane ROWS of the wave: open = all ones in a lane that CONTINUES the run of the
// lane below it, 0 in the first lane of a run (and in the first lane of every row: a run that crosses a row boundary is cut there and
// issues one more record -- the two cross-row carries would cost as much as two more steps for a handful of records per wave).
// Hillis-Steele with DPP row shifts 1, 2, 4, 8.  Written in assembly because the select costs nothing this way: the incoming value is
// ANDed with the lane's `open` mask in the DPP instruction that fetches it (5 VALU per step instead of 9 from the compiler's mov_dpp +
// cndmask + add), and `open &= open[i - N]` leaves lanes without a source untouched (no bound_ctrl: the lane is not written).  s_nop: a
// VALU write needs two wait states before a DPP read of the register (the compiler cannot see into the blocks, so every block keeps the
// distance itself).  A step is skipped (wave-uniform) when no open lane has a source for it: runs of two or three samples (the middle
// levels) finish after one or two steps.
#define NGP_SEG_ROW_STEP(N)                                                                       \
    asm volatile("s_nop 1\n\t"                                                                    \
                 "v_and_b32_dpp %3, %0, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                 "v_and_b32_dpp %4, %1, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                 "v_and_b32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"          \
                 "v_add_f32 %0, %0, %3\n\t"                                                       \
                 "v_add_f32 %1, %1, %4\n\t"                                                       \
                 : "+v"(v0), "+v"(v1), "+v"(open), "=&v"(t0), "=&v"(t1))
__device__ __forceinline__ void seg_scan_rows(float& v0, float& v1, const LaneMask& continues) {
    uint32_t open = __builtin_amdgcn_inverse_ballot_w64(continues.v) ? 0xffffffffu : 0u, t0, t1;
    auto still_open = [&]() { return LaneMask::of(__builtin_amdgcn_ballot_w64(open != 0u)); };
    NGP_SEG_ROW_STEP(1);
    if (still_open().any_of(0xFFFCFFFCu)) {  // an open lane at row position >= 2 has a source two lanes below
        NGP_SEG_ROW_STEP(2);
        if (still_open().any_of(0xFFF0FFF0u)) {
            NGP_SEG_ROW_STEP(4);
            if (still_open().any_of(0xFF00FF00u)) NGP_SEG_ROW_STEP(8);
        }
    }
}
#undef NGP_SEG_ROW_STEP

struct BinItem {  // wave-uniform description of one (chunk, level) item
    uint32_t li, chunk_x, n_bins, desc_base;
    float scale;
    bool interleaved, plan_ok;
    half_t* gtable;
};

// LDS counters addressed by BYTE offset from the start of the dynamic LDS segment (one v_add less per counter than pointer arithmetic)
__device__ __forceinline__ void lds_add_u32(uint32_t byte_addr) {
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(byte_addr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t byte_addr) {
    return __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(byte_addr), 1u, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int D>
struct BinLane {  // what a lane carries from the count pass to the place pass
    uint32_t addr[1 << D];  // table index of corner slot s (slot = ABSOLUTE parities of the vertex coordinates)
    float wp[D][2];         // per dimension: weight of the cell's vertex with EVEN coordinate ([d][0]) and with ODD coordinate ([d][1]); a dead
                            // lane has 0 in dimension 0
};

template <bool FAST>
__device__ __forceinline__ uint32_t bin_counter_addr(uint32_t addr, const BinItem& it, uint32_t row_base) {  // LDS byte address of the bin's counter
    // (rows are 512-byte aligned: the base is ORed in)
    if constexpr (FAST) return ((addr >> (BIN_SLICE_BITS - 2)) & (uint32_t)((BIN_MAX_BINS - 1) << 2)) | row_base;
    else return ((it.interleaved ? (addr & (uint32_t)(BIN_DENSE_BINS - 1)) : (addr >> BIN_SLICE_BITS)) << 2) | row_base;
}

// Lanes of a corner slot that CONTINUE the run of the lane below them: same table index, both live, same 16-lane row (`pairs` holds the
// last two conditions and the wave's merge decision).  Computed wherever it is needed (two VALU) rather than carried in scalar registers.
__device__ __forceinline__ LaneMask bin_run_mask(uint32_t addr, const LaneMask& pairs) {
    return LaneMask{LaneMask::of(__builtin_amdgcn_ballot_w64(wave_shr1(addr) == addr)).v & pairs.v};
}
// the lanes that issue a record: live and not continued by the lane above (the LAST lane of a run)
__device__ __forceinline__ bool bin_issues(const LaneMask& live, const LaneMask& m) {
    return __builtin_amdgcn_inverse_ballot_w64(live.v & ~(m.v >> 1));
}
// `pairs` of a wave: this lane and the one below it are live and sit in the same 16-lane row -- or nothing at all when fewer than
// NGP_BIN_MERGE_MIN lanes of the wave continue a run in corner slot 0 (fine levels: consecutive samples share no vertex; the decision is
// made once per wave and pass, from the same data in both passes)
__device__ __forceinline__ LaneMask bin_pairs(const LaneMask& live, uint32_t addr0) {
    LaneMask pairs{live.v & (live.v << 1) & 0xFFFEFFFEFFFEFFFEull};
    if (bin_run_mask(addr0, pairs).count() < NGP_BIN_MERGE_MIN) pairs.v = 0ull;
    return pairs;
}

// COUNT pass.  FAST = hashed level with a power-of-two table in contiguous slices (every fine level of the instant-ngp configuration)
template <int D, bool FAST>
__device__ __forceinline__ void bin_pass_count(const float (&x)[D], uint32_t gbits, bool in_range, const BinItem& it, const LevelIndexer<D>& ix,
                                               bool align_corners, uint32_t interp, InputMap im, uint32_t hrow_base, BinLane<D>& bl,
                                               LaneMask& live_mask) {
    constexpr int NS = 1 << D;
    float frac[D], deriv[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; d++) { frac[d] = 0.0f; cell[d] = 0u; }
    bool live = in_range && (gbits & 0x7fff7fffu) != 0u;  // an exactly-zero gradient (samples behind an early-terminated ray) adds nothing
    if (live) live = locate<D>(x, it.scale, align_corners, interp, frac, deriv, cell, im);
    live_mask = LaneMask::of(__builtin_amdgcn_ballot_w64(live));
#pragma unroll
    for (int s = 0; s < NS; s++) bl.addr[s] = 0u;
#pragma unroll
    for (int d = 0; d < D; d++) bl.wp[d][0] = bl.wp[d][1] = 0.0f;
    if (!live_mask.any()) return;  // (wave-uniform: every lane executes the cross-lane reads below)
    uint32_t tp[D][2];  // per dimension: index term of the cell's vertex with EVEN coordinate ([d][0]) and with ODD coordinate ([d][1])
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t lower = FAST ? cell[d] * kPrimes[d] : ix.term(d, cell[d]);
        const uint32_t upper = lower + (FAST ? kPrimes[d] : ix.step(d));
        const bool odd = (cell[d] & 1u) != 0u;
        tp[d][0] = odd ? upper : lower;
        tp[d][1] = odd ? lower : upper;
        const float f = frac[d], nf = 1.0f - frac[d];
        bl.wp[d][0] = odd ? f : nf;
        bl.wp[d][1] = odd ? nf : f;
    }
    if (!live) bl.wp[0][0] = bl.wp[0][1] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        uint32_t t[D];
#pragma unroll
        for (int d = 0; d < D; d++) t[d] = tp[d][(s >> d) & 1];
        if constexpr (FAST) {  // hashed, power-of-two table
            uint32_t a = t[0];
#pragma unroll
            for (int d = 1; d < D; d++) a ^= t[d];
            bl.addr[s] = a & ix.mask;
        } else {
            bl.addr[s] = ix.combine(t);
        }
    }
    if (!(FAST || it.plan_ok)) return;
    const LaneMask pairs = bin_pairs(live_mask, bl.addr[0]);
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (bin_issues(live_mask, bin_run_mask(bl.addr[s], pairs))) {
            NGP_BOUNDS(bl.addr[s] < ix.size && bin_counter_addr<FAST>(bl.addr[s], it, hrow_base) - hrow_base < 4u * it.n_bins);
            lds_add_u32(bin_counter_addr<FAST>(bl.addr[s], it, hrow_base));
        }
}

// PLACE pass: staging slots from the wave's own cursors (eight returning LDS adds in flight), then values, run sums and the records
template <int D, bool FAST>
__device__ __forceinline__ void bin_pass_place(uint32_t gbits, const BinItem& it, const BinLane<D>& bl, const LaneMask& live_mask_in,
                                               uint32_t prow_base, uint32_t staging_base) {
    constexpr int NS = 1 << D;
    // (pinned to scalar registers again: carried across the barrier the mask may sit in a vector register, and everything derived from
    // it would then be computed by the vector unit)
    const LaneMask live_mask = LaneMask::of(live_mask_in.v);
    if (!live_mask.any()) return;
    const LaneMask pairs = bin_pairs(live_mask, bl.addr[0]);
    const bool binned = FAST || it.plan_ok;
    uint32_t slot[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        slot[s] = 0u;
        if (binned && bin_issues(live_mask, bin_run_mask(bl.addr[s], pairs))) slot[s] = lds_add_rtn_u32(bin_counter_addr<FAST>(bl.addr[s], it, prow_base));
    }
    const half2_t gh = __builtin_bit_cast(half2_t, gbits);
    const float g0 = (float)gh.x, g1 = (float)gh.y;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const uint32_t addr = bl.addr[s];
        float w = bl.wp[0][s & 1];
#pragma unroll
        for (int d = 1; d < D; d++) w *= bl.wp[d][(s >> d) & 1];
        float v0 = w * g0, v1 = w * g1;
        const LaneMask m = bin_run_mask(addr, pairs);
        uint32_t key = FAST ? (addr & (uint32_t)(BIN_SLICE - 1)) : (it.interleaved ? (addr >> BIN_DENSE_BITS) : (addr & (uint32_t)(BIN_SLICE - 1)));
        if (m.any()) {  // segmented inclusive scan: the last lane of a run ends up with the run's sum
            seg_scan_rows(v0, v1, m);
            // a run of up to 16 finite fp16-range terms can leave the fp16 range although the sum over the whole batch need not: such a
            // record carries sum / 64 in that channel and says so in bit 31 / 30 of its key (the accumulate shifts the exact fixed-point
            // addend back).  Rare: one test for the wave first.
            if (__builtin_amdgcn_ballot_w64(fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)) >= 32768.0f) != 0ull) {
                if (__builtin_fabsf(v0) >= 32768.0f) { v0 *= 0.015625f; key |= BIN_KEY_SCALED0; }
                if (__builtin_fabsf(v1) >= 32768.0f) { v1 *= 0.015625f; key |= BIN_KEY_SCALED1; }
            }
        }
        const uint32_t packed = pack_half2(v0, v1);
        if (bin_issues(live_mask, m)) {
            NGP_BOUNDS(!binned || slot[s] < (uint32_t)(BIN_PPB * NS) + (uint32_t)BIN_MAX_BINS);
            if (binned)  // record `slot` of the chunk (bins start at even slots: the copy-out packs slots 2u, 2u + 1 into pair unit u)
                *(__attribute__((address_space(3))) unsigned long long*)(uintptr_t)(staging_base + slot[s] * 8u) = ((unsigned long long)packed << 32) | key;
            else atomic_add_packed(it.gtable + (size_t)addr * 2, packed);
        }
    }
}

// After the count barrier: a wave reads every wave's counter row (BPL consecutive bins per lane), scans the bin totals and writes ITS OWN
// cursor row: chunk offset of the bin + the records of the waves before it.  Wave 0 also writes the descriptors.  Returns all records.
template <int BPL, int WAVES>
__device__ __forceinline__ uint32_t bin_offsets(const uint32_t* __restrict__ hist, uint32_t* __restrict__ prow, int lane, int wid, const BinItem& it,
                                                uint32_t n_chunks, uint32_t* __restrict__ descriptors, uint32_t staging_base) {
    constexpr int CAP = 64 * BPL;
    uint32_t tot[BPL], mine[BPL], run = 0u;
#pragma unroll
    for (int k = 0; k < BPL; k++) tot[k] = mine[k] = 0u;
#pragma unroll BPL <= 2 ? WAVES : 1  // (8 x 8 loads in flight would cost the kernel its occupancy)
    for (int w = 0; w < WAVES; w++) {
#pragma unroll
        for (int k = 0; k < BPL; k++) tot[k] += hist[w * CAP + lane * BPL + k];
    }
#pragma unroll 1
    for (int w = 0; w < wid; w++) {  // (wave-uniform trip count) the records of the waves before this one
#pragma unroll
        for (int k = 0; k < BPL; k++) mine[k] += hist[w * CAP + lane * BPL + k];
    }
    // cursors count RECORDS from an EVEN base per bin: record c of the chunk sits in pair unit c >> 1, half c & 1
    uint32_t begin = 0u;
    {
        uint32_t even[BPL], erun = 0u;
#pragma unroll
        for (int k = 0; k < BPL; k++) { even[k] = (tot[k] + 1u) & ~1u; erun += even[k]; }
        uint32_t incl = erun;
        incl += row_shr<1>(incl);
        incl += row_shr<2>(incl);
        incl += row_shr<4>(incl);
        incl += row_shr<8>(incl);
        incl += bcast15_rows13(incl);
        incl += bcast31_rows23(incl);
        begin = incl - erun;
        (void)run;
        const uint32_t total_units = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) >> 1;
#pragma unroll
        for (int k = 0; k < BPL; k++) {
            prow[lane * BPL + k] = begin + mine[k];
            const uint32_t bin = (uint32_t)(lane * BPL + k);
            NGP_BOUNDS(begin + even[k] <= (uint32_t)(BIN_PPB * 8) + (uint32_t)CAP && it.chunk_x < n_chunks);
            if (wid == 0 && bin < it.n_bins) {
                descriptors[it.desc_base + (size_t)bin * n_chunks + it.chunk_x] = (begin >> 1) | (tot[k] << 16);  // first unit | records
                if (tot[k] & 1u)  // the unused second half of the bin's last pair: key 0, value +0
                    *(volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(staging_base + (begin + tot[k]) * 8u) = 0ull;
            }
            begin += even[k];
        }
        return total_units;
    }
}

#ifndef NGP_BIN_WAVES_PER_EU
#define NGP_BIN_WAVES_PER_EU 6
#endif
template <int D, int AMERGE /* run merge of the atomic workgroups: 3 = DPP over the wave, 1 = ds_bpermute */,
          int BPL /* bin counters per lane and wave: 2 (levels of <= 128 bins, 40 KiB of LDS) or 8 (<= 512 bins, 64 KiB) */>
__global__ __launch_bounds__(BIN_THREADS) __attribute__((amdgpu_waves_per_eu(BPL == 2 ? NGP_BIN_WAVES_PER_EU : 4, BPL == 2 ? NGP_BIN_WAVES_PER_EU : 4)))
void k_grid_backward_bin(const half_t* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
                         half_t* __restrict__ grad_grid, uint32_t B, GridLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp,
                         InputMap im, BinPlan plan, uint32_t* __restrict__ descriptors, uint2* __restrict__ records, AtomicPart ap) {
    uint32_t bin_block = blockIdx.x, n_bin_blocks = gridDim.x;
    if (ap.n_blocks) {
        const uint64_t total = gridDim.x;
        const uint32_t a0 = (uint32_t)(((uint64_t)blockIdx.x * ap.n_blocks) / total);
        const uint32_t a1 = (uint32_t)((((uint64_t)blockIdx.x + 1u) * ap.n_blocks) / total);
        if (a1 > a0) {  // this workgroup is the a0-th of the atomic kind
            backward_atomic_block<half_t, D, 2, AMERGE, BIN_THREADS>(grad, inputs, offsets, grad_grid, B, ap.levels.level[a0 / ap.blocks_per_level], lv,
                                                                gridtype, align_corners, interp, ap.points_per_block, im,
                                                                a0 % ap.blocks_per_level);
            return;
        }
        bin_block = blockIdx.x - a0;
        n_bin_blocks = gridDim.x - ap.n_blocks;
    }
    constexpr int C = 2;
    constexpr int NS = 1 << D;               // corner slots = records per sample
    constexpr int MAX_REC = BIN_PPB * NS;    // records per workgroup item = slots per chunk
    constexpr int WAVES = BIN_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char bin_smem[];
    unsigned char* staging = bin_smem;                                                // [MAX_REC + cap] records of 8 bytes (a pad record per odd bin)
    const uint32_t staging_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)staging;
    uint32_t* hist = reinterpret_cast<uint32_t*>(bin_smem + sizeof(uint2) * (MAX_REC + 64 * BPL));  // [WAVES][cap] records per (wave, bin)
    constexpr uint32_t cap = 64 * BPL;
    static_assert((MAX_REC / 2 + 32 * BPL) * BIN_UNIT_BYTES <= MAX_REC * sizeof(uint2), "pair units (plus half a unit of padding per bin) fit the chunk");
    uint32_t* pos = hist + WAVES * cap;                                               // [WAVES][cap] staging cursors per (wave, bin)
    const int tid = threadIdx.x, lane = tid & 63, wid = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* __restrict__ hrow = hist + wid * cap;
    uint32_t* __restrict__ prow = pos + wid * cap;
    // LDS byte addresses of the two rows
    const uint32_t hrow_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)hrow;
    const uint32_t prow_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)prow;
    const uint32_t n_levels = plan.n_levels;
    const uint32_t n_items = plan.n_chunks * n_levels;
    // this workgroup's items: [item, item_end), item = chunk * n_levels + li
    uint32_t item = (uint32_t)(((uint64_t)bin_block * n_items) / n_bin_blocks);
    const uint32_t item_end = (uint32_t)((((uint64_t)bin_block + 1u) * n_items) / n_bin_blocks);
    BinItem it;
    it.chunk_x = item / n_levels;
    it.li = item - it.chunk_x * n_levels;

    for (uint32_t i = lane; i < cap; i += 64) hrow[i] = 0u;
    // binned levels whose size on the device differs from the host copy the plan was made from (none, normally): one check per wave here
    // instead of two dependent scalar loads in front of every item
    unsigned long long levels_differ;
    {
        bool differ = false;
        if ((uint32_t)lane < n_levels) {
            const uint32_t level = plan.lc[lane][0];
            differ = (uint32_t)offsets[level + 1] - (uint32_t)offsets[level] != plan.lc[lane][3];
        }
        levels_differ = __builtin_amdgcn_ballot_w64(differ);
    }
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) x[d] = 0.0f;
    uint32_t g_cur = 0u, chunk_loaded = 0xffffffffu;
    auto fetch_g = [&](uint32_t chunk_x, uint32_t li) -> uint32_t {
        const uint32_t b = chunk_x * BIN_PPB + (uint32_t)tid;
        return b < B ? *reinterpret_cast<const uint32_t*>(grad + ((size_t)plan.lc[li][0] * B + b) * C) : 0u;
    };
    if (item < item_end) g_cur = fetch_g(it.chunk_x, it.li);
    __syncthreads();
#ifdef NGP_BIN_PHASE_PROBE
    unsigned long long t_probe = __builtin_amdgcn_s_memtime(), t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; item < item_end; item++) {
        NGP_PROBE_T(0)
        const uint32_t b = it.chunk_x * BIN_PPB + (uint32_t)tid;
        const bool in_range = b < B;
        if (it.chunk_x != chunk_loaded) {  // (wave-uniform) the positions: once per chunk
            chunk_loaded = it.chunk_x;
            if (in_range) {
#pragma unroll
                for (int d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
            }
        }
        uint32_t li_next = it.li + 1u, chunk_next = it.chunk_x;
        if (li_next == n_levels) { li_next = 0u; chunk_next++; }
        const uint32_t g_next = item + 1u < item_end ? fetch_g(chunk_next, li_next) : 0u;  // in flight during this item
        const uint32_t* __restrict__ lc = plan.lc[it.li];
        const uint32_t level = lc[0], flags = lc[2];
        it.n_bins = lc[1];
        it.scale = __builtin_bit_cast(float, lc[4]);
        it.desc_base = lc[7];
        it.interleaved = (flags & 1u) != 0u;
        LevelIndexer<D> indexer;
        indexer.size = lc[3];
        indexer.mask = lc[5];
        indexer.hashed = (flags & 4u) != 0u;
        indexer.need_mod = (flags & 8u) != 0u;
#pragma unroll
        for (int d = 0; d < D; d++) indexer.stride[d] = lc[8 + (d < 3 ? d : 2)];
        // the plan was made from the caller's HOST copy of the offsets; the device offsets are authoritative: where they describe another
        // level size (checked once per wave, above) index with them, and stay in bounds (records of a level that outgrew its bins take the atomic)
        it.plan_ok = true;
        it.gtable = grad_grid;
        if ((levels_differ >> it.li) & 1ull) {
            const uint32_t off0 = (uint32_t)offsets[level], size_dev = (uint32_t)offsets[level + 1] - off0;
            indexer.init(gridtype, align_corners, size_dev, lv.res[level]);
            it.plan_ok = size_dev <= (it.n_bins << BIN_SLICE_BITS);
            it.gtable = grad_grid + (size_t)off0 * C;
        }
        NGP_PROBE_T(1)
        BinLane<D> bl;
        LaneMask live_mask;
        const bool fast = indexer.hashed && indexer.mask != 0u && !it.interleaved && it.plan_ok;
        if (fast) bin_pass_count<D, true>(x, g_cur, in_range, it, indexer, align_corners, interp, im, hrow_base, bl, live_mask);
        else bin_pass_count<D, false>(x, g_cur, in_range, it, indexer, align_corners, interp, im, hrow_base, bl, live_mask);
        NGP_PROBE_T(2)
        __syncthreads();  // all counts of this item are in
        NGP_PROBE_T(3)
        const uint32_t total_units = bin_offsets<BPL, WAVES>(hist, prow, lane, wid, it, plan.n_chunks, descriptors, staging_base);
        NGP_PROBE_T(4)
        if (fast) bin_pass_place<D, true>(g_cur, it, bl, live_mask, prow_base, staging_base);
        else bin_pass_place<D, false>(g_cur, it, bl, live_mask, prow_base, staging_base);
        NGP_PROBE_T(5)
        __syncthreads();  // the sorted chunk is complete in LDS (and every wave has read every counter row)
        NGP_PROBE_T(6)
        for (uint32_t i = lane; i < cap; i += 64) hrow[i] = 0u;  // own counters for the next item
        // copy-out: records 2u, 2u + 1 of the staging area (16 bytes) become pair unit u of the chunk (12 bytes: the two 16-bit keys share a word)
        uint32_t* __restrict__ chunk = reinterpret_cast<uint32_t*>(records + ((size_t)it.li * plan.n_chunks + it.chunk_x) * MAX_REC);
        const uint4* staging2 = reinterpret_cast<const uint4*>(staging);
        for (uint32_t u = tid; u < total_units; u += BIN_THREADS) {
            const uint4 q = staging2[u];
            uint32_t* dst = chunk + 3u * u;
            const uint32_t keys = (q.x & 0xffffu) | (q.z << 16);
            const uint32_t unit[3] = {keys, q.y, q.w};
            __builtin_memcpy(dst, unit, sizeof(unit));  // 4-byte aligned: global_store_dwordx3
        }
        NGP_PROBE_T(7)
        g_cur = g_next;
        it.li = li_next;
        it.chunk_x = chunk_next;
#ifdef NGP_BIN_PHASE_PROBE
        asm volatile("" ::"v"(g_cur));
        NGP_PROBE_T(8)
        t_acc[9] += 1ull;
#endif
        // (the next item's staging stores come after ITS first barrier, i.e. after every wave has finished this copy)
    }
#ifdef NGP_BIN_PHASE_PROBE
    if (tid == 0) {
        for (int i = 0; i < 9; i++) atomicAdd(&g_bin_probe[i], t_acc[i]);
        atomicAdd(&g_bin_probe[15], t_acc[9]);
    }
#endif
}

// Pass 2, round 4: DENSE lanes.  The kernel is bound by its instruction stream (34 M VALU per launch, PMC), and the round-3 walk -- a group
// of 16 lanes per run, two records per lane, a second trip for the part of a run beyond 32 records -- kept about a third of the lanes
// busy: runs are Poisson-distributed around 32 records, so most waves made the second trip for a handful of records.  Now a wave takes
// a contiguous share of the slice's runs (one per chunk), scans their lengths, and walks the CONCATENATION of its runs 64 record pairs at
// a time: lane l of window w handles pair 64 w + l.  Which run a pair belongs to comes from a scatter of the run heads that fall into
// the window (one LDS byte array per wave) followed by a DPP max-scan over the lanes -- about 15 VALU per window of 128 records, whatever
// the run lengths.  Non-finite and scaled records (rare) leave the straight-line addend behind one wave-uniform branch.
// ADAM (round 6): the flush of a slice is also the table's Adam sweep for that slice (ngp_table_adam_t, include/ngp_hip.h): the final
// gradient of 4096 entries sits in LDS, most of the chip is waiting on record walks, and the separate 28 B/parameter sweep over the whole
// table that used to follow (k_adam: 61 us, HBM-bound) is exactly the kind of stream that fits under a latency-bound kernel.  The update is
// speculative -- parameters / moments of buffer set state[5] are read, the OTHER set is written; the commit flips the parity only when no
// gradient of the step was non-finite -- and uses the fp16-rounded gradient, i.e. the bits k_adam would have read from the stored table.
#ifndef NGP_TADAM_STORE_GRADIENT
#define NGP_TADAM_STORE_GRADIENT 0
#endif
#ifndef NGP_TADAM_PROBE
#define NGP_TADAM_PROBE 0   // timing probes only (tools/table_adam_probe.py): 1 = no Adam on the round-robin bins of the dense levels, 2 = no stores, 4 = no loads
#endif
template <int D, bool ADAM>
__global__ __launch_bounds__(ACC_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_grid_backward_accumulate(const int32_t* __restrict__ offsets, half_t* __restrict__ grad_grid,
                                                                          BinPlan plan, const uint32_t* __restrict__ descriptors,
                                                                          const uint2* __restrict__ records, float* __restrict__ found_inf,
                                                                          SlabSets slabs, bool overwrite, TableAdam ta) {
    constexpr int MAX_REC = BIN_PPB * (1 << D);
    constexpr int WAVES = ACC_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char acc_smem[];
    // one more row of the grid than there are levels: the caller's carried reductions (the two MLPs' weight-gradient slabs and the loss sum
    // of the training step, ngp_grid_encode_backward_checked_slabs) -- ~270 small blocks beside the first slices instead of a launch of their own
    const uint32_t slab_row = slabs.total_blocks() != 0u ? 1u : 0u;   // row 0 when present: dispatched first, done in ~4 us
    if (slab_row && blockIdx.y == 0u) {
        carried_block(slabs, blockIdx.x, reinterpret_cast<float (*)[RS_PARAMS]>(acc_smem), found_inf);
        return;
    }
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(acc_smem);                                // [BIN_SLICE][2]
    uint32_t* poison = reinterpret_cast<uint32_t*>(acc_smem + sizeof(unsigned long long) * 2 * BIN_SLICE);  // [BIN_SLICE / 16], 2 bits per entry
    uint2* run_table = reinterpret_cast<uint2*>(poison + BIN_SLICE / 16);                                   // [WAVES][64] {first record, first pair | records << 18}
    uint32_t* head_at = reinterpret_cast<uint32_t*>(run_table + WAVES * 64);                                // [WAVES][64] run (1-based) whose first pair sits at this lane of the window
    // levels in REVERSE order: the sort wrote the last level's records last, so they are the ones still in the memory-side cache
    // (same-box A/B: -4 us per iteration)
    // (Round 6 tried other row orders through a byte table in the plan -- dense levels first / alternating, to stagger the flush phases of the
    // Adam-carrying workgroups: no gain.  That build also stopped being reproducible at loss-scaled magnitudes (tests/test_gpu_grid.py caught
    // it).  The table was not the cause: the build happened to need all 40 registers of its allocation and put the addend's shift amount into
    // v39 -- and on gfx950 a 64-bit shift whose amount sits in the LAST allocated register misreads it now and then (isolated in
    // tools/probes/vgpr_last_probe.hip; guarded for every kernel by tests/test_isa_invariants.py; EXPERIMENTS.md round 6).)
    const uint32_t li = plan.n_levels - 1u - (blockIdx.y - slab_row), bin = blockIdx.x;
    if (bin >= plan.n_bins(li)) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * BIN_SLICE; i += ACC_THREADS) acc[i] = 0ull;
    if (tid < BIN_SLICE / 16) poison[tid] = 0u;
    __syncthreads();
    const uint32_t n_chunks = plan.n_chunks;
    const uint32_t* __restrict__ desc = descriptors + plan.desc_base(li) + (size_t)bin * n_chunks;
    const uint32_t* __restrict__ words = reinterpret_cast<const uint32_t*>(records + (size_t)li * n_chunks * MAX_REC);  // 2 words per record
    const bool interleaved = plan.interleaved(li);
    // ADAM: the slice's master weights and moments are REQUESTED HERE, in front of the record walk, and consumed by the flush behind it --
    // the walk is bound by latency (descriptors -> record windows), the requests ride under it; the step's constants (two powf, a sqrt,
    // a division) are computed under the same latency and parked in LDS.  The round-robin bins of the dense levels (entries 128 apart: 8-byte
    // accesses at a stride of 1 KiB) do NOT take part: their flush stores the gradient as usual and the step's closing launch
    // (ngp_optim_adam_small_commit) sweeps that prefix of the table contiguously -- measured: 45 of the fused flush's 94 us were those bins.
    constexpr int TRIPS = BIN_SLICE / (2 * ACC_THREADS);
    static_assert(BIN_SLICE % (2 * ACC_THREADS) == 0, "slice entries divide evenly over lane pairs");
    float* adam_lds = reinterpret_cast<float*>(head_at + WAVES * 64);   // [4]: inv_scale, step_size, bc2_sqrt, source set
    const bool adam_here = ADAM && !interleaved;
    float4_t pp[TRIPS], pm[TRIPS], pv[TRIPS];
    bool wide[TRIPS];
    // (requested in FRONT of the walk, straight-line.  Requesting behind the first two record windows -- so that the descriptors and those
    // windows are not queued behind the burst: vector-memory returns are in order -- needs the values live across a branchy region: the
    // compiler spilled 65 registers and the kernel went from 95 to 125 us, EXPERIMENTS.md round 6)
    if constexpr (ADAM) {
        if (adam_here) {
            const uint32_t level_ = plan.level(li);
            const uint32_t off0_ = (uint32_t)offsets[level_], size_ = (uint32_t)offsets[level_ + 1] - off0_;
            const uint32_t src = ta.state[5] != 0.0f ? 1u : 0u;
            const float* __restrict__ p_in = ta.p[src] + (size_t)off0_ * 2;
            const float* __restrict__ m_in = ta.m[src] + (size_t)off0_ * 2;
            const float* __restrict__ v_in = ta.v[src] + (size_t)off0_ * 2;
#pragma unroll
            for (int k = 0; k < TRIPS; k++) {
                const uint32_t e0 = bin * BIN_SLICE + 2u * ((uint32_t)tid + (uint32_t)k * ACC_THREADS);
                wide[k] = e0 + 1u < size_ && ((off0_ + e0) & 1u) == 0u;   // a 16-byte aligned pair of entries inside the level
                pp[k] = pm[k] = pv[k] = float4_t{0.0f, 0.0f, 0.0f, 0.0f};
                if (wide[k] && !(NGP_TADAM_PROBE & 4)) {
                    pp[k] = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(p_in + (size_t)e0 * 2));
                    pm[k] = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(m_in + (size_t)e0 * 2));
                    pv[k] = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(v_in + (size_t)e0 * 2));
                }
            }
            if (wid == 0) {   // one wave computes the constants (every lane the same values), one lane parks them
                const AdamConsts ac = adam_consts(ta.state, ta.beta1, ta.beta2, 1.0f);
                if (lane == 0) {
                    adam_lds[0] = ac.inv_scale;
                    adam_lds[1] = ta.lr * ac.lr_mult / ac.bc1;
                    adam_lds[2] = ac.bc2_sqrt;
                    adam_lds[3] = (float)src;
                }
            }
        }
    }
    // Exact fixed-point addend of a finite fp16 value v (11 significant bits), straight-line for BOTH magnitude ranges:
    //   |v| <  128: v * 2^24 is an integer below 2^31                      -> q = (int32) (v * 2^24), addend = q
    //   |v| >= 128: v is a multiple of 2^-3 (ulp of the binade of 128)     -> q = (int32) (v * 8) (<= 524032), addend = q << 21
    // one multiply, one conversion, one 64-bit shift by a selected amount; both channels are added unconditionally (a zero addend is
    // harmless).
    auto fixed_addend = [](float v) -> unsigned long long {
        const bool big = __builtin_fabsf(v) >= 128.0f;
        const int32_t q = (int32_t)(v * (big ? 8.0f : 0x1p24f));
        return (unsigned long long)(long long)q << (big ? 21 : 0);  // (shifted as unsigned: two's complement, exact mod 2^64)
    };
    auto entry_of = [&](uint32_t key) { return key & (BIN_SLICE - 1u); };  // (16-bit keys: the entry inside the slice, two scale flags)
    auto add_plain = [&](const uint32_t key, const uint32_t val) {  // finite, unscaled
        const uint32_t idx = entry_of(key);
        NGP_BOUNDS(idx < (uint32_t)BIN_SLICE);
        const half2_t hv = __builtin_bit_cast(half2_t, val);
        __hip_atomic_fetch_add(&acc[2 * idx], fixed_addend((float)hv.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&acc[2 * idx + 1], fixed_addend((float)hv.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto add_general = [&](const uint32_t key, const uint32_t val) {  // inf / NaN poison their channel; a scaled channel carries 1/64
        const uint32_t idx = entry_of(key);
        NGP_BOUNDS(idx < (uint32_t)BIN_SLICE);
        const int up0 = (key & BIN_KEY_SCALED0) ? 6 : 0, up1 = (key & BIN_KEY_SCALED1) ? 6 : 0;
        const half2_t hv = __builtin_bit_cast(half2_t, val);
        const bool fin0 = (val & 0x7c00u) != 0x7c00u, fin1 = (val & 0x7c000000u) != 0x7c000000u;
        __hip_atomic_fetch_add(&acc[2 * idx], fixed_addend(fin0 ? (float)hv.x : 0.0f) << up0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&acc[2 * idx + 1], fixed_addend(fin1 ? (float)hv.y : 0.0f) << up1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!(fin0 && fin1)) atomicOr(&poison[idx >> 4], ((fin0 ? 0u : 1u) | (fin1 ? 0u : 2u)) << ((idx & 15u) * 2u));
    };
    auto special = [](uint32_t key, uint32_t val) {  // a record the straight-line addend does not cover
        return (key & BIN_KEY_SCALED) != 0u || (val & 0x7c00u) == 0x7c00u || (val & 0x7c000000u) == 0x7c000000u;
    };
    // every workgroup of a level walks the same chunks: start each one somewhere else
    const uint32_t rot = (bin * 97u) % n_chunks;
    auto chunk_of = [&](uint32_t i) { const uint32_t k = i + rot; return k >= n_chunks ? k - n_chunks : k; };
    // (lanes of the wave talk to each other through these two arrays with nothing but the LDS's in-order execution between a store and
    // the load of another lane: the accesses are volatile, otherwise the compiler forwards a lane's OWN earlier store to its load)
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_words_t;
    lds_words_t my_runs = (lds_words_t)(__attribute__((address_space(3))) void*)(run_table + wid * 64);
    lds_words_t my_heads = (lds_words_t)(__attribute__((address_space(3))) void*)(head_at + wid * 64);
    const uint32_t per_wave = (n_chunks + WAVES - 1) / WAVES;
    const uint32_t run_end = min(n_chunks, (uint32_t)(wid + 1) * per_wave);
    for (uint32_t run0 = (uint32_t)wid * per_wave; run0 < run_end; run0 += 64u) {  // this wave's runs, 64 at a time (a lane each)
        const uint32_t i = run0 + (uint32_t)lane;
        uint32_t first = 0u, cnt = 0u;
        if (i < run_end) {
            const uint32_t k = chunk_of(i), d = desc[k];
            cnt = d >> 16;
            first = k * (uint32_t)(2 * MAX_REC) + (d & 0xffffu) * 3u;  // WORD index of the run's first pair unit inside the level: < 2^31
        }
        const uint32_t pairs = (cnt + 1u) >> 1;  // 16-byte accesses: two records per lane (an odd run ends in a half-used pair)
        uint32_t incl = pairs;
        incl += row_shr<1>(incl);
        incl += row_shr<2>(incl);
        incl += row_shr<4>(incl);
        incl += row_shr<8>(incl);
        incl += bcast15_rows13(incl);
        incl += bcast31_rows23(incl);
        const uint32_t total_pairs = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t start = incl - pairs;
        my_runs[2 * lane] = first;
        my_runs[2 * lane + 1] = start | (cnt << 18);  // (start < 64 * 2048 = 2^17, cnt <= 4096)
        uint32_t carry = 0u;  // run (1-based) that covers the pair before the window
        // one window: which run does pair w0 + lane belong to (heads inside the window are scattered, the rest follows by a max-scan),
        // then the 16-byte load of the pair.  valid: bit 0 = first record, bit 1 = second record of the pair exists
        auto fetch = [&](uint32_t w0, uint4& q, uint32_t& valid) {
            my_heads[lane] = 0u;
            if (pairs != 0u && start - w0 < 64u) my_heads[start - w0] = (uint32_t)lane + 1u;  // (unsigned: start >= w0)
            uint32_t r = my_heads[lane];
            r = max(r, row_shr<1>(r));
            r = max(r, row_shr<2>(r));
            r = max(r, row_shr<4>(r));
            r = max(r, row_shr<8>(r));
            r = max(r, bcast15_rows13(r));
            r = max(r, bcast31_rows23(r));
            r = max(r, carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
            const uint32_t p = w0 + (uint32_t)lane;
            const bool have = p < total_pairs;  // (r >= 1 then: pair 0 is the head of the first non-empty run)
            const uint32_t ri = have ? r - 1u : 0u;
            const uint32_t r_first = my_runs[2 * ri], r_word = my_runs[2 * ri + 1];
            const uint32_t r_start = r_word & 0x3ffffu, r_cnt = r_word >> 18;
            const uint32_t rec = 2u * (p - r_start);  // first record of the pair inside its run
            q = make_uint4(0u, 0u, 0u, 0u);
            NGP_BOUNDS(!have || (r >= 1u && r <= 64u && rec < r_cnt && r_first + 3u * (p - r_start) + 2u < n_chunks * (uint32_t)(2 * MAX_REC)));
            if (have) {  // one pair unit: {keys, value 0, value 1} (4-byte aligned: global_load_dwordx3)
                uint32_t u[3];
                __builtin_memcpy(u, words + (r_first + 3u * (p - r_start)), sizeof(u));
                q = make_uint4(u[0] & 0xffffu, u[1], u[0] >> 16, u[2]);
            }
            valid = have ? (rec + 1u < r_cnt ? 3u : 1u) : 0u;
        };
        // two windows ahead: the loads of windows w + 1 and w + 2 are in flight during the adds of window w (the kernel waits on memory)
        uint4 q1, q2;
        uint32_t valid1 = 0u, valid2 = 0u;
        if (total_pairs != 0u) fetch(0u, q1, valid1);
        if (total_pairs > 64u) fetch(64u, q2, valid2);
        for (uint32_t w0 = 0u; w0 < total_pairs; w0 += 64u) {
            const uint4 q = q1;
            const uint32_t valid = valid1;
            q1 = q2;
            valid1 = valid2;
            valid2 = 0u;
            if (w0 + 128u < total_pairs) fetch(w0 + 128u, q2, valid2);
            const bool have = (valid & 1u) != 0u, second = (valid & 2u) != 0u;
            if (__builtin_amdgcn_ballot_w64((have && special(q.x, q.y)) || (second && special(q.z, q.w))) == 0ull) {
                if (have) add_plain(q.x, q.y);
                if (second) add_plain(q.z, q.w);
            } else {
                if (have) add_general(q.x, q.y);
                if (second) add_general(q.z, q.w);
            }
        }
    }
    __syncthreads();
    const uint32_t level = plan.level(li);
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    half2_t* __restrict__ gtable = reinterpret_cast<half2_t*>(grad_grid + (size_t)off0 * 2);
    bool nonfinite = false;
    if constexpr (ADAM) {
        if (adam_here) {
            // Two ADJACENT entries per lane and trip: 16-byte accesses on the fp32 streams (8-byte global accesses run at 0.5-0.7x that rate
            // on this chip, optim.hip).  The pairs that were not requested ahead (a level whose end is not pair-aligned) are fetched here.
            const float inv_scale = adam_lds[0], step_size = adam_lds[1], bc2_sqrt = adam_lds[2];
            const uint32_t src = adam_lds[3] != 0.0f ? 1u : 0u, dst = src ^ 1u;
            float* __restrict__ p_out = ta.p[dst] + (size_t)off0 * 2;
            float* __restrict__ m_out = ta.m[dst] + (size_t)off0 * 2;
            float* __restrict__ v_out = ta.v[dst] + (size_t)off0 * 2;
            half_t* __restrict__ h_out = ta.p16[dst] + (size_t)off0 * 2;
#pragma unroll
            for (int k = 0; k < TRIPS; k++) {
                const uint32_t i = 2u * ((uint32_t)tid + (uint32_t)k * ACC_THREADS);
                const uint32_t e0 = bin * BIN_SLICE + i;
                const bool ok0 = e0 < hashmap_size, ok1 = e0 + 1u < hashmap_size;
                if (!wide[k]) {
                    const float* __restrict__ p_in = ta.p[src] + (size_t)off0 * 2;
                    const float* __restrict__ m_in = ta.m[src] + (size_t)off0 * 2;
                    const float* __restrict__ v_in = ta.v[src] + (size_t)off0 * 2;
                    if (ok0) {
                        const float2_t a = *reinterpret_cast<const float2_t*>(p_in + (size_t)e0 * 2), b = *reinterpret_cast<const float2_t*>(m_in + (size_t)e0 * 2),
                                       c = *reinterpret_cast<const float2_t*>(v_in + (size_t)e0 * 2);
                        pp[k].x = a.x; pp[k].y = a.y; pm[k].x = b.x; pm[k].y = b.y; pv[k].x = c.x; pv[k].y = c.y;
                    }
                    if (ok1) {
                        const float2_t a = *reinterpret_cast<const float2_t*>(p_in + (size_t)e0 * 2 + 2), b = *reinterpret_cast<const float2_t*>(m_in + (size_t)e0 * 2 + 2),
                                       c = *reinterpret_cast<const float2_t*>(v_in + (size_t)e0 * 2 + 2);
                        pp[k].z = a.x; pp[k].w = a.y; pm[k].z = b.x; pm[k].w = b.y; pv[k].z = c.x; pv[k].w = c.y;
                    }
                }
                const float nan = __builtin_nanf("");
                half4_t g16, h16;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t ii = i + (uint32_t)(c >> 1);
                    const long long sum = (long long)acc[2 * ii + (c & 1)];
                    const uint32_t bad = (poison[ii >> 4] >> ((ii & 15u) * 2u + (uint32_t)(c & 1))) & 1u;
                    // the gradient as it would be stored: the exact sum rounded once to fp16 (0 + sum: what adding into a zeroed entry gives)
                    g16[c] = (half_t)(0.0f + (bad ? nan : (float)sum * 0x1p-24f));
                    const bool live = (c >> 1) ? ok1 : ok0;
                    nonfinite = nonfinite || (live && !__builtin_isfinite((float)g16[c]));
                    const float g = (float)g16[c] * inv_scale;
                    float m_ = pm[k][c], v_ = pv[k][c], p_ = pp[k][c];
                    adam_element(g, m_, v_, p_, ta.beta1, ta.beta2, ta.eps, step_size, bc2_sqrt);
                    pm[k][c] = m_; pv[k][c] = v_; pp[k][c] = p_;
                    h16[c] = (half_t)p_;
                }
                if ((NGP_TADAM_PROBE & 2) && pp[k][0] != 123.456f) continue;
                if (wide[k]) {
                    __builtin_nontemporal_store(pm[k], reinterpret_cast<float4_t*>(m_out + (size_t)e0 * 2));
                    __builtin_nontemporal_store(pv[k], reinterpret_cast<float4_t*>(v_out + (size_t)e0 * 2));
                    __builtin_nontemporal_store(pp[k], reinterpret_cast<float4_t*>(p_out + (size_t)e0 * 2));
                    *reinterpret_cast<half4_t*>(h_out + (size_t)e0 * 2) = h16;
                    // (the gradient is consumed right here: it is not stored -- 23 MB of writes per step; -DNGP_TADAM_STORE_GRADIENT for debugging)
                    if (NGP_TADAM_STORE_GRADIENT && grad_grid) *reinterpret_cast<half4_t*>(gtable + e0) = g16;
                } else {
                    if (ok0) {
                        *reinterpret_cast<float2_t*>(m_out + (size_t)e0 * 2) = float2_t{pm[k].x, pm[k].y};
                        *reinterpret_cast<float2_t*>(v_out + (size_t)e0 * 2) = float2_t{pv[k].x, pv[k].y};
                        *reinterpret_cast<float2_t*>(p_out + (size_t)e0 * 2) = float2_t{pp[k].x, pp[k].y};
                        *reinterpret_cast<half2_t*>(h_out + (size_t)e0 * 2) = half2_t{h16[0], h16[1]};
                        if (NGP_TADAM_STORE_GRADIENT && grad_grid) gtable[e0] = half2_t{g16[0], g16[1]};
                    }
                    if (ok1) {
                        *reinterpret_cast<float2_t*>(m_out + (size_t)e0 * 2 + 2) = float2_t{pm[k].z, pm[k].w};
                        *reinterpret_cast<float2_t*>(v_out + (size_t)e0 * 2 + 2) = float2_t{pv[k].z, pv[k].w};
                        *reinterpret_cast<float2_t*>(p_out + (size_t)e0 * 2 + 2) = float2_t{pp[k].z, pp[k].w};
                        *reinterpret_cast<half2_t*>(h_out + (size_t)e0 * 2 + 2) = half2_t{h16[2], h16[3]};
                        if (NGP_TADAM_STORE_GRADIENT && grad_grid) gtable[e0 + 1u] = half2_t{g16[2], g16[3]};
                    }
                }
            }
            if (found_inf && __any(nonfinite) && (tid & 63) == 0) found_inf[0] = 1.0f;
            return;
        }
    }
    // every thread owns BIN_SLICE / ACC_THREADS entries: their old values are loaded TOGETHER, then added to and stored (written as one
    // load-add-store per loop trip the compiler has to order each load behind the previous store -- it cannot see that the entries
    // differ -- and the tail of every workgroup became that many dependent round trips to memory)
    constexpr int PER_THREAD = BIN_SLICE / ACC_THREADS;
    static_assert(BIN_SLICE % ACC_THREADS == 0, "slice entries divide evenly over the workgroup");
    uint32_t ent[PER_THREAD];
    long long sum0[PER_THREAD], sum1[PER_THREAD];
    uint32_t badb[PER_THREAD];
    half2_t oldv[PER_THREAD];
    bool touch[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * ACC_THREADS;
        ent[k] = interleaved ? (i << BIN_DENSE_BITS) + bin : bin * BIN_SLICE + i;
        sum0[k] = (long long)acc[2 * i];
        sum1[k] = (long long)acc[2 * i + 1];
        badb[k] = (poison[i >> 4] >> ((i & 15u) * 2u)) & 3u;
        // overwrite: every entry of the slice is written (a zero sum stores +0: what adding it to a zeroed entry gives) and none is read
        touch[k] = ent[k] < hashmap_size && (overwrite || !(sum0[k] == 0 && sum1[k] == 0 && !badb[k]));
        oldv[k] = half2_t{(half_t)0.0f, (half_t)0.0f};
    }
    if (!overwrite) {
#pragma unroll
        for (int k = 0; k < PER_THREAD; k++)
            if (touch[k]) oldv[k] = gtable[ent[k]];
    }
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        if (!touch[k]) continue;
        const float nan = __builtin_nanf("");
        half2_t nu;
        nu.x = (half_t)((float)oldv[k].x + ((badb[k] & 1u) ? nan : (float)sum0[k] * 0x1p-24f));
        nu.y = (half_t)((float)oldv[k].y + ((badb[k] & 2u) ? nan : (float)sum1[k] * 0x1p-24f));
        gtable[ent[k]] = nu;
        nonfinite = nonfinite || !__builtin_isfinite((float)nu.x) || !__builtin_isfinite((float)nu.y);
    }
    // the optimizer's non-finite sweep over the table, done where the final values are produced (benign race: every writer stores 1)
    if (found_inf && __any(nonfinite) && (tid & 63) == 0) found_inf[0] = 1.0f;
}

// gridencoder.cu:343-369
template <typename T>
__global__ void k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                      uint32_t B, uint32_t L, uint32_t D, uint32_t C) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t c = 0; c < C; c++)
            r = __builtin_fmaf((float)grad[((size_t)l * B + b) * C + c], (float)dd[(l * D + d) * C + c], r);
    grad_inputs[t] = (T)r;
}

// gridencoder.cu:506-610
template <typename T, int D, int C>
__global__ __launch_bounds__(FWD_THREADS) void k_grad_tv(const T* __restrict__ inputs, const T* __restrict__ grid,
                                                         T* __restrict__ grad, const int32_t* __restrict__ offsets,
                                                         float weight, uint32_t B, GridLevels lv, uint32_t gridtype,
                                                         bool align_corners) {
    const uint32_t level = blockIdx.y;
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const uint32_t resolution = lv.res[level];
    LevelIndexer<D> indexer;
    indexer.init(gridtype, align_corners, hashmap_size, resolution);
    const T* __restrict__ table = grid + (size_t)off0 * C;
    T* __restrict__ gtable = grad + (size_t)off0 * C;
    const float w = weight / (float)(2 * D);
    for (uint32_t b = blockIdx.x * FWD_THREADS + threadIdx.x; b < B; b += gridDim.x * FWD_THREADS) {
        float x[D];
        bool inside = true;
#pragma unroll
        for (int d = 0; d < D; d++) {
            x[d] = (float)inputs[(size_t)b * D + d];
            inside = inside && !(x[d] < 0.0f || x[d] > 1.0f);
        }
        if (!inside) continue;
        uint32_t pg[D];
#pragma unroll
        for (int d = 0; d < D; d++) pg[d] = (uint32_t)floorf(__builtin_fmaf(x[d], lv.scale[level], align_corners ? 0.0f : 0.5f));
        const uint32_t idx = indexer(pg) * C;
        Vec<T, C> ctr;
        ctr.load(table + idx);
        float res[C], idelta[C];
#pragma unroll
        for (int c = 0; c < C; c++) res[c] = idelta[c] = 0.0f;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t cur = pg[d];
            if (cur < resolution) {
                pg[d] = cur + 1u;
                Vec<T, C> o;
                o.load(table + (size_t)indexer(pg) * C);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    float gv = ctr.v[c] - o.v[c];
                    res[c] += gv;
                    idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
                }
            }
            if (cur > 0u) {
                pg[d] = cur - 1u;
                Vec<T, C> o;
                o.load(table + (size_t)indexer(pg) * C);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    float gv = ctr.v[c] - o.v[c];
                    res[c] += gv;
                    idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
                }
            }
            pg[d] = cur;
        }
        float upd[C];
#pragma unroll
        for (int c = 0; c < C; c++) upd[c] = w * res[c] * rsqrtf(idelta[c] + 1e-9f);
        scatter_add<T, C>(gtable + idx, upd, 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static void fill_levels(GridLevels& lv, uint32_t L, float S, uint32_t H) {
    ngp_grid_level_table(L, S, H, lv.scale, lv.res);
}

static uint32_t fwd_blocks(uint32_t B) {
    // >= 256 workgroups per level already at B = 64k; cap so the level-major order stays meaningful
    uint32_t nb = cdiv(B, FWD_THREADS);
    return nb < 1 ? 1 : (nb > 65535u ? 65535u : nb);
}

#ifndef NGP_FWD_BALANCE
#define NGP_FWD_BALANCE 1
#endif
// level_cost[l]: relative time of one tile of level l (nullptr: unknown -> every level costs the same -> whole levels only, the plain
// (level mod 8) placement).  Returns the number of slots of the longest work list.
static_assert(NGP_MAX_LEVELS <= 8 * FWD_MAX_SEG, "whole levels alone must fit the per-XCD work lists");
static uint32_t build_forward_schedule(FwdSchedule& sc, uint32_t L, uint32_t tiles, const float* level_cost) {
    struct Seg { uint32_t level, tile0, n; };
    // (a cost vector that is not positive and finite is ignored here -- the entry points that take one from a caller refuse it first)
    for (uint32_t l = 0; level_cost && l < L; l++)
        if (!(level_cost[l] > 0.0f && level_cost[l] < 1e6f)) { level_cost = nullptr; break; }
    std::vector<Seg> list[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto cost = [&](uint32_t l) { return level_cost ? (double)level_cost[l] : 1.0; };
    for (uint32_t l = 0; l < L; l++) {
#ifdef NGP_FWD_LEVEL_MASK_PROBE  // timing probe only (profiles/r03_grid_forward_levels.txt): launch only the levels of a bit mask
        if (const char* mk = getenv("NGP_FWD_LEVEL_MASK")) { if (!((strtoul(mk, nullptr, 0) >> l) & 1ul)) continue; }
#endif
        list[l & 7u].push_back({l, 0u, tiles});
        load[l & 7u] += cost(l) * tiles;
    }
    if (NGP_FWD_BALANCE && level_cost && L > 8) {
        double target = 0.0;
        for (int x = 0; x < 8; x++) target += load[x] / 8.0;
        for (int round = 0; round < 16; round++) {
            int hi = 0, lo = 0;
            for (int x = 1; x < 8; x++) {
                if (load[x] > load[hi]) hi = x;
                if (load[x] < load[lo]) lo = x;
            }
            if (load[hi] - target < 0.02 * target || list[lo].size() >= (size_t)FWD_MAX_SEG || list[hi].empty()) break;
            Seg& last = list[hi].back();               // the donor gives away the END of the level it processes last
            const double c = cost(last.level);
            const double want = std::min(load[hi] - target, target - load[lo]) / c;
            uint32_t n = want >= (double)tiles ? tiles : (want > 0.0 ? (uint32_t)want : 0u);
            if (n == 0) break;
            if (n >= last.n) n = last.n > 1 ? last.n - 1 : 0;
            if (n == 0) break;
            last.n -= n;
            list[lo].push_back({last.level, last.tile0 + last.n, n});
            load[hi] -= c * n;
            load[lo] += c * n;
        }
    }
    uint32_t max_slots = 0;
    for (int x = 0; x < 8; x++) {
        uint32_t e = 0;
        for (int sg = 0; sg < FWD_MAX_SEG; sg++) {
            if (sg < (int)list[x].size()) {
                sc.level[x][sg] = (uint16_t)list[x][sg].level;
                sc.tile0[x][sg] = list[x][sg].tile0;
                e += list[x][sg].n;
            } else {
                sc.level[x][sg] = 0xffffu;
                sc.tile0[x][sg] = 0u;
            }
            sc.end[x][sg] = e;
        }
        max_slots = e > max_slots ? e : max_slots;
    }
    return max_slots;
}

template <typename T, int D, int C>
static int launch_forward(const float* inputs, const void* emb, const int32_t* offsets, void* outputs, uint32_t B,
                          uint32_t L, const GridLevels& lv, void* dy_dx, uint32_t gridtype, bool ac, uint32_t interp,
                          InputMap im, const float* level_cost, TableSel sel, hipStream_t st) {
    if (dy_dx) {
        if (sel.parity) {
            set_error("grid_encode_forward: the double-buffered table selection does not serve the dy_dx kernel");
            return NGP_ERR_INVALID;
        }
        dim3 grid(fwd_blocks(B), L, 1);
        hipLaunchKernelGGL((k_grid_forward<T, D, C, true>), grid, dim3(FWD_THREADS), 0, st, inputs, (const T*)emb, offsets,
                           (T*)outputs, B, L, lv, (T*)dy_dx, gridtype, ac, interp);
        return check_launch("grid_encode_forward(dy_dx)");
    }
    // >= ~8 workgroups per CU over the whole launch; a workgroup covers ppb consecutive points of one level
    uint32_t ppb = 1024;
    while (ppb > 128 && (uint64_t)cdiv(B, ppb) * L < 2048) ppb >>= 1;
    const uint32_t tiles = cdiv(B, ppb);
    FwdSchedule sched;
    const uint32_t max_slots = build_forward_schedule(sched, L, tiles, level_cost);
    if constexpr (sizeof(T) == 2 && D == 3 && C == 2) {
        if (!ac) {  // the instant-ngp shape: index mode resolved per workgroup, dense levels one lane per point (k_grid_forward_fast)
            const bool smooth = interp == 1u, mapped = im.scale != 0.0f;
#define NGP_FWD_FAST_LAUNCH(S, M)                                                                                                        \
            hipLaunchKernelGGL((k_grid_forward_fast<S, M>), dim3(8u * max_slots), dim3(FWD_THREADS), 0, st, inputs, (const half_t*)emb, \
                               offsets, (half_t*)outputs, B, L, lv, gridtype, interp, sched, ppb, im, sel)
            if (smooth && mapped) NGP_FWD_FAST_LAUNCH(true, true);
            else if (smooth) NGP_FWD_FAST_LAUNCH(true, false);
            else if (mapped) NGP_FWD_FAST_LAUNCH(false, true);
            else NGP_FWD_FAST_LAUNCH(false, false);
#undef NGP_FWD_FAST_LAUNCH
            return check_launch("grid_encode_forward");
        }
    }
    hipLaunchKernelGGL((k_grid_forward_pair<T, D, C>), dim3(8u * max_slots), dim3(FWD_THREADS), 0, st, inputs, (const T*)emb, offsets,
                       (T*)outputs, B, L, lv, gridtype, ac, interp, sched, ppb, im, sel);
    return check_launch("grid_encode_forward");
}

// Compile-time experiment knobs of the backward (no run-time switches in the product path):
//   -DNGP_GRID_BWD_VARIANT=1 no run merge in the atomic kernel, 2 merge inside 16-lane rows, 3 wave-wide merge through ds_bpermute; 0 = default
//   -DNGP_GRID_BWD_BIN_FROM=<level> first level that may be binned (99 = never); -1 = every eligible level
//   -DNGP_GRID_BWD_SEPARATE=1 atomic levels in a launch of their own instead of riding in the record sort's
#ifndef NGP_GRID_BWD_VARIANT
#define NGP_GRID_BWD_VARIANT 0
#endif
#ifndef NGP_GRID_BWD_BIN_FROM
#define NGP_GRID_BWD_BIN_FROM -1
#endif
#ifndef NGP_GRID_BWD_SEPARATE
#define NGP_GRID_BWD_SEPARATE 0
#endif
static constexpr int grid_backward_variant() { return NGP_GRID_BWD_VARIANT; }

struct BackwardPlan {
    LevelList atomic_levels;
    uint32_t n_atomic = 0;
    BinPlan bins;
    uint32_t n_binned = 0, total_desc = 0, max_bins = 0;
    uint64_t total_records = 0;
    float* found_inf = nullptr;  // optional: set to 1 when a gradient value this call produced is not finite
    bool overwrite = false;      // the table gradient is written, not added to (every entry: needs every level on the record-sort path)
    SlabSets slabs = {};         // optional: slab reduction carried by the accumulate launch (blocks[] = 0: none)
    bool adam = false;           // the accumulate's flush applies Adam to the table (needs overwrite; ngp_table_adam_t)
    TableAdam table_adam = {};
    mutable bool slabs_done = false;  // set by the launch that carried them
    size_t desc_bytes() const { return ((size_t)total_desc * sizeof(uint32_t) + 255) & ~(size_t)255; }
    // (+64: the 16-byte load of a one-record run at the very end of the last chunk reads 8 bytes past it)
    size_t workspace_bytes() const { return n_binned ? desc_bytes() + (size_t)total_records * sizeof(uint2) + 64 : 0; }
};

constexpr uint32_t BIN_MIN_SAMPLES = 16384;       // below this the launch overheads of the two extra kernels win
constexpr uint32_t BIN_MAX_SAMPLES = 1u << 24;

static constexpr int bin_first_level() { return NGP_GRID_BWD_BIN_FROM; }

// Every level of an eligible call is binned: hashed levels in contiguous slices, dense levels in round-robin bins.
static uint32_t float_bits(float v) {
    uint32_t u;
    memcpy(&u, &v, sizeof(u));
    return u;
}

static void plan_backward(BackwardPlan& p, const int32_t* offsets_host, const GridLevels& lv, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                          int dtype, uint32_t gridtype, bool align_corners, bool have_workspace) {
    const bool eligible = have_workspace && offsets_host && dtype == NGP_F16 && C == 2 && (D == 2 || D == 3) && B >= BIN_MIN_SAMPLES &&
                          B <= BIN_MAX_SAMPLES;
    const int first = bin_first_level();
    p.bins.n_chunks = cdiv(B, (uint32_t)BIN_PPB);
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t size = eligible ? (uint32_t)(offsets_host[l + 1] - offsets_host[l]) : 0u;
        double dense = 1.0;
        for (uint32_t d = 0; d < D; d++) dense *= (double)(align_corners ? lv.res[l] : lv.res[l] + 1u);
        const bool hashed = gridtype == 0u && dense > (double)size;
        // hashed levels: contiguous 4096-entry slices (records spread evenly by construction); dense levels: 128 round-robin bins
        const bool interleave = !hashed && size <= (uint32_t)BIN_DENSE_BINS * BIN_SLICE;
        const uint32_t n_bins = interleave ? (uint32_t)BIN_DENSE_BINS : (size + BIN_SLICE - 1) >> BIN_SLICE_BITS;
        const bool binned = eligible && size >= 1 && n_bins <= (uint32_t)BIN_MAX_BINS && (first >= 0 ? (int)l >= first : (hashed || interleave));
        if (!binned) {
            p.atomic_levels.level[p.n_atomic++] = (uint8_t)l;
            continue;
        }
        const uint32_t i = p.n_binned++;
        uint32_t* lc = p.bins.lc[i];
        for (int k = 0; k < BIN_LC_WORDS; k++) lc[k] = 0u;
        lc[0] = l;
        lc[1] = n_bins;
        lc[3] = size;
        lc[4] = float_bits(lv.scale[l]);
        lc[7] = p.total_desc;
        {   // the indexer's constants exactly as LevelIndexer::init derives them on the device (32-bit arithmetic)
            uint32_t s = 1, stride[3] = {0u, 0u, 0u};
            for (uint32_t d = 0; d < D && d < 3; d++) {
                if (s <= size) {
                    stride[d] = s;
                    s *= align_corners ? lv.res[l] : (lv.res[l] + 1u);
                }
            }
            const bool hashed_dev = gridtype == 0u && s > size;
            const bool need_mod = hashed_dev || s > size || align_corners;
            lc[2] = (interleave ? 1u : 0u) | (hashed_dev ? 4u : 0u) | (need_mod ? 8u : 0u);
            lc[5] = ((size & (size - 1u)) == 0u) ? size - 1u : 0u;
            lc[8] = stride[0]; lc[9] = stride[1]; lc[10] = stride[2];
        }
        p.total_desc += n_bins * p.bins.n_chunks;
        p.total_records += (uint64_t)p.bins.n_chunks * BIN_PPB * (1u << D);
        p.max_bins = n_bins > p.max_bins ? n_bins : p.max_bins;
    }
}

template <int D, int AMERGE>
static int launch_backward_bins(const void* grad, const float* inputs, const int32_t* offsets, void* grad_emb, uint32_t B,
                                const GridLevels& lv, uint32_t gridtype, bool ac, uint32_t interp, InputMap im, const BackwardPlan& p,
                                void* workspace, bool with_atomic_levels, hipStream_t st) {
    constexpr size_t acc_smem = sizeof(unsigned long long) * 2 * BIN_SLICE + sizeof(uint32_t) * (BIN_SLICE / 16) +
                                (ACC_THREADS / 64) * 64 * (sizeof(uint2) + sizeof(uint32_t)) + 4 * sizeof(float);
    const uint32_t bins_cap = p.max_bins <= 128u ? 128u : (uint32_t)BIN_MAX_BINS;
    constexpr size_t bin_smem_max = sizeof(uint2) * (BIN_PPB * (1 << D) + BIN_MAX_BINS) + sizeof(uint32_t) * 2 * (BIN_THREADS / 64) * BIN_MAX_BINS;
    const size_t bin_smem = sizeof(uint2) * (BIN_PPB * (1 << D) + bins_cap) + sizeof(uint32_t) * 2 * (BIN_THREADS / 64) * bins_cap;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_backward_accumulate<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)acc_smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_backward_accumulate<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)acc_smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_backward_bin<D, AMERGE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bin_smem_max) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_backward_bin<D, AMERGE, BIN_MAX_BINS / 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bin_smem_max) != hipSuccess) {
            set_error("grid_encode_backward: hipFuncSetAttribute(LDS size) failed");
            return NGP_ERR_LAUNCH;
        }
        configured = true;
    }
    uint32_t* descriptors = reinterpret_cast<uint32_t*>(workspace);
    uint2* records = reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(workspace) + p.desc_bytes());
    AtomicPart ap;
    ap.levels = p.atomic_levels;
    ap.points_per_block = 1024;
    ap.blocks_per_level = cdiv(B, ap.points_per_block);
    ap.n_blocks = with_atomic_levels ? ap.blocks_per_level * p.n_atomic : 0u;
    // persistent sort workgroups: as many as are resident at once (LDS and registers allow BIN_RESIDENT per CU), never more than items
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("grid_encode_backward: hipGetDeviceProperties failed");
            return NGP_ERR_LAUNCH;
        }
        n_cus = prop.multiProcessorCount;
    }
    const uint32_t n_items = p.bins.n_chunks * p.n_binned;
    // (128 counters per wave: 40 KiB of LDS per workgroup, four of them on a CU; 512: 64 KiB, two)
    const uint32_t persistent = std::min<uint32_t>(n_items, (uint32_t)n_cus * (bins_cap <= 128u ? BIN_RESIDENT : 2u));
    const uint32_t blocks = persistent + ap.n_blocks;
    BinPlan bins = p.bins;
    bins.n_levels = p.n_binned;
    if (bins_cap <= 128u)
        hipLaunchKernelGGL((k_grid_backward_bin<D, AMERGE, 2>), dim3(blocks), dim3(BIN_THREADS), bin_smem, st, (const half_t*)grad, inputs, offsets,
                           (half_t*)grad_emb, B, lv, gridtype, ac, interp, im, bins, descriptors, records, ap);
    else
        hipLaunchKernelGGL((k_grid_backward_bin<D, AMERGE, BIN_MAX_BINS / 64>), dim3(blocks), dim3(BIN_THREADS), bin_smem, st, (const half_t*)grad, inputs,
                           offsets, (half_t*)grad_emb, B, lv, gridtype, ac, interp, im, bins, descriptors, records, ap);
    int rc = check_launch("grid_encode_backward(bin)");
    if (rc) return rc;
    static_assert(ACC_THREADS == RS_PARAMS * RS_GROUPS, "the slab reduction's blocks have the accumulate's shape");
    const uint32_t slab_blocks = p.slabs.total_blocks();
    const dim3 acc_grid(std::max(p.max_bins, slab_blocks), p.n_binned + (slab_blocks ? 1u : 0u));
    if (p.adam)
        hipLaunchKernelGGL((k_grid_backward_accumulate<D, true>), acc_grid, dim3(ACC_THREADS), acc_smem, st, offsets, (half_t*)grad_emb, bins,
                           (const uint32_t*)descriptors, (const uint2*)records, p.found_inf, p.slabs, p.overwrite, p.table_adam);
    else
        hipLaunchKernelGGL((k_grid_backward_accumulate<D, false>), acc_grid, dim3(ACC_THREADS), acc_smem, st, offsets, (half_t*)grad_emb, bins,
                           (const uint32_t*)descriptors, (const uint2*)records, p.found_inf, p.slabs, p.overwrite, p.table_adam);
    p.slabs_done = slab_blocks != 0;
    return check_launch("grid_encode_backward(accumulate)");
}

template <typename T, int D, int C>
static int launch_backward(const void* grad, const float* inputs, const int32_t* offsets, void* grad_emb, uint32_t B,
                           uint32_t L, const GridLevels& lv, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                           bool ac, uint32_t interp, InputMap im, const BackwardPlan& plan, void* workspace, hipStream_t st) {
    int rc = NGP_OK;
    constexpr bool can_bin = sizeof(T) == 2 && C == 2 && (D == 2 || D == 3);
    // the atomic levels ride in the launch of the record sort when there is one (and the merge variant is the default one)
    const int variant = grid_backward_variant();
    const bool mixed = can_bin && plan.n_binned > 0 && (variant == 0 || variant == 3) && !NGP_GRID_BWD_SEPARATE;
    if (plan.n_atomic && !mixed) {
        // one block covers `ppb` consecutive points of one level; >= ~4 blocks per CU over all levels fills the chip
        uint32_t ppb = 2048;
        while (ppb > 128 && (uint64_t)cdiv(B, ppb) * plan.n_atomic < 2048) ppb >>= 1;
        dim3 grid(cdiv(B, ppb), plan.n_atomic, 1);
        if constexpr (BwdLanes<T, C>::LPP == 2) {  // two lanes per sample: the run merge is pure DPP (default: over the whole wave)
            if (variant == 2 || variant == 0) {
                if (variant == 2)
                    hipLaunchKernelGGL((k_grid_backward<T, D, C, 2>), grid, dim3(BWD_THREADS), 0, st, (const T*)grad, inputs, offsets,
                                       (T*)grad_emb, B, plan.atomic_levels, lv, gridtype, ac, interp, ppb, im);
                else
                    hipLaunchKernelGGL((k_grid_backward<T, D, C, 3>), grid, dim3(BWD_THREADS), 0, st, (const T*)grad, inputs, offsets,
                                       (T*)grad_emb, B, plan.atomic_levels, lv, gridtype, ac, interp, ppb, im);
                rc = check_launch("grid_encode_backward");
                if (rc) return rc;
                goto atomic_done;
            }
        }
        if (variant == 1)
            hipLaunchKernelGGL((k_grid_backward<T, D, C, 0>), grid, dim3(BWD_THREADS), 0, st, (const T*)grad, inputs, offsets,
                               (T*)grad_emb, B, plan.atomic_levels, lv, gridtype, ac, interp, ppb, im);
        else
            hipLaunchKernelGGL((k_grid_backward<T, D, C, 1>), grid, dim3(BWD_THREADS), 0, st, (const T*)grad, inputs, offsets,
                               (T*)grad_emb, B, plan.atomic_levels, lv, gridtype, ac, interp, ppb, im);
        rc = check_launch("grid_encode_backward");
        if (rc) return rc;
    }
atomic_done:
    if (plan.n_binned) {
        if constexpr (can_bin) {
            rc = variant == 3 ? launch_backward_bins<D, 1>(grad, inputs, offsets, grad_emb, B, lv, gridtype, ac, interp, im, plan, workspace,
                                                           mixed && plan.n_atomic > 0, st)
                              : launch_backward_bins<D, 3>(grad, inputs, offsets, grad_emb, B, lv, gridtype, ac, interp, im, plan, workspace,
                                                           mixed && plan.n_atomic > 0, st);
            if (rc) return rc;
        }
    }
    if (dy_dx && grad_inputs) {
        hipLaunchKernelGGL((k_grid_input_backward<T>), dim3(cdiv(B * D, 256)), dim3(256), 0, st, (const T*)grad,
                           (const T*)dy_dx, (T*)grad_inputs, B, L, (uint32_t)D, (uint32_t)C);
        rc = check_launch("grid_encode_backward(input)");
    }
    return rc;
}

// the carried reductions on their own (a call without an accumulate launch)
__global__ __launch_bounds__(RS_PARAMS * RS_GROUPS) void k_carried_reductions(SlabSets sets, float* __restrict__ found_inf) {
    __shared__ float part[RS_GROUPS][RS_PARAMS];
    carried_block(sets, blockIdx.x, part, found_inf);
}

// found_inf for gradient entries written by atomics (their results are never observed by the writer): one sweep over the table
template <typename T>
__global__ __launch_bounds__(256) void k_flag_nonfinite(const T* __restrict__ g, uint64_t n, float* __restrict__ found_inf) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) bad = bad || !__builtin_isfinite((float)g[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;
}

template <typename T, int D, int C>
static int launch_tv(const void* inputs, const void* emb, void* grad, const int32_t* offsets, float weight, uint32_t B,
                     uint32_t L, const GridLevels& lv, uint32_t gridtype, bool ac, hipStream_t st) {
    dim3 grid(fwd_blocks(B), L, 1);
    hipLaunchKernelGGL((k_grad_tv<T, D, C>), grid, dim3(FWD_THREADS), 0, st, (const T*)inputs, (const T*)emb, (T*)grad,
                       offsets, weight, B, lv, gridtype, ac);
    return check_launch("grad_total_variation");
}

#define NGP_DISPATCH_DC(FN, T, ...)                                                               \
    switch (D * 16 + C) {                                                                         \
        case 2 * 16 + 1: return FN<T, 2, 1>(__VA_ARGS__);                                          \
        case 2 * 16 + 2: return FN<T, 2, 2>(__VA_ARGS__);                                          \
        case 2 * 16 + 4: return FN<T, 2, 4>(__VA_ARGS__);                                          \
        case 2 * 16 + 8: return FN<T, 2, 8>(__VA_ARGS__);                                          \
        case 3 * 16 + 1: return FN<T, 3, 1>(__VA_ARGS__);                                          \
        case 3 * 16 + 2: return FN<T, 3, 2>(__VA_ARGS__);                                          \
        case 3 * 16 + 4: return FN<T, 3, 4>(__VA_ARGS__);                                          \
        case 3 * 16 + 8: return FN<T, 3, 8>(__VA_ARGS__);                                          \
        case 4 * 16 + 1: return FN<T, 4, 1>(__VA_ARGS__);                                          \
        case 4 * 16 + 2: return FN<T, 4, 2>(__VA_ARGS__);                                          \
        case 4 * 16 + 4: return FN<T, 4, 4>(__VA_ARGS__);                                          \
        case 4 * 16 + 8: return FN<T, 4, 8>(__VA_ARGS__);                                          \
        case 5 * 16 + 1: return FN<T, 5, 1>(__VA_ARGS__);                                          \
        case 5 * 16 + 2: return FN<T, 5, 2>(__VA_ARGS__);                                          \
        case 5 * 16 + 4: return FN<T, 5, 4>(__VA_ARGS__);                                          \
        case 5 * 16 + 8: return FN<T, 5, 8>(__VA_ARGS__);                                          \
        default: break;                                                                            \
    }

static int check_grid_args(const char* fn, uint32_t B, uint32_t D, uint32_t C, uint32_t L, int dtype) {
    (void)B;
    // the reference throws std::runtime_error{"GridEncoding: C must be 1, 2, 4, or 8."} for both (gridencoder.cu:381,398)
    NGP_REQUIRE(D >= 2 && D <= 5, NGP_ERR_INVALID, "%s: GridEncoding: input dim D must be 2, 3, 4 or 5 (got %u)", fn, D);
    NGP_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8, NGP_ERR_INVALID, "%s: GridEncoding: C must be 1, 2, 4, or 8. (got %u)", fn, C);
    NGP_REQUIRE(L >= 1 && L <= NGP_MAX_LEVELS, NGP_ERR_INVALID, "%s: number of levels must be in [1, %d] (got %u)", fn, NGP_MAX_LEVELS, L);
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_ERR_INVALID, "%s: embeddings must be float32 or float16", fn);
    return NGP_OK;
}

}  // namespace ngp

using namespace ngp;

#ifdef NGP_BIN_PHASE_PROBE
extern "C" int ngp_debug_bin_probe(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ngp::g_bin_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ngp::g_bin_probe), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int ngp_grid_level_table(uint32_t L, float S, uint32_t H, float* scale_out, uint32_t* resolution_out) {
    NGP_REQUIRE(scale_out && resolution_out, NGP_ERR_INVALID, "ngp_grid_level_table: NULL output");
    for (uint32_t l = 0; l < L; l++) {
        const float a = (float)l * S;
        const float e = (float)exp2((double)a);
        const float sc = fmaf(e, (float)H, -1.0f);
        scale_out[l] = sc;
        resolution_out[l] = (uint32_t)ceilf(sc) + 1u;
    }
    return NGP_OK;
}

static InputMap make_input_map(float bound) {
    InputMap im{0.0f, 0.0f};
    if (bound > 0.0f) {
        im.shift = bound;
        im.scale = 1.0f / (float)(2.0 * (double)bound);
    }
    return im;
}

// The forward's per-XCD work lists as the launch would build them (host computation, no device work): 8 x 8 segments
// (level, first tile, cumulative slot end), level 0xffff = unused.  Returns the slots of the longest list.  For tests of the scheduler's
// invariant: every (level, tile) of the call appears in exactly one segment.
extern "C" uint32_t ngp_grid_forward_work_lists(uint32_t L, uint32_t tiles, const float* level_cost_host, uint16_t* level_out, uint32_t* tile0_out,
                                                uint32_t* end_out) {
    if (!level_out || !tile0_out || !end_out || L < 1 || L > NGP_MAX_LEVELS) return 0;
    FwdSchedule sc;
    const uint32_t max_slots = build_forward_schedule(sc, L, tiles, level_cost_host);
    for (int x = 0; x < 8; x++)
        for (int sg = 0; sg < FWD_MAX_SEG; sg++) {
            level_out[x * FWD_MAX_SEG + sg] = sc.level[x][sg];
            tile0_out[x * FWD_MAX_SEG + sg] = sc.tile0[x][sg];
            end_out[x * FWD_MAX_SEG + sg] = sc.end[x][sg];
        }
    return max_slots;
}

static int grid_encode_forward_impl(const float* inputs, const void* embeddings, TableSel sel, const int32_t* offsets, void* outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                    uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                    const float* level_cost_host, ngp_stream_t stream) {
    int rc = check_grid_args("grid_encode_forward", B, D, C, L, dtype);
    NGP_REQUIRE(!(bound > 0.0f && dy_dx), NGP_ERR_INVALID, "grid_encode_forward: the fused input mapping does not provide dy_dx");
    const InputMap im = make_input_map(bound);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && offsets && outputs, NGP_ERR_INVALID, "grid_encode_forward: NULL tensor");
    GridLevels lv;
    fill_levels(lv, L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    const float* level_cost = level_cost_host;
    for (uint32_t l = 0; level_cost && l < L; l++)
        NGP_REQUIRE(level_cost[l] > 0.0f && level_cost[l] < 1e6f, NGP_ERR_INVALID, "grid_encode_forward: level_cost[%u] must be positive and finite", l);
    if (dtype == NGP_F16) {
        NGP_DISPATCH_DC(launch_forward, half_t, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, im, level_cost, sel, st)
    } else {
        NGP_DISPATCH_DC(launch_forward, float, inputs, embeddings, offsets, outputs, B, L, lv, dy_dx, gridtype, ac, interp, im, level_cost, sel, st)
    }
    set_error("grid_encode_forward: unsupported (D=%u, C=%u)", D, C);
    return NGP_ERR_INVALID;
}

extern "C" int ngp_grid_encode_forward_sched(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                             uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                             const float* level_cost_host, ngp_stream_t stream) {
    return grid_encode_forward_impl(inputs, embeddings, TableSel{nullptr, nullptr, nullptr}, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                                    interp, dtype, bound, level_cost_host, stream);
}

extern "C" int ngp_grid_encode_forward_sel(const float* inputs, const void* embeddings, const void* embeddings_alt, const float* parity,
                                           const uint32_t* rows_dev, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D, uint32_t C,
                                           uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                           float bound, const float* level_cost_host, ngp_stream_t stream) {
    NGP_REQUIRE(dtype == NGP_F16 || !(embeddings_alt && parity), NGP_ERR_INVALID, "grid_encode_forward_sel: the double-buffered table is fp16");
    TableSel sel = embeddings_alt && parity ? TableSel{reinterpret_cast<const _Float16*>(embeddings_alt), parity, nullptr} : TableSel{nullptr, nullptr, nullptr};
    sel.rows = rows_dev;
    return grid_encode_forward_impl(inputs, embeddings, sel, offsets, outputs, B, D, C, L, S, H, nullptr, gridtype, align_corners, interp, dtype,
                                    bound, level_cost_host, stream);
}

#ifdef NGP_DEBUG_BOUNDS
// self-test of the range traps (debug library only, tests/test_gpu_debug_bounds.py): a forward launch whose work list names a tile
// BEHIND the last point -- NGP_BOUNDS in k_grid_forward_pair must abort it
extern "C" int ngp_debug_forward_bad_tile(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                                          ngp_stream_t stream) {
    GridLevels lv;
    fill_levels(lv, 2, 1.0f, 16);
    FwdSchedule sc;
    for (int x = 0; x < 8; x++)
        for (int sg = 0; sg < FWD_MAX_SEG; sg++) {
            sc.level[x][sg] = 0xffffu;
            sc.tile0[x][sg] = 0u;
            sc.end[x][sg] = x == 0 ? 1u : 0u;
        }
    sc.level[0][0] = 0;
    sc.tile0[0][0] = cdiv(B, 1024u);  // one past the last tile
    hipLaunchKernelGGL((k_grid_forward_pair<half_t, 3, 2>), dim3(8u), dim3(FWD_THREADS), 0, as_stream(stream), inputs, (const half_t*)embeddings,
                       offsets, (half_t*)outputs, B, 2u, lv, 0u, false, 0u, sc, 1024u, InputMap{0.0f, 0.0f}, TableSel{nullptr, nullptr, nullptr});
    return check_launch("debug_forward_bad_tile");
}
#endif

extern "C" int ngp_grid_encode_forward_ex(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                          uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                          uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                          ngp_stream_t stream) {
    return ngp_grid_encode_forward_sched(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype,
                                         bound, nullptr, stream);
}

extern "C" int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                       uint32_t gridtype, int align_corners, uint32_t interp, int dtype, ngp_stream_t stream) {
    return ngp_grid_encode_forward_ex(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                                      dtype, 0.0f, stream);
}

extern "C" size_t ngp_grid_backward_workspace_bytes(const int32_t* offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                                    uint32_t H, uint32_t gridtype, int align_corners, int dtype) {
    if (!offsets_host || L < 1 || L > NGP_MAX_LEVELS || D < 2 || D > 5) return 0;
    GridLevels lv;
    fill_levels(lv, L, S, H);
    BackwardPlan plan;
    plan_backward(plan, offsets_host, lv, B, D, C, L, dtype, gridtype, align_corners != 0, true);
    return plan.workspace_bytes();
}

extern "C" uint32_t ngp_grid_table_adam_prefix(const int32_t* offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                               uint32_t gridtype, int align_corners, int dtype) {
    if (!offsets_host || L < 1 || L > NGP_MAX_LEVELS || D < 2 || D > 5) return 0xffffffffu;
    GridLevels lv;
    fill_levels(lv, L, S, H);
    BackwardPlan plan;
    plan_backward(plan, offsets_host, lv, B, D, C, L, dtype, gridtype, align_corners != 0, true);
    if (plan.n_atomic != 0 || plan.n_binned != L || dtype != NGP_F16 || C != 2) return 0xffffffffu;
    uint32_t k = 0;
    while (k < L && plan.bins.interleaved(k)) k++;
    for (uint32_t j = k; j < L; j++)
        if (plan.bins.interleaved(j)) return 0xffffffffu;
    return (uint32_t)offsets_host[k];   // (binned levels are listed in level order when every level is binned: index == level)
}

template <typename T>
static int dispatch_backward(uint32_t D, uint32_t C, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                             uint32_t L, const GridLevels& lv, const void* dy_dx, void* grad_inputs, uint32_t gridtype, bool ac, uint32_t interp,
                             InputMap im, const BackwardPlan& plan, void* workspace, hipStream_t st) {
    NGP_DISPATCH_DC(launch_backward, T, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp, im, plan, workspace, st)
    set_error("grid_encode_backward: unsupported (D=%u, C=%u)", D, C);
    return NGP_ERR_INVALID;
}

extern "C" int ngp_grid_encode_backward_checked(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                                uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                                uint32_t interp, int dtype, float bound, const int32_t* offsets_host, void* workspace,
                                                size_t workspace_bytes, float* found_inf, ngp_stream_t stream) {
    return ngp_grid_encode_backward_checked_slabs(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                                                  align_corners, interp, dtype, bound, offsets_host, workspace, workspace_bytes, found_inf, nullptr,
                                                  stream);
}

extern "C" int ngp_grid_encode_backward_checked_slabs(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                                      void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                                      uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                                      uint32_t interp, int dtype, float bound, const int32_t* offsets_host, void* workspace,
                                                      size_t workspace_bytes, float* found_inf, const ngp_slab_sets_t* slab_sets,
                                                      ngp_stream_t stream) {
    (void)embeddings;
    // the carried reductions ride in the accumulate launch when this call has one; otherwise (few samples, no workspace, an empty batch) they
    // are launched on their own -- the caller gets the reduced gradients and the loss either way
    SlabSets carried = {};
    if (slab_sets) {
        // same rule as ngp_ffmlp_reduce_slabs_pair: a set without slabs (gradients stored directly) still gets blocks when found_inf asks for the sweep
        const uint32_t ba = (slab_sets->n_slabs_a || found_inf) && slab_sets->n_params_a ? cdiv(slab_sets->n_params_a, (uint32_t)RS_PARAMS) : 0u;
        const uint32_t bb = (slab_sets->n_slabs_b || found_inf) && slab_sets->n_params_b ? cdiv(slab_sets->n_params_b, (uint32_t)RS_PARAMS) : 0u;
        NGP_REQUIRE((!ba || ((slab_sets->slabs_a || !slab_sets->n_slabs_a) && slab_sets->grad_weights_a)) &&
                        (!bb || ((slab_sets->slabs_b || !slab_sets->n_slabs_b) && slab_sets->grad_weights_b)),
                    NGP_ERR_INVALID, "grid_encode_backward: NULL tensor in the slab sets");
        NGP_REQUIRE(!slab_sets->loss || (slab_sets->ray_err && slab_sets->n_rays > 0), NGP_ERR_INVALID, "grid_encode_backward: loss sum without per-ray errors");
        carried.slabs[0] = (const float*)slab_sets->slabs_a; carried.n_slabs[0] = slab_sets->n_slabs_a; carried.n_params[0] = slab_sets->n_params_a;
        carried.grad_weights[0] = (half_t*)slab_sets->grad_weights_a; carried.blocks[0] = ba;
        carried.slabs[1] = (const float*)slab_sets->slabs_b; carried.n_slabs[1] = slab_sets->n_slabs_b; carried.n_params[1] = slab_sets->n_params_b;
        carried.grad_weights[1] = (half_t*)slab_sets->grad_weights_b; carried.blocks[1] = bb;
        carried.ray_err = slab_sets->ray_err; carried.n_rays = slab_sets->n_rays; carried.loss = slab_sets->loss;
    }
    auto reduce_alone = [&]() -> int {
        if (carried.total_blocks() == 0u) return NGP_OK;
        hipLaunchKernelGGL(k_carried_reductions, dim3(carried.total_blocks()), dim3(RS_PARAMS * RS_GROUPS), 0, as_stream(stream), carried, found_inf);
        return check_launch("grid_encode_backward(carried reductions)");
    };
    NGP_REQUIRE(!(bound > 0.0f && dy_dx), NGP_ERR_INVALID, "grid_encode_backward: the fused input mapping does not provide grad_inputs");
    NGP_REQUIRE(!found_inf || offsets_host, NGP_ERR_INVALID, "grid_encode_backward: found_inf needs the host copy of the offsets");
    const InputMap im = make_input_map(bound);
    int rc = check_grid_args("grid_encode_backward", B, D, C, L, dtype);
    if (rc) return rc;
    if (B == 0) {
        NGP_REQUIRE(!(slab_sets && slab_sets->table_adam), NGP_ERR_INVALID, "grid_encode_backward: table_adam with an empty batch");
        if (slab_sets && slab_sets->overwrite_table) {
            NGP_REQUIRE(offsets_host && grad_embeddings, NGP_ERR_INVALID, "grid_encode_backward: overwrite_table needs the table and the host copy of the offsets");
            hipError_t e = hipMemsetAsync(grad_embeddings, 0, (size_t)offsets_host[L] * C * (dtype == NGP_F16 ? 2 : 4), as_stream(stream));
            NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "grid_encode_backward: hipMemsetAsync failed: %s", hipGetErrorString(e));
        }
        return reduce_alone();
    }
    const ngp_table_adam_t* tadam = slab_sets ? slab_sets->table_adam : nullptr;
    NGP_REQUIRE(grad && inputs && offsets && (grad_embeddings || tadam), NGP_ERR_INVALID, "grid_encode_backward: NULL tensor");
    GridLevels lv;
    fill_levels(lv, L, S, H);
    BackwardPlan plan;
    plan_backward(plan, offsets_host, lv, B, D, C, L, dtype, gridtype, align_corners != 0, workspace != nullptr);
    plan.found_inf = found_inf;
    plan.slabs = carried;
    if (tadam) {
        // the table's Adam sweep rides in the accumulate's flush: only when EVERY entry of the table comes out of a flush (all levels on the
        // record-sort path, overwrite mode) -- refused otherwise, before anything is launched (the caller falls back to ngp_optim_adam_step_ex)
        NGP_REQUIRE(slab_sets->overwrite_table && plan.n_atomic == 0 && plan.n_binned == L && dtype == NGP_F16 && C == 2, NGP_ERR_INVALID,
                    "grid_encode_backward: table_adam needs overwrite_table and every level on the record-sort path (fp16, C = 2, >= %u samples, "
                    "a workspace and the host offsets)", BIN_MIN_SAMPLES);
        NGP_REQUIRE(tadam->state, NGP_ERR_INVALID, "grid_encode_backward: table_adam without the optimizer's state");
        {   // the dense levels' round-robin bins leave their entries to ngp_optim_adam_small_commit: they must form a prefix of the table
            // (ngp_grid_table_adam_prefix tells the caller how long it is), and their gradient has to be stored for it
            uint32_t k = 0;
            while (k < plan.n_binned && plan.bins.interleaved(k)) k++;
            for (uint32_t j = k; j < plan.n_binned; j++)
                NGP_REQUIRE(!plan.bins.interleaved(j), NGP_ERR_INVALID, "grid_encode_backward: table_adam: a dense level behind a hashed one (level %u)",
                            plan.bins.level(j));
            NGP_REQUIRE(k == 0 || grad_embeddings, NGP_ERR_INVALID, "grid_encode_backward: table_adam: the dense levels' gradient needs grad_embeddings");
        }
        for (int k = 0; k < 2; k++)
            NGP_REQUIRE(tadam->param[k] && tadam->exp_avg[k] && tadam->exp_avg_sq[k] && tadam->param_fp16[k], NGP_ERR_INVALID,
                        "grid_encode_backward: table_adam: NULL buffer in set %d", k);
        plan.adam = true;
        for (int k = 0; k < 2; k++) {
            plan.table_adam.p[k] = tadam->param[k]; plan.table_adam.m[k] = tadam->exp_avg[k]; plan.table_adam.v[k] = tadam->exp_avg_sq[k];
            plan.table_adam.p16[k] = reinterpret_cast<_Float16*>(tadam->param_fp16[k]);
        }
        plan.table_adam.state = tadam->state;
        plan.table_adam.lr = tadam->lr; plan.table_adam.beta1 = tadam->beta1; plan.table_adam.beta2 = tadam->beta2; plan.table_adam.eps = tadam->eps;
    }
    if (slab_sets && slab_sets->overwrite_table) {
        // every entry comes out of the accumulate only when every level is sorted; otherwise: zero the table, then add as usual
        if (plan.n_atomic == 0 && plan.n_binned == L && dtype == NGP_F16) {
            plan.overwrite = true;
        } else {
            NGP_REQUIRE(offsets_host, NGP_ERR_INVALID, "grid_encode_backward: overwrite_table needs the host copy of the offsets");
            const size_t bytes = (size_t)offsets_host[L] * C * (dtype == NGP_F16 ? 2 : 4);
            hipError_t e = hipMemsetAsync(grad_embeddings, 0, bytes, as_stream(stream));
            NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "grid_encode_backward: hipMemsetAsync failed: %s", hipGetErrorString(e));
        }
    }
    NGP_REQUIRE(plan.workspace_bytes() <= workspace_bytes, NGP_ERR_INVALID,
                "grid_encode_backward: workspace of %zu bytes, ngp_grid_backward_workspace_bytes() asks for %zu", workspace_bytes,
                plan.workspace_bytes());
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    rc = dtype == NGP_F16 ? dispatch_backward<half_t>(D, C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac,
                                                      interp, im, plan, workspace, st)
                          : dispatch_backward<float>(D, C, grad, inputs, offsets, grad_embeddings, B, L, lv, dy_dx, grad_inputs, gridtype, ac, interp,
                                                     im, plan, workspace, st);
    if (!rc && slab_sets && !plan.slabs_done) rc = reduce_alone();
    if (rc || !found_inf || plan.n_atomic == 0) return rc;
    // some levels went through atomics: their sums are swept here (the record-sort path flags its own)
    const uint64_t n = (uint64_t)offsets_host[L] * C;
    const uint32_t blocks = (uint32_t)(cdiv64(n, 256 * 8) > 2048 ? 2048 : cdiv64(n, 256 * 8));
    if (dtype == NGP_F16)
        hipLaunchKernelGGL(k_flag_nonfinite<half_t>, dim3(blocks ? blocks : 1), dim3(256), 0, st, (const half_t*)grad_embeddings, n, found_inf);
    else
        hipLaunchKernelGGL(k_flag_nonfinite<float>, dim3(blocks ? blocks : 1), dim3(256), 0, st, (const float*)grad_embeddings, n, found_inf);
    return check_launch("grid_encode_backward(non-finite sweep)");
}

extern "C" int ngp_grid_encode_backward_ws(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                           void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                           uint32_t interp, int dtype, float bound, const int32_t* offsets_host, void* workspace,
                                           size_t workspace_bytes, ngp_stream_t stream) {
    return ngp_grid_encode_backward_checked(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                                            align_corners, interp, dtype, bound, offsets_host, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int ngp_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                           void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                           uint32_t interp, int dtype, float bound, ngp_stream_t stream) {
    return ngp_grid_encode_backward_ws(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                                       align_corners, interp, dtype, bound, nullptr, nullptr, 0, stream);
}

extern "C" int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                        void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                        uint32_t interp, int dtype, ngp_stream_t stream) {
    return ngp_grid_encode_backward_ex(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                                       align_corners, interp, dtype, 0.0f, stream);
}

extern "C" int ngp_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                                        float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        uint32_t gridtype, int align_corners, int dtype, ngp_stream_t stream) {
    int rc = check_grid_args("grad_total_variation", B, D, C, L, dtype);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && grad && offsets, NGP_ERR_INVALID, "grad_total_variation: NULL tensor");
    GridLevels lv;
    fill_levels(lv, L, S, H);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (dtype == NGP_F16) {
        NGP_DISPATCH_DC(launch_tv, half_t, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st)
    } else {
        NGP_DISPATCH_DC(launch_tv, float, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, ac, st)
    }
    set_error("grad_total_variation: unsupported (D=%u, C=%u)", D, C);
    return NGP_ERR_INVALID;
}

extern "C" int ngp_grid_corner_indices(const float* inputs, const int32_t* offsets, uint32_t* indices, uint32_t B, uint32_t D,
                                       uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                       ngp_stream_t stream) {
    int rc = check_grid_args("grid_corner_indices", B, D, 1, L, NGP_F32);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && offsets && indices, NGP_ERR_INVALID, "grid_corner_indices: NULL tensor");
    GridLevels lv;
    fill_levels(lv, L, S, H);
    hipStream_t st = as_stream(stream);
    dim3 grid(fwd_blocks(B), L, 1);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: hipLaunchKernelGGL((k_grid_corner_indices<2>), grid, dim3(FWD_THREADS), 0, st, inputs, offsets, indices, B, lv, gridtype, ac); break;
        case 3: hipLaunchKernelGGL((k_grid_corner_indices<3>), grid, dim3(FWD_THREADS), 0, st, inputs, offsets, indices, B, lv, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((k_grid_corner_indices<4>), grid, dim3(FWD_THREADS), 0, st, inputs, offsets, indices, B, lv, gridtype, ac); break;
        default: hipLaunchKernelGGL((k_grid_corner_indices<5>), grid, dim3(FWD_THREADS), 0, st, inputs, offsets, indices, B, lv, gridtype, ac); break;
    }
    return check_launch("grid_corner_indices");
}
