// Density-grid ray marching and volume compositing for gfx950 (MI355X).
//
// Behaviour restated from raymarching/src/raymarching.cu of the reference (line ranges per kernel below).
//
// MI355X design (see DESIGN.md "raymarching"):
//   * sample slots of march_rays_train are handed out by a deterministic two-kernel prefix sum in ray
//     order (count -> block sums -> scan -> write) instead of the reference's two global atomics per
//     ray: the layout of xyzs/dirs/deltas/rays is reproducible and equals what a sequential run of the
//     reference produces, and the only cross-workgroup traffic is one 4-byte block sum per 256 rays;
//   * training compositing runs one 64-lane wavefront per ray: samples are read as coalesced 64-wide
//     rows, transmittance is a wave-level prefix product (6 __shfl_up steps), colour/depth/weight are
//     wave reductions; the T < T_thresh early stop becomes a per-lane predicate on the exclusive
//     prefix product, so a ray stops after the 64-sample row in which it saturates;
//   * every floating-point operation that decides an integer (cell index, occupancy bit, step count)
//     is written with explicit fmaf so that it rounds exactly like oracle/ngp_oracle.c
//     (both are compiled with -ffp-contract=off; division is IEEE).
#include "common.h"
#include <math.h>
#include <float.h>

namespace ngp {

constexpr int RM_THREADS = 256;

// ---------------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------------
// raymarching.cu:92-145
__device__ __forceinline__ void near_far_from_aabb(float ox, float oy, float oz, float dx, float dy, float dz, const float* __restrict__ aabb,
                                                   float min_near, float& near_out, float& far_out) {
    const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, t;
    near_out = far_out = FLT_MAX;
    if (near > far) { t = near; near = far; far = t; }
    float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
    if (ny > fy) { t = ny; ny = fy; fy = t; }
    if (near > fy || ny > far) return;
    if (ny > near) near = ny;
    if (fy < far) far = fy;
    float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
    if (nz > fz) { t = nz; nz = fz; fz = t; }
    if (near > fz || nz > far) return;
    if (nz > near) near = nz;
    if (fz < far) far = fz;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

__global__ void k_near_far_from_aabb(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                     const float* __restrict__ aabb, uint32_t N, float min_near, float* __restrict__ nears,
                                     float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near, far;
    near_far_from_aabb(rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], aabb, min_near,
                       near, far);
    nears[n] = near;
    fars[n] = far;
}

// raymarching.cu:163-198
__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius, uint32_t N,
                               float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float Bh = ox * dx + oy * dy + oz * dz;
    const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    const float RPI = 0.3183098861837907f;
    coords[n * 2] = 2.0f * theta * RPI - 1.0f;
    coords[n * 2 + 1] = phi * RPI;
}

// raymarching.cu:56-81
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D_1(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}
__global__ void k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3D_1((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];
    coords[n * 3 + 0] = (int32_t)compact_bits((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)compact_bits((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)compact_bits((uint32_t)(ind >> 2));
}

// raymarching.cu:268-289.  One lane packs one byte from two 16-byte loads (the grid is a pure stream:
// 4.125 B per cell).
__global__ void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, const float* __restrict__ thresh_cap,
                           uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (thresh_cap) thresh = fminf(thresh, thresh_cap[0]);  // min(mean_density, density_thresh) with the mean still on the device
    const float4_t a = *reinterpret_cast<const float4_t*>(grid + (size_t)n * 8);
    const float4_t b = *reinterpret_cast<const float4_t*>(grid + (size_t)n * 8 + 4);
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ---------------------------------------------------------------------------------------------
// occupancy-grid refresh, apply half (nerf/renderer.py:515-529): the reference builds a full `tmp_grid` of -1, index-assigns the fresh
// densities, masks, takes the EMA-max, clamps, means and packs -- about fifteen PyTorch launches over the whole grid.  Here:
//   k_density_scatter    : scratch[cell] = density_scale * sigma             (`tmp_grid[cas, indices] = sigmas`; duplicates: any one wins,
//                                                                             as with index_put_)
//   k_density_apply_mean : one streaming pass over all cells: `grid = max(grid * decay, scratch)` where both are >= 0, the written scratch
//                          entries go back to -1 (the state the buffer is kept in between calls), mean of max(grid, 0) in double, fixed order
//   k_packbits           : against min(density_thresh, mean) with the mean still on the device
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RM_THREADS) void k_density_scatter(const float* __restrict__ sigmas, const int64_t* __restrict__ cells, uint32_t n,
                                                                  float scale, float* __restrict__ scratch, uint32_t n_cells) {
    const uint32_t i = blockIdx.x * RM_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = (uint64_t)cells[i];
    if (c < n_cells) scratch[c] = sigmas[i] * scale;
}

// one pass over ALL cells (a stream of 2 x 4 B per cell; a per-sample pass would need a fabric atomic per sample to find the first
// visitor of a cell): cells whose scratch entry was written take the EMA-max and hand the entry back as -1; the same pass sums
// max(grid, 0) in double -- block partials, the last block (ticket) adds them in a fixed order and writes the mean
constexpr int DM_THREADS = 256, DM_PER_THREAD = 32;  // 8192 cells per block
__global__ __launch_bounds__(DM_THREADS) void k_density_apply_mean(float* __restrict__ grid, float* __restrict__ scratch, uint32_t n_cells, float decay,
                                                                     double* __restrict__ partials, uint32_t* __restrict__ ticket,
                                                                     float* __restrict__ mean_out) {
    const uint32_t base = blockIdx.x * (DM_THREADS * DM_PER_THREAD);
    double acc = 0.0;
    auto one = [&](float g, float f, bool& touched) {
        touched = !(f == -1.0f);                                  // written by the scatter (a NaN density counts as written: it is reset, not applied)
        if (f >= 0.0f && g >= 0.0f) g = fmaxf(g * decay, f);      // renderer.py:517-518: both sides valid
        acc += (double)fmaxf(g, 0.0f);
        return g;
    };
#pragma unroll 2
    for (int k = 0; k < DM_PER_THREAD / 4; k++) {
        const uint32_t i = base + (uint32_t)(k * DM_THREADS + threadIdx.x) * 4u;
        if (i + 3u < n_cells) {
            float4_t g = *reinterpret_cast<const float4_t*>(grid + i);
            const float4_t f = *reinterpret_cast<const float4_t*>(scratch + i);
            bool t0, t1, t2, t3;
            g.x = one(g.x, f.x, t0); g.y = one(g.y, f.y, t1); g.z = one(g.z, f.z, t2); g.w = one(g.w, f.w, t3);
            if (t0 || t1 || t2 || t3) {
                *reinterpret_cast<float4_t*>(grid + i) = g;
                *reinterpret_cast<float4_t*>(scratch + i) = float4_t{-1.0f, -1.0f, -1.0f, -1.0f};
            }
        } else {
            for (uint32_t j = i; j < n_cells; j++) {
                bool t;
                const float g = one(grid[j], scratch[j], t);
                if (t) { grid[j] = g; scratch[j] = -1.0f; }
            }
        }
    }
    __shared__ double part[DM_THREADS / 64];
    __shared__ bool last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0;
#pragma unroll
        for (int w = 0; w < DM_THREADS / 64; w++) b += part[w];
        // write-through store + own-store wait, then the ticket: the last block reads every partial from memory (see k_composite_train_loss_bwd)
        __hip_atomic_store(&partials[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!last) return;
    double t = 0.0;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += DM_THREADS) t += __hip_atomic_load(&partials[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < DM_THREADS / 64; w++) sum += part[w];
        mean_out[0] = (float)(sum / (double)n_cells);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// the marcher (shared by training and inference)          raymarching.cu:312-480, 701-805
// ---------------------------------------------------------------------------------------------
struct Ray {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};
struct MarchParams {  // wave-uniform
    float bound, rbound, dt_gamma, dt_min, dt_max, Hf, rH, H3f, Hm1, Cm1;
    uint32_t H3;
    const uint8_t* grid;
};

__device__ __forceinline__ MarchParams make_params(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                   const uint8_t* grid) {
    MarchParams p;
    const float SQRT3 = 1.7320508075688772f;
    p.bound = bound;
    p.dt_gamma = dt_gamma;
    p.Hf = (float)H;
    p.rH = 1.0f / p.Hf;
    p.H3f = (float)(H * H * H);
    p.H3 = H * H * H;
    p.rbound = 1.0f / bound;
    p.Hm1 = (float)(H - 1);
    p.Cm1 = (float)C - 1.0f;
    p.dt_min = 2.0f * SQRT3 / (float)max_steps;
    p.dt_max = 2.0f * SQRT3 * (float)(1u << (C - 1)) / p.Hf;
    p.grid = grid;
    return p;
}

__device__ __forceinline__ Ray load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t i) {
    Ray r;
    r.ox = rays_o[i * 3]; r.oy = rays_o[i * 3 + 1]; r.oz = rays_o[i * 3 + 2];
    r.dx = rays_d[i * 3]; r.dy = rays_d[i * 3 + 1]; r.dz = rays_d[i * 3 + 2];
    r.rdx = 1.0f / r.dx; r.rdy = 1.0f / r.dy; r.rdz = 1.0f / r.dz;
    return r;
}

// clamp of a value to [lo, hi] with lo <= hi as ONE instruction (v_med3_f32).  `fminf(hi, fmaxf(lo, x))` is two, plus a v_max(x, x) per
// operand the compiler cannot prove quiet (IEEE mode) -- recomputed in every iteration of the march loop for the loop-invariant bounds too.
// Same result for every x incl. NaN (both forms return lo); NOT for lo > hi, which is why step_dt keeps the two-instruction form.
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

__device__ __forceinline__ int mip_exponent(float mx, float Cm1) {
    int e;
    (void)frexpf(mx, &e);
    return (int)clamp_med3((float)e, 0.0f, Cm1);
}

__device__ __forceinline__ float step_dt(const MarchParams& p, float t) { return clampf(t * p.dt_gamma, p.dt_min, p.dt_max); }

// Evaluate the occupancy grid at parameter t.  Returns true when the voxel is occupied; otherwise tt
// receives the parameter of the far face of the voxel (raymarching.cu:359-400).  Split in two so that a caller can keep the
// bitfield read of the NEXT batch of terms in flight while it finishes the current one.
struct ProbeGeom {
    float x, y, z, dt, mip_bound;
    int nx, ny, nz;
    uint32_t index;  // bit index in the occupancy bitfield
};

__device__ __forceinline__ ProbeGeom probe_geom(const MarchParams& p, const Ray& r, float t) {
    ProbeGeom g;
    g.x = clamp_med3(__builtin_fmaf(t, r.dx, r.ox), -p.bound, p.bound);
    g.y = clamp_med3(__builtin_fmaf(t, r.dy, r.oy), -p.bound, p.bound);
    g.z = clamp_med3(__builtin_fmaf(t, r.dz, r.oz), -p.bound, p.bound);
    g.dt = step_dt(p, t);
    int level = 0;  // a single cascade: both mip exponents clamp to 0 (min(C - 1, .) in raymarching.cu:367-369), skip the frexp work
    if (p.Cm1 > 0.0f) {
        const float mx = fmaxf(fabsf(g.x), fmaxf(fabsf(g.y), fabsf(g.z)));
        const int lp = mip_exponent(mx, p.Cm1);
        const int ld = mip_exponent((g.dt * p.Hf) * 0.5f, p.Cm1);
        level = lp > ld ? lp : ld;
    }
    // mip_bound = min(2^level, bound) and its reciprocal (raymarching.cu:371-372: `1 / mip_bound`, an IEEE division -- a dozen
    // instructions per term) without dividing: 1 / 2^level IS 2^-level, 1 / bound is loop-invariant
    const float pw = scalbnf(1.0f, level);
    const bool capped = !(pw < p.bound);
    g.mip_bound = capped ? p.bound : pw;
    const float mip_rbound = capped ? p.rbound : scalbnf(1.0f, -level);
    g.nx = (int)clamp_med3((0.5f * __builtin_fmaf(g.x, mip_rbound, 1.0f)) * p.Hf, 0.0f, p.Hm1);
    g.ny = (int)clamp_med3((0.5f * __builtin_fmaf(g.y, mip_rbound, 1.0f)) * p.Hf, 0.0f, p.Hm1);
    g.nz = (int)clamp_med3((0.5f * __builtin_fmaf(g.z, mip_rbound, 1.0f)) * p.Hf, 0.0f, p.Hm1);
    g.index = (uint32_t)level * p.H3 + morton3D_1((uint32_t)g.nx, (uint32_t)g.ny, (uint32_t)g.nz);
    return g;
}

__device__ __forceinline__ bool probe_resolve(const MarchParams& p, const Ray& r, float t, const ProbeGeom& g, uint32_t byte, float& tt) {
    const bool occ = (byte & (1u << (g.index & 7u))) != 0;
    if (!occ) {
        const float tx = __builtin_fmaf(((float)g.nx + 0.5f + 0.5f * copysignf(1.0f, r.dx)) * p.rH * 2.0f - 1.0f, g.mip_bound, -g.x) * r.rdx;
        const float ty = __builtin_fmaf(((float)g.ny + 0.5f + 0.5f * copysignf(1.0f, r.dy)) * p.rH * 2.0f - 1.0f, g.mip_bound, -g.y) * r.rdy;
        const float tz = __builtin_fmaf(((float)g.nz + 0.5f + 0.5f * copysignf(1.0f, r.dz)) * p.rH * 2.0f - 1.0f, g.mip_bound, -g.z) * r.rdz;
        tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

__device__ __forceinline__ bool probe(const MarchParams& p, const Ray& r, float t, float& x, float& y, float& z, float& dt,
                                      float& tt) {
    const ProbeGeom g = probe_geom(p, r, t);
    x = g.x; y = g.y; z = g.z; dt = g.dt;
    return probe_resolve(p, r, t, g, p.grid[g.index >> 3], tt);
}

// advance in whole steps to the far face of an empty voxel (raymarching.cu:396-399).  The extra
// `t < far` bound cannot change any output (nothing is emitted once t >= far) but keeps a degenerate
// ray (zero direction => tt = +inf) from spinning forever, which the reference would.
__device__ __forceinline__ float skip_to(const MarchParams& p, float t, float tt, float far) {
    do {
        t += step_dt(p, t);
    } while (t < tt && t < far);
    return t;
}

// The same walk for the lane-per-ray marcher, eight steps per trip: the partial sums t + dt, (t + dt) + dt, ... are the SAME sequence of
// fp32 additions (a closed form t + k dt rounds differently), the first one that is not below min(tt, far) is selected.  One step per trip
// was 6 vector + 4 scalar instructions and a taken branch; a cascade-0 voxel is ~5 steps wide, and a wave leaves the loop with its
// slowest lane.  CONST_DT: dt_gamma == 0, dt(t) is the constant clamp(0, dt_min, dt_max).
template <bool CONST_DT>
__device__ __forceinline__ float skip_to_unrolled(const MarchParams& p, float t, float tt, float far, float dt_const) {
    const float stop = fminf(tt, far);   // (t < tt && t < far) == (t < min(tt, far)); a NaN tt cannot occur: it is t + max(0, .)
    constexpr int U = 8;
    for (;;) {
        float s[U];
        float c = t;
#pragma unroll
        for (int i = 0; i < U; i++) {
            c += CONST_DT ? dt_const : step_dt(p, c);
            s[i] = c;
        }
        float pick = s[U - 1];
#pragma unroll
        for (int i = U - 2; i >= 0; i--) pick = (s[i] < stop) ? pick : s[i];
        // no progress over a whole trip (t has grown until t + dt == t: only a ray with tt = far = +inf gets here, the reference's loop would
        // spin forever): leave with `stop`, which ends the caller's `t < far` loop as well
        if (!(pick > t)) return stop;
        t = pick;
        if (!(t < stop)) return t;
    }
}

// block-wide sum of one uint32 per thread (RM_THREADS threads); result valid in every thread
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* lds4) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __syncthreads();
    if (lane == 0) lds4[wid] = s;
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < RM_THREADS / 64; w++) tot += lds4[w];
    return tot;
}

// ---------------------------------------------------------------------------------------------
// march_rays_train: one WAVEFRONT per ray                                 raymarching.cu:312-480
//
// The reference walks a ray sequentially: probe the voxel at t; occupied -> emit a sample and t += dt(t); empty
// -> compute the far face tt of the voxel and repeat t += dt(t) until t >= tt.  Every t it ever visits therefore
// belongs to ONE occupancy-independent sequence  t_0 = near + dt(near)*noise,  t_{k+1} = t_k (+) dt(t_k)  (fp32).
// A lane-per-ray marcher is a chain of ~200-1000 dependent (position -> byte load -> branch) iterations with 64
// diverging lanes; on gfx950 that is ~2 us per iteration and only 64 wavefronts for a 4096-ray batch.  Here a
// wavefront owns a ray and processes its sequence 64 terms at a time:
//   1. the 64 terms are produced by the same fp32 recurrence the reference uses (wave-uniform chain, each lane
//      keeps its own term), so every t is bit-identical to the sequential walk;
//   2. all 64 positions are probed AT ONCE (one coalesced burst of bitfield reads, no divergence); empty lanes
//      also compute tt and, by a 6-step binary search over the (monotone) terms, the index the reference's
//      "advance until t >= tt" loop would land on;
//   3. which of the probed terms the sequential walk really visits is then a pointer chase over wave-uniform
//      scalars (ballot masks + v_readlane), consuming whole runs of occupied terms with one bit scan; jumps that
//      leave the 64-term window are carried into the next window as a threshold;
//   4. (write pass) emitted terms are compacted with ballot/popcount prefix ranks and stored as coalesced rows.
// Sample slots are handed out in ray order by an exclusive scan between the two passes (count -> scan -> write):
// reproducible layout, no atomics (the reference's completion-order atomics are "parity unpinned").
// ---------------------------------------------------------------------------------------------
constexpr int MW_WAVES = 4;  // rays per workgroup
// The fused composite / loss / backward kernel finds its last workgroup with tickets.  ONE ticket word taken by every workgroup serialises
// ~1000 device-scope atomics on one address (measured: 8.5 of the kernel's 22 us on the 4096-ray batch), so there are two levels: the
// workgroups are dealt round-robin onto CT_GROUPS group tickets, 128 bytes apart, and only a group's last arrival takes the final
// ticket (workspace word 1).  The group tickets live behind the marcher's per-ray words in ITS workspace and are cleared by the marcher
// together with word 1 (every ticket also returns to 0 by itself).
constexpr uint32_t CT_GROUPS = 32, CT_GROUP_STRIDE = 32;  // tickets, words between them
constexpr uint32_t MARCH_MASK_WINDOWS = 20;   // emit masks kept per ray (64 terms each); longer rays re-probe in the write pass

// word index of group ticket 0 in the marcher's workspace: behind [0] fit_end, [1] final ticket, [2 .. 2+N) windows per ray and the
// N x MARCH_MASK_WINDOWS 64-bit emit masks, rounded up to a 128-byte boundary
__host__ __device__ inline size_t march_ws_ticket_word(uint32_t N) {
    const size_t words = (size_t)(2 + ((N + 1u) & ~1u)) + 2 * (size_t)N * MARCH_MASK_WINDOWS;
    return (words + CT_GROUP_STRIDE - 1) / CT_GROUP_STRIDE * CT_GROUP_STRIDE;
}

__device__ __forceinline__ float readlane_f(float v, uint32_t l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)l));
}

// The 64 terms t_base, t_base (+) dt, ... of one window (lane j keeps term j) and the base of the next window.
// With a constant step the fp32 recurrence has a closed form inside one binade: every term is a multiple of the binade's ulp u, so
// each step adds the SAME increment  inc = fl(t + dt) - t  (dt rounded to a multiple of u, to nearest), unless dt sits exactly half
// way between two multiples (round-to-even would alternate).  So when all 65 terms stay in the binade of t_base and there is no
// tie, term j = t_base + j * inc EXACTLY (j * inc < 2^23 u is exact, the sum is a multiple of u inside the binade) -- bit-identical to
// the 64 dependent additions, without the dependent chain.  Otherwise (binade crossing: a handful of windows per ray; dt_gamma != 0)
// the recurrence is evaluated as written.
template <bool CONST_DT>
__device__ __forceinline__ void window_terms(const MarchParams& p, float t_base, float dt_const, uint32_t lane, float& mine,
                                             float& t_next_base, float& inc_out, bool& closed_out) {
    if (CONST_DT) {
        const float t1 = t_base + dt_const;
        const float inc = t1 - t_base;                    // exact (Sterbenz)
        const float err = dt_const - inc;                 // exact: the rounding error of the first step
        const float t_end = __builtin_fmaf(64.0f, inc, t_base);
        const uint32_t e0 = __builtin_bit_cast(uint32_t, t_base) >> 23, e1 = __builtin_bit_cast(uint32_t, t_end) >> 23;
        const float half_ulp = __builtin_bit_cast(float, (e0 > 24u ? e0 - 24u : 0u) << 23);  // u / 2 of the binade (0: give up)
        const bool closed = inc > 0.0f && e0 == e1 && e0 > 24u && e0 < 255u && fabsf(err) != half_ulp;
        if (__builtin_amdgcn_readfirstlane((int)closed)) {  // wave-uniform by construction
            mine = __builtin_fmaf((float)lane, inc, t_base);
            t_next_base = t_end;
            inc_out = inc;
            closed_out = true;
            return;
        }
    }
    inc_out = 0.0f;
    closed_out = false;
    float t = t_base;
    mine = t_base;
#pragma unroll
    for (uint32_t j = 1; j < 64; j++) {
        t += CONST_DT ? dt_const : step_dt(p, t);
        mine = (lane == j) ? t : mine;
    }
    t_next_base = t + (CONST_DT ? dt_const : step_dt(p, t));
}

// NGP_MARCH_NOISE_FROM_SEED: the per-ray start offset in [0, 1) without a noise tensor -- a counter-based draw from (ray index, seed),
// the seed read from device memory (any word that changes from step to step, e.g. the optimizer's step count), so that a captured
// graph needs neither torch's generator bookkeeping (two fill launches per replay) nor a separate rand kernel.
__device__ __forceinline__ float seeded_noise(uint32_t n, uint32_t seed) {
    uint32_t x = (n * 0x9E3779B9u) ^ (seed * 0x85EBCA6Bu + 0x27D4EB2Fu);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (float)(x >> 8) * 0x1p-24f;
}

template <bool WRITE, bool CONST_DT>
__global__ __launch_bounds__(MW_WAVES * 64) void k_march_train_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                                    const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                                    float* __restrict__ nears, float* __restrict__ fars,
                                                                    float* __restrict__ xyzs, float* __restrict__ dirs,
                                                                    float* __restrict__ deltas, int32_t* __restrict__ rays,
                                                                    const float* __restrict__ noises, uint32_t* __restrict__ n_windows,
                                                                    uint64_t* __restrict__ masks, bool noise_from_seed,
                                                                    const float* __restrict__ aabb, float min_near,
                                                                    uint32_t* __restrict__ fit_end, uint32_t ray_blocks,
                                                                    int32_t* __restrict__ self_scan_counter, bool zero_tail) {
    const uint32_t lane = threadIdx.x & 63;
    // Write pass with self_scan_counter != NULL: no scan launch between the passes.  Sample slots are still handed out in ray order: a
    // workgroup adds up the counts of all rays in front of its own (a few thousand L2-resident words), and the workgroups behind the
    // ray ones redo the whole prefix to find the end of the rows that will be written (fit_end), publish it together with the counter,
    // and zero the unowned tail.  (Used for N <= 8192 rays with the counter reset in-kernel; otherwise k_march_train_scan runs.)
    __shared__ uint32_t part[MW_WAVES], part2[MW_WAVES];
    if (WRITE && blockIdx.x >= ray_blocks) {
        uint32_t first_row;
        if (self_scan_counter) {
            constexpr uint32_t T = MW_WAVES * 64;
            const uint32_t per = (N + T - 1) / T, lo = min(N, threadIdx.x * per), hi = min(N, lo + per);
            uint32_t seg = 0;
            for (uint32_t i = lo; i < hi; i++) seg += (uint32_t)rays[i * 3 + 2];
            const uint32_t incl = wave_inclusive_scan(seg);
            if (lane == 63) part[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < MW_WAVES; w++) { before += (uint32_t)w < (threadIdx.x >> 6) ? part[w] : 0u; total += part[w]; }
            uint32_t off = before + incl - seg, unfit = 0xFFFFFFFFu;  // offset of my segment's first ray; first ray that does not fit
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t c = (uint32_t)rays[i * 3 + 2];
                if (c != 0u && off + c > M && unfit == 0xFFFFFFFFu) unfit = off;
                off += c;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) unfit = min(unfit, (uint32_t)__shfl_xor((int)unfit, o, 64));
            if (lane == 0) part2[threadIdx.x >> 6] = unfit;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < MW_WAVES; w++) unfit = min(unfit, part2[w]);
            const uint32_t end = unfit != 0xFFFFFFFFu ? unfit : total;
            first_row = end < M ? end : M;
            if (blockIdx.x == ray_blocks && threadIdx.x == 0) {
                self_scan_counter[0] = (int32_t)total;
                self_scan_counter[1] = (int32_t)N;
                fit_end[0] = first_row;
                fit_end[1] = 0u;  // the ticket of k_composite_train_loss_bwd
            }
            if (blockIdx.x == ray_blocks && threadIdx.x < CT_GROUPS) fit_end[march_ws_ticket_word(N) + threadIdx.x * CT_GROUP_STRIDE] = 0u;
        } else {
            first_row = fit_end[0];
        }
        if (zero_tail) {
            // the sample rows [fit_end, M) that no ray writes: the extra workgroups of the write pass zero them
            const uint32_t stride = (gridDim.x - ray_blocks) * MW_WAVES * 64;
            for (uint32_t row = first_row + (blockIdx.x - ray_blocks) * MW_WAVES * 64 + threadIdx.x; row < M; row += stride) {
                xyzs[(size_t)row * 3] = 0.0f; xyzs[(size_t)row * 3 + 1] = 0.0f; xyzs[(size_t)row * 3 + 2] = 0.0f;
                dirs[(size_t)row * 3] = 0.0f; dirs[(size_t)row * 3 + 1] = 0.0f; dirs[(size_t)row * 3 + 2] = 0.0f;
                *reinterpret_cast<float2_t*>(deltas + (size_t)row * 2) = float2_t{0.0f, 0.0f};
            }
        }
        return;
    }
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = blockIdx.x * MW_WAVES + wid;  // wave-uniform
    uint32_t scanned_offset = 0;
    if (WRITE && self_scan_counter) {
        uint32_t sum = 0;
        const uint32_t first = blockIdx.x * MW_WAVES;
        for (uint32_t i = threadIdx.x; i < first; i += MW_WAVES * 64) sum += (uint32_t)rays[i * 3 + 2];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, 64);
        if (lane == 0) part[wid] = sum;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < MW_WAVES; w++) scanned_offset += part[w];
        for (uint32_t w = 0; w < wid; w++) scanned_offset += first + w < N ? (uint32_t)rays[(first + w) * 3 + 2] : 0u;
        if (n < N && lane == 0) rays[n * 3 + 1] = (int32_t)scanned_offset;
    }
    if (n >= N) return;
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    const Ray r = load_ray(rays_o, rays_d, n);
    float far, near;
    if (!WRITE && aabb) {  // near_far_from_aabb folded into the count pass: same arithmetic, one launch less; the write pass reads it back
        near_far_from_aabb(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, aabb, min_near, near, far);
        if (lane == 0) { nears[n] = near; fars[n] = far; }
    } else {
        far = fars[n];
        near = nears[n];
    }
    const float dt_const = step_dt(p, 0.0f);  // value of dt(t) when dt_gamma == 0

    uint32_t limit = max_steps, offset = 0;
    if (WRITE) {
        offset = self_scan_counter ? scanned_offset : (uint32_t)rays[n * 3 + 1];
        limit = (uint32_t)rays[n * 3 + 2];
        if (limit == 0 || offset + limit > M) return;  // raymarching.cu:405-416: recorded, nothing written
    }
    uint32_t num_steps = 0;
    const float noise = noise_from_seed ? seeded_noise(n, reinterpret_cast<const uint32_t*>(noises)[0]) : noises[n];
    float t_base = __builtin_fmaf(step_dt(p, near), noise, near);
    float last_t = t_base;
    uint32_t window = 0;

    if (WRITE) {
        // Fast path: the count pass recorded, per 64-term window, which terms the walk emitted (MARCH_MASK_WINDOWS words per
        // ray).  Re-create the terms with the same recurrence and store the marked ones: no probing, no search, no walk.
        const uint32_t nw = n_windows[n];
        if (nw <= MARCH_MASK_WINDOWS) {
            const uint64_t* my_masks = masks + (size_t)n * MARCH_MASK_WINDOWS;
            for (; window < nw; window++) {
                float mine, t_next, inc_unused;
                bool closed_unused;
                window_terms<CONST_DT>(p, t_base, dt_const, lane, mine, t_next, inc_unused, closed_unused);
                t_base = t_next;
                const uint64_t emit = my_masks[window];
                if (emit == 0ull) continue;
                const float x = clampf(__builtin_fmaf(mine, r.dx, r.ox), -p.bound, p.bound);
                const float y = clampf(__builtin_fmaf(mine, r.dy, r.oy), -p.bound, p.bound);
                const float z = clampf(__builtin_fmaf(mine, r.dz, r.oz), -p.bound, p.bound);
                const float dt = step_dt(p, mine);
                const bool e = (emit >> lane) & 1ull;
                const uint64_t below = emit & ((1ull << lane) - 1ull);
                const uint32_t rank = (uint32_t)__builtin_popcountll(below);
                const float t_after = mine + dt;
                const int src = below ? 63 - __builtin_clzll(below) : 0;
                const float prev_after = __shfl(t_after, src, 64);
                const float lt = below ? prev_after : last_t;
                if (e) {
                    const size_t o = (size_t)offset + rank;
                    float* xo = xyzs + o * 3;
                    float* dd = dirs + o * 3;
                    xo[0] = x; xo[1] = y; xo[2] = z;
                    dd[0] = r.dx; dd[1] = r.dy; dd[2] = r.dz;
                    *reinterpret_cast<float2_t*>(deltas + o * 2) = float2_t{dt, t_after - lt};
                }
                last_t = readlane_f(t_after, 63u - (uint32_t)__builtin_clzll(emit));
                offset += (uint32_t)__builtin_popcountll(emit);
            }
            return;
        }
    }
    float carry = -INFINITY;  // the walk enters a window at its first term that is not < carry
    bool done = !(t_base < far);

    // The windows of a ray are a dependent chain (carry, step budget), and each one needs a bitfield read: software-pipelined -- the
    // terms of window w+1 do not depend on the walk of window w, so their positions are computed and their reads issued before the
    // walk of window w starts.
    float mine_n = 0.0f, t_after_n = t_base, base_n = t_base, inc_n = 0.0f;
    bool closed_n = false;
    ProbeGeom g_n = {};
    uint32_t byte_n = 0u;
    bool valid_n = false;
    auto issue_window = [&](float base) {
        base_n = base;
        window_terms<CONST_DT>(p, base, dt_const, lane, mine_n, t_after_n, inc_n, closed_n);
        valid_n = mine_n < far;
        g_n = probe_geom(p, r, valid_n ? mine_n : near);
        byte_n = p.grid[g_n.index >> 3];  // unconditional (a valid address for every lane): nothing waits on it here
    };
    if (!done) issue_window(t_base);

    while (!done) {
        // ---- 1. + 2. this window's terms and probes were issued one iteration ago; start the next window's ----
        const float mine = mine_n;
        const float t_next_base = t_after_n;
        const float win_base = base_n, win_inc = inc_n;
        const bool win_closed = closed_n;
        const bool valid = valid_n;
        const ProbeGeom g = g_n;
        const uint32_t byte = byte_n;
        if (__builtin_amdgcn_readfirstlane((int)(t_next_base < far))) issue_window(t_next_base);
        else valid_n = false;  // a further window (entered when all 64 terms were in front of `far`) has nothing to probe
        bool occ = false;
        float tt = -INFINITY;
        const float x = g.x, y = g.y, z = g.z, dt = g.dt;
        if (valid) occ = probe_resolve(p, r, mine, g, byte, tt);
        const uint64_t validmask = __ballot(valid);
        const uint64_t occmask = __ballot(valid && occ);
        const uint64_t entrymask = __ballot(!(mine < carry));
        uint32_t cur = entrymask ? (uint32_t)__builtin_ctzll(entrymask) : 64u;
        bool ray_done = false;
        uint64_t emit = 0;

        if (cur < 64u) {
            // landing index of "do t += dt(t) while (t < tt)" started at this lane: first j > lane with !(t_j < tt)
            const float target = (valid && !occ) ? tt : -INFINITY;
            uint32_t land = lane + 1u;
            if (CONST_DT && win_closed) {
                // closed-form window: term q = win_base + q * win_inc exactly, so the landing index is ceil((target - base) / inc) up
                // to the rounding of that quotient -- start one below the estimate and confirm against the exact terms (the
                // estimate is off by less than one, so two confirmations reach the first q with !(t_q < target))
                const float est = fminf(fmaxf(ceilf((target - win_base) / win_inc) - 1.0f, 0.0f), 64.0f);  // (-inf target -> 0)
                const uint32_t q0 = (uint32_t)est;
                land = q0 > land ? q0 : land;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const float tq = __builtin_fmaf((float)land, win_inc, win_base);
                    land = (land < 64u && tq < target) ? land + 1u : land;
                }
                land = land < 64u ? land : 64u;
            } else {
#pragma unroll
                for (uint32_t s = 32; s >= 1; s >>= 1) {
                    const uint32_t q = land + s - 1u;
                    const float tq = __shfl(mine, (int)(q & 63u), 64);
                    const bool adv = (q < 64u) && (tq < target);
                    land = adv ? land + s : land;
                }
            }
            carry = -INFINITY;
            // ---- 3. the sequential walk over wave-uniform scalars ----
            while (cur < 64u) {
                if (!((validmask >> cur) & 1ull) || num_steps >= limit) { ray_done = true; break; }
                if ((occmask >> cur) & 1ull) {
                    const uint64_t rest = ~(occmask >> cur);
                    uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;  // rest == 0 only for a fully occupied window at cur == 0
                    const uint32_t room = limit - num_steps;
                    run = run < room ? run : room;
                    emit |= (run >= 64u ? ~0ull : ((1ull << run) - 1ull)) << cur;
                    num_steps += run;
                    cur += run;
                } else {
                    const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)land, (int)cur);
                    if (nx >= 64u) carry = readlane_f(target, cur);
                    cur = nx;
                }
            }
        }

        // ---- 4. emit ----
        if (WRITE && emit != 0ull) {
            const bool e = (emit >> lane) & 1ull;
            const uint64_t below = emit & ((1ull << lane) - 1ull);
            const uint32_t rank = (uint32_t)__builtin_popcountll(below);
            const float t_after = mine + dt;
            const int src = below ? 63 - __builtin_clzll(below) : 0;
            const float prev_after = __shfl(t_after, src, 64);
            const float lt = below ? prev_after : last_t;
            if (e) {
                const size_t o = (size_t)offset + rank;
                NGP_BOUNDS(o < (size_t)M);
                float* xo = xyzs + o * 3;
                float* dd = dirs + o * 3;
                xo[0] = x; xo[1] = y; xo[2] = z;
                dd[0] = r.dx; dd[1] = r.dy; dd[2] = r.dz;
                *reinterpret_cast<float2_t*>(deltas + o * 2) = float2_t{dt, t_after - lt};
            }
            last_t = readlane_f(t_after, 63u - (uint32_t)__builtin_clzll(emit));
            offset += (uint32_t)__builtin_popcountll(emit);
        }
        if (!WRITE && lane == 0 && window < MARCH_MASK_WINDOWS) masks[(size_t)n * MARCH_MASK_WINDOWS + window] = emit;
        window++;
        done = ray_done || (validmask != ~0ull);
        t_base = t_next_base;
    }
    if (!WRITE && lane == 0) {
        rays[n * 3] = (int32_t)n;
        rays[n * 3 + 2] = (int32_t)num_steps;
        n_windows[n] = window;
    }
}

// exclusive scan of the per-ray sample counts in ray order -> rays[:,1]; counter[0] += total, counter[1] += N.
// reset != 0: the counter is treated as {0, 0} first (saves the caller's memset launch).
// ws[0] receives `fit_end`: the end of the contiguous prefix of sample rows that will actually be written -- the offset of the
// first ray that does not fit in M rows (every later ray starts even further back, raymarching.cu:405-416), or the total.
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void k_march_train_scan(int32_t* __restrict__ rays, int32_t* __restrict__ counter, uint32_t N,
                                                                   uint32_t M, int reset, uint32_t* __restrict__ ws) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ uint32_t first_unfit;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t running = reset ? 0u : (uint32_t)counter[0];
    const int32_t rays_before = reset ? 0 : counter[1];
    if (threadIdx.x == 0) first_unfit = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t tile = 0; tile < N; tile += SCAN_THREADS) {
        const uint32_t n = tile + threadIdx.x;
        const uint32_t c = n < N ? (uint32_t)rays[n * 3 + 2] : 0u;
        const uint32_t incl = wave_inclusive_scan(c);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            const uint32_t v = wsum[w];
            wbase += w < wid ? v : 0u;
            total += v;
        }
        const uint32_t offset = running + wbase + incl - c;
        if (n < N) {
            rays[n * 3 + 1] = (int32_t)offset;
            if (c != 0u && offset + c > M) atomicMin(&first_unfit, offset);
        }
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counter[0] = (int32_t)running;
        counter[1] = rays_before + (int32_t)N;
        const uint32_t fit_end = first_unfit != 0xFFFFFFFFu ? first_unfit : running;
        ws[0] = fit_end < M ? fit_end : M;
        ws[1] = 0u;  // the ticket of k_composite_train_loss_bwd
    }
    if (threadIdx.x < CT_GROUPS) ws[march_ws_ticket_word(N) + threadIdx.x * CT_GROUP_STRIDE] = 0u;  // ... and its group tickets
}

// samples per ray and iteration of the inference loop (renderer.py:349: max(min(N // n_alive, 8), 1)).  cap: the 8 of that rule; the
// on-device loop may raise it for the tail of a frame (cap = 0 means 8): a handful of surviving rays then finish in a few iterations
// instead of one launch set per 8 samples -- a ray's samples and their compositing order do not depend on the chunking
__device__ __host__ __forceinline__ uint32_t loop_n_step(uint32_t n_total, uint32_t n_alive, uint32_t cap) {
    if (cap == 0u) cap = 8u;
    if (n_alive == 0) return 1u;
    const uint32_t q = n_total / n_alive;
    return q > cap ? cap : (q < 1u ? 1u : q);
}

// raymarching.cu:701-805
// zero_rows > 0: the kernel also zeroes every sample slot it does not fill (the unused tail of each ray's n_step slots and the
// rows between n_alive * n_step and zero_rows), so the caller may pass uninitialised buffers (extension; the reference contract,
// zero_rows == 0, expects pre-zeroed buffers -- raymarching.py:334-336).
__global__ __launch_bounds__(RM_THREADS) void k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                           const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                           uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                           const float* __restrict__ fars, float* __restrict__ xyzs,
                                                           float* __restrict__ dirs, float* __restrict__ deltas,
                                                           const float* __restrict__ noises, uint32_t zero_rows,
                                                           const int32_t* __restrict__ dev_state, uint32_t n_total, uint32_t n_step_cap,
                                                           uint32_t* __restrict__ rows_used) {
    const uint32_t n = blockIdx.x * RM_THREADS + threadIdx.x;
    if (dev_state) {  // on-device render loop: the alive count lives on the device, n_step follows the renderer's rule
        n_alive = (uint32_t)dev_state[0];
        n_step = loop_n_step(n_total, n_alive, n_step_cap);
    }
    if (rows_used) {
        // the rows that can carry a sample in this iteration, padded by the marchers' rule (raymarching.py:328-331) -- published for the
        // encoder / network launches behind this one (ngp_grid_encode_forward_sel, ngp_network_forward_rows), which are sized for the
        // caller's stale bound and stop here; only these rows are zero-filled
        const uint32_t used = n_alive * n_step;
        const uint32_t padded = min(zero_rows, used + 128u - used % 128u);
        if (n == 0u) rows_used[0] = padded;
        zero_rows = padded;
    }
    if (zero_rows > 0) {  // padding rows behind the last ray's slots: at most `align` of them, spread over the first lanes
        for (uint32_t row = n_alive * n_step + n; row < zero_rows; row += gridDim.x * RM_THREADS) {
            xyzs[(size_t)row * 3] = 0.0f; xyzs[(size_t)row * 3 + 1] = 0.0f; xyzs[(size_t)row * 3 + 2] = 0.0f;
            dirs[(size_t)row * 3] = 0.0f; dirs[(size_t)row * 3 + 1] = 0.0f; dirs[(size_t)row * 3 + 2] = 0.0f;
            deltas[(size_t)row * 2] = 0.0f; deltas[(size_t)row * 2 + 1] = 0.0f;
        }
    }
    if (n >= n_alive) return;
    const uint32_t index = (uint32_t)rays_alive[n];
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    const Ray r = load_ray(rays_o, rays_d, index);
    const float far = fars[index];
    float t = rays_t[index];
    t = __builtin_fmaf(step_dt(p, t), noises ? noises[n] : 0.0f, t);
    float last_t = t;
    float* xo = xyzs + (size_t)n * n_step * 3;
    float* dd = dirs + (size_t)n * n_step * 3;
    float* de = deltas + (size_t)n * n_step * 2;
    uint32_t step = 0;
    float x, y, z, dt, tt;
    const bool const_dt = dt_gamma == 0.0f;
    const float dt_const = step_dt(p, 0.0f);
    while (t < far && step < n_step) {
        if (probe(p, r, t, x, y, z, dt, tt)) {
            xo[0] = x; xo[1] = y; xo[2] = z;
            dd[0] = r.dx; dd[1] = r.dy; dd[2] = r.dz;
            t += dt;
            de[0] = dt; de[1] = t - last_t;
            last_t = t;
            xo += 3; dd += 3; de += 2;
            step++;
        } else {
            t = const_dt ? skip_to_unrolled<true>(p, t, tt, far, dt_const) : skip_to_unrolled<false>(p, t, tt, far, dt_const);
        }
    }
    if (zero_rows > 0) {
        for (; step < n_step; step++) {
            xo[0] = 0.0f; xo[1] = 0.0f; xo[2] = 0.0f;
            dd[0] = 0.0f; dd[1] = 0.0f; dd[2] = 0.0f;
            de[0] = 0.0f; de[1] = 0.0f;
            xo += 3; dd += 3; de += 2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// compositing (training): one wavefront per ray           raymarching.cu:501-577, 602-682
// ---------------------------------------------------------------------------------------------
// Wave-wide inclusive scans on the VALU (DPP: row shifts inside the 16-lane rows, then row_bcast:15 / row_bcast:31 carry the row totals
// across -- profiles/r01_dpp_probe.txt), no LDS crossbar: the compositing kernels are one latency chain per ray, and a ds_bpermute
// round trip per scan step (__shfl_up) was most of it.  A lane without a source keeps `identity`.
// (dpp_or, wave_incl_sum, lane63, wave_total: common.h)
__device__ __forceinline__ float wave_incl_prod(float v, int) {
    v *= dpp_or<0x111, 0xF>(1.0f, v);  // row_shr:1
    v *= dpp_or<0x112, 0xF>(1.0f, v);  // row_shr:2
    v *= dpp_or<0x114, 0xF>(1.0f, v);  // row_shr:4
    v *= dpp_or<0x118, 0xF>(1.0f, v);  // row_shr:8
    v *= dpp_or<0x142, 0xA>(1.0f, v);  // row_bcast:15 into rows 1 and 3
    v *= dpp_or<0x143, 0xC>(1.0f, v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float prev_lane(float identity, float v) { return dpp_or<0x138, 0xF>(identity, v); }  // wave_shr:1, lane 0 keeps identity

constexpr int CT_WAVES = 4;  // rays per workgroup

// optional fused epilogue/prologue of the renderer (mode 0: off = the reference op; 1: scalar background; 2: per-ray [N,3])
struct Finish {
    int mode;
    float bg_scalar;
    const float* bg;
    const float* nears;
    const float* fars;
    float* image_out;
    float* depth_out;
};

__global__ __launch_bounds__(CT_WAVES * 64) void k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                       const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                       uint32_t M, uint32_t N, float T_thresh,
                                                                       float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                       float* __restrict__ image, Finish fin) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num = (uint32_t)rays[n * 3 + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (num != 0 && offset + num <= M) {
        float T = 1.0f, tcarry = 0.0f;  // transmittance / accumulated real-delta before this row
        for (uint32_t s0 = 0; s0 < num; s0 += 64) {
            const uint32_t s = s0 + lane;
            const bool valid = s < num;
            float sg = 0.0f, d0 = 0.0f, d1 = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
            if (valid) {
                sg = sigmas[offset + s];
                const float2_t dl = *reinterpret_cast<const float2_t*>(deltas + (size_t)(offset + s) * 2);
                d0 = dl.x; d1 = dl.y;
                cr = rgbs[(size_t)(offset + s) * 3];
                cg = rgbs[(size_t)(offset + s) * 3 + 1];
                cb = rgbs[(size_t)(offset + s) * 3 + 2];
            }
            const float alpha = valid ? 1.0f - __expf(-sg * d0) : 0.0f;
            const float om = 1.0f - alpha;
            const float pin = wave_incl_prod(om, lane);          // prod_{j<=lane} (1-alpha_j)
                        const float T_before = T * prev_lane(1.0f, pin);  // transmittance in front of this sample
            const float tt = tcarry + wave_incl_sum(d1, lane);    // t after this sample
            // the sample that drives T below the threshold is still composited (raymarching.cu:557-560)
            const bool live = valid && !(T_before < T_thresh);
            const float w = live ? alpha * T_before : 0.0f;
            r += w * cr; g += w * cg; b += w * cb; ws += w; d += w * tt;
            T = T * lane63(pin);
            tcarry = lane63(tt);
            if (T < T_thresh) break;  // wave-uniform
        }
        r = wave_total(r); g = wave_total(g); b = wave_total(b); ws = wave_total(ws); d = wave_total(d);
    }
    if (lane == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
        if (fin.mode != 0) {
            // NeRFRenderer.run_cuda's epilogue (renderer.py:316-318): image + (1 - weights_sum) * bg ; clamp(depth - near, 0) / (far - near)
            const float t1 = 1.0f - ws;
            const float b0 = fin.mode == 2 ? fin.bg[index * 3] : fin.bg_scalar, b1 = fin.mode == 2 ? fin.bg[index * 3 + 1] : fin.bg_scalar,
                        b2 = fin.mode == 2 ? fin.bg[index * 3 + 2] : fin.bg_scalar;
            fin.image_out[index * 3] = r + t1 * b0;
            fin.image_out[index * 3 + 1] = g + t1 * b1;
            fin.image_out[index * 3 + 2] = b + t1 * b2;
            const float nr = fin.nears[index], fr = fin.fars[index];
            fin.depth_out[index] = fmaxf(d - nr, 0.0f) / (fr - nr);
        }
    }
}

__global__ __launch_bounds__(CT_WAVES * 64) void k_composite_train_bwd(const float* __restrict__ grad_ws, const float* __restrict__ grad_image,
                                                                       const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                       const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                       const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                       uint32_t M, uint32_t N, float T_thresh, float* __restrict__ grad_sigmas,
                                                                       float* __restrict__ grad_rgbs, Finish fin,
                                                                       const uint32_t* __restrict__ rows_used, uint32_t ray_blocks) {
    const int lane = threadIdx.x & 63;
    // rows_used != NULL: the outputs arrive UNINITIALISED and every row the compositing does not reach is zeroed here -- the rows of
    // a ray behind its early termination (below) and the rows >= *rows_used that no ray owns (the workgroups after the ray ones)
    if (blockIdx.x >= ray_blocks) {
        const uint32_t first = min(rows_used[0], M);
        const uint32_t stride = (gridDim.x - ray_blocks) * CT_WAVES * 64;
        for (uint32_t o = first + (blockIdx.x - ray_blocks) * CT_WAVES * 64 + threadIdx.x; o < M; o += stride) {
            grad_sigmas[o] = 0.0f;
            grad_rgbs[(size_t)o * 3] = 0.0f; grad_rgbs[(size_t)o * 3 + 1] = 0.0f; grad_rgbs[(size_t)o * 3 + 2] = 0.0f;
        }
        return;
    }
    const bool zero_fill = rows_used != nullptr;
    const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num = (uint32_t)rays[n * 3 + 2];
    if (num == 0 || offset + num > M) return;
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    float gw = grad_ws ? grad_ws[index] : 0.0f;
    if (fin.mode != 0) {  // grad_image is the gradient of the FINISHED image: d/d(weights_sum) picks up -sum_c g_c * bg_c
        const float b0 = fin.mode == 2 ? fin.bg[index * 3] : fin.bg_scalar, b1 = fin.mode == 2 ? fin.bg[index * 3 + 1] : fin.bg_scalar,
                    b2 = fin.mode == 2 ? fin.bg[index * 3 + 2] : fin.bg_scalar;
        gw -= gi0 * b0 + gi1 * b1 + gi2 * b2;
    }
    const float rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2], wsf = weights_sum[index];
    float T = 1.0f, rc = 0.0f, gc = 0.0f, bc = 0.0f;  // carries from previous rows
    for (uint32_t s0 = 0; s0 < num; s0 += 64) {
        const uint32_t s = s0 + lane;
        const bool valid = s < num;
        float sg = 0.0f, d0 = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (valid) {
            sg = sigmas[offset + s];
            d0 = deltas[(size_t)(offset + s) * 2];
            cr = rgbs[(size_t)(offset + s) * 3];
            cg = rgbs[(size_t)(offset + s) * 3 + 1];
            cb = rgbs[(size_t)(offset + s) * 3 + 2];
        }
        const float alpha = valid ? 1.0f - __expf(-sg * d0) : 0.0f;
        const float pin = wave_incl_prod(1.0f - alpha, lane);
                const float T_before = T * prev_lane(1.0f, pin);
        const float T_after = T * pin;
        const bool live = valid && !(T_before < T_thresh);
        const float w = live ? alpha * T_before : 0.0f;
        const float ra = rc + wave_incl_sum(w * cr, lane);
        const float ga = gc + wave_incl_sum(w * cg, lane);
        const float ba = bc + wave_incl_sum(w * cb, lane);
        if (live) {
            const uint32_t o = offset + s;
            grad_rgbs[(size_t)o * 3] = gi0 * w;
            grad_rgbs[(size_t)o * 3 + 1] = gi1 * w;
            grad_rgbs[(size_t)o * 3 + 2] = gi2 * w;
            grad_sigmas[o] = d0 * (gi0 * (T_after * cr - (rf - ra)) + gi1 * (T_after * cg - (gf - ga)) +
                                   gi2 * (T_after * cb - (bf - ba)) + gw * (1.0f - wsf));
        } else if (zero_fill && valid) {
            const uint32_t o = offset + s;
            grad_rgbs[(size_t)o * 3] = 0.0f; grad_rgbs[(size_t)o * 3 + 1] = 0.0f; grad_rgbs[(size_t)o * 3 + 2] = 0.0f;
            grad_sigmas[o] = 0.0f;
        }
        T = T * lane63(pin);
        rc = lane63(ra); gc = lane63(ga); bc = lane63(ba);
        if (T < T_thresh) {
            if (zero_fill)
                for (uint32_t z = s0 + 64 + lane; z < num; z += 64) {
                    const uint32_t o = offset + z;
                    grad_rgbs[(size_t)o * 3] = 0.0f; grad_rgbs[(size_t)o * 3 + 1] = 0.0f; grad_rgbs[(size_t)o * 3 + 2] = 0.0f;
                    grad_sigmas[o] = 0.0f;
                }
            break;
        }
    }
}

// The image-space middle of a training iteration in ONE launch, one wavefront per ray:
//   k_composite_train_fwd (+ finish)  ->  the Trainer's MSE loss and its scaled gradient (k_mse_loss)  ->  k_composite_train_bwd  ->
//   the colour head's sigmoid backward (k_rgb_backward),
// with the arithmetic of those four kernels expression for expression, so the gradients are the same bits.  What a ray needs from the
// loss is its own three pixels, so nothing crosses rays except the loss VALUE (a logged scalar): every ray deposits its squared error,
// the last workgroup to finish (a ticket) adds them up in a fixed order -> deterministic.  Four launches, their tails and the
// [N,3] / [M,3] fp32 intermediates (grad_image, grad_rgbs) are gone; the second sweep re-reads sigma / rgb / delta from L2.
// ticket[0] and the group tickets must be 0 on entry (the marcher clears them every step); the kernel leaves them 0.
__global__ __launch_bounds__(CT_WAVES * 64) void k_composite_train_loss_bwd(
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
    uint32_t M, uint32_t N, float T_thresh, float* __restrict__ weights_sum, Finish fin, const float* __restrict__ target,
    const float* __restrict__ loss_scale, float* __restrict__ ray_err, uint32_t* __restrict__ ticket, float* __restrict__ loss,
    float* __restrict__ grad_sigmas, half_t* __restrict__ grad_out16, const uint32_t* __restrict__ rows_used, uint32_t ray_blocks,
    uint32_t* __restrict__ group_tickets) {
    const int lane = threadIdx.x & 63;
    auto zero_row = [&](uint32_t o) {
        half8_t z;
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = (half_t)0.0f;
        half8_t* dst = reinterpret_cast<half8_t*>(grad_out16 + (size_t)o * 16);
        dst[0] = z; dst[1] = z;
        grad_sigmas[o] = 0.0f;
    };
    if (blockIdx.x >= ray_blocks) {  // rows >= *rows_used that no ray owns
        const uint32_t first = min(rows_used[0], M);
        const uint32_t stride = (gridDim.x - ray_blocks) * CT_WAVES * 64;
        for (uint32_t o = first + (blockIdx.x - ray_blocks) * CT_WAVES * 64 + threadIdx.x; o < M; o += stride) zero_row(o);
        return;
    }
    const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
    if (n < N) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num = (uint32_t)rays[n * 3 + 2];
        const bool has_rows = num != 0 && offset + num <= M;
        // ---- forward sweep (k_composite_train_fwd) ----
        float r = 0, g = 0, b = 0, ws = 0, d = 0;
        if (has_rows) {
            float T = 1.0f, tcarry = 0.0f;
            for (uint32_t s0 = 0; s0 < num; s0 += 64) {
                const uint32_t s = s0 + lane;
                const bool valid = s < num;
                float sg = 0.0f, d0 = 0.0f, d1 = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
                if (valid) {
                    sg = sigmas[offset + s];
                    const float2_t dl = *reinterpret_cast<const float2_t*>(deltas + (size_t)(offset + s) * 2);
                    d0 = dl.x; d1 = dl.y;
                    cr = rgbs[(size_t)(offset + s) * 3];
                    cg = rgbs[(size_t)(offset + s) * 3 + 1];
                    cb = rgbs[(size_t)(offset + s) * 3 + 2];
                }
                const float alpha = valid ? 1.0f - __expf(-sg * d0) : 0.0f;
                const float om = 1.0f - alpha;
                const float pin = wave_incl_prod(om, lane);
                                const float T_before = T * prev_lane(1.0f, pin);
                const float tt = tcarry + wave_incl_sum(d1, lane);
                const bool live = valid && !(T_before < T_thresh);
                const float w = live ? alpha * T_before : 0.0f;
                r += w * cr; g += w * cg; b += w * cb; ws += w; d += w * tt;
                T = T * lane63(pin);
                tcarry = lane63(tt);
                if (T < T_thresh) break;
            }
            r = wave_total(r); g = wave_total(g); b = wave_total(b); ws = wave_total(ws); d = wave_total(d);
        }
        // ---- finish (renderer.py:316-318) + loss (nerf/utils.py:516,557), every lane the same values ----
        const float b0 = fin.mode == 2 ? fin.bg[index * 3] : fin.bg_scalar, b1 = fin.mode == 2 ? fin.bg[index * 3 + 1] : fin.bg_scalar,
                    b2 = fin.mode == 2 ? fin.bg[index * 3 + 2] : fin.bg_scalar;
        const float t1 = 1.0f - ws;
        const float i0 = r + t1 * b0, i1 = g + t1 * b1, i2 = b + t1 * b2;
        const float scale = loss_scale ? loss_scale[0] : 1.0f;
        const float norm = 2.0f / (float)(3u * N);
        const float e0 = i0 - target[index * 3], e1 = i1 - target[index * 3 + 1], e2 = i2 - target[index * 3 + 2];
        const float gi0 = (norm * e0) * scale, gi1 = (norm * e1) * scale, gi2 = (norm * e2) * scale;
        if (lane == 0) {
            weights_sum[index] = ws;
            fin.image_out[index * 3] = i0; fin.image_out[index * 3 + 1] = i1; fin.image_out[index * 3 + 2] = i2;
            const float nr = fin.nears[index], fr = fin.fars[index];
            fin.depth_out[index] = fmaxf(d - nr, 0.0f) / (fr - nr);
            // write-through store (agent scope): the last workgroup reads it from another XCD without anybody flushing an L2
            __hip_atomic_store(&ray_err[n], __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- backward sweep (k_composite_train_bwd with grad_weights_sum = 0, then k_rgb_backward) ----
        if (has_rows) {
            const float gw = 0.0f - (gi0 * b0 + gi1 * b1 + gi2 * b2);
            const float rf = r, gf = g, bf = b, wsf = ws;
            float T = 1.0f, rc = 0.0f, gc = 0.0f, bc = 0.0f;
            for (uint32_t s0 = 0; s0 < num; s0 += 64) {
                const uint32_t s = s0 + lane;
                const bool valid = s < num;
                float sg = 0.0f, d0 = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
                if (valid) {
                    sg = sigmas[offset + s];
                    d0 = deltas[(size_t)(offset + s) * 2];
                    cr = rgbs[(size_t)(offset + s) * 3];
                    cg = rgbs[(size_t)(offset + s) * 3 + 1];
                    cb = rgbs[(size_t)(offset + s) * 3 + 2];
                }
                const float alpha = valid ? 1.0f - __expf(-sg * d0) : 0.0f;
                const float pin = wave_incl_prod(1.0f - alpha, lane);
                                const float T_before = T * prev_lane(1.0f, pin);
                const float T_after = T * pin;
                const bool live = valid && !(T_before < T_thresh);
                const float w = live ? alpha * T_before : 0.0f;
                const float ra = rc + wave_incl_sum(w * cr, lane);
                const float ga = gc + wave_incl_sum(w * cg, lane);
                const float ba = bc + wave_incl_sum(w * cb, lane);
                if (live) {
                    const uint32_t o = offset + s;
                    half8_t lo, hi;
#pragma unroll
                    for (int i = 0; i < 8; i++) { lo[i] = (half_t)0.0f; hi[i] = (half_t)0.0f; }
                    lo[0] = to_half_rne((gi0 * w) * (cr * (1.0f - cr)));
                    lo[1] = to_half_rne((gi1 * w) * (cg * (1.0f - cg)));
                    lo[2] = to_half_rne((gi2 * w) * (cb * (1.0f - cb)));
                    half8_t* dst = reinterpret_cast<half8_t*>(grad_out16 + (size_t)o * 16);
                    dst[0] = lo; dst[1] = hi;
                    grad_sigmas[o] = d0 * (gi0 * (T_after * cr - (rf - ra)) + gi1 * (T_after * cg - (gf - ga)) +
                                           gi2 * (T_after * cb - (bf - ba)) + gw * (1.0f - wsf));
                } else if (valid) {
                    zero_row(offset + s);
                }
                T = T * lane63(pin);
                rc = lane63(ra); gc = lane63(ga); bc = lane63(ba);
                if (T < T_thresh) {
                    for (uint32_t z = s0 + 64 + lane; z < num; z += 64) zero_row(offset + z);
                    break;
                }
            }
        }
    }
    // ---- the loss value: the last workgroup sums the per-ray errors in a fixed order ----
    // loss == NULL: the caller has the sum carried by a later launch (ngp_grid_encode_backward_checked_slabs: common.h loss_sum_block, the
    // same routine) -- no ticket, no device-scope round trip at the end of every workgroup, no serial tail: 19 -> 14 us.
    if (loss == nullptr) return;
    // (no agent-scope release fence: on this chip it writes back the XCD's whole dirty L2, once per workgroup -- measured 15 -> 130 us.
    // The per-ray errors are write-through atomic stores; a workgroup-scope release waits for them to complete before the ticket moves.)
    __shared__ float part[CT_WAVES];
    __shared__ bool last;
    // every lane waits for ITS OWN outstanding stores (the write-through ray_err store among them) before the barrier: a workgroup-scope
    // release alone need not emit s_waitcnt vmcnt(0) outside tgsplit mode, and the tickets below are relaxed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        // two levels (see CT_GROUPS): my group's ticket; the group's last arrival puts it back to 0 and takes the final ticket.
        // (Taking the group ticket early -- right after the forward sweep, its answer read here -- was measured: no gain, 19.9 vs 19.0 us;
        // what is left over a kernel without any ticket, 13.7 us, is one device-scope round trip and the last workgroup's sum.)
        const uint32_t groups = ray_blocks < CT_GROUPS ? ray_blocks : CT_GROUPS;
        const uint32_t group = blockIdx.x % groups, members = (ray_blocks - group + groups - 1u) / groups;
        uint32_t* gt = group_tickets + group * CT_GROUP_STRIDE;
        bool l = false;
        if (__hip_atomic_fetch_add(gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
        }
        last = l;
    }
    __syncthreads();
    if (!last) return;
    static_assert(CT_WAVES == 4, "loss_sum_block adds four wave sums");
    loss_sum_block(ray_err, N, loss, part);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// raymarching.cu:819-905 -- at most 8 samples per ray and call: one lane per alive ray
__global__ __launch_bounds__(RM_THREADS) void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* __restrict__ rays_alive,
                                                               float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                               const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                               float* __restrict__ weights_sum, float* __restrict__ depth,
                                                               float* __restrict__ image, const int32_t* __restrict__ dev_state, uint32_t n_total,
                                                               uint32_t n_step_cap) {
    const uint32_t n = blockIdx.x * RM_THREADS + threadIdx.x;
    if (dev_state) {
        n_alive = (uint32_t)dev_state[0];
        n_step = loop_n_step(n_total, n_alive, n_step_cap);
    }
    if (n >= n_alive) return;
    const uint32_t index = (uint32_t)rays_alive[n];
    const float* sg = sigmas + (size_t)n * n_step;
    const float* rg = rgbs + (size_t)n * n_step * 3;
    const float* de = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index], ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        const float d0 = de[step * 2];
        if (d0 == 0.0f) break;
        const float alpha = 1.0f - __expf(-sg[step] * d0);
        const float T = 1.0f - ws;
        const float w = alpha * T;
        ws += w;
        t += de[step * 2 + 1];
        d += w * t;
        r += w * rg[step * 3]; g += w * rg[step * 3 + 1]; b += w * rg[step * 3 + 2];
        if (T < T_thresh) break;
        step++;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// ---------------------------------------------------------------------------------------------
// order-preserving compaction of the alive list (extension; replaces a host-side boolean index)
// workspace (uint32): per-block survivor counts
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RM_THREADS) void k_compact_count(const int32_t* __restrict__ rays_alive, uint32_t n_alive, uint32_t* __restrict__ ws,
                                                              const int32_t* __restrict__ dev_state) {
    __shared__ uint32_t lds4[RM_THREADS / 64];
    if (dev_state) n_alive = (uint32_t)dev_state[0];
    const uint32_t n = blockIdx.x * RM_THREADS + threadIdx.x;
    const uint32_t keep = (n < n_alive && rays_alive[n] >= 0) ? 1u : 0u;
    const uint32_t tot = block_sum(keep, lds4);
    if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}
__global__ __launch_bounds__(RM_THREADS) void k_compact_write(const int32_t* __restrict__ rays_alive, uint32_t n_alive,
                                                              int32_t* __restrict__ out_alive, int32_t* __restrict__ out_count,
                                                              const uint32_t* __restrict__ ws, const int32_t* __restrict__ dev_state,
                                                              uint32_t n_total, uint32_t max_steps, uint32_t n_step_cap) {
    __shared__ uint32_t lds4[RM_THREADS / 64];
    __shared__ uint32_t wave_excl[RM_THREADS / 64];
    uint32_t steps_done = 0;
    if (dev_state) {  // out_count is the next iteration's state {n_alive, steps marched so far}
        n_alive = (uint32_t)dev_state[0];
        steps_done = (uint32_t)dev_state[1] + loop_n_step(n_total, n_alive, n_step_cap);
    }
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += RM_THREADS) part += ws[j];
    const uint32_t block_offset = block_sum(part, lds4);
    const uint32_t n = blockIdx.x * RM_THREADS + threadIdx.x;
    const int32_t id = (n < n_alive) ? rays_alive[n] : -1;
    const uint32_t keep = id >= 0 ? 1u : 0u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan(keep);
    __syncthreads();
    if (lane == 63) wave_excl[wid] = incl;
    __syncthreads();
    uint32_t wbase = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < RM_THREADS / 64; w++) {
        const uint32_t c = wave_excl[w];
        if (w < wid) wbase += c;
        block_total += c;
    }
    if (keep) out_alive[block_offset + wbase + incl - 1] = id;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (dev_state) {  // the loop ends when no ray is alive or max_steps samples were marched (renderer.py:341-346)
            out_count[0] = steps_done >= max_steps ? 0 : (int32_t)(block_offset + block_total);
            out_count[1] = (int32_t)steps_done;
        } else {
            *out_count = (int32_t)(block_offset + block_total);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Empty-ray culling for the inference loop (extension, round 5).  In the first iteration of an 800x800 frame every ray is alive and more
// than half of them never meet an occupied voxel: each of those walks the whole box voxel by voxel (~90 instructions per voxel, 284 us
// for the iteration) to emit nothing.  Which rays those are can be decided CONSERVATIVELY without the walk:
//   k_coarse_occupancy: per cascade a (H/4)^3 grid, cell = 1 when any voxel of its 4^3 block OR OF ANY OF THE 26 NEIGHBOURING BLOCKS is
//       occupied (the bitfield is in Morton order: a 4^3 block is 8 consecutive bytes).  (8^3 blocks were tried first: the dilated hull of
//       the lego-shaped scene then holds 76 % of the frame's rays -- 45 % really meet a voxel);
//   k_cull_rays: a ray is sampled from near to far at steps of (a little under) ONE coarse cell (of the finest cascade whose extent holds the point);
//       a sample looks up its cell in every cascade the marcher could select there (level >= the position's exponent).  Every point of
//       the segment lies within half a cell of a sample, i.e. in that sample's cell or a neighbour of it, which the dilation covers: a ray that keeps
//       finding 0 cannot touch an occupied voxel of any cascade, with a whole coarse cell of margin against rounding differences between this test
//       and the marcher (4 voxels; the two disagree by rounding only).  Such rays get -1 in the alive list (they would have produced no sample: same image, bit for bit); everything
//       else -- including every doubtful case -- is marched as before.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t COARSE_SHIFT = 2;   // coarse cell = (1 << COARSE_SHIFT)^3 voxels
__global__ __launch_bounds__(RM_THREADS) void k_coarse_occupancy(const uint8_t* __restrict__ grid, uint32_t C, uint32_t H, uint8_t* __restrict__ coarse) {
    const uint32_t R = H >> COARSE_SHIFT, cells = R * R * R;
    const uint32_t i = blockIdx.x * RM_THREADS + threadIdx.x;
    if (i >= C * cells) return;
    const uint32_t c = i / cells, k = i - c * cells;
    const int cx = (int)(k % R), cy = (int)((k / R) % R), cz = (int)(k / (R * R));
    const uint8_t* __restrict__ g = grid + (size_t)c * ((size_t)H * H * H / 8);
    uint32_t any = 0u;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x < 0 || y < 0 || z < 0 || x >= (int)R || y >= (int)R || z >= (int)R) continue;
                static_assert(COARSE_SHIFT == 2, "a 4^3 block of the Morton-ordered bitfield is 64 bits");
                const uint2 v = *reinterpret_cast<const uint2*>(g + (size_t)morton3D_1((uint32_t)x, (uint32_t)y, (uint32_t)z) * 8);
                any |= v.x | v.y;
            }
    coarse[i] = any ? 1u : 0u;
}

__global__ __launch_bounds__(RM_THREADS) void k_cull_rays(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const float* __restrict__ nears, const float* __restrict__ fars, uint32_t N, float bound,
                                                          uint32_t C, uint32_t H, const uint8_t* __restrict__ coarse, int32_t* __restrict__ rays_alive) {
    const uint32_t n = blockIdx.x * RM_THREADS + threadIdx.x;
    if (n >= N) return;
    const uint32_t R = H >> COARSE_SHIFT, cells = R * R * R;
    const float Rf = (float)R, Cm1 = (float)C - 1.0f;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float near = nears[n], far = fars[n];
    const float dlen = sqrtf(dx * dx + dy * dy + dz * dz);
    const float rdlen = 1.0f / dlen, rbound = 1.0f / bound;
    // (a ray with near >= far -- it misses the box -- is not marched at all by k_march_rays: no sample either way; a DEGENERATE ray that the
    // marcher would still take up -- zero or non-finite direction, far = +inf -- is kept, not judged)
    const bool regular = dlen > 0.0f && __builtin_isfinite(far) && __builtin_isfinite(dlen);
    bool keep = (near < far) && !regular;
    if (near < far && regular) {
        float t = near;
        for (uint32_t it = 0; it < 4096u && !keep; it++) {      // (bounded: a degenerate ray is kept, never spun on)
            const float tc = fminf(t, far);
            const float x = clamp_med3(__builtin_fmaf(tc, dx, ox), -bound, bound);
            const float y = clamp_med3(__builtin_fmaf(tc, dy, oy), -bound, bound);
            const float z = clamp_med3(__builtin_fmaf(tc, dz, oz), -bound, bound);
            int lp = 0;
            if (C > 1u) lp = mip_exponent(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), Cm1);
            // (from one cascade BELOW the sample's own: a point of the next half cell may already lie inside that finer cascade's box; the
            // sample itself is outside it and clamps to its boundary cell, which is the neighbour of that point's cell)
            for (int c = lp > 0 ? lp - 1 : 0; c < (int)C && !keep; c++) {
                const float pw = scalbnf(1.0f, c);
                const float rmb = pw < bound ? scalbnf(1.0f, -c) : rbound;   // 1 / min(2^c, bound) without dividing (as the marcher)
                const int ix = (int)clamp_med3((0.5f * __builtin_fmaf(x, rmb, 1.0f)) * Rf, 0.0f, Rf - 1.0f);
                const int iy = (int)clamp_med3((0.5f * __builtin_fmaf(y, rmb, 1.0f)) * Rf, 0.0f, Rf - 1.0f);
                const int iz = (int)clamp_med3((0.5f * __builtin_fmaf(z, rmb, 1.0f)) * Rf, 0.0f, Rf - 1.0f);
                keep = coarse[(size_t)c * cells + ((uint32_t)iz * R + (uint32_t)iy) * R + (uint32_t)ix] != 0u;
            }
            if (!(t < far)) break;
            // a little under ONE coarse cell of the finest cascade that holds this point, as a step of the ray parameter: every point of the
            // segment then lies within 0.45 cells of a sample, i.e. in the sample's cell or one of its 26 neighbours -- which the dilated
            // grid covers -- in that cascade and every coarser one; in the next finer cascade (cells half the size, looked up from
            // `lp - 1`) within 0.9 of its cells of the sample clamped into its box: again a neighbour at most.  (Round 5 started with half
            // this step: twice the samples, 74 us per 800x800 frame for the same verdicts.)
            t += 0.9f * (2.0f * fminf(scalbnf(1.0f, lp), bound) / Rf) * rdlen;
            if (it == 4095u) keep = true;
        }
    }
    rays_alive[n] = keep ? (int32_t)n : -1;
}

}  // namespace ngp

using namespace ngp;

#define RM_LAUNCH_1D(kernel, count, st, ...) hipLaunchKernelGGL(kernel, dim3(cdiv((count), RM_THREADS)), dim3(RM_THREADS), 0, st, __VA_ARGS__)

extern "C" int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                                      float* nears, float* fars, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb && nears && fars, NGP_ERR_INVALID, "near_far_from_aabb: NULL tensor");
    RM_LAUNCH_1D(k_near_far_from_aabb, N, as_stream(stream), rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch("near_far_from_aabb");
}

extern "C" int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                                ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && coords, NGP_ERR_INVALID, "sph_from_ray: NULL tensor");
    RM_LAUNCH_1D(k_sph_from_ray, N, as_stream(stream), rays_o, rays_d, radius, N, coords);
    return check_launch("sph_from_ray");
}

extern "C" int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, NGP_ERR_INVALID, "morton3D: NULL tensor");
    RM_LAUNCH_1D(k_morton3D, N, as_stream(stream), coords, N, indices);
    return check_launch("morton3D");
}

extern "C" int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, NGP_ERR_INVALID, "morton3D_invert: NULL tensor");
    RM_LAUNCH_1D(k_morton3D_invert, N, as_stream(stream), indices, N, coords);
    return check_launch("morton3D_invert");
}

extern "C" int ngp_packbits_ex(const float* grid, uint32_t N, float density_thresh, const float* thresh_cap, uint8_t* bitfield,
                               ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grid && bitfield, NGP_ERR_INVALID, "packbits: NULL tensor");
    NGP_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15) == 0, NGP_ERR_INVALID, "packbits: grid must be 16-byte aligned");
    RM_LAUNCH_1D(k_packbits, N, as_stream(stream), grid, N, density_thresh, thresh_cap, bitfield);
    return check_launch("packbits");
}

extern "C" int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, ngp_stream_t stream) {
    return ngp_packbits_ex(grid, N, density_thresh, nullptr, bitfield, stream);
}

extern "C" size_t ngp_density_grid_update_workspace_bytes(uint32_t n_cells) {
    return sizeof(double) * (size_t)cdiv(n_cells, DM_THREADS * DM_PER_THREAD) + 64;
}

// nerf/renderer.py:515-529 in three launches (see k_density_scatter).  cells: global cell index (cascade * H^3 + morton index) of every
// queried point; scratch [n_cells] fp32 must hold -1 everywhere before the FIRST call and is left that way; workspace
// (ngp_density_grid_update_workspace_bytes) must be zeroed before the first call.  mean_out[0] = mean(max(grid, 0)) after the update,
// bitfield = packbits(grid, min(density_thresh, mean)).
extern "C" int ngp_density_grid_update(const float* sigmas, const int64_t* cells, uint32_t n, float density_scale, float decay, float* density_grid,
                                       uint32_t n_cells, float* scratch, float density_thresh, float* mean_out, uint8_t* bitfield, void* workspace,
                                       ngp_stream_t stream) {
    NGP_REQUIRE(density_grid && scratch && mean_out && bitfield && workspace, NGP_ERR_INVALID, "density_grid_update: NULL tensor");
    NGP_REQUIRE(n_cells >= 8 && n_cells % 8 == 0, NGP_ERR_INVALID, "density_grid_update: the grid must hold a multiple of 8 cells (got %u)", n_cells);
    NGP_REQUIRE(((reinterpret_cast<uintptr_t>(density_grid) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0, NGP_ERR_INVALID,
                "density_grid_update: grid and scratch must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (n) {
        NGP_REQUIRE(sigmas && cells, NGP_ERR_INVALID, "density_grid_update: NULL tensor");
        RM_LAUNCH_1D(k_density_scatter, n, st, sigmas, cells, n, density_scale, scratch, n_cells);
    }
    const uint32_t blocks = cdiv(n_cells, DM_THREADS * DM_PER_THREAD);
    double* partials = reinterpret_cast<double*>(workspace);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(partials + blocks);
    hipLaunchKernelGGL(k_density_apply_mean, dim3(blocks), dim3(DM_THREADS), 0, st, density_grid, scratch, n_cells, decay, partials, ticket, mean_out);
    RM_LAUNCH_1D(k_packbits, n_cells / 8, st, (const float*)density_grid, n_cells / 8, density_thresh, (const float*)mean_out, bitfield);
    return check_launch("density_grid_update");
}

// workspace: [0] fit_end, [1] final ticket of the fused composite/loss/backward kernel (cleared by the scan), [2 .. 2+N) windows per ray,
// N x MARCH_MASK_WINDOWS 64-bit emit masks, then that kernel's CT_GROUPS group tickets (march_ws_ticket_word)
extern "C" size_t ngp_march_rays_train_workspace_bytes(uint32_t N) {
    return sizeof(uint32_t) * (march_ws_ticket_word(N) + (size_t)CT_GROUPS * CT_GROUP_STRIDE);
}

static int check_march_args(const char* fn, uint32_t C, uint32_t H, uint32_t max_steps) {
    NGP_REQUIRE(C >= 1 && C <= 8, NGP_ERR_INVALID, "%s: cascade count C must be in [1, 8] (got %u)", fn, C);
    NGP_REQUIRE(H >= 2 && H <= 1024, NGP_ERR_INVALID, "%s: grid size H must be in [2, 1024] (got %u)", fn, H);
    NGP_REQUIRE(max_steps >= 1, NGP_ERR_INVALID, "%s: max_steps must be positive", fn);
    return NGP_OK;
}

static int march_rays_train_impl(const float* rays_o, const float* rays_d, const uint8_t* grid_in, float bound, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* aabb, float min_near,
                                 float* nears, float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                 const float* noises, void* workspace, uint32_t flags, ngp_stream_t stream) {
    int rc = check_march_args("march_rays_train", C, H, max_steps);
    if (rc) return rc;
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && grid_in && nears && fars && xyzs && dirs && deltas && rays && counter && noises && workspace,
                NGP_ERR_INVALID, "march_rays_train: NULL tensor");
    hipStream_t st = as_stream(stream);
    uint32_t* ws = reinterpret_cast<uint32_t*>(workspace);
    const uint32_t ray_blocks = cdiv(N, MW_WAVES);
    const bool zero_tail = (flags & NGP_MARCH_ZERO_TAIL) && M > 0;
    // few rays and an in-kernel counter reset: the write pass hands out the sample slots itself (no scan launch between the passes)
    const bool self_scan = (flags & NGP_MARCH_RESET_COUNTER) && !(flags & NGP_MARCH_SCAN_LAUNCH) && N <= 8192u;
    const uint32_t extra = zero_tail ? 16u : (self_scan ? 1u : 0u);
    const dim3 block(MW_WAVES * 64);
    const bool const_dt = dt_gamma == 0.0f;
#define MARCH_WAVE(WRITE, CDT)                                                                                                              \
    hipLaunchKernelGGL((k_march_train_wave<WRITE, CDT>), dim3(ray_blocks + ((WRITE) ? extra : 0u)), block, 0, st, rays_o, rays_d, grid_bits,   \
                       bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, noises, ws_windows, ws_masks,              \
                       (flags & NGP_MARCH_NOISE_FROM_SEED) != 0, aabb, min_near, ws, ray_blocks, self_scan ? counter : (int32_t*)nullptr,       \
                       zero_tail)
    const uint8_t* grid_bits = grid_in;
    uint32_t* ws_windows = ws + 2;                                                               // [N]
    uint64_t* ws_masks = reinterpret_cast<uint64_t*>(ws + 2 + ((N + 1u) & ~1u));                 // [N][MARCH_MASK_WINDOWS], 8-byte aligned
    if (const_dt) MARCH_WAVE(false, true); else MARCH_WAVE(false, false);
    rc = check_launch("march_rays_train(count)");
    if (rc) return rc;
    if (!self_scan) {
        hipLaunchKernelGGL(k_march_train_scan, dim3(1), dim3(SCAN_THREADS), 0, st, rays, counter, N, M, (int)((flags & NGP_MARCH_RESET_COUNTER) != 0), ws);
        rc = check_launch("march_rays_train(scan)");
        if (rc) return rc;
    }
    if (const_dt) MARCH_WAVE(true, true); else MARCH_WAVE(true, false);
#undef MARCH_WAVE
    return check_launch("march_rays_train(write)");
}

extern "C" int ngp_march_rays_train_ex(const float* rays_o, const float* rays_d, const uint8_t* grid_in, float bound, float dt_gamma,
                                       uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                       const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                       const float* noises, void* workspace, uint32_t flags, ngp_stream_t stream) {
    // (nears / fars are only read when no box is given)
    return march_rays_train_impl(rays_o, rays_d, grid_in, bound, dt_gamma, max_steps, N, C, H, M, nullptr, 0.0f, const_cast<float*>(nears),
                                 const_cast<float*>(fars), xyzs, dirs, deltas, rays, counter, noises, workspace, flags, stream);
}

extern "C" int ngp_march_rays_train_aabb(const float* rays_o, const float* rays_d, const uint8_t* grid_in, float bound, float dt_gamma,
                                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* aabb, float min_near,
                                         float* nears, float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                         const float* noises, void* workspace, uint32_t flags, ngp_stream_t stream) {
    NGP_REQUIRE(aabb || N == 0, NGP_ERR_INVALID, "march_rays_train_aabb: NULL box");
    return march_rays_train_impl(rays_o, rays_d, grid_in, bound, dt_gamma, max_steps, N, C, H, M, aabb, min_near, nears, fars, xyzs, dirs, deltas,
                                 rays, counter, noises, workspace, flags, stream);
}

extern "C" int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid_in, float bound, float dt_gamma,
                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                    const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                    const float* noises, void* workspace, ngp_stream_t stream) {
    return ngp_march_rays_train_ex(rays_o, rays_d, grid_in, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                   counter, noises, workspace, 0u, stream);
}

static int make_finish(const char* fn, int bg_mode, float bg_scalar, const float* bg, const float* nears, const float* fars, float* image_out,
                       float* depth_out, bool forward, Finish* fin) {
    NGP_REQUIRE(bg_mode >= 0 && bg_mode <= 2, NGP_ERR_INVALID, "%s: bg_mode must be 0, 1 or 2", fn);
    NGP_REQUIRE(bg_mode != 2 || bg, NGP_ERR_INVALID, "%s: bg_mode 2 needs a per-ray background tensor", fn);
    NGP_REQUIRE(bg_mode == 0 || !forward || (nears && fars && image_out && depth_out), NGP_ERR_INVALID, "%s: NULL finishing tensor", fn);
    *fin = Finish{bg_mode, bg_scalar, bg, nears, fars, image_out, depth_out};
    return NGP_OK;
}

extern "C" int ngp_composite_rays_train_forward_ex(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                   uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth, float* image,
                                                   int bg_mode, float bg_scalar, const float* bg, const float* nears, const float* fars,
                                                   float* image_out, float* depth_out, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image, NGP_ERR_INVALID,
                "composite_rays_train_forward: NULL tensor");
    Finish fin;
    int rc = make_finish("composite_rays_train_forward", bg_mode, bg_scalar, bg, nears, fars, image_out, depth_out, true, &fin);
    if (rc) return rc;
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(cdiv(N, CT_WAVES)), dim3(CT_WAVES * 64), 0, as_stream(stream), sigmas, rgbs, deltas,
                       rays, M, N, T_thresh, weights_sum, depth, image, fin);
    return check_launch("composite_rays_train_forward");
}

extern "C" int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                                float* image, ngp_stream_t stream) {
    return ngp_composite_rays_train_forward_ex(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image, 0, 0.0f, nullptr,
                                               nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int ngp_composite_rays_train_backward_ex(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                    const float* rgbs, const float* deltas, const int32_t* rays,
                                                    const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                    float T_thresh, float* grad_sigmas, float* grad_rgbs, int bg_mode, float bg_scalar,
                                                    const float* bg, const uint32_t* rows_used, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs, NGP_ERR_INVALID,
                "composite_rays_train_backward: NULL tensor");
    NGP_REQUIRE(grad_weights_sum || bg_mode != 0, NGP_ERR_INVALID, "composite_rays_train_backward: NULL tensor");
    Finish fin;
    int rc = make_finish("composite_rays_train_backward", bg_mode, bg_scalar, bg, nullptr, nullptr, nullptr, nullptr, false, &fin);
    if (rc) return rc;
    if (N == 0) return NGP_OK;
    const uint32_t ray_blocks = cdiv(N, CT_WAVES);
    const uint32_t tail_blocks = rows_used ? 32u : 0u;
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(ray_blocks + tail_blocks), dim3(CT_WAVES * 64), 0, as_stream(stream), grad_weights_sum,
                       grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, fin, rows_used,
                       ray_blocks);
    return check_launch("composite_rays_train_backward");
}

extern "C" int ngp_composite_train_loss_backward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M,
                                                 uint32_t N, float T_thresh, int bg_mode, float bg_scalar, const float* bg, const float* nears,
                                                 const float* fars, const float* target, const float* loss_scale, float* weights_sum,
                                                 float* image_out, float* depth_out, float* loss, float* ray_err, float* grad_sigmas,
                                                 void* grad_out16, void* march_workspace, size_t march_workspace_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(N > 0, NGP_ERR_INVALID, "composite_train_loss_backward: no rays");
    NGP_REQUIRE(march_workspace_bytes >= ngp_march_rays_train_workspace_bytes(N), NGP_ERR_INVALID,
                "composite_train_loss_backward: march_workspace of %zu bytes, needs ngp_march_rays_train_workspace_bytes(%u) = %zu (the group tickets sit at its end)",
                march_workspace_bytes, N, ngp_march_rays_train_workspace_bytes(N));
    // (loss may be NULL: the sum of ray_err is then left to the caller -- ngp_grid_encode_backward_checked_slabs carries it)
    NGP_REQUIRE(sigmas && rgbs && deltas && rays && target && weights_sum && ray_err && grad_sigmas && grad_out16 && march_workspace,
                NGP_ERR_INVALID, "composite_train_loss_backward: NULL tensor");
    NGP_REQUIRE(bg_mode == 1 || bg_mode == 2, NGP_ERR_INVALID, "composite_train_loss_backward: bg_mode must be 1 (scalar) or 2 (per ray)");
    NGP_REQUIRE((uint64_t)N * 3u <= 0xffffffffull, NGP_ERR_INVALID, "composite_train_loss_backward: too many rays");
    Finish fin;
    int rc = make_finish("composite_train_loss_backward", bg_mode, bg_scalar, bg, nears, fars, image_out, depth_out, true, &fin);
    if (rc) return rc;
    uint32_t* ws = reinterpret_cast<uint32_t*>(march_workspace);
    const uint32_t ray_blocks = cdiv(N, CT_WAVES), tail_blocks = 32u;
    hipLaunchKernelGGL(k_composite_train_loss_bwd, dim3(ray_blocks + tail_blocks), dim3(CT_WAVES * 64), 0, as_stream(stream), sigmas, rgbs,
                       deltas, rays, M, N, T_thresh, weights_sum, fin, target, loss_scale, ray_err, ws + 1, loss, grad_sigmas,
                       (half_t*)grad_out16, (const uint32_t*)ws, ray_blocks, ws + march_ws_ticket_word(N));
    return check_launch("composite_train_loss_backward");
}

extern "C" int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                 const float* rgbs, const float* deltas, const int32_t* rays,
                                                 const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                 float T_thresh, float* grad_sigmas, float* grad_rgbs, ngp_stream_t stream) {
    NGP_REQUIRE(grad_weights_sum || N == 0, NGP_ERR_INVALID, "composite_rays_train_backward: NULL tensor");
    return ngp_composite_rays_train_backward_ex(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                                                grad_sigmas, grad_rgbs, 0, 0.0f, nullptr, nullptr, stream);
}

extern "C" int ngp_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                                 const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                 const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                 const float* noises, uint32_t zero_rows, ngp_stream_t stream) {
    (void)nears;  // read but unused by the reference kernel as well (raymarching.cu:741)
    int rc = check_march_args("march_rays", C, H, max_steps);
    if (rc) return rc;
    if ((n_alive == 0 || n_step == 0) && zero_rows == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && (noises || zero_rows), NGP_ERR_INVALID,
                "march_rays: NULL tensor");
    NGP_REQUIRE(zero_rows == 0 || zero_rows >= n_alive * n_step, NGP_ERR_INVALID, "march_rays: zero_rows is smaller than n_alive * n_step");
    const uint32_t lanes = n_alive > 0 ? n_alive : 1u;
    RM_LAUNCH_1D(k_march_rays, lanes, as_stream(stream), n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma,
                 max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises, zero_rows, (const int32_t*)nullptr, 0u, 0u, (uint32_t*)nullptr);
    return check_launch("march_rays");
}

extern "C" int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                              const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                              const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                              const float* noises, ngp_stream_t stream) {
    if (n_alive == 0 || n_step == 0) return NGP_OK;
    NGP_REQUIRE(noises, NGP_ERR_INVALID, "march_rays: NULL tensor");
    return ngp_march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs,
                             dirs, deltas, noises, 0u, stream);
}

extern "C" int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                  const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                                  float* image, ngp_stream_t stream) {
    if (n_alive == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, NGP_ERR_INVALID,
                "composite_rays: NULL tensor");
    RM_LAUNCH_1D(k_composite_rays, n_alive, as_stream(stream), n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas,
                 weights_sum, depth, image, (const int32_t*)nullptr, 0u, 0u);
    return check_launch("composite_rays");
}

extern "C" size_t ngp_coarse_occupancy_bytes(uint32_t C, uint32_t H) { const size_t R = H >> COARSE_SHIFT; return (size_t)C * R * R * R; }

extern "C" int ngp_coarse_occupancy(const uint8_t* grid, uint32_t C, uint32_t H, uint8_t* coarse, ngp_stream_t stream) {
    NGP_REQUIRE(grid && coarse, NGP_ERR_INVALID, "coarse_occupancy: NULL tensor");
    NGP_REQUIRE(C >= 1 && C <= 8 && H >= 16 && (H & (H - 1)) == 0, NGP_ERR_INVALID, "coarse_occupancy: needs 1..8 cascades and a power-of-two grid >= 16 (got %u, %u)", C, H);
    RM_LAUNCH_1D(k_coarse_occupancy, (uint32_t)ngp_coarse_occupancy_bytes(C, H), as_stream(stream), grid, C, H, coarse);
    return check_launch("coarse_occupancy");
}

extern "C" int ngp_cull_rays(const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N, float bound, uint32_t C,
                             uint32_t H, const uint8_t* coarse, int32_t* rays_alive, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && nears && fars && coarse && rays_alive, NGP_ERR_INVALID, "cull_rays: NULL tensor");
    NGP_REQUIRE(C >= 1 && C <= 8 && H >= 16 && (H & (H - 1)) == 0 && bound > 0.0f, NGP_ERR_INVALID, "cull_rays: bad grid description");
    RM_LAUNCH_1D(k_cull_rays, N, as_stream(stream), rays_o, rays_d, nears, fars, N, bound, C, H, coarse, rays_alive);
    return check_launch("cull_rays");
}

extern "C" size_t ngp_compact_rays_workspace_bytes(uint32_t n_alive) { return sizeof(uint32_t) * (size_t)cdiv(n_alive ? n_alive : 1, RM_THREADS); }

extern "C" int ngp_compact_rays(const int32_t* rays_alive, uint32_t n_alive, int32_t* out_alive, int32_t* out_count, void* workspace,
                                ngp_stream_t stream) {
    NGP_REQUIRE(rays_alive && out_alive && out_count && workspace, NGP_ERR_INVALID, "compact_rays: NULL tensor");
    hipStream_t st = as_stream(stream);
    if (n_alive == 0) {
        hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t), st);
        NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "compact_rays: memset failed");
        return NGP_OK;
    }
    uint32_t* ws = reinterpret_cast<uint32_t*>(workspace);
    RM_LAUNCH_1D(k_compact_count, n_alive, st, rays_alive, n_alive, ws, (const int32_t*)nullptr);
    int rc = check_launch("compact_rays(count)");
    if (rc) return rc;
    RM_LAUNCH_1D(k_compact_write, n_alive, st, rays_alive, n_alive, out_alive, out_count, (const uint32_t*)ws, (const int32_t*)nullptr, 0u, 0u, 0u);
    return check_launch("compact_rays(write)");
}

// ---------------------------------------------------------------------------------------------
// On-device inference loop: march_rays / composite_rays / the alive-list compaction of NeRFRenderer.run_cuda's eval
// branch (renderer.py:322-367) with the alive count, the per-iteration n_step = max(min(N // n_alive, 8), 1) and the marched-step total
// kept in a device word pair `state` = {n_alive, steps marched}.  The host launches for an UPPER BOUND of the alive count
// (`alive_bound`, the value it last read back) and may run many iterations between read-backs; every kernel takes the true count from
// `state`, lanes and sample rows beyond it do nothing / are zero-filled up to `rows`.  Same slot layout, same n_step sequence and same
// order-preserving compaction as the host-driven loop: identical results.
// ---------------------------------------------------------------------------------------------
extern "C" int ngp_march_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, const int32_t* rays_alive, const float* rays_t,
                                  const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                  const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                  const float* noises, uint32_t rows, ngp_stream_t stream) {
    return ngp_march_rays_dev_rows(state, alive_bound, n_total, n_step_cap, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
                                   xyzs, dirs, deltas, noises, rows, nullptr, stream);
}

extern "C" int ngp_march_rays_dev_rows(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, const int32_t* rays_alive, const float* rays_t,
                                  const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                  uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs,
                                  float* deltas, const float* noises, uint32_t rows, uint32_t* rows_used, ngp_stream_t stream) {
    (void)nears;
    int rc = check_march_args("march_rays", C, H, max_steps);
    if (rc) return rc;
    NGP_REQUIRE(state && rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas, NGP_ERR_INVALID, "march_rays: NULL tensor");
    NGP_REQUIRE(rows > 0 && n_total > 0, NGP_ERR_INVALID, "march_rays_dev: rows and n_total must be positive");
    NGP_REQUIRE(n_step_cap <= 1024u, NGP_ERR_INVALID, "march_rays_dev: n_step_cap must be in [0, 1024] (0 = the reference's 8)");
    const uint32_t lanes = alive_bound > 0 ? alive_bound : 1u;
    RM_LAUNCH_1D(k_march_rays, lanes, as_stream(stream), alive_bound, 1u, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                 grid, fars, xyzs, dirs, deltas, noises, rows, state, n_total, n_step_cap, rows_used);
    return check_launch("march_rays_dev");
}

extern "C" int ngp_composite_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, float T_thresh, int32_t* rays_alive,
                                      float* rays_t, const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                                      float* depth, float* image, ngp_stream_t stream) {
    if (alive_bound == 0) return NGP_OK;
    NGP_REQUIRE(state && rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, NGP_ERR_INVALID,
                "composite_rays: NULL tensor");
    RM_LAUNCH_1D(k_composite_rays, alive_bound, as_stream(stream), alive_bound, 1u, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas,
                 weights_sum, depth, image, state, n_total, n_step_cap);
    return check_launch("composite_rays_dev");
}

/* compaction + loop bookkeeping: out_alive / out_state {n_alive, steps marched} describe the next iteration; the count becomes 0 once
 * max_steps samples per ray were marched */
extern "C" int ngp_compact_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, uint32_t max_steps, const int32_t* rays_alive,
                                    int32_t* out_alive, int32_t* out_state, void* workspace, ngp_stream_t stream) {
    NGP_REQUIRE(state && rays_alive && out_alive && out_state && workspace, NGP_ERR_INVALID, "compact_rays: NULL tensor");
    hipStream_t st = as_stream(stream);
    const uint32_t lanes = alive_bound > 0 ? alive_bound : 1u;
    uint32_t* ws = reinterpret_cast<uint32_t*>(workspace);
    RM_LAUNCH_1D(k_compact_count, lanes, st, rays_alive, alive_bound, ws, state);
    int rc = check_launch("compact_rays_dev(count)");
    if (rc) return rc;
    RM_LAUNCH_1D(k_compact_write, lanes, st, rays_alive, alive_bound, out_alive, out_state, (const uint32_t*)ws, state, n_total, max_steps, n_step_cap);
    return check_launch("compact_rays_dev(write)");
}
