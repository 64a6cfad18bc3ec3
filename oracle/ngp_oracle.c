/*
 * oracle/ngp_oracle.c -- CPU restatement of torch-ngp's instant-ngp hot path.
 *
 *   TEST INFRASTRUCTURE ONLY.  Nothing under torch-ngp_amd/ may link, import or call this file.
 *   It exists to CHECK the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
 *
 * Every function cites the reference lines whose behaviour it restates (paths relative to
 * /root/reference).  The reference kernels are CUDA-only and cannot be built here (no nvcc; ffmlp
 * additionally needs the un-vendored CUTLASS submodule), so this file is a from-the-algorithm
 * restatement in scalar C, executed in the order a sequential run of the reference kernels would
 * produce (ray order, level order), which is one of the schedules the reference's atomics allow.
 *
 * Pinning status (see DESIGN.md "oracle pinning"):
 *   - SH basis:      pinned (reference testing/test_shencoder.py torch implementation -> tests/golden,
 *                    plus scipy.special.sph_harm for all 64 components).
 *   - FFMLP:         pinned (reference testing/test_ffmlp.py `MLP` -> tests/golden); lives in
 *                    oracle/__init__.py (numpy), not in this file.
 *   - grid offsets:  pinned (reference gridencoder/grid.py ctor run with a stub backend).
 *   - grid encode kernels, ray marching, compositing, packbits, morton: PARITY UNPINNED by the
 *     reference (it ships no fixture or assertion for them); pinned only by the known-answer tests
 *     derived from SURVEY.md section 8(c).
 *
 * Floating-point contract shared with the HIP kernels (compile BOTH with -ffp-contract=off):
 *   every multiply-add that decides an integer (cell index, step count) is written as an explicit
 *   fmaf() here and in the kernels, so the two sides round identically; divisions and sqrt are IEEE.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <stdlib.h>

#include "sh_table.inc"

/* =====================================================================================
 *  grid encoder        (gridencoder/src/gridencoder.cu)
 * ===================================================================================== */

/* hash primes: gridencoder.cu:50-63 (fast_hash) */
static const uint32_t ORC_PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};

/* Per-level scale / resolution, gridencoder.cu:137-139:
 *     scale = exp2f(level * S) * H - 1 ;  resolution = ceil(scale) + 1
 * The device exp2f of the reference is not reproducible off-device, so the framework DEFINES the
 * table by this host recipe (fp32 product level*S, correctly rounded exp2, fused *H-1) and feeds the
 * same table to kernel and oracle. */
void orc_grid_level_table(uint32_t L, float S, uint32_t H, float *scale, uint32_t *resolution) {
    for (uint32_t l = 0; l < L; l++) {
        float a = (float)l * S;
        float e = (float)exp2((double)a);
        float sc = fmaf(e, (float)H, -1.0f);
        scale[l] = sc;
        resolution[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

/* gridencoder.cu:66-84 (get_grid_index) without the "*C + ch" tail */
static uint32_t orc_grid_index(uint32_t gridtype, int align_corners, uint32_t D, uint32_t hashmap_size,
                               uint32_t resolution, const uint32_t *pg) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pg[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1u);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
        for (uint32_t d = 0; d < D; d++) index ^= pg[d] * ORC_PRIMES[d];
    }
    return index % hashmap_size;
}

static float orc_smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
static float orc_smoothstep_d(float v) { return 6.0f * v * (1.0f - v); }

/* locate a point in one level: gridencoder.cu:146-159.  returns 0 if out of [0,1]^D (cu:110-135) */
static int orc_grid_locate(const float *x, uint32_t D, float scale, int align_corners, uint32_t interp,
                           float *frac, float *deriv, uint32_t *cell) {
    for (uint32_t d = 0; d < D; d++)
        if (x[d] < 0.0f || x[d] > 1.0f) return 0;
    for (uint32_t d = 0; d < D; d++) {
        float p = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1) {
            deriv[d] = orc_smoothstep_d(p);
            p = orc_smoothstep(p);
        } else {
            deriv[d] = 1.0f;
        }
        frac[d] = p;
    }
    return 1;
}

/* Corner indices (entry index inside the level, before *C), [L, B, 2^D] uint32; 0xFFFFFFFF for
 * out-of-range points.  Diagnostic used for the "bit-exact grid indexing" parity test. */
void orc_grid_corner_indices(const float *inputs, const int32_t *offsets, uint32_t *out, uint32_t B,
                             uint32_t D, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             int align_corners) {
    float scale[64];
    uint32_t res[64];
    orc_grid_level_table(L, S, H, scale, res);
    const uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            float frac[8], der[8];
            uint32_t cell[8], pg[8];
            uint32_t *o = out + ((size_t)l * B + b) * nc;
            if (!orc_grid_locate(inputs + (size_t)b * D, D, scale[l], align_corners, 0, frac, der, cell)) {
                for (uint32_t c = 0; c < nc; c++) o[c] = 0xFFFFFFFFu;
                continue;
            }
            for (uint32_t c = 0; c < nc; c++) {
                for (uint32_t d = 0; d < D; d++) pg[d] = cell[d] + ((c >> d) & 1u);
                o[c] = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg);
            }
        }
    }
}

/* forward: gridencoder.cu:87-245 (kernel_grid).  embeddings are fp32 here; the caller rounds them to
 * fp16 first when it wants the autocast path.  outputs [L,B,C] fp32 (accumulated in double),
 * dy_dx [B,L,D,C] or NULL. */
void orc_grid_forward(const float *inputs, const float *emb, const int32_t *offsets, float *outputs,
                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float *dy_dx,
                      uint32_t gridtype, int align_corners, uint32_t interp) {
    float scale[64];
    uint32_t res[64];
    orc_grid_level_table(L, S, H, scale, res);
    const uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        const float *grid = emb + (size_t)(uint32_t)offsets[l] * C;
        uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            float frac[8], der[8];
            uint32_t cell[8], pg[8];
            float *o = outputs + ((size_t)l * B + b) * C;
            float *g = dy_dx ? dy_dx + ((size_t)b * L + l) * D * C : NULL;
            if (!orc_grid_locate(inputs + (size_t)b * D, D, scale[l], align_corners, interp, frac, der, cell)) {
                for (uint32_t c = 0; c < C; c++) o[c] = 0.0f;
                if (g) memset(g, 0, sizeof(float) * D * C);
                continue;
            }
            double acc[8] = {0};
            for (uint32_t k = 0; k < nc; k++) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (((k >> d) & 1u) == 0) { w *= 1.0f - frac[d]; pg[d] = cell[d]; }
                    else { w *= frac[d]; pg[d] = cell[d] + 1u; }
                }
                uint32_t idx = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                for (uint32_t c = 0; c < C; c++) acc[c] += (double)w * (double)grid[idx + c];
            }
            for (uint32_t c = 0; c < C; c++) o[c] = (float)acc[c];
            if (!g) continue;
            /* cu:201-244 */
            for (uint32_t gd = 0; gd < D; gd++) {
                double ga[8] = {0};
                for (uint32_t k = 0; k < (nc >> 1); k++) {
                    float w = scale[l];
                    for (uint32_t nd = 0; nd + 1 < D; nd++) {
                        uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if (((k >> nd) & 1u) == 0) { w *= 1.0f - frac[d]; pg[d] = cell[d]; }
                        else { w *= frac[d]; pg[d] = cell[d] + 1u; }
                    }
                    pg[gd] = cell[gd];
                    uint32_t il = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                    pg[gd] = cell[gd] + 1u;
                    uint32_t ir = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                    for (uint32_t c = 0; c < C; c++)
                        ga[c] += (double)w * ((double)grid[ir + c] - (double)grid[il + c]) * (double)der[gd];
                }
                for (uint32_t c = 0; c < C; c++) g[gd * C + c] = (float)ga[c];
            }
        }
    }
}

/* backward: gridencoder.cu:248-340 (scatter-add into grad_embeddings, double accumulators so the
 * result is the order-independent exact sum) and cu:343-369 (grad_inputs from dy_dx).
 * grad is [L,B,C]; grad_emb must be zeroed by the caller (double, size sO*C). */
void orc_grid_backward(const float *grad, const float *inputs, const int32_t *offsets, double *grad_emb,
                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                       const float *dy_dx, float *grad_inputs, uint32_t gridtype, int align_corners,
                       uint32_t interp) {
    float scale[64];
    uint32_t res[64];
    orc_grid_level_table(L, S, H, scale, res);
    const uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        double *gg = grad_emb + (size_t)(uint32_t)offsets[l] * C;
        uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            float frac[8], der[8];
            uint32_t cell[8], pg[8];
            if (!orc_grid_locate(inputs + (size_t)b * D, D, scale[l], align_corners, interp, frac, der, cell))
                continue;
            const float *gr = grad + ((size_t)l * B + b) * C;
            for (uint32_t k = 0; k < nc; k++) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (((k >> d) & 1u) == 0) { w *= 1.0f - frac[d]; pg[d] = cell[d]; }
                    else { w *= frac[d]; pg[d] = cell[d] + 1u; }
                }
                uint32_t idx = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                for (uint32_t c = 0; c < C; c++) gg[idx + c] += (double)w * (double)gr[c];
            }
        }
    }
    if (dy_dx && grad_inputs) {
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < D; d++) {
                double r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t c = 0; c < C; c++)
                        r += (double)grad[((size_t)l * B + b) * C + c] *
                             (double)dy_dx[(((size_t)b * L + l) * D + d) * C + c];
                grad_inputs[(size_t)b * D + d] = (float)r;
            }
    }
}

/* total-variation gradient: gridencoder.cu:506-610 (kernel_grad_tv); adds into grad (double). */
void orc_grid_grad_tv(const float *inputs, const float *emb, double *grad, const int32_t *offsets,
                      float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                      uint32_t gridtype, int align_corners) {
    float scale[64];
    uint32_t res[64];
    orc_grid_level_table(L, S, H, scale, res);
    for (uint32_t l = 0; l < L; l++) {
        const float *grid = emb + (size_t)(uint32_t)offsets[l] * C;
        double *gg = grad + (size_t)(uint32_t)offsets[l] * C;
        uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0.0f || x[d] > 1.0f) oob = 1;
            if (oob) continue;
            uint32_t pg[8];
            for (uint32_t d = 0; d < D; d++)
                pg[d] = (uint32_t)floorf(fmaf(x[d], scale[l], align_corners ? 0.0f : 0.5f));
            double r[8] = {0}, id[8] = {0};
            uint32_t idx = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
            double w = (double)weight / (2.0 * D);
            for (uint32_t d = 0; d < D; d++) {
                uint32_t cur = pg[d];
                if (cur < res[l]) {
                    pg[d] = cur + 1u;
                    uint32_t ir = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                    for (uint32_t c = 0; c < C; c++) {
                        double gv = (double)grid[idx + c] - (double)grid[ir + c];
                        r[c] += gv; id[c] += gv * gv;
                    }
                }
                if (cur > 0) {
                    pg[d] = cur - 1u;
                    uint32_t il = orc_grid_index(gridtype, align_corners, D, hs, res[l], pg) * C;
                    for (uint32_t c = 0; c < C; c++) {
                        double gv = (double)grid[idx + c] - (double)grid[il + c];
                        r[c] += gv; id[c] += gv * gv;
                    }
                }
                pg[d] = cur;
            }
            for (uint32_t c = 0; c < C; c++) gg[idx + c] += w * r[c] / sqrt(id[c] + 1e-9);
        }
    }
}

/* =====================================================================================
 *  spherical harmonics   (shencoder/src/shencoder.cu:27-382)
 * ===================================================================================== */
/* outputs [B, bands^2]; dy_dx [B, 3, bands^2] or NULL */
void orc_sh_forward(const float *inputs, float *outputs, uint32_t B, uint32_t bands, float *dy_dx) {
    const uint32_t n = bands * bands;
    double Y[64], dx[64], dy[64], dz[64];
    for (uint32_t b = 0; b < B; b++) {
        const float *p = inputs + (size_t)b * 3;
        orc_sh_eval(p[0], p[1], p[2], Y, dy_dx ? dx : NULL, dy, dz);
        for (uint32_t i = 0; i < n; i++) outputs[(size_t)b * n + i] = (float)Y[i];
        if (dy_dx) {
            float *g = dy_dx + (size_t)b * 3 * n;
            for (uint32_t i = 0; i < n; i++) { g[i] = (float)dx[i]; g[n + i] = (float)dy[i]; g[2 * n + i] = (float)dz[i]; }
        }
    }
}
/* shencoder.cu:358-382: grad_inputs[b,d] += sum_ch grad[b,ch]*dy_dx[b,d,ch] */
void orc_sh_backward(const float *grad, uint32_t B, uint32_t bands, const float *dy_dx, float *grad_inputs) {
    const uint32_t n = bands * bands;
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < 3; d++) {
            double r = grad_inputs[(size_t)b * 3 + d];
            for (uint32_t i = 0; i < n; i++)
                r += (double)grad[(size_t)b * n + i] * (double)dy_dx[((size_t)b * 3 + d) * n + i];
            grad_inputs[(size_t)b * 3 + d] = (float)r;
        }
}

/* =====================================================================================
 *  ray marching          (raymarching/src/raymarching.cu)
 * ===================================================================================== */
static float orc_clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
static float orc_signf(float x) { return copysignf(1.0f, x); }

/* raymarching.cu:92-145 */
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                            float min_near, float *nears, float *fars) {
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + (size_t)n * 3, *d = rays_d + (size_t)n * 3;
        float rdx = 1.0f / d[0], rdy = 1.0f / d[1], rdz = 1.0f / d[2];
        float near = (aabb[0] - o[0]) * rdx, far = (aabb[3] - o[0]) * rdx, t;
        if (near > far) { t = near; near = far; far = t; }
        float ny = (aabb[1] - o[1]) * rdy, fy = (aabb[4] - o[1]) * rdy;
        if (ny > fy) { t = ny; ny = fy; fy = t; }
        if (near > fy || ny > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (aabb[2] - o[2]) * rdz, fz = (aabb[5] - o[2]) * rdz;
        if (nz > fz) { t = nz; nz = fz; fz = t; }
        if (near > fz || nz > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (nz > near) near = nz;
        if (fz < far) far = fz;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:163-198.  Computed in double, narrowed at the end (fp tolerance test only). */
void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + (size_t)n * 3, *d = rays_d + (size_t)n * 3;
        double A = (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2];
        double Bh = (double)o[0] * d[0] + (double)o[1] * d[1] + (double)o[2] * d[2];
        double Cc = (double)o[0] * o[0] + (double)o[1] * o[1] + (double)o[2] * o[2] - (double)radius * radius;
        double t = (-Bh + sqrt(Bh * Bh - A * Cc)) / A;
        double x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
        double theta = atan2(sqrt(x * x + z * z), y);
        double phi = atan2(z, x);
        coords[(size_t)n * 2] = (float)(2.0 * theta / M_PI - 1.0);
        coords[(size_t)n * 2 + 1] = (float)(phi / M_PI);
    }
}

/* raymarching.cu:56-81 */
static uint32_t orc_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static uint32_t orc_morton3D_1(uint32_t x, uint32_t y, uint32_t z) {
    return orc_expand_bits(x) | (orc_expand_bits(y) << 1) | (orc_expand_bits(z) << 2);
}
static uint32_t orc_compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}
/* raymarching.cu:214-232 / 237-260 */
void orc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)orc_morton3D_1((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
void orc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)orc_compact_bits((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)orc_compact_bits((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)orc_compact_bits((uint32_t)(ind >> 2));
    }
}
/* raymarching.cu:268-289 ; N = number of output bytes */
void orc_packbits(const float *grid, uint32_t N, float thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* One marching state shared by the three marchers (raymarching.cu:312-480 and 701-805). */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, bound, dt_gamma, dt_min, dt_max, Hf, H3f;
    uint32_t C, H;
    const uint8_t *grid;
} orc_ray_t;

static void orc_ray_init(orc_ray_t *r, const float *o, const float *d, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid) {
    r->ox = o[0]; r->oy = o[1]; r->oz = o[2];
    r->dx = d[0]; r->dy = d[1]; r->dz = d[2];
    r->rdx = 1.0f / d[0]; r->rdy = 1.0f / d[1]; r->rdz = 1.0f / d[2];
    r->Hf = (float)H;
    r->rH = 1.0f / r->Hf;
    r->H3f = (float)(H * H * H); /* cu:339: float H3 = H*H*H (uint32 product converted once) */
    r->bound = bound;
    r->dt_gamma = dt_gamma;
    const float SQRT3 = 1.7320508075688772f;
    r->dt_min = 2.0f * SQRT3 / (float)max_steps;
    r->dt_max = 2.0f * SQRT3 * (float)(1u << (C - 1)) / r->Hf;
    r->C = C; r->H = H; r->grid = grid;
}

static int orc_mip_exponent(float mx, uint32_t C) {
    int e;
    frexpf(mx, &e);
    /* cu:46: fminf(max_cascade-1, fmaxf(0, exponent)) */
    float v = fminf((float)C - 1.0f, fmaxf(0.0f, (float)e));
    return (int)v;
}

/* evaluates occupancy at parameter t; returns dt; writes clamped point, and (if empty) the t to skip to */
static int orc_ray_probe(const orc_ray_t *r, float t, float *px, float *py, float *pz, float *dt_out,
                         float *tt_out) {
    const float x = orc_clampf(fmaf(t, r->dx, r->ox), -r->bound, r->bound);
    const float y = orc_clampf(fmaf(t, r->dy, r->oy), -r->bound, r->bound);
    const float z = orc_clampf(fmaf(t, r->dz, r->oz), -r->bound, r->bound);
    const float dt = orc_clampf(t * r->dt_gamma, r->dt_min, r->dt_max);
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int lp = orc_mip_exponent(mx, r->C);
    int ld = orc_mip_exponent((dt * r->Hf) * 0.5f, r->C);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), r->bound);
    const float mip_rbound = 1.0f / mip_bound;
    /* cu:374-376: 0.5 * (x*mip_rbound + 1) * H evaluated in double then narrowed; the double product of
     * a float and a small integer is exact, so it equals the fp32 product (0.5f*u)*H rounded once. */
    const float Hm1 = (float)(r->H - 1);
    const int nx = (int)orc_clampf((0.5f * fmaf(x, mip_rbound, 1.0f)) * r->Hf, 0.0f, Hm1);
    const int ny = (int)orc_clampf((0.5f * fmaf(y, mip_rbound, 1.0f)) * r->Hf, 0.0f, Hm1);
    const int nz = (int)orc_clampf((0.5f * fmaf(z, mip_rbound, 1.0f)) * r->Hf, 0.0f, Hm1);
    /* cu:378: uint32 index = level*H3 + morton  (float sum, then converted) */
    const uint32_t index = (uint32_t)((float)level * r->H3f + (float)orc_morton3D_1((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (r->grid[index / 8] & (1u << (index % 8))) != 0;
    *px = x; *py = y; *pz = z; *dt_out = dt;
    if (!occ) {
        /* cu:390-394 */
        const float tx = (fmaf(((float)nx + 0.5f + 0.5f * orc_signf(r->dx)) * r->rH * 2.0f - 1.0f, mip_bound, -x)) * r->rdx;
        const float ty = (fmaf(((float)ny + 0.5f + 0.5f * orc_signf(r->dy)) * r->rH * 2.0f - 1.0f, mip_bound, -y)) * r->rdy;
        const float tz = (fmaf(((float)nz + 0.5f + 0.5f * orc_signf(r->dz)) * r->rH * 2.0f - 1.0f, mip_bound, -z)) * r->rdz;
        *tt_out = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

static float orc_ray_skip(const orc_ray_t *r, float t, float tt) {
    do {
        t += orc_clampf(t * r->dt_gamma, r->dt_min, r->dt_max);
    } while (t < tt);
    return t;
}

/* raymarching.cu:312-480, executed ray by ray in index order (so rays[n] = (n, offset, count) with
 * offsets an exclusive prefix sum -- the schedule a sequential run of the reference yields).
 * counter[0] += sum of counts, counter[1] += N.  Rays whose samples would exceed M are recorded
 * in `rays` but write nothing (cu:405-416). */
void orc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                          int32_t *rays, int32_t *counter, const float *noises) {
    for (uint32_t n = 0; n < N; n++) {
        orc_ray_t r;
        orc_ray_init(&r, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float near = nears[n], far = fars[n];
        /* cu:348-351: t0 = near + clamp(near*dt_gamma)*noise, one fused multiply-add */
        const float t0 = fmaf(orc_clampf(near * dt_gamma, r.dt_min, r.dt_max), noises[n], near);
        float t = t0, px, py, pz, dt, tt;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            if (orc_ray_probe(&r, t, &px, &py, &pz, &dt, &tt)) { num_steps++; t += dt; }
            else t = orc_ray_skip(&r, t, tt);
        }
        uint32_t point_index = (uint32_t)counter[0];
        counter[0] += (int32_t)num_steps;
        uint32_t ray_index = (uint32_t)counter[1];
        counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;
        float *xo = xyzs + (size_t)point_index * 3, *dd = dirs + (size_t)point_index * 3, *de = deltas + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        float last_t = t;
        while (t < far && step < num_steps) {
            if (orc_ray_probe(&r, t, &px, &py, &pz, &dt, &tt)) {
                xo[0] = px; xo[1] = py; xo[2] = pz;
                dd[0] = r.dx; dd[1] = r.dy; dd[2] = r.dz;
                t += dt;
                de[0] = dt; de[1] = t - last_t;
                last_t = t;
                xo += 3; dd += 3; de += 2; step++;
            } else t = orc_ray_skip(&r, t, tt);
        }
    }
}

/* raymarching.cu:701-805 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                    float *xyzs, float *dirs, float *deltas, const float *noises) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        orc_ray_t r;
        orc_ray_init(&r, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
        float *xo = xyzs + (size_t)n * n_step * 3, *dd = dirs + (size_t)n * n_step * 3, *de = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;
        t = fmaf(orc_clampf(t * dt_gamma, r.dt_min, r.dt_max), noises[n], t); /* cu:741 */
        float last_t = t, px, py, pz, dt, tt;
        uint32_t step = 0;
        while (t < far && step < n_step) {
            if (orc_ray_probe(&r, t, &px, &py, &pz, &dt, &tt)) {
                xo[0] = px; xo[1] = py; xo[2] = pz;
                dd[0] = r.dx; dd[1] = r.dy; dd[2] = r.dz;
                t += dt;
                de[0] = dt; de[1] = t - last_t;
                last_t = t;
                xo += 3; dd += 3; de += 2; step++;
            } else t = orc_ray_skip(&r, t, tt);
        }
    }
}

/* raymarching.cu:501-577 */
void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                      float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num = (uint32_t)rays[n * 3 + 2];
        if (num == 0 || offset + num > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3, *de = deltas + (size_t)offset * 2;
        double T = 1.0, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t s = 0; s < num; s++) {
            double alpha = 1.0 - exp(-(double)sg[s] * (double)de[s * 2]);
            double w = alpha * T;
            r += w * rg[s * 3]; g += w * rg[s * 3 + 1]; b += w * rg[s * 3 + 2];
            t += de[s * 2 + 1];
            d += w * t;
            ws += w;
            T *= 1.0 - alpha;
            if (T < (double)T_thresh) break;
        }
        weights_sum[index] = (float)ws; depth[index] = (float)d;
        image[index * 3] = (float)r; image[index * 3 + 1] = (float)g; image[index * 3 + 2] = (float)b;
    }
}

/* raymarching.cu:602-682 ; grad_sigmas/grad_rgbs zeroed by the caller */
void orc_composite_rays_train_backward(const float *grad_ws, const float *grad_image, const float *sigmas,
                                       const float *rgbs, const float *deltas, const int32_t *rays,
                                       const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                       float T_thresh, float *grad_sigmas, float *grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num = (uint32_t)rays[n * 3 + 2];
        if (num == 0 || offset + num > M) continue;
        const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3, *de = deltas + (size_t)offset * 2;
        const float *gi = grad_image + (size_t)index * 3;
        const double gw = grad_ws[index];
        const double rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2], wsf = weights_sum[index];
        double T = 1.0, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t s = 0; s < num; s++) {
            double alpha = 1.0 - exp(-(double)sg[s] * (double)de[s * 2]);
            double w = alpha * T;
            r += w * rg[s * 3]; g += w * rg[s * 3 + 1]; b += w * rg[s * 3 + 2];
            ws += w;
            T *= 1.0 - alpha;
            grad_rgbs[(size_t)(offset + s) * 3 + 0] = (float)(gi[0] * w);
            grad_rgbs[(size_t)(offset + s) * 3 + 1] = (float)(gi[1] * w);
            grad_rgbs[(size_t)(offset + s) * 3 + 2] = (float)(gi[2] * w);
            grad_sigmas[offset + s] = (float)((double)de[s * 2] * (gi[0] * (T * rg[s * 3] - (rf - r)) +
                                                                  gi[1] * (T * rg[s * 3 + 1] - (gf - g)) +
                                                                  gi[2] * (T * rg[s * 3 + 2] - (bf - b)) +
                                                                  gw * (1.0 - wsf)));
            if (T < (double)T_thresh) break;
        }
    }
}

/* raymarching.cu:819-905 (in place on weights_sum/depth/image/rays_alive/rays_t) */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                        float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        const float *sg = sigmas + (size_t)n * n_step, *rg = rgbs + (size_t)n * n_step * 3, *de = deltas + (size_t)n * n_step * 2;
        double t = rays_t[index], ws = weights_sum[index], d = depth[index];
        double r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (de[step * 2] == 0.0f) break;
            double alpha = 1.0 - exp(-(double)sg[step] * (double)de[step * 2]);
            double T = 1.0 - ws;
            double w = alpha * T;
            ws += w;
            t += de[step * 2 + 1];
            d += w * t;
            r += w * rg[step * 3]; g += w * rg[step * 3 + 1]; b += w * rg[step * 3 + 2];
            if (T < (double)T_thresh) break;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = (float)t;
        weights_sum[index] = (float)ws; depth[index] = (float)d;
        image[index * 3] = (float)r; image[index * 3 + 1] = (float)g; image[index * 3 + 2] = (float)b;
    }
}
