// Fused optimizer + loss-scaling step for the instant-ngp parameter set.
//
// The reference trains with torch.optim.Adam + torch.cuda.amp.GradScaler (main_nerf.py:132, nerf/utils.py:393,557-560).  On the
// 12.24 M-entry hash table that is, per iteration: zero a fp16 gradient table, scatter into it, cast it to fp32 (autograd), sweep
// it for non-finite values, run Adam (read p, g, m, v; write p, m, v and the unscaled g), and cast the fp32 table back to fp16 for
// the next forward -- six table-sized passes.  Here the gradient stays in the fp16 buffer the scatter kernel wrote:
//   k_check_finite : one read of the fp16 gradients -> found_inf flag
//   k_adam         : g (fp16, still loss-scaled) * inv_scale -> Adam moments and fp32 master weights, plus the fp16 shadow copy the
//                    next forward reads, and the gradient buffer is zeroed in the same sweep (nothing else has to touch it)
//   k_update_scale : GradScaler's dynamic loss scale (grow after `growth_interval` clean steps, back off on overflow) and the
//                    Adam step counter, on device -- the whole step is free of host synchronisation and graph-capturable.
// Update rule = PyTorch's Adam (amsgrad = False, weight_decay = 0, maximize = False), restated from its documented formula:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// A step with a non-finite gradient anywhere is skipped as a whole (moments, weights and t untouched), as GradScaler.step does; with more
// than 8 tensors the caller runs the phases separately (check every chunk, then update every chunk, then commit once) so that "as a
// whole" holds across chunks.
// Optional, in the same sweep: the exponential moving average of the parameters the reference Trainer keeps with torch_ema
// (nerf/utils.py:388-391,760-761,891-892): shadow -= (1 - decay) * (shadow - param), computed on the UPDATED parameter -- one extra
// fp32 read-modify-write stream instead of a separate pass over every parameter.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace ngp {

// adam_element on components of vector registers (a vector element does not bind to a reference)
#define NGP_ADAM_EL(G, M, V, P)                                                    \
    {                                                                              \
        float m_ = (M), v_ = (V), p_ = (P);                                        \
        adam_element((G), m_, v_, p_, beta1, beta2, eps, step_size, bc2_sqrt);     \
        (M) = m_; (V) = v_; (P) = p_;                                              \
    }

constexpr int OPT_THREADS = 256;
constexpr int OPT_MAX_TENSORS = 8;

struct OptTensors {
    int count;
    uint64_t n[OPT_MAX_TENSORS];
    float* p[OPT_MAX_TENSORS];
    float* m[OPT_MAX_TENSORS];
    float* v[OPT_MAX_TENSORS];
    void* g[OPT_MAX_TENSORS];       // fp16 or fp32 gradient (g_is_half)
    half_t* p16[OPT_MAX_TENSORS];   // optional fp16 shadow of the weights
    int g_is_half[OPT_MAX_TENSORS];
    float lr[OPT_MAX_TENSORS];
    float* ema[OPT_MAX_TENSORS];    // optional EMA shadow (fp32), updated when ema_omd > 0
};

// state[0] = loss scale, state[1] = growth tracker, state[2] = found_inf (0/1), state[3] = Adam step count t, state[4] = lr multiplier
__global__ __launch_bounds__(OPT_THREADS) void k_check_finite(OptTensors ts, float* __restrict__ state) {
    bool bad = false;
    for (int k = 0; k < ts.count; k++) {
        const uint64_t n = ts.n[k];
        if (ts.g_is_half[k] & 1) {
            const half8_t* g = reinterpret_cast<const half8_t*>(ts.g[k]);
            const uint64_t n8 = n / 8;
            for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * OPT_THREADS) {
                const half8_t x = g[i];
#pragma unroll
                for (int j = 0; j < 8; j++) bad = bad || !__builtin_isfinite((float)x[j]);
            }
            const half_t* gt = reinterpret_cast<const half_t*>(ts.g[k]);
            for (uint64_t i = n8 * 8 + (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * OPT_THREADS)
                bad = bad || !__builtin_isfinite((float)gt[i]);
        } else {
            const float* g = reinterpret_cast<const float*>(ts.g[k]);
            for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * OPT_THREADS)
                bad = bad || !__builtin_isfinite(g[i]);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0f;  // benign race: every writer stores the same value
}

// (Folding k_update_scale into this kernel behind a last-workgroup ticket was measured and dropped: the kernel took 74 instead of 62 us --
// presumably the 2048 workgroups, which finish together, queueing on ONE device-scope atomic -- for 4 us of launch saved.)
__global__ __launch_bounds__(OPT_THREADS) void k_adam(OptTensors ts, const float* __restrict__ state, float beta1, float beta2, float eps,
                                                      float grad_mult, float ema_omd) {
    // A loss scale that has underflowed (a long run of overflowing steps halves it to a denormal, then to 0): GradScaler unscales BEFORE it
    // checks, so its 1 / scale = inf turns every gradient into inf or NaN and the step is skipped -- for good, the run is dead, but the weights
    // stay what they were.  Checking the SCALED gradient (the producers do) would let a zero gradient through and 0 * inf = NaN into every
    // parameter (seen in a 200 000-step soak of the bench workload, tools/soak_train.py): the unusable scale is an overflow of its own
    // (adam_consts, common.h).
    const AdamConsts ac = adam_consts(state, beta1, beta2, grad_mult);
    const bool skip = ac.skip;
    const float inv_scale = ac.inv_scale, bc1 = ac.bc1, bc2_sqrt = ac.bc2_sqrt, lr_mult = ac.lr_mult;
    for (int k = 0; k < ts.count; k++) {
        const uint64_t n = ts.n[k];
        const float step_size = ts.lr[k] * lr_mult / bc1;
        float* __restrict__ p = ts.p[k];
        float* __restrict__ m = ts.m[k];
        float* __restrict__ v = ts.v[k];
        half_t* __restrict__ p16 = ts.p16[k];
        float* __restrict__ ema = ema_omd > 0.0f ? ts.ema[k] : nullptr;
        const bool gh = (ts.g_is_half[k] & 1) != 0;
        // bit 1: the gradient's producer OVERWRITES the whole buffer every step (grid backward in overwrite mode): nothing to zero here,
        // and a skipped step has nothing to do for this tensor at all (unless it keeps an average)
        const bool keep = (ts.g_is_half[k] & 2) != 0;
        if (keep && skip && !ema) continue;
        half_t* g16 = reinterpret_cast<half_t*>(ts.g[k]);
        float* g32 = reinterpret_cast<float*>(ts.g[k]);
        // Fast path of the training configuration (fp16 gradient + fp16 shadow, no EMA, length a multiple of 4, 16-byte aligned streams):
        // four elements per lane and iteration, 16-byte accesses (8-byte global accesses run at 0.5-0.7x the 16-byte rate on this chip).
        // Same arithmetic per element as the general loop below; -7 us per iteration in a same-box A/B.
        const bool wide = gh && p16 && !ema && (n & 3u) == 0 &&
                          ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0 &&
                          ((reinterpret_cast<uintptr_t>(g16) | reinterpret_cast<uintptr_t>(p16)) & 7u) == 0;
#ifndef NGP_ADAM_UNROLL
#define NGP_ADAM_UNROLL 1
#endif
        if (wide && NGP_ADAM_UNROLL == 2) {
            // two independent 4-element groups per lane and trip: twice the loads in flight before the first dependent instruction
            const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * OPT_THREADS;
            for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += 2 * stride) {
                const uint64_t j = i + stride;
                const bool two = j < n4;
                const uint64_t jj = two ? j : i;
                const half4_t x0 = reinterpret_cast<half4_t*>(g16)[i], x1 = reinterpret_cast<half4_t*>(g16)[jj];
                float4_t pm0 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(m) + i), pv0 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(v) + i),
                         pp0 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(p) + i);
                float4_t pm1 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(m) + jj), pv1 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(v) + jj),
                         pp1 = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(p) + jj);
                if (!keep) {
                    reinterpret_cast<half4_t*>(g16)[i] = half4_t{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                    if (two) reinterpret_cast<half4_t*>(g16)[j] = half4_t{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                }
                if (skip) continue;
                half4_t ph0, ph1;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float g0 = (float)x0[c] * inv_scale, g1 = (float)x1[c] * inv_scale;
                    NGP_ADAM_EL(g0, pm0[c], pv0[c], pp0[c]);
                    ph0[c] = (half_t)pp0[c];
                    NGP_ADAM_EL(g1, pm1[c], pv1[c], pp1[c]);
                    ph1[c] = (half_t)pp1[c];
                }
                __builtin_nontemporal_store(pm0, reinterpret_cast<float4_t*>(m) + i);
                __builtin_nontemporal_store(pv0, reinterpret_cast<float4_t*>(v) + i);
                __builtin_nontemporal_store(pp0, reinterpret_cast<float4_t*>(p) + i);
                reinterpret_cast<half4_t*>(p16)[i] = ph0;
                if (two) {
                    __builtin_nontemporal_store(pm1, reinterpret_cast<float4_t*>(m) + j);
                    __builtin_nontemporal_store(pv1, reinterpret_cast<float4_t*>(v) + j);
                    __builtin_nontemporal_store(pp1, reinterpret_cast<float4_t*>(p) + j);
                    reinterpret_cast<half4_t*>(p16)[j] = ph1;
                }
            }
            continue;
        }
        if (wide) {
            const uint64_t n4 = n / 4;
            for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * OPT_THREADS) {
                const half4_t x = reinterpret_cast<half4_t*>(g16)[i];
                if (!keep) reinterpret_cast<half4_t*>(g16)[i] = half4_t{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                if (skip) continue;
                float4_t pm = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(m) + i), pv = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(v) + i),
                         pp = __builtin_nontemporal_load(reinterpret_cast<float4_t*>(p) + i);
                half4_t ph;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float g = (float)x[c] * inv_scale;
                    NGP_ADAM_EL(g, pm[c], pv[c], pp[c]);
                    ph[c] = (half_t)pp[c];
                }
                __builtin_nontemporal_store(pm, reinterpret_cast<float4_t*>(m) + i);
                __builtin_nontemporal_store(pv, reinterpret_cast<float4_t*>(v) + i);
                __builtin_nontemporal_store(pp, reinterpret_cast<float4_t*>(p) + i);
                reinterpret_cast<half4_t*>(p16)[i] = ph;
            }
            continue;
        }
        // two elements per lane and iteration (all our tensors have an even length; a trailing odd element is handled below)
        const uint64_t n2 = n / 2;
        for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * OPT_THREADS) {
            float g0, g1;
            if (gh) {
                const half2_t x = reinterpret_cast<half2_t*>(g16)[i];
                g0 = (float)x.x; g1 = (float)x.y;
                if (!keep) reinterpret_cast<half2_t*>(g16)[i] = half2_t{(half_t)0.0f, (half_t)0.0f};
            } else {
                const float2_t x = reinterpret_cast<float2_t*>(g32)[i];
                g0 = x.x; g1 = x.y;
                if (!keep) reinterpret_cast<float2_t*>(g32)[i] = float2_t{0.0f, 0.0f};
            }
            if (skip) {
                if (ema) {  // the average moves towards the (unchanged) parameters as torch_ema's update() would
                    const float2_t pp = reinterpret_cast<float2_t*>(p)[i];
                    float2_t e = reinterpret_cast<float2_t*>(ema)[i];
                    e.x -= ema_omd * (e.x - pp.x); e.y -= ema_omd * (e.y - pp.y);
                    reinterpret_cast<float2_t*>(ema)[i] = e;
                }
                continue;
            }
            g0 *= inv_scale; g1 *= inv_scale;
            // the moment / master-weight streams pass exactly once per step: non-temporal, they do not displace the L2 (measured with the
            // activation stores of the network forward: -11 us per iteration on one box)
            float2_t pm = __builtin_nontemporal_load(reinterpret_cast<float2_t*>(m) + i), pv = __builtin_nontemporal_load(reinterpret_cast<float2_t*>(v) + i),
                     pp = __builtin_nontemporal_load(reinterpret_cast<float2_t*>(p) + i);
            NGP_ADAM_EL(g0, pm.x, pv.x, pp.x);
            NGP_ADAM_EL(g1, pm.y, pv.y, pp.y);
            __builtin_nontemporal_store(pm, reinterpret_cast<float2_t*>(m) + i);
            __builtin_nontemporal_store(pv, reinterpret_cast<float2_t*>(v) + i);
            __builtin_nontemporal_store(pp, reinterpret_cast<float2_t*>(p) + i);
            if (p16) reinterpret_cast<half2_t*>(p16)[i] = half2_t{(half_t)pp.x, (half_t)pp.y};
            if (ema) {
                float2_t e = reinterpret_cast<float2_t*>(ema)[i];
                e.x -= ema_omd * (e.x - pp.x); e.y -= ema_omd * (e.y - pp.y);
                reinterpret_cast<float2_t*>(ema)[i] = e;
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const uint64_t i = n - 1;
            float g0 = gh ? (float)g16[i] : g32[i];
            if (!keep) { if (gh) g16[i] = (half_t)0.0f; else g32[i] = 0.0f; }
            if (!skip) {
                g0 *= inv_scale;
                float nm = m[i], nv = v[i], np_ = p[i];
                adam_element(g0, nm, nv, np_, beta1, beta2, eps, step_size, bc2_sqrt);
                m[i] = nm; v[i] = nv; p[i] = np_;
                if (p16) p16[i] = (half_t)np_;
            }
            if (ema) ema[i] -= ema_omd * (ema[i] - p[i]);
        }
    }
}

// torch.amp.GradScaler.update (_amp_update_scale_): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and, when it reaches
// growth_interval, scale *= growth (only if the result is finite) and tracker = 0.  Also commits the Adam step count.
__device__ __forceinline__ void update_scale(float* state, float growth, float backoff, float growth_interval, bool flip_parity = false) {
    // (all eight words are read in ONE request and the changed ones written back: written as read-modify-writes of single words this was a
    // chain of six dependent memory round trips -- 4 us of a 4 us kernel)
    float4_t a = *reinterpret_cast<const float4_t*>(state), b = *reinterpret_cast<const float4_t*>(state + 4);
    float scale = a.x, tracker = a.y, t = a.w, parity = b.y, dead = b.w;
    const bool scale_dead = !__builtin_isfinite(1.0f / scale);
    if (a.z != 0.0f || scale_dead) {   // (an underflowed scale counts as an overflow: see k_adam)
        // GradScaler has no lower bound either: a scale that has underflowed stays 0 and every later step is skipped -- the run is dead,
        // silently.  state[7] is the signal (sticky): optim.NGPAdam.scale_is_dead() / bench.py report it.
        if (scale_dead) dead = 1.0f;
        scale *= backoff;
        tracker = 0.0f;
    } else {
        t += 1.0f;
        const float tr = tracker + 1.0f;
        if (tr >= growth_interval) {
            const float grown = scale * growth;
            if (__builtin_isfinite(grown)) scale = grown;
            tracker = 0.0f;
        } else {
            tracker = tr;
        }
        // the step stands: the buffer set the grid backward wrote speculatively (ngp_table_adam_t) becomes the current one.  A skipped
        // step leaves the parity -- and with it the table, its moments and its fp16 shadow -- exactly as they were.
        if (flip_parity) parity = parity != 0.0f ? 0.0f : 1.0f;
    }
    *reinterpret_cast<float4_t*>(state) = float4_t{scale, tracker, 0.0f, t};
    state[5] = parity;
    state[7] = dead;
}
__global__ void k_update_scale(float* __restrict__ state, float growth, float backoff, float growth_interval, int flip_parity) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    update_scale(state, growth, backoff, growth_interval, flip_parity != 0);
}

// The rest of a step whose TABLE was updated inside the grid backward (ngp_table_adam_t): Adam on what is left -- the dense-level prefix of
// the double-buffered table (contiguous here; the accumulate's round-robin bins could only have reached it 8 bytes at a stride of 1 KiB)
// and the few small tensors (the two MLPs' weights: 18 k parameters, in place) -- and the commit, in ONE launch of a few workgroups: every
// workgroup has read the scalars before it draws its ticket (state[6]); the last one commits behind it.  ~8 us where k_adam (60 us over
// the table) + k_update_scale used to be.  Same arithmetic per element as k_adam.
constexpr int OPT_SMALL_THREADS = 1024;
__global__ __launch_bounds__(OPT_SMALL_THREADS) void k_adam_small_commit(OptTensors ts, float* __restrict__ state, float beta1, float beta2, float eps,
                                                                        float grad_mult, float growth, float backoff, float growth_interval,
                                                                        int flip_parity, TableAdam ta, const half_t* __restrict__ tgrad,
                                                                        uint64_t tprefix, uint32_t prefix_blocks) {
    const AdamConsts ac = adam_consts(state, beta1, beta2, grad_mult);
    const float bc2_sqrt = ac.bc2_sqrt;
    const float beta1_s = beta1, beta2_s = beta2, eps_s = eps;   // (the small tensors'; the table prefix brings its own in `ta` -- the same values)
    // two kinds of workgroups -- the first `prefix_blocks` sweep the table prefix, the others the small tensors -- so that the two dependent
    // load -> store chains run side by side instead of one behind the other in every lane
    const bool prefix_role = blockIdx.x < prefix_blocks;
    const uint32_t role_blocks = prefix_role ? prefix_blocks : gridDim.x - prefix_blocks;
    const uint64_t gtid = (uint64_t)(prefix_role ? blockIdx.x : blockIdx.x - prefix_blocks) * OPT_SMALL_THREADS + threadIdx.x;
    const uint64_t gstride = (uint64_t)role_blocks * OPT_SMALL_THREADS;
    if (prefix_role && tprefix && !ac.skip) {   // (a skipped step writes nothing: the current set stays what it is, the other one is never read)
        const uint32_t src = state[5] != 0.0f ? 1u : 0u, dst = src ^ 1u;
        const float step_size = ta.lr * ac.lr_mult / ac.bc1;
        beta1 = ta.beta1; beta2 = ta.beta2; eps = ta.eps;
        const float4_t* __restrict__ p_in = reinterpret_cast<const float4_t*>(ta.p[src]);
        const float4_t* __restrict__ m_in = reinterpret_cast<const float4_t*>(ta.m[src]);
        const float4_t* __restrict__ v_in = reinterpret_cast<const float4_t*>(ta.v[src]);
        float4_t* __restrict__ p_out = reinterpret_cast<float4_t*>(ta.p[dst]);
        float4_t* __restrict__ m_out = reinterpret_cast<float4_t*>(ta.m[dst]);
        float4_t* __restrict__ v_out = reinterpret_cast<float4_t*>(ta.v[dst]);
        half4_t* __restrict__ h_out = reinterpret_cast<half4_t*>(ta.p16[dst]);
        const half4_t* __restrict__ g4 = reinterpret_cast<const half4_t*>(tgrad);
        // two independent 4-element groups per lane and trip (all eight loads in flight before the first dependent instruction)
        const uint64_t n4 = tprefix / 4;
        for (uint64_t i = gtid; i < n4; i += 2 * gstride) {
            const uint64_t j = i + gstride;
            const bool two = j < n4;
            const uint64_t jj = two ? j : i;
            const half4_t x0 = g4[i], x1 = g4[jj];
            float4_t pm0 = m_in[i], pv0 = v_in[i], pp0 = p_in[i];
            float4_t pm1 = m_in[jj], pv1 = v_in[jj], pp1 = p_in[jj];
            half4_t ph0, ph1;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float g0 = (float)x0[c] * ac.inv_scale, g1 = (float)x1[c] * ac.inv_scale;
                NGP_ADAM_EL(g0, pm0[c], pv0[c], pp0[c]);
                ph0[c] = (half_t)pp0[c];
                NGP_ADAM_EL(g1, pm1[c], pv1[c], pp1[c]);
                ph1[c] = (half_t)pp1[c];
            }
            m_out[i] = pm0; v_out[i] = pv0; p_out[i] = pp0; h_out[i] = ph0;
            if (two) { m_out[j] = pm1; v_out[j] = pv1; p_out[j] = pp1; h_out[j] = ph1; }
        }
    }
    for (int k = 0; !prefix_role && k < ts.count; k++) {
        const uint64_t n = ts.n[k];
        const float step_size = ts.lr[k] * ac.lr_mult / ac.bc1;
        float* __restrict__ p = ts.p[k];
        float* __restrict__ m = ts.m[k];
        float* __restrict__ v = ts.v[k];
        half_t* __restrict__ p16 = ts.p16[k];
        const bool gh = (ts.g_is_half[k] & 1) != 0, keep = (ts.g_is_half[k] & 2) != 0;
        half_t* g16 = reinterpret_cast<half_t*>(ts.g[k]);
        float* g32 = reinterpret_cast<float*>(ts.g[k]);
        for (uint64_t i = gtid; i < n; i += gstride) {
            float g = gh ? (float)g16[i] : g32[i];
            if (!keep) { if (gh) g16[i] = (half_t)0.0f; else g32[i] = 0.0f; }
            if (ac.skip) continue;
            g *= ac.inv_scale;
            float nm = m[i], nv = v[i], np_ = p[i];
            adam_element(g, nm, nv, np_, beta1_s, beta2_s, eps_s, step_size, bc2_sqrt);
            m[i] = nm; v[i] = nv; p[i] = np_;
            if (p16) p16[i] = (half_t)np_;
        }
    }
    // every lane of this workgroup has read state[] (adam_consts, parity) before the workgroup draws its ticket; the last ticket commits
    // (No __threadfence() around the ticket: on this chip it writes back the XCD's whole dirty L2 -- full of the accumulate's Adam output at
    // this point -- once per workgroup, and nothing here needs it: what must be ordered are this workgroup's READS of state[], and a
    // load whose value has been consumed -- every one of them steers a branch above -- cannot be served later; what the last workgroup
    // writes is for the NEXT launch.  The ticket itself is a device-scope atomic.)
    __syncthreads();
    if (threadIdx.x == 0) {
        const float old = __hip_atomic_fetch_add(&state[6], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (float)(gridDim.x - 1u)) {
            update_scale(state, growth, backoff, growth_interval, flip_parity != 0);
            __hip_atomic_store(&state[6], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Data-parallel sharded update, the "skipped as a whole" verdict WITHOUT a collective of its own: a rank whose local gradient is not finite
// (found_inf = state[2], set by the producers or by the CHECK phase) writes NaN into the first element of EVERY shard of its flat fp16
// gradient before the reduce-scatter; the average of a shard that any rank poisoned is NaN on its owner, so after the exchange every rank
// reads the global verdict off the first element of ITS OWN shard (k_shard_verdict) -- the 4-byte all-reduce(MAX) that used to sit on the
// critical path next to the reduce-scatter is gone.  (The poisoned step is skipped, so the clobbered gradient elements are never used.)
__global__ void k_poison_shards(half_t* __restrict__ grad, uint32_t shards, uint64_t payload, const float* __restrict__ state) {
    if (state[2] == 0.0f) return;
    for (uint32_t r = threadIdx.x; r < shards; r += blockDim.x) grad[(uint64_t)r * payload] = (half_t)__builtin_nanf("");
}
// ... and takes the poison out of the flat buffer again (the exchange has consumed it): with a producer that OVERWRITES its gradients and an
// optimizer that therefore does not zero the flat buffer, a poisoned element that lies in padding would otherwise stay NaN forever
__global__ void k_shard_verdict(const half_t* __restrict__ shard, float* __restrict__ state, half_t* __restrict__ flat, uint32_t shards,
                                uint64_t payload) {
    if (threadIdx.x == 0 && !__builtin_isfinite((float)shard[0])) state[2] = 1.0f;
    if (flat)
        for (uint32_t r = threadIdx.x; r < shards; r += blockDim.x)
            if (!__builtin_isfinite((float)flat[(uint64_t)r * payload])) flat[(uint64_t)r * payload] = (half_t)0.0f;
}

// torch_ema's update() on its own (the Trainer calls it once per epoch, not per step): shadow -= omd * (shadow - param)
__global__ __launch_bounds__(OPT_THREADS) void k_ema(OptTensors ts, float omd) {
    for (int k = 0; k < ts.count; k++) {
        const uint64_t n = ts.n[k];
        const float* __restrict__ p = ts.p[k];
        float* __restrict__ e = ts.ema[k];
        for (uint64_t i = (uint64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * OPT_THREADS) e[i] -= omd * (e[i] - p[i]);
    }
}

}  // namespace ngp

using namespace ngp;

static uint32_t opt_blocks(uint64_t total) {
    uint32_t blocks = (uint32_t)cdiv64(total / 2 + 1, OPT_THREADS * 4);  // ~8 elements per lane
// grid cap: 768 workgroups = 3 per CU.  Swept in round 5 (EXPERIMENTS.md): 256 / 384 / 512 / 768 / 1024 / 2048 workgroups -> step 0.4419 /
// 0.4335 / 0.4255 / 0.4245 / 0.4291 / 0.4297 ms, k_adam itself 63.3 us at 512-1024 against 66.0 at 2048: fewer, longer-lived workgroups stream as
// fast and leave room for the lookahead march that starts beside this kernel
#ifndef NGP_ADAM_MAX_BLOCKS
#define NGP_ADAM_MAX_BLOCKS 768u
#endif
    if (blocks > NGP_ADAM_MAX_BLOCKS) blocks = NGP_ADAM_MAX_BLOCKS;
    if (blocks < 1u) blocks = 1u;
    return blocks;
}

extern "C" int ngp_optim_adam_step_ex(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                                      void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                                      float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor, float growth_interval,
                                      float* state, float* const* ema, float ema_one_minus_decay, uint32_t phases, ngp_stream_t stream) {
    NGP_REQUIRE(state, NGP_ERR_INVALID, "optim_adam_step: NULL state");
    NGP_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15u) == 0, NGP_ERR_INVALID, "optim_adam_step: state (8 floats) must be 16-byte aligned");
    NGP_REQUIRE((phases & ~15u) == 0 && (phases & 7u) != 0, NGP_ERR_INVALID, "optim_adam_step: phases must be a non-empty subset of CHECK|UPDATE|COMMIT (+FLIP)");
    NGP_REQUIRE(!(phases & NGP_OPT_PHASE_FLIP) || (phases & NGP_OPT_PHASE_COMMIT), NGP_ERR_INVALID, "optim_adam_step: FLIP rides in the COMMIT phase");
    hipStream_t st = as_stream(stream);
    if (phases & (NGP_OPT_PHASE_CHECK | NGP_OPT_PHASE_UPDATE)) {
        NGP_REQUIRE(count >= 1 && count <= OPT_MAX_TENSORS, NGP_ERR_INVALID, "optim_adam_step: between 1 and %d tensors per call (got %d)",
                    OPT_MAX_TENSORS, count);
        const bool upd = (phases & NGP_OPT_PHASE_UPDATE) != 0;
        NGP_REQUIRE(n && grads && grad_is_half && (!upd || (params && exp_avg && exp_avg_sq && lr)), NGP_ERR_INVALID,
                    "optim_adam_step: NULL argument");
        NGP_REQUIRE(!(ema_one_minus_decay > 0.0f) || ema, NGP_ERR_INVALID, "optim_adam_step: EMA decay given without shadow tensors");
        OptTensors ts;
        ts.count = count;
        uint64_t total = 0;
        for (int k = 0; k < count; k++) {
            NGP_REQUIRE(grads[k] && (!upd || (params[k] && exp_avg[k] && exp_avg_sq[k])), NGP_ERR_INVALID, "optim_adam_step: NULL tensor %d", k);
            ts.n[k] = n[k];
            ts.p[k] = upd ? params[k] : nullptr;   // a CHECK-only call needs the gradients alone
            ts.m[k] = upd ? exp_avg[k] : nullptr;
            ts.v[k] = upd ? exp_avg_sq[k] : nullptr;
            ts.g[k] = grads[k];
            ts.p16[k] = params_fp16 ? reinterpret_cast<half_t*>(params_fp16[k]) : nullptr;
            ts.g_is_half[k] = grad_is_half[k];
            ts.lr[k] = upd ? lr[k] : 0.0f;
            ts.ema[k] = ema ? ema[k] : nullptr;
            total += n[k];
        }
        const uint32_t blocks = opt_blocks(total);
        if (phases & NGP_OPT_PHASE_CHECK) {
            hipLaunchKernelGGL(k_check_finite, dim3(blocks), dim3(OPT_THREADS), 0, st, ts, state);
            const int rc = check_launch("optim_adam_step(check)");
            if (rc) return rc;
        }
        if (phases & NGP_OPT_PHASE_UPDATE) {
            hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(OPT_THREADS), 0, st, ts, (const float*)state, beta1, beta2, eps, grad_mult,
                               ema_one_minus_decay > 0.0f ? ema_one_minus_decay : 0.0f);
            const int rc = check_launch("optim_adam_step(adam)");
            if (rc) return rc;
        }
    }
    if (phases & NGP_OPT_PHASE_COMMIT) {
        hipLaunchKernelGGL(k_update_scale, dim3(1), dim3(64), 0, st, state, growth_factor, backoff_factor, growth_interval,
                           (phases & NGP_OPT_PHASE_FLIP) ? 1 : 0);
        return check_launch("optim_adam_step(scale)");
    }
    return NGP_OK;
}

extern "C" int ngp_optim_adam_step(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                                   void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                                   float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor, float growth_interval,
                                   float* state, ngp_stream_t stream) {
    const uint32_t phases = NGP_OPT_PHASE_CHECK | NGP_OPT_PHASE_UPDATE | (growth_interval < 0.0f ? 0u : NGP_OPT_PHASE_COMMIT);
    return ngp_optim_adam_step_ex(count, n, params, exp_avg, exp_avg_sq, grads, params_fp16, grad_is_half, lr, beta1, beta2, eps, grad_mult,
                                  growth_factor, backoff_factor, growth_interval, state, nullptr, 0.0f, phases, stream);
}

extern "C" int ngp_optim_adam_small_commit(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                                           void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                                           float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor,
                                           float growth_interval, float* state, int flip_parity, const ngp_table_adam_t* table,
                                           const void* table_grad_fp16, uint64_t table_prefix_params, ngp_stream_t stream) {
    NGP_REQUIRE(state, NGP_ERR_INVALID, "optim_adam_small_commit: NULL state");
    NGP_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15u) == 0, NGP_ERR_INVALID, "optim_adam_small_commit: state (8 floats) must be 16-byte aligned");
    NGP_REQUIRE(count >= 0 && count <= OPT_MAX_TENSORS, NGP_ERR_INVALID, "optim_adam_small_commit: at most %d tensors per call (got %d)", OPT_MAX_TENSORS,
                count);
    NGP_REQUIRE(count == 0 || (n && params && exp_avg && exp_avg_sq && grads && grad_is_half && lr), NGP_ERR_INVALID,
                "optim_adam_small_commit: NULL argument");
    OptTensors ts = {};
    ts.count = count;
    uint64_t total = 0;
    for (int k = 0; k < count; k++) {
        NGP_REQUIRE(grads[k] && params[k] && exp_avg[k] && exp_avg_sq[k], NGP_ERR_INVALID, "optim_adam_small_commit: NULL tensor %d", k);
        ts.n[k] = n[k];
        ts.p[k] = params[k];
        ts.m[k] = exp_avg[k];
        ts.v[k] = exp_avg_sq[k];
        ts.g[k] = grads[k];
        ts.p16[k] = params_fp16 ? reinterpret_cast<half_t*>(params_fp16[k]) : nullptr;
        ts.g_is_half[k] = grad_is_half[k];
        ts.lr[k] = lr[k];
        total += n[k];
    }
    TableAdam ta = {};
    if (table_prefix_params) {
        NGP_REQUIRE(table && table_grad_fp16 && table->state == state, NGP_ERR_INVALID,
                    "optim_adam_small_commit: a table prefix needs the table's buffer sets (on the same state) and its stored fp16 gradient");
        NGP_REQUIRE((table_prefix_params & 3u) == 0, NGP_ERR_INVALID, "optim_adam_small_commit: table_prefix_params must be a multiple of 4");
        for (int k = 0; k < 2; k++) {
            NGP_REQUIRE(table->param[k] && table->exp_avg[k] && table->exp_avg_sq[k] && table->param_fp16[k], NGP_ERR_INVALID,
                        "optim_adam_small_commit: NULL buffer in table set %d", k);
            ta.p[k] = table->param[k]; ta.m[k] = table->exp_avg[k]; ta.v[k] = table->exp_avg_sq[k];
            ta.p16[k] = reinterpret_cast<_Float16*>(table->param_fp16[k]);
        }
        ta.state = state;
        ta.lr = table->lr; ta.beta1 = table->beta1; ta.beta2 = table->beta2; ta.eps = table->eps;
        total += table_prefix_params;
    }
    // a few workgroups: meant for what is left when the table is updated elsewhere (the 12 M-entry table itself belongs in ngp_optim_adam_step_ex)
    NGP_REQUIRE(total <= (1u << 22), NGP_ERR_INVALID, "optim_adam_small_commit: %llu parameters -- this entry serves small tensors (<= 4 M); use "
                "ngp_optim_adam_step_ex", (unsigned long long)total);
    // (every workgroup ends with one atomic on the ticket word: ~100 of them keep that queue short)
    // workgroups of two kinds (see the kernel): one trip per lane where the cap allows it -- eight prefix parameters, one small-tensor element
    uint32_t prefix_blocks = (uint32_t)cdiv64(table_prefix_params, (uint64_t)OPT_SMALL_THREADS * 8);
    if (prefix_blocks > 192u) prefix_blocks = 192u;
    uint32_t small_blocks = (uint32_t)cdiv64(total - table_prefix_params, (uint64_t)OPT_SMALL_THREADS);
    small_blocks = small_blocks < 1u ? 1u : (small_blocks > 32u ? 32u : small_blocks);   // (>= 1: somebody has to draw the last ticket)
    hipLaunchKernelGGL(k_adam_small_commit, dim3(prefix_blocks + small_blocks), dim3(OPT_SMALL_THREADS), 0, as_stream(stream), ts, state, beta1, beta2, eps,
                       grad_mult, growth_factor, backoff_factor, growth_interval, flip_parity, ta, reinterpret_cast<const half_t*>(table_grad_fp16),
                       table_prefix_params, prefix_blocks);
    return check_launch("optim_adam_small_commit");
}

extern "C" int ngp_optim_poison_shards(void* flat_grad_fp16, uint32_t shards, uint64_t payload, const float* state, ngp_stream_t stream) {
    NGP_REQUIRE(flat_grad_fp16 && state && shards >= 1 && payload >= 1, NGP_ERR_INVALID, "optim_poison_shards: NULL / empty argument");
    hipLaunchKernelGGL(k_poison_shards, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<half_t*>(flat_grad_fp16), shards, payload, state);
    return check_launch("optim_poison_shards");
}

extern "C" int ngp_optim_shard_verdict(const void* shard_grad_fp16, float* state, void* flat_grad_fp16, uint32_t shards, uint64_t payload,
                                       ngp_stream_t stream) {
    NGP_REQUIRE(shard_grad_fp16 && state, NGP_ERR_INVALID, "optim_shard_verdict: NULL argument");
    NGP_REQUIRE(!flat_grad_fp16 || (shards >= 1 && payload >= 1), NGP_ERR_INVALID, "optim_shard_verdict: empty shard layout");
    hipLaunchKernelGGL(k_shard_verdict, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<const half_t*>(shard_grad_fp16), state,
                       reinterpret_cast<half_t*>(flat_grad_fp16), shards, payload);
    return check_launch("optim_shard_verdict");
}

extern "C" int ngp_optim_ema_update(int count, const uint64_t* n, float* const* params, float* const* ema, float one_minus_decay,
                                    ngp_stream_t stream) {
    NGP_REQUIRE(count >= 1 && count <= OPT_MAX_TENSORS, NGP_ERR_INVALID, "optim_ema_update: between 1 and %d tensors per call (got %d)",
                OPT_MAX_TENSORS, count);
    NGP_REQUIRE(n && params && ema, NGP_ERR_INVALID, "optim_ema_update: NULL argument");
    OptTensors ts = {};
    ts.count = count;
    uint64_t total = 0;
    for (int k = 0; k < count; k++) {
        NGP_REQUIRE(params[k] && ema[k], NGP_ERR_INVALID, "optim_ema_update: NULL tensor %d", k);
        ts.n[k] = n[k];
        ts.p[k] = params[k];
        ts.ema[k] = ema[k];
        total += n[k];
    }
    hipLaunchKernelGGL(k_ema, dim3(opt_blocks(total)), dim3(OPT_THREADS), 0, as_stream(stream), ts, one_minus_decay);
    return check_launch("optim_ema_update");
}
