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

namespace ngp {

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
    // parameter (seen in a 200 000-step soak of the bench workload, tools/soak_train.py): the unusable scale is an overflow of its own.
    const bool scale_dead = !__builtin_isfinite(1.0f / state[0]);
    const bool skip = state[2] != 0.0f || scale_dead;
    const float inv_scale = scale_dead ? 0.0f : grad_mult / state[0];
    const float t = state[3] + 1.0f;  // this step's count (k_update_scale commits it)
    const float bc1 = 1.0f - powf(beta1, t);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, t));
    const float lr_mult = state[4];
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
                    pm0[c] = beta1 * pm0[c] + (1.0f - beta1) * g0;
                    pv0[c] = beta2 * pv0[c] + (1.0f - beta2) * g0 * g0;
                    pp0[c] -= step_size * pm0[c] / (sqrtf(pv0[c]) / bc2_sqrt + eps);
                    ph0[c] = (half_t)pp0[c];
                    pm1[c] = beta1 * pm1[c] + (1.0f - beta1) * g1;
                    pv1[c] = beta2 * pv1[c] + (1.0f - beta2) * g1 * g1;
                    pp1[c] -= step_size * pm1[c] / (sqrtf(pv1[c]) / bc2_sqrt + eps);
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
                    pm[c] = beta1 * pm[c] + (1.0f - beta1) * g;
                    pv[c] = beta2 * pv[c] + (1.0f - beta2) * g * g;
                    pp[c] -= step_size * pm[c] / (sqrtf(pv[c]) / bc2_sqrt + eps);
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
            pm.x = beta1 * pm.x + (1.0f - beta1) * g0; pm.y = beta1 * pm.y + (1.0f - beta1) * g1;
            pv.x = beta2 * pv.x + (1.0f - beta2) * g0 * g0; pv.y = beta2 * pv.y + (1.0f - beta2) * g1 * g1;
            pp.x -= step_size * pm.x / (sqrtf(pv.x) / bc2_sqrt + eps);
            pp.y -= step_size * pm.y / (sqrtf(pv.y) / bc2_sqrt + eps);
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
                const float nm = beta1 * m[i] + (1.0f - beta1) * g0, nv = beta2 * v[i] + (1.0f - beta2) * g0 * g0;
                const float np_ = p[i] - step_size * nm / (sqrtf(nv) / bc2_sqrt + eps);
                m[i] = nm; v[i] = nv; p[i] = np_;
                if (p16) p16[i] = (half_t)np_;
            }
            if (ema) ema[i] -= ema_omd * (ema[i] - p[i]);
        }
    }
}

// torch.amp.GradScaler.update (_amp_update_scale_): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and, when it reaches
// growth_interval, scale *= growth (only if the result is finite) and tracker = 0.  Also commits the Adam step count.
__device__ __forceinline__ void update_scale(float* state, float growth, float backoff, float growth_interval) {
    if (state[2] != 0.0f || !__builtin_isfinite(1.0f / state[0])) {   // (an underflowed scale counts as an overflow: see k_adam)
        state[0] *= backoff;
        state[1] = 0.0f;
    } else {
        state[3] += 1.0f;
        const float tr = state[1] + 1.0f;
        if (tr >= growth_interval) {
            const float grown = state[0] * growth;
            if (__builtin_isfinite(grown)) state[0] = grown;
            state[1] = 0.0f;
        } else {
            state[1] = tr;
        }
    }
    state[2] = 0.0f;
}
__global__ void k_update_scale(float* __restrict__ state, float growth, float backoff, float growth_interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    update_scale(state, growth, backoff, growth_interval);
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
    NGP_REQUIRE((phases & ~7u) == 0 && phases != 0, NGP_ERR_INVALID, "optim_adam_step: phases must be a non-empty subset of CHECK|UPDATE|COMMIT");
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
        hipLaunchKernelGGL(k_update_scale, dim3(1), dim3(64), 0, st, state, growth_factor, backoff_factor, growth_interval);
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
