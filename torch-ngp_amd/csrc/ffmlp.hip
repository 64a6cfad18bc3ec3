// Fully fused fp16 MLP (bias-free, ReLU hidden layers) on the gfx950 matrix cores.
//
// Behaviour restated from the reference (paths relative to the reference checkout):
//   forward / inference  ffmlp/src/ffmlp.cu:331-407 (kernel_mlp_fused), hosts :635-709
//   backward             ffmlp/src/ffmlp.cu:410-518 (kernel_mlp_fused_backward) plus the CUTLASS split-K
//                        weight-gradient GEMMs and the dL/dinput GEMM launched from :749-895
//   weight layout        ffmlp/src/ffmlp.cu:631-634: W_in [hid,in] | (num_layers-1) x W_h [hid,hid] | W_out [16,hid],
//                        each row-major [out,in];  y = x . W^T, hidden activation on every hidden layer
//   activations          ffmlp/src/utils.h:29-37, 424-582
// Numerics: fp16 operands, fp32 accumulation everywhere (the reference accumulates in fp16, including
// the batch reductions of the weight gradients); hidden activations and back-propagated hidden
// gradients are rounded to fp16 between layers (they are fp16 tensors in the reference as well).
//
// MI355X design (see DESIGN.md "ffmlp"):
//   * The network is evaluated TRANSPOSED on v_mfma_f32_32x32x16_f16:  H_l^T [feat, sample] =
//     W_l [feat, k] . H_{l-1}^T [k, sample].  The weights are the A operand, a tile of 32 samples is the
//     B operand (one sample per lane column, lane>>5 selects the k half).  The C/D layout of that
//     instruction leaves lane (sample n, half h) holding, for every 32-feature block, features
//     {8q + 4h + c : q,c in 0..3} of ITS OWN sample -- and an MFMA contraction is indifferent to how
//     k slots are numbered as long as A and B agree.  So the A fragments are built with k slot
//     (kb, h, j) := feature 16kb + 8(j>>2) + 4h + (j&3), and the fp32 accumulators of layer l, packed
//     to fp16, ARE the B fragments of layer l+1: activations never leave the lane's registers between
//     layers -- no LDS round trip, no shuffle, no transposition.
//   * Weights live in LDS for the whole kernel, pre-swizzled into MFMA-fragment order (one 1 KiB
//     lane-linear ds_read_b128 per fragment); workgroups are persistent so the 14-22 KiB of weights are
//     fetched from L2 once per workgroup, not once per tile.
//   * The training forward streams each layer's post-activation fragments to forward_buffer in
//     fragment order (four fully coalesced 1 KiB stores per layer and tile); backward reads them back
//     the same way.  The layout of forward_buffer is private to this file.
//   * Backward is ONE kernel: the dgrad chain runs exactly like the forward (A = W^T fragments); the weight gradients
//     dW_l = dZ_l^T . X_l are accumulated per wave in fp32 MFMA accumulators over all of the wave's tiles.  The operand
//     transposition this needs (the batch dimension becomes the MFMA k dimension) runs on the matrix core itself (selector
//     MFMAs, see below) -- no LDS stage; the next tile's operands are streamed into per-wave LDS buffers by global->LDS DMA
//     while the current tile is computed (the accumulators leave one wave per SIMD, so latency must be hidden explicitly).
//     Per workgroup the four waves' partial sums are combined in LDS and written as one fp32 slab into the caller's
//     backward_buffer; a second tiny kernel adds the slabs in a fixed order and rounds once to fp16 -- deterministic, no
//     atomics, no side streams (this replaces the reference's num_layers+1 split-K GEMMs).
#include "common.h"
#include "sh_poly.inc"

namespace ngp {

constexpr int FF_WAVES = 4;
constexpr int FF_THREADS = FF_WAVES * 64;
constexpr int FF_TILE = 32;  // samples per wave tile

enum Act : uint32_t { ACT_RELU = 0, ACT_EXP = 1, ACT_SINE = 2, ACT_SIGMOID = 3, ACT_SQUAREPLUS = 4, ACT_SOFTPLUS = 5, ACT_NONE = 6 };
constexpr float K_ACT = 10.0f;  // utils.h:41

__device__ __forceinline__ float act_forward(uint32_t a, float x) {
    switch (a) {
        case ACT_RELU: return x > 0.0f ? x : 0.0f;
        case ACT_EXP: return __expf(x);
        case ACT_SINE: return __sinf(x);
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        case ACT_SQUAREPLUS: { const float s = x * K_ACT; return 0.5f * (s + sqrtf(s * s + 4.0f)) / K_ACT; }
        case ACT_SOFTPLUS: return __logf(__expf(x * K_ACT) + 1.0f) / K_ACT;
        default: return x;
    }
}
// derivative factor from the STORED post-activation y (utils.h:532-582, warp_activation_backward)
__device__ __forceinline__ float act_backward_factor(uint32_t a, float y) {
    switch (a) {
        case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case ACT_EXP: return y;
        case ACT_SIGMOID: return y * (1.0f - y);
        case ACT_SQUAREPLUS: { const float s = y * K_ACT; return s * s / (s * s + 1.0f); }
        case ACT_SOFTPLUS: return 1.0f - __expf(-y * K_ACT);
        default: return 1.0f;  // None; Sine has no transfer from post-activations in the reference either
    }
}

__device__ __forceinline__ float16_t mfma(half8_t a, half8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t zero16() {
    float16_t z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// k slot (kb, h, j) of a hidden-activation operand <-> hidden feature
__device__ __host__ __forceinline__ int slot_feature(int kb, int h, int j) { return 16 * kb + 8 * (j >> 2) + 4 * h + (j & 3); }
// accumulator register r of lane-half h in 32-row block ib <-> output row
__device__ __host__ __forceinline__ int acc_row(int ib, int h, int r) { return 32 * ib + 8 * (r >> 2) + 4 * h + (r & 3); }

// ------------------------------------------------------------------------------------------------
// LDS weight-fragment images.  A fragment is 64 lanes x 8 halves (1 KiB), stored lane-linear.
// Forward image order:  layer 0: [ib][kb<in/16] ; layers 1..NL-1: [ib][kb<NKB] ; output: [kb<NKB]
// ------------------------------------------------------------------------------------------------
template <int WIDTH>
struct Shape {
    static constexpr int NIB = (WIDTH + 31) / 32;  // 32-row output blocks of a hidden layer (WIDTH = 16: one block, rows 16..31 zero)
    static constexpr int NKB = (WIDTH + 15) / 16;  // 16-wide k blocks of a hidden operand
};

template <int WIDTH>
__device__ __forceinline__ uint32_t fwd_frag_count(uint32_t in_dim, uint32_t num_layers) {
    return Shape<WIDTH>::NIB * (in_dim / 16) + (num_layers - 1) * Shape<WIDTH>::NIB * Shape<WIDTH>::NKB + Shape<WIDTH>::NKB;
}

// Fill the forward image.  weights: flat fp16 as in ffmlp.cu:631-634.
// Every workgroup pays for this before its first tile, so the loop is organised around memory latency: the sources of IMG_BATCH
// fragments per thread are worked out first, ALL their loads are issued (unconditionally, at clamped addresses), and only then is
// anything stored -- one round trip to L2 per batch instead of one per fragment (measured: ~12 us of fixed cost per launch before).
constexpr int IMG_BATCH = 8;  // (10 = the fused network's whole image in one round: measured, no gain -- EXPERIMENTS.md)

// where the two 8-byte halves of fragment element e come from (nullptr: zeros)
template <int WIDTH>
__device__ __forceinline__ void forward_fragment_source(uint32_t e, const half_t* __restrict__ w, uint32_t in_dim, uint32_t num_layers,
                                                        const half_t*& lo, const half_t*& hi) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const uint32_t in_kb = in_dim / 16;
    const uint32_t frag = e >> 6, lane = e & 63;
    const int i = lane & 31, h = lane >> 5;
    uint32_t f = frag;
    if (f < NIB * in_kb) {  // input layer: natural k order, slot (kb,h,j) = input feature 16kb + 8h + j
        const uint32_t ib = f / in_kb, kb = f % in_kb;
        if (32 * ib + i >= (uint32_t)WIDTH) { lo = hi = nullptr; return; }  // WIDTH = 16: rows 16..31 of the block are zero
        lo = w + (size_t)(32 * ib + i) * in_dim + 16 * kb + 8 * h;
        hi = lo + 4;
        return;
    }
    f -= NIB * in_kb;
    const half_t* base = w + (size_t)WIDTH * in_dim;
    if (f < (num_layers - 1) * NIB * NKB) {  // hidden layer
        const uint32_t l = f / (NIB * NKB), rem = f % (NIB * NKB);
        const uint32_t ib = rem / NKB, kb = rem % NKB;
        if (32 * ib + i >= (uint32_t)WIDTH) { lo = hi = nullptr; return; }
        const half_t* row = base + (size_t)l * WIDTH * WIDTH + (size_t)(32 * ib + i) * WIDTH;
        lo = row + 16 * kb + 4 * h;
        hi = lo + 8;
        return;
    }
    // output layer: 16 real rows, rows 16..31 of the block are zero
    const uint32_t kb = f - (num_layers - 1) * NIB * NKB;
    const half_t* row = base + (size_t)(num_layers - 1) * WIDTH * WIDTH + (size_t)i * WIDTH;
    lo = i < 16 ? row + 16 * kb + 4 * h : nullptr;
    hi = i < 16 ? lo + 8 : nullptr;
}

// The WHOLE image from coalesced reads (round 4).  The gather below fetches every fragment element where it lies -- 64 lanes of a wave
// read 8 bytes from 32 different weight rows: 32 cache lines per wave instruction, and the L1 serves one line per clock; four workgroups
// per CU each fetching 20-24 KiB that way was most of the ~6 us every launch of the fused forward spent before its first tile.  Here the
// weights are read as they lie, 16 bytes per lane (a wave instruction = 1 KiB contiguous = 8-16 lines), and each lane drops its eight
// values into the image: one 16-byte LDS store on the input layer (an aligned run of 8 input features IS one fragment element), two
// 8-byte stores on the hidden / output layers (features 16kb + 8q + 0..3 belong to half-wave 0, + 4..7 to half-wave 1, both at
// element half q).  Rows that exist only as padding (output rows 16..31, rows >= WIDTH of a narrow network) are stored as zeros.
template <int WIDTH>
__device__ void build_forward_image_coalesced(half8_t* img, const half_t* __restrict__ w, uint32_t in_dim, uint32_t num_layers) {
    constexpr uint32_t NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const uint32_t in_kb = in_dim / 16, in_segs = in_dim / 8;
    constexpr uint32_t hid_segs = WIDTH / 8;
    const uint32_t units_in = NIB * 32u * in_segs, units_hid = NIB * 32u * hid_segs, units_out = 32u * hid_segs;
    const uint32_t total = units_in + (num_layers - 1) * units_hid + units_out;
    const half_t* w_hid = w + (size_t)WIDTH * in_dim;
    const half_t* w_out = w_hid + (size_t)(num_layers - 1) * WIDTH * WIDTH;
    const half8_t zero8 = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
    constexpr int BATCH = 6;  // the fused network's images are 6 units per thread: all loads of a thread in flight before its first store
    for (uint32_t u0 = threadIdx.x; u0 < total; u0 += BATCH * blockDim.x) {
        half8_t v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const uint32_t u = u0 + b * blockDim.x;
            v[b] = zero8;
            if (u < units_in) {
                const uint32_t r = u / in_segs, sg = u % in_segs;
                if (r < (uint32_t)WIDTH) v[b] = *reinterpret_cast<const half8_t*>(w + (size_t)r * in_dim + 8 * sg);
            } else if (u < total - units_out) {
                const uint32_t q = u - units_in, l = q / units_hid, rem = q % units_hid, r = rem / hid_segs, sg = rem % hid_segs;
                if (r < (uint32_t)WIDTH) v[b] = *reinterpret_cast<const half8_t*>(w_hid + ((size_t)l * WIDTH + r) * WIDTH + 8 * sg);
            } else if (u < total) {
                const uint32_t q = u - (total - units_out), r = q / hid_segs, sg = q % hid_segs;
                if (r < 16u) v[b] = *reinterpret_cast<const half8_t*>(w_out + (size_t)r * WIDTH + 8 * sg);
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const uint32_t u = u0 + b * blockDim.x;
            if (u >= total) continue;
            if (u < units_in) {
                const uint32_t r = u / in_segs, sg = u % in_segs, ib = r / 32u, i = r % 32u, kb = sg / 2u, h = sg % 2u;
                img[(ib * in_kb + kb) * 64u + h * 32u + i] = v[b];
            } else {
                uint32_t frag0, r, sg;
                if (u < total - units_out) {
                    const uint32_t q = u - units_in, l = q / units_hid, rem = q % units_hid;
                    r = rem / hid_segs; sg = rem % hid_segs;
                    frag0 = NIB * in_kb + l * NIB * NKB + (r / 32u) * NKB;
                } else {
                    const uint32_t q = u - (total - units_out);
                    r = q / hid_segs; sg = q % hid_segs;
                    frag0 = NIB * in_kb + (num_layers - 1) * NIB * NKB;
                }
                const uint32_t i = r % 32u, kb = sg / 2u, q2 = sg % 2u;
                half4_t* e0 = reinterpret_cast<half4_t*>(img + (frag0 + kb) * 64u + i) + q2;         // half-wave 0, element half q2
                half4_t* e1 = reinterpret_cast<half4_t*>(img + (frag0 + kb) * 64u + 32u + i) + q2;   // half-wave 1
                *e0 = half4_t{v[b][0], v[b][1], v[b][2], v[b][3]};
                *e1 = half4_t{v[b][4], v[b][5], v[b][6], v[b][7]};
            }
        }
    }
}

// fragments [first_frag, first_frag + n_frags) of the image order land at img[0 ...] (default: the whole image)
template <int WIDTH>
__device__ void build_forward_image(half8_t* img, const half_t* __restrict__ w, uint32_t in_dim, uint32_t num_layers, uint32_t first_frag = 0,
                                    uint32_t n_frags = 0xFFFFFFFFu) {
#ifndef NGP_FF_IMAGE_GATHER
    if (first_frag == 0u && n_frags == 0xFFFFFFFFu && (in_dim & 15u) == 0u) {
        build_forward_image_coalesced<WIDTH>(img, w, in_dim, num_layers);
        return;
    }
#endif
    const uint32_t all = fwd_frag_count<WIDTH>(in_dim, num_layers);
    const uint32_t total = (n_frags == 0xFFFFFFFFu ? all : n_frags) * 64;
    const uint32_t shift = first_frag * 64;
    for (uint32_t e0 = threadIdx.x; e0 < total; e0 += IMG_BATCH * blockDim.x) {
        half4_t lo[IMG_BATCH], hi[IMG_BATCH];
        bool real[IMG_BATCH];
#pragma unroll
        for (int b = 0; b < IMG_BATCH; b++) {
            const uint32_t e = e0 + b * blockDim.x;
            const half_t *pl = nullptr, *ph = nullptr;
            if (e < total) forward_fragment_source<WIDTH>(e + shift, w, in_dim, num_layers, pl, ph);
            real[b] = pl != nullptr;
            lo[b] = *reinterpret_cast<const half4_t*>(pl ? pl : w);
            hi[b] = *reinterpret_cast<const half4_t*>(ph ? ph : w);
        }
#pragma unroll
        for (int b = 0; b < IMG_BATCH; b++) {
            const uint32_t e = e0 + b * blockDim.x;
            if (e < total) {
                const half4_t z = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                const half4_t l4 = real[b] ? lo[b] : z, h4 = real[b] ? hi[b] : z;
                img[e] = half8_t{l4.x, l4.y, l4.z, l4.w, h4.x, h4.y, h4.z, h4.w};
            }
        }
    }
}

// pack the 16 fp32 accumulators of the NIB blocks into NKB fp16 operand fragments
template <int WIDTH>
__device__ __forceinline__ void pack_hidden(const float16_t (&acc)[Shape<WIDTH>::NIB], half8_t (&frag)[Shape<WIDTH>::NKB]) {
#pragma unroll
    for (int kb = 0; kb < Shape<WIDTH>::NKB; kb++) {
#pragma unroll
        for (int j = 0; j < 8; j++) frag[kb][j] = (half_t)acc[kb >> 1][(kb & 1) * 8 + j];
    }
}

// ReLU + pack in two instructions per PAIR of accumulators: round the pair to fp16 (v_cvt_pk_f16_f32, the rounding of pack_hidden), then
// a packed SIGNED-INTEGER max with 0 on the fp16 bit patterns -- every pattern with the sign bit set (negative values, -0) becomes +0,
// every other one is kept: relu(half(x)) = half(relu(x)) bit for bit (rounding is monotonic and keeps the sign).  The fp32 form
// `fmaxf(acc, 0)` costs FOUR per pair (the compiler quiets a possible signalling NaN with a v_max(x, x) of its own before the max with 0)
// plus the conversion: 88 vector instructions per 64-wide layer and tile against 32 here -- more than half of the network forward's
// vector stream was this (SQ_INSTS_VALU 808 per tile, EXPERIMENTS.md round 5).  NaN: a positive NaN now propagates (fmaxf returned 0).
__device__ __forceinline__ uint32_t relu_pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const half2_t hv = __builtin_convertvector(f32x2{a, b}, half2_t);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, hv), i16x2{0, 0}));
}
template <int WIDTH>
__device__ __forceinline__ void pack_hidden_relu(const float16_t (&acc)[Shape<WIDTH>::NIB], half8_t (&frag)[Shape<WIDTH>::NKB]) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int kb = 0; kb < Shape<WIDTH>::NKB; kb++) {
        u32x4 w;
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = relu_pack2(acc[kb >> 1][(kb & 1) * 8 + 2 * q], acc[kb >> 1][(kb & 1) * 8 + 2 * q + 1]);
        frag[kb] = __builtin_bit_cast(half8_t, w);
    }
}


// 8 consecutive input features f0 .. f0+7 (f0 a multiple of 8) of sample s.
//   row-major: inputs[s][in_dim]                      (the reference layout)
//   planar   : inputs[in_dim/2][rows][2]              (the grid encoder's level-major [L, B, C=2] output: no permute copy
//              between encoder and MLP; a wave still reads/writes 128 contiguous bytes per plane)
__device__ __forceinline__ half8_t load_features8(const half_t* __restrict__ inputs, bool planar, size_t rows, size_t s, uint32_t in_dim,
                                                  uint32_t f0) {
    if (!planar) return *reinterpret_cast<const half8_t*>(inputs + s * in_dim + f0);
    half8_t v;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const half2_t t = *reinterpret_cast<const half2_t*>(inputs + ((size_t)(f0 / 2 + q) * rows + s) * 2);
        v[2 * q] = t.x;
        v[2 * q + 1] = t.y;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// forward / inference
// ------------------------------------------------------------------------------------------------
// The activation streams are written once here and read once by the backward launches, hundreds of megabytes later: the stores are
// marked non-temporal, so they stream past the L2 instead of displacing it (same-box A/B together with Adam's streams: -11 us per iteration).
template <typename V>
__device__ __forceinline__ void stream_store(V* dst, V v) {
    __builtin_nontemporal_store(v, dst);
}

template <int WIDTH, bool TRAIN, bool PLAIN /* ReLU hidden layers, no output activation: the only case FFMLP produces (ffmlp.py:107) */>
__device__ __forceinline__ void ffmlp_forward_body(const half_t* __restrict__ inputs, const half_t* __restrict__ weights,
                                                   half_t* __restrict__ forward_buffer, half_t* __restrict__ outputs, uint32_t n_tiles,
                                                   uint32_t in_dim, uint32_t num_layers, uint32_t act, uint32_t out_act, bool in_planar) {
#ifdef NGP_FF_DIAG  // timing experiments of tools/bench_kernels.py (compile-time only) -- bit 0: no activation/output stores, bit 1: no input loads
    const uint32_t diag = NGP_FF_DIAG;
#else
    constexpr uint32_t diag = 0u;
#endif
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img = reinterpret_cast<half8_t*>(smem);
    build_forward_image<WIDTH>(img, weights, in_dim, num_layers);
    const size_t rows = (size_t)n_tiles * FF_TILE;
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const uint32_t in_kb = in_dim / 16;
    const half8_t* img_l0 = img + lane;
    const half8_t* img_hid = img_l0 + (size_t)NIB * in_kb * 64;
    const half8_t* img_out = img_hid + (size_t)(num_layers - 1) * NIB * NKB * 64;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;  // in half8 units

    for (uint32_t tile = blockIdx.x * FF_WAVES + wid; tile < n_tiles; tile += gridDim.x * FF_WAVES) {
        const size_t srow = (size_t)tile * FF_TILE + n;
        float16_t acc[NIB];
#pragma unroll
        for (int ib = 0; ib < NIB; ib++) acc[ib] = zero16();
        for (uint32_t kb = 0; kb < in_kb; kb++) {
            const half8_t x = (diag & 2u) ? img_l0[kb * 64] : load_features8(inputs, in_planar, rows, srow, in_dim, 16 * kb + 8 * h);
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(img_l0[(ib * in_kb + kb) * 64], x, acc[ib]);
        }
        half8_t hid[NKB];
        for (uint32_t l = 0;; l++) {
            // activation of hidden layer l (fp32), then round to fp16 operand fragments
            if (PLAIN || act == ACT_RELU) {
                pack_hidden_relu<WIDTH>(acc, hid);
            } else {
                if (act != ACT_NONE) {
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[ib][r] = act_forward(act, acc[ib][r]);
                }
                pack_hidden<WIDTH>(acc, hid);
            }
            if (TRAIN && !(diag & 1u)) {
                half8_t* dst = reinterpret_cast<half8_t*>(forward_buffer) + l * layer_stride + (size_t)tile * NKB * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) stream_store(dst + kb * 64, hid[kb]);
            }
            if (l + 1 == num_layers) break;
            const half8_t* wl = img_hid + (size_t)l * NIB * NKB * 64;
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) {
                acc[ib] = zero16();
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) acc[ib] = mfma(wl[(ib * NKB + kb) * 64], hid[kb], acc[ib]);
            }
        }
        // output layer: one 32-row block of which rows 0..15 are real
        float16_t o = zero16();
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) o = mfma(img_out[kb * 64], hid[kb], o);
        half4_t lo, hi;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            lo[c] = (half_t)(PLAIN ? o[c] : act_forward(out_act, o[c]));          // out features 4h + c
            hi[c] = (half_t)(PLAIN ? o[4 + c] : act_forward(out_act, o[4 + c]));  // out features 8 + 4h + c
        }
        half_t* orow = outputs + ((size_t)tile * FF_TILE + n) * 16 + 4 * h;
        if (!(diag & 1u) || tile == 0) {
            *reinterpret_cast<half4_t*>(orow) = lo;
            *reinterpret_cast<half4_t*>(orow + 8) = hi;
        }
    }
}

// widths <= 64: four waves per SIMD without AGPR copies (<= 128 registers); 128-wide layers keep 64 accumulators + 32 operand registers
// per lane and run two waves per SIMD
template <int WIDTH, bool TRAIN, bool PLAIN>
__global__ __launch_bounds__(FF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ffmlp_forward(
    const half_t* __restrict__ inputs, const half_t* __restrict__ weights, half_t* __restrict__ forward_buffer, half_t* __restrict__ outputs,
    uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, uint32_t act, uint32_t out_act, bool in_planar) {
    ffmlp_forward_body<WIDTH, TRAIN, PLAIN>(inputs, weights, forward_buffer, outputs, n_tiles, in_dim, num_layers, act, out_act, in_planar);
}
template <int WIDTH, bool TRAIN, bool PLAIN>
__global__ __launch_bounds__(FF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffmlp_forward_wide(
    const half_t* __restrict__ inputs, const half_t* __restrict__ weights, half_t* __restrict__ forward_buffer, half_t* __restrict__ outputs,
    uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, uint32_t act, uint32_t out_act, bool in_planar) {
    ffmlp_forward_body<WIDTH, TRAIN, PLAIN>(inputs, weights, forward_buffer, outputs, n_tiles, in_dim, num_layers, act, out_act, in_planar);
}


// ------------------------------------------------------------------------------------------------
// FUSED NETWORK FORWARD (extension): nerf/network_ff.py:51-74 between the encoder and the compositor in ONE launch --
//   sigma FFMLP (32 -> 64 x nl_s -> 16) -> [sigma = density_scale * exp(h0); colour input = half(SH_4(dir)) | h1..h15 | 0]
//   -> colour FFMLP (32 -> 64 x nl_c -> 16) -> rgb = half(sigmoid(out[:3])) as fp32.
// The four-kernel sequence it replaces (ffmlp_forward, mid_forward, ffmlp_forward, rgb_forward) spends most of its time outside the
// matrix core: every launch rebuilds a weight image (~8 us), a wave sees ~2 tiles per launch, and h16 / colour input / out16 round-trip
// through HBM.  Here both weight images are built once, a tile's activations stay in the lane's registers from the encoder output to
// the colour, and the sigma-net output reaches the colour net's B operand through one half-wave exchange (the C/D layout leaves lane
// (n, h) with output features {4h..4h+3, 8+4h..11+4h}; the colour input's k slots want features 8h..8h+7 contiguous).
// Same MFMA sequence and the same fp16 rounding points as the separate kernels: bit-identical sigma / rgb / stored activations.
// TRAIN additionally stores what the backward kernels read: both forward buffers (fragment order), h16 and the colour input (row-major).
// ------------------------------------------------------------------------------------------------
#ifdef NGP_NETFWD_DIAG  // timing experiments of tools/netfwd_probe.py (compile-time only, results are garbage): bit 0: no MFMA (one dependent vector
                        // instruction instead), bit 1: no ReLU / pack, bit 2: no SH polynomials, bit 3: no exp / sigmoid epilogue
__device__ __forceinline__ float16_t net_mfma(half8_t a, half8_t b, float16_t c) {
    if (NGP_NETFWD_DIAG & 1) { c[0] += (float)a[0] * (float)b[0]; return c; }
    return mfma(a, b, c);
}
template <int WIDTH>
__device__ __forceinline__ void net_pack(const float16_t (&acc)[Shape<WIDTH>::NIB], half8_t (&frag)[Shape<WIDTH>::NKB]) {
    if (NGP_NETFWD_DIAG & 2) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int kb = 0; kb < Shape<WIDTH>::NKB; kb++) {
            f32x4 w = {acc[kb >> 1][(kb & 1) * 8], acc[kb >> 1][(kb & 1) * 8 + 2], acc[kb >> 1][(kb & 1) * 8 + 4], acc[kb >> 1][(kb & 1) * 8 + 6]};
            frag[kb] = __builtin_bit_cast(half8_t, w);
        }
        return;
    }
    pack_hidden_relu<WIDTH>(acc, frag);
}
#else
#define net_mfma mfma
#define net_pack pack_hidden_relu
#endif
// hidden layers + output layer of a 64-wide ReLU network for the NT tiles a wave works on together: first-layer accumulators in, output
// accumulators out; the hidden post-activations are streamed to `fb` (fragment order) when TRAIN.  Every weight fragment is read from
// the LDS image ONCE and multiplied into all NT tiles: NT independent accumulator chains per wave (the vector work of one tile issues
// under the matrix work of the other), half the LDS reads per MFMA at NT = 2.
template <bool TRAIN, int NT>
__device__ __forceinline__ void relu_network_tail(float16_t (&acc)[NT][2], float16_t (&o)[NT], const half8_t* __restrict__ hid_img,
                                                  const half8_t* __restrict__ out_img, uint32_t nl, half_t* __restrict__ fb, size_t layer_stride,
                                                  const uint32_t (&tile)[NT], const bool (&live)[NT], int lane) {
    constexpr int WIDTH = 64, NIB = 2, NKB = 4;
    half8_t hid[NT][NKB];
    for (uint32_t l = 0;; l++) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            net_pack<WIDTH>(acc[t], hid[t]);
            if (TRAIN && fb && live[t]) {  // (fb == NULL: the backward recomputes the activations, NGP_FF_RECOMPUTE)
                half8_t* dst = reinterpret_cast<half8_t*>(fb) + l * layer_stride + (size_t)tile[t] * NKB * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) stream_store(dst + kb * 64, hid[t][kb]);
            }
        }
        if (l + 1 == nl) break;
        const half8_t* wl = hid_img + (size_t)l * NIB * NKB * 64;
#pragma unroll
        for (int ib = 0; ib < NIB; ib++) {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t][ib] = zero16();
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) {
                const half8_t a = wl[(ib * NKB + kb) * 64];
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ib] = net_mfma(a, hid[t][kb], acc[t][ib]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) o[t] = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) {
        const half8_t a = out_img[kb * 64];
#pragma unroll
        for (int t = 0; t < NT; t++) o[t] = net_mfma(a, hid[t][kb], o[t]);
    }
}

template <bool TRAIN, int WAVES /* per workgroup: they share one pair of weight images */, int NT /* tiles a wave works on together */>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(NT == 1 ? 4 : 2, NT == 1 ? 4 : 3))) void k_network_forward(
    const half_t* __restrict__ enc, bool enc_planar, const float* __restrict__ dirs, uint32_t M_valid, const half_t* __restrict__ w_sigma,
    const half_t* __restrict__ w_color, half_t* __restrict__ fb_s, half_t* __restrict__ h16, float* __restrict__ sigma,
    half_t* __restrict__ color_in, half_t* __restrict__ fb_c, float* __restrict__ rgb, uint32_t n_tiles_cap, uint32_t nl_s, uint32_t nl_c,
    float density_scale, const uint32_t* __restrict__ rows_dev) {
    constexpr int WIDTH = 64, NIB = 2, NKB = 4;
    constexpr uint32_t in_kb = 2;  // both networks take 32 inputs
    // n_tiles_cap: the buffers' rows / 32 (strides of the planar input and of the stored activations); n_tiles: the tiles that carry work --
    // all of them, or (rows_dev: the eval loop's device-side count of emitted rows) the first ceil(*rows_dev / 32)
    const uint32_t n_tiles = rows_dev ? min(n_tiles_cap, (rows_dev[0] + (uint32_t)FF_TILE - 1u) / (uint32_t)FF_TILE) : n_tiles_cap;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img_s = reinterpret_cast<half8_t*>(smem);
    const uint32_t frags_s = NIB * in_kb + (nl_s - 1) * NIB * NKB + NKB;
    half8_t* img_c = img_s + (size_t)frags_s * 64;
    const size_t rows = (size_t)n_tiles_cap * FF_TILE;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const uint32_t n_groups = (n_tiles + NT - 1) / NT;   // a wave iteration = NT consecutive tiles (the last group repeats the last tile)
    // a group's inputs (encoder features, direction) are requested one group ahead -- the first one's before the weight images are built,
    // so that round trip runs under the build, the later ones under the previous group's arithmetic
    half8_t x_next[NT][in_kb];
    float dir_next[NT][3];
#pragma unroll
    for (int t = 0; t < NT; t++) dir_next[t][0] = dir_next[t][1] = dir_next[t][2] = 0.0f;
    auto request = [&](uint32_t group) {
        if (group < n_groups) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint32_t tl = group * NT + t < n_tiles ? group * NT + t : n_tiles - 1;
                const size_t srow = (size_t)tl * FF_TILE + n;
#pragma unroll
                for (uint32_t kb = 0; kb < in_kb; kb++) x_next[t][kb] = load_features8(enc, enc_planar, rows, srow, 32, 16 * kb + 8 * h);
                dir_next[t][0] = dir_next[t][1] = dir_next[t][2] = 0.0f;
                if (srow < M_valid) { dir_next[t][0] = dirs[srow * 3]; dir_next[t][1] = dirs[srow * 3 + 1]; dir_next[t][2] = dirs[srow * 3 + 2]; }
            }
        }
    };
    request(blockIdx.x * WAVES + wid);
    build_forward_image<WIDTH>(img_s, w_sigma, 32, nl_s);
    build_forward_image<WIDTH>(img_c, w_color, 32, nl_c);
    __syncthreads();

    const size_t layer_stride = (size_t)n_tiles_cap * NKB * 64;  // half8 units
    const half8_t* s_l0 = img_s + lane;
    const half8_t* s_hid = s_l0 + (size_t)NIB * in_kb * 64;
    const half8_t* s_out = s_hid + (size_t)(nl_s - 1) * NIB * NKB * 64;
    const half8_t* c_l0 = img_c + lane;
    const half8_t* c_hid = c_l0 + (size_t)NIB * in_kb * 64;
    const half8_t* c_out = c_hid + (size_t)(nl_c - 1) * NIB * NKB * 64;

    for (uint32_t group = blockIdx.x * WAVES + wid; group < n_groups; group += gridDim.x * WAVES) {
        uint32_t tile[NT];
        bool live[NT];
        size_t srow[NT];
        half8_t x_cur[NT][in_kb];
        float dir_x[NT], dir_y[NT], dir_z[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            live[t] = NT == 1 || group * NT + t < n_tiles;
            tile[t] = live[t] ? group * NT + t : n_tiles - 1;
            srow[t] = (size_t)tile[t] * FF_TILE + n;
#pragma unroll
            for (uint32_t kb = 0; kb < in_kb; kb++) x_cur[t][kb] = x_next[t][kb];
            dir_x[t] = dir_next[t][0]; dir_y[t] = dir_next[t][1]; dir_z[t] = dir_next[t][2];
        }
        request(group + gridDim.x * WAVES);
        // ---- sigma network ----
        float16_t acc[NT][NIB], o[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) acc[t][ib] = zero16();
#pragma unroll
        for (uint32_t kb = 0; kb < in_kb; kb++)
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) {
                const half8_t a = s_l0[(ib * in_kb + kb) * 64];
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ib] = net_mfma(a, x_cur[t][kb], acc[t][ib]);
            }
        relu_network_tail<TRAIN, NT>(acc, o, s_hid, s_out, nl_s, fb_s, layer_stride, tile, live, lane);
        half8_t cin[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            // h16 = half(o): this lane owns output features 4h + c (lo) and 8 + 4h + c (hi) of its sample
            half4_t lo, hi;
#pragma unroll
            for (int c = 0; c < 4; c++) { lo[c] = (half_t)o[t][c]; hi[c] = (half_t)o[t][4 + c]; }
            if (TRAIN && live[t]) {
                half_t* hrow = h16 + srow[t] * 16 + 4 * h;
                *reinterpret_cast<half4_t*>(hrow) = lo;
                *reinterpret_cast<half4_t*>(hrow + 8) = hi;
            }
#if defined(NGP_NETFWD_DIAG) && (NGP_NETFWD_DIAG & 8)
            if (h == 0 && live[t]) sigma[srow[t]] = density_scale * (float)lo[0];
#else
            if (h == 0 && live[t]) sigma[srow[t]] = density_scale * expf((float)lo[0]);
#endif  // trunc_exp forward on the fp16 output (activation.py:9-10)
            // ---- colour-net input: k block 0 = half(SH_4(dir))[8h .. 8h+7], k block 1 = h16[1 + 8h + j] (j < 8; feature 16 -> the zero pad) ----
            const float x = dir_x[t], y = dir_y[t], z = dir_z[t];
            // component i goes to half-wave i >> 3, slot i & 7: all 16 polynomials as fp32 values (pinned: the conversion must round the
            // fp32 VALUE, see to_half_rne), then one select per slot and four packed conversions.  (A conditional store per component --
            // the first version -- compiled to sixteen divergent branches per tile, both sides taken by every wave.)
            float sh_0, sh_1, sh_2, sh_3, sh_4, sh_5, sh_6, sh_7, sh_8, sh_9, sh_10, sh_11, sh_12, sh_13, sh_14, sh_15;
#define SH_OUT(i, v) { sh_##i = (v); asm volatile("" : "+v"(sh_##i)); }
#if defined(NGP_NETFWD_DIAG) && (NGP_NETFWD_DIAG & 4)
            sh_0 = sh_1 = sh_2 = sh_3 = sh_4 = sh_5 = sh_6 = sh_7 = x; sh_8 = sh_9 = sh_10 = sh_11 = sh_12 = sh_13 = sh_14 = sh_15 = y + z;
#else
            SH_BAND_0_VALUES;
            SH_BAND_1_VALUES;
            SH_BAND_2_VALUES;
            SH_BAND_3_VALUES;
#endif
#undef SH_OUT
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SH_PAIR(q, a0, a1, b0, b1) { const half2_t pr = __builtin_convertvector(f32x2{h ? b0 : a0, h ? b1 : a1}, half2_t); cin[t][0][2 * (q)] = pr.x; cin[t][0][2 * (q) + 1] = pr.y; }
            SH_PAIR(0, sh_0, sh_1, sh_8, sh_9);
            SH_PAIR(1, sh_2, sh_3, sh_10, sh_11);
            SH_PAIR(2, sh_4, sh_5, sh_12, sh_13);
            SH_PAIR(3, sh_6, sh_7, sh_14, sh_15);
#undef SH_PAIR
            // exchange the eight fp16 outputs with the partner half-wave (lane ^ 32): mine[q] = feature (q < 4 ? 4h + q : 8 + 4h + q - 4)
            const uint32_t m0 = __builtin_bit_cast(uint32_t, half2_t{lo[0], lo[1]}), m1 = __builtin_bit_cast(uint32_t, half2_t{lo[2], lo[3]});
            const uint32_t m2 = __builtin_bit_cast(uint32_t, half2_t{hi[0], hi[1]}), m3 = __builtin_bit_cast(uint32_t, half2_t{hi[2], hi[3]});
            const half2_t p0 = __builtin_bit_cast(half2_t, (uint32_t)__shfl_xor((int)m0, 32, 64)), p1 = __builtin_bit_cast(half2_t, (uint32_t)__shfl_xor((int)m1, 32, 64));
            const half2_t p2 = __builtin_bit_cast(half2_t, (uint32_t)__shfl_xor((int)m2, 32, 64)), p3 = __builtin_bit_cast(half2_t, (uint32_t)__shfl_xor((int)m3, 32, 64));
            // p0, p1 = the partner's lo quad (features 4(1-h) + 0..3), p2, p3 = its hi quad (8 + 4(1-h) + 0..3)
            // feature f of the full row: half-wave (f >> 2) & 1 holds it, in lo (f < 8) or hi, at position f & 3
            // h == 0 wants features 1..8 : mine lo[1..3], partner lo[0..3] (4..7), mine hi[0] (8)
            // h == 1 wants features 9..15, pad: partner hi[1..3] (9..11), mine hi[0..3] (12..15), 0
            if (h == 0) {
                cin[t][1][0] = lo[1]; cin[t][1][1] = lo[2]; cin[t][1][2] = lo[3];
                cin[t][1][3] = p0.x; cin[t][1][4] = p0.y; cin[t][1][5] = p1.x; cin[t][1][6] = p1.y;
                cin[t][1][7] = hi[0];
            } else {
                cin[t][1][0] = p2.y; cin[t][1][1] = p3.x; cin[t][1][2] = p3.y;
                cin[t][1][3] = hi[0]; cin[t][1][4] = hi[1]; cin[t][1][5] = hi[2]; cin[t][1][6] = hi[3];
                cin[t][1][7] = (half_t)0.0f;
            }
            if (TRAIN && live[t]) {
                half_t* crow = color_in + srow[t] * 32 + 8 * h;
                stream_store(reinterpret_cast<half8_t*>(crow), cin[t][0]);
                stream_store(reinterpret_cast<half8_t*>(crow + 16), cin[t][1]);
            }
        }
        // ---- colour network ----
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) acc[t][ib] = zero16();
#pragma unroll
        for (uint32_t kb = 0; kb < in_kb; kb++)
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) {
                const half8_t a = c_l0[(ib * in_kb + kb) * 64];
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ib] = net_mfma(a, cin[t][kb], acc[t][ib]);
            }
        relu_network_tail<TRAIN, NT>(acc, o, c_hid, c_out, nl_c, fb_c, layer_stride, tile, live, lane);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (h == 0 && live[t]) {  // rgb = fp16-rounded sigmoid of the fp16-rounded outputs 0..2 (network_ff.py:72)
                float* prgb = rgb + srow[t] * 3;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float v = (float)(half_t)o[t][c];
#if defined(NGP_NETFWD_DIAG) && (NGP_NETFWD_DIAG & 8)
                    prgb[c] = v;
#else
                    prgb[c] = (float)to_half_rne(1.0f / (1.0f + expf(-v)));
#endif
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Backward image order (A = W^T fragments for the dgrad chain):
//   out : [ib<NIB]              A[i=hidden feat][slot(h,j)] = W_out[8h+j][feat]            (K = 16 real outputs)
//   hid : layers NL-1 .. 1, each [ib<NIB][kb<NKB]   A[i][slot(kb,h,j)] = W_l[slot_feature][i]
//   in  : [ib<in/32 rounded up][kb<NKB]             A[i=input feat][slot] = W_in[slot_feature][i]   (only if dL/dx wanted)
template <int WIDTH>
__device__ __forceinline__ uint32_t bwd_frag_count(uint32_t in_dim, uint32_t num_layers, bool with_dx) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    return NIB + (num_layers - 1) * NIB * NKB + (with_dx ? ((in_dim + 31) / 32) * NKB : 0);
}

// element j of fragment element e is read from src + row_of(j) * stride (rows 0..3 then 8..11 in slot order, 0..7 for the output
// block); src == nullptr: zeros
template <int WIDTH>
__device__ __forceinline__ void backward_fragment_source(uint32_t e, const half_t* __restrict__ w, uint32_t in_dim, uint32_t num_layers,
                                                         const half_t*& src, uint32_t& stride, bool& slot_order) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const half_t* w_hid = w + (size_t)WIDTH * in_dim;
    const half_t* w_out = w_hid + (size_t)(num_layers - 1) * WIDTH * WIDTH;
    const uint32_t frag = e >> 6, lane = e & 63;
    const int i = lane & 31, h = lane >> 5;
    uint32_t f = frag;
    if (f < NIB) {  // W_out^T: rows 8h + j
        src = 32 * f + i < (uint32_t)WIDTH ? w_out + (size_t)(8 * h) * WIDTH + (32 * f + i) : nullptr;
        stride = WIDTH;
        slot_order = false;
    } else if ((f -= NIB) < (num_layers - 1) * NIB * NKB) {
        const uint32_t li = f / (NIB * NKB), rem = f % (NIB * NKB);  // li = 0 -> layer NL-1
        const uint32_t ib = rem / NKB, kb = rem % NKB;
        const half_t* wl = w_hid + (size_t)(num_layers - 2 - li) * WIDTH * WIDTH;
        src = 32 * ib + i < (uint32_t)WIDTH ? wl + (size_t)slot_feature(kb, h, 0) * WIDTH + (32 * ib + i) : nullptr;
        stride = WIDTH;
        slot_order = true;
    } else {
        f -= (num_layers - 1) * NIB * NKB;
        const uint32_t ib = f / NKB, kb = f % NKB;
        const uint32_t col = 32 * ib + i;
        src = col < in_dim ? w + (size_t)slot_feature(kb, h, 0) * in_dim + col : nullptr;
        stride = in_dim;
        slot_order = true;
    }
}

template <int WIDTH>
__device__ void build_backward_image(half8_t* img, const half_t* __restrict__ w, uint32_t in_dim, uint32_t num_layers, bool with_dx,
                                     uint32_t first_frag = 0, uint32_t n_frags = 0xFFFFFFFFu) {
    constexpr int BATCH = 3;  // x 8 two-byte gathers per fragment element, all in flight before the first store
    const uint32_t total = (n_frags == 0xFFFFFFFFu ? bwd_frag_count<WIDTH>(in_dim, num_layers, with_dx) : n_frags) * 64;
    const uint32_t shift = first_frag * 64;
    for (uint32_t e0 = threadIdx.x; e0 < total; e0 += BATCH * blockDim.x) {
        half_t v[BATCH][8];
        bool real[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const uint32_t e = e0 + b * blockDim.x;
            const half_t* src = nullptr;
            uint32_t stride = 0;
            bool slot_order = false;
            if (e < total) backward_fragment_source<WIDTH>(e + shift, w, in_dim, num_layers, src, stride, slot_order);
            real[b] = src != nullptr;
            const half_t* p = src ? src : w;
            const uint32_t st = src ? stride : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t row = slot_order ? (uint32_t)((j & 3) + 8 * (j >> 2)) : (uint32_t)j;  // slot_feature(kb,h,j) - slot_feature(kb,h,0)
                v[b][j] = p[(size_t)row * st];
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const uint32_t e = e0 + b * blockDim.x;
            if (e < total) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = real[b] ? v[b][j] : (half_t)0.0f;
                img[e] = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Transposition on the matrix core.
// The weight gradient dW[o,i] = sum_s dZ[s,o] * X[s,i] contracts over SAMPLES, so both MFMA operands need "8 consecutive
// samples of one feature per lane", the transpose of what a lane holds after the dgrad chain (lane = sample, 8 features).
// Instead of a round trip through LDS (8 two-byte ds_writes per fragment + fences) the fragment is multiplied by a 0/1
// selector matrix: used as the A operand, a fragment reads as [row = sample, k slot = feature]; with B[k slot, n] = 1 iff
// slot k holds feature n of the wanted 32-feature block, D[sample, feature] comes out in the C/D layout -- lane = feature,
// registers = samples -- which, packed to fp16 (exact: the products are x*1 and the sums have one non-zero term), IS the
// operand layout of the weight-gradient MFMA.  As in the forward pass the numbering of the k slots (here: samples) is free
// as long as both operands agree: slot (g, h, j) of 16-sample group g := sample 16g + 8(j>>2) + 4h + (j&3) = accumulator
// register 8g + j.  Two selector fragments per slot order serve all blocks.
// ------------------------------------------------------------------------------------------------
struct Selectors {
    half8_t hid[2];  // slot order of hidden operands (slot_feature): k block 2*ib + e -> features 32*ib + 16*e + ...
    half8_t nat[2];  // natural order (input features, 16*kb + 8*h + j)
    half8_t out;     // output gradient: features 8*h + j, 16 real columns
};

__device__ __forceinline__ Selectors make_selectors(int n, int h) {
    Selectors s;
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool hid = ((n >> 4) == e) && (((n >> 2) & 1) == h) && (j == ((n >> 3) & 1) * 4 + (n & 3));
            const bool nat = ((n >> 4) == e) && (((n >> 3) & 1) == h) && (j == (n & 7));
            s.hid[e][j] = hid ? (half_t)1.0f : (half_t)0.0f;
            s.nat[e][j] = nat ? (half_t)1.0f : (half_t)0.0f;
        }
#pragma unroll
    for (int j = 0; j < 8; j++) s.out[j] = ((n < 16) && ((n >> 3) == h) && (j == (n & 7))) ? (half_t)1.0f : (half_t)0.0f;
    return s;
}

// pack a transposed 32x32 block (fp32, exact fp16 values) into the two 16-sample operand fragments
__device__ __forceinline__ void pack_transposed(const float16_t& t, half8_t (&frag)[2]) {
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) frag[g][j] = (half_t)t[8 * g + j];
}

// transpose the NKB fragments of a 32-sample x WIDTH-feature operand (hidden slot order) into NIB blocks x 2 sample groups
template <int WIDTH>
__device__ __forceinline__ void transpose_hidden(const half8_t (&v)[Shape<WIDTH>::NKB], const Selectors& sel,
                                                 half8_t (&out)[Shape<WIDTH>::NIB][2]) {
#pragma unroll
    for (int ib = 0; ib < Shape<WIDTH>::NIB; ib++) {
        float16_t t = mfma(v[2 * ib], sel.hid[0], zero16());
        if constexpr (Shape<WIDTH>::NKB > 1) t = mfma(v[2 * ib + 1], sel.hid[1], t);
        pack_transposed(t, out[ib]);
    }
}

// dZ = dH * act'(stored post-activation), rounded to fp16.  The activation id is wave-uniform: branch once, not per element
// (ReLU, the only hidden activation FFMLP can select -- ffmlp.py:107 -- is a compare + select).
__device__ __noinline__ float act_backward_factor_slow(uint32_t a, float y) { return act_backward_factor(a, y); }

template <int WIDTH, bool RELU>
__device__ __forceinline__ void activation_transfer(uint32_t act, const float16_t (&acc)[Shape<WIDTH>::NIB],
                                                    const half8_t (&post)[Shape<WIDTH>::NKB], half8_t (&dz)[Shape<WIDTH>::NKB]) {
    if (RELU) {
        // dZ = post > 0 ? half(dH) : 0 on PAIRS of fp16 values, without compares or selects (round 4: the compare + convert + select +
        // pack sequence was 3.5 VALU per element and a third of the backward's instruction stream; this is 2): a stored ReLU output
        // is +0 or positive, so "non-zero" is the sign of (0 - bits) as a 16-bit integer -- an arithmetic shift spreads it over the
        // half, an AND applies it.  Bit-identical to the select for every finite stored activation (a NaN activation, which the select
        // treats as "not positive", keeps its gradient here: the step is skipped by the loss scaler either way).
        typedef short short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int kb = 0; kb < Shape<WIDTH>::NKB; kb++) {
            uint32_t out[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const half2_t h = {(half_t)acc[kb >> 1][(kb & 1) * 8 + 2 * q], (half_t)acc[kb >> 1][(kb & 1) * 8 + 2 * q + 1]};
                const half2_t y = {post[kb][2 * q], post[kb][2 * q + 1]};
                const short2_t live = ((short2_t){0, 0} - __builtin_bit_cast(short2_t, y)) >> (short2_t){15, 15};  // 0xffff where the unit fired
                out[q] = __builtin_bit_cast(uint32_t, h) & __builtin_bit_cast(uint32_t, live);
            }
            uint32_t packed[4] = {out[0], out[1], out[2], out[3]};
            __builtin_memcpy(&dz[kb], packed, sizeof(packed));
        }
    } else {
#pragma unroll
        for (int kb = 0; kb < Shape<WIDTH>::NKB; kb++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                dz[kb][j] = (half_t)(acc[kb >> 1][(kb & 1) * 8 + j] * act_backward_factor_slow(act, (float)post[kb][j]));
    }
}

// Number of fp32 words of one weight-gradient slab = number of parameters.
__host__ __device__ inline uint32_t ff_param_count(uint32_t in_dim, uint32_t hidden, uint32_t num_layers) {
    return hidden * (in_dim + hidden * (num_layers - 1) + 16);
}

// ------------------------------------------------------------------------------------------------
// Asynchronous tile prefetch (global -> LDS DMA, no VGPR round trip).
// The backward kernel keeps every weight-gradient accumulator in registers (128-192 fp32 registers), which leaves one
// wavefront per SIMD: nothing hides the HBM latency of a tile's operands (dY, the stored activations of every layer, X).
// Each wave therefore owns two LDS tile buffers and streams the NEXT tile into the idle one with global_load_lds_dwordx4
// (16 bytes per lane, landing lane-linear, which is exactly the fragment order the data is consumed in) while it computes on
// the current one.  Order per iteration: wait vmcnt(0) (current tile has landed) -> issue the next tile's loads -> compute.
// Every lane reads back only bytes it requested itself, so no cross-lane synchronisation is needed.
// Buffer layout in 1 KiB fragments: [0] dY | [1 + l*NKB + kb] post-activations of hidden layer l | then X (in_kb fragments;
// planar inputs arrive as four 4-byte rows per fragment).
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gmem_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// cache policy of the operand DMA: 2 = nt on gfx94x / gfx950.  Every operand of the backward (output gradient, stored activations, inputs) is
// read exactly once; streamed past the L2 they stop evicting what the neighbouring kernels reuse: -30 us per iteration in a same-box A/B
// (0.617 -> 0.586 ms), of which only ~5 us are this kernel's own.
constexpr int DMA_CPOL = 2;
__device__ __forceinline__ void dma16(const void* src_lane, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_ptr_t)src_lane, (lds_ptr_t)lds_wave_base, 16, 0, DMA_CPOL);
}
__device__ __forceinline__ void dma4(const void* src_lane, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_ptr_t)src_lane, (lds_ptr_t)lds_wave_base, 4, 0, DMA_CPOL);
}

// DMA instructions prefetch_tile issues for one tile -- the ONE statement of that number: the counted `s_waitcnt vmcnt(N)` of the three-deep
// pipeline (k_ffmlp_backward_paired) waits by it, and prefetch_tile returns what it really issued so that the debug build
// (make DEBUG_BOUNDS=1) traps when the two drift apart (an edit that merges or splits loads would otherwise let the barrier pass
// before the oldest tile has landed -- a silent LDS race).
template <int WIDTH>
__host__ __device__ constexpr uint32_t tile_dma_count(uint32_t stored_layers, uint32_t in_dim, bool in_planar) {
    return 1u + stored_layers * (uint32_t)Shape<WIDTH>::NKB + (in_dim / 16u) * (in_planar ? 4u : 1u);
}

template <int WIDTH>
__device__ __forceinline__ uint32_t prefetch_tile(unsigned char* buf, uint32_t tile, const half_t* __restrict__ grad, const half8_t* __restrict__ fb,
                                                  const half_t* __restrict__ inputs, uint32_t num_layers, size_t layer_stride, size_t rows,
                                                  uint32_t in_dim, bool in_planar, int lane, int n, int h) {
    constexpr int NKB = Shape<WIDTH>::NKB;
    uint32_t issued = 0u;   // (dead code unless a caller checks it)
    const size_t srow = (size_t)tile * FF_TILE + n;
    dma16(grad + srow * 16 + 8 * h, buf); issued++;
    const half8_t* frag = fb + (size_t)tile * NKB * 64 + lane;
    for (uint32_t l = 0; l < num_layers; l++)
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) { dma16(frag + l * layer_stride + kb * 64, buf + (size_t)(1 + l * NKB + kb) * 1024); issued++; }
    unsigned char* xb = buf + (size_t)(1 + num_layers * NKB) * 1024;
    const uint32_t in_kb = in_dim / 16;
    if (!in_planar) {
        for (uint32_t kb = 0; kb < in_kb; kb++) { dma16(inputs + srow * in_dim + 16 * kb + 8 * h, xb + (size_t)kb * 1024); issued++; }
    } else {
        for (uint32_t kb = 0; kb < in_kb; kb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dma4(inputs + ((size_t)((16 * kb + 8 * h) / 2 + q) * rows + srow) * 2, xb + (size_t)kb * 1024 + q * 256);
                issued++;
            }
    }
    return issued;
}

// X fragment kb of this lane from the tile buffer (see layout above)
__device__ __forceinline__ half8_t tile_x(const unsigned char* xb, uint32_t kb, bool in_planar, int lane) {
    if (!in_planar) return *reinterpret_cast<const half8_t*>(xb + (size_t)kb * 1024 + lane * 16);
    half8_t v;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const half2_t t = *reinterpret_cast<const half2_t*>(xb + (size_t)kb * 1024 + q * 256 + lane * 4);
        v[2 * q] = t.x;
        v[2 * q + 1] = t.y;
    }
    return v;
}

template <int WIDTH, int IN_JB /* ceil(in_dim/32) */, int NHM /* hidden matmuls = num_layers-1 */, bool RELU /* hidden activation is ReLU */>
__global__ __launch_bounds__(FF_THREADS) void k_ffmlp_backward(const half_t* __restrict__ grad, const half_t* __restrict__ inputs,
                                                               const half_t* __restrict__ weights, const half_t* __restrict__ forward_buffer,
                                                               uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                                                               bool with_dx, half_t* __restrict__ grad_inputs, float* __restrict__ slabs,
                                                               half_t* __restrict__ grad_weights_direct, bool in_planar, bool dx_planar, uint32_t pf_depth) {
#ifdef NGP_FF_BWD_DIAG  // timing experiments (compile-time only): 1 = no loads after the first tile, 2 = loads only
    constexpr uint32_t diag = NGP_FF_BWD_DIAG;
#else
    constexpr uint32_t diag = 0u;
#endif
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img = reinterpret_cast<half8_t*>(smem);
    const uint32_t nfrag = bwd_frag_count<WIDTH>(in_dim, num_layers, with_dx);
    build_backward_image<WIDTH>(img, weights, in_dim, num_layers, with_dx);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    (void)nfrag;
    const Selectors sel = make_selectors(n, h);
    __syncthreads();

    const uint32_t in_kb = in_dim / 16;
    const half8_t* img_out = img + lane;
    const half8_t* img_hid = img_out + (size_t)NIB * 64;
    const half8_t* img_in = img_hid + (size_t)(num_layers - 1) * NIB * NKB * 64;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;
    const half8_t* fb = reinterpret_cast<const half8_t*>(forward_buffer);
    const size_t rows = (size_t)n_tiles * FF_TILE;

    // weight-gradient accumulators (fp32): output layer [jb<NIB], hidden layers [l][ib][jb], input layer [ib][jb<IN_JB]
    float16_t gw_out[NIB];
    float16_t gw_in[NIB][IN_JB];
#pragma unroll
    for (int jb = 0; jb < NIB; jb++) gw_out[jb] = zero16();
#pragma unroll
    for (int ib = 0; ib < NIB; ib++)
#pragma unroll
        for (int jb = 0; jb < IN_JB; jb++) gw_in[ib][jb] = zero16();
    float16_t gw_hid[NHM][NIB][NIB];
#pragma unroll
    for (int l = 0; l < NHM; l++)
#pragma unroll
        for (int ib = 0; ib < NIB; ib++)
#pragma unroll
            for (int jb = 0; jb < NIB; jb++) gw_hid[l][ib][jb] = zero16();

    // per-wave tile buffers behind the weight image
    const uint32_t tile_frags = 1 + num_layers * NKB + in_dim / 16;
    unsigned char* pf_base = smem + (size_t)nfrag * 1024 + (size_t)wid * pf_depth * tile_frags * 1024;
    const uint32_t tile_step = gridDim.x * FF_WAVES;
    uint32_t cur = 0;
    {
        const uint32_t first = blockIdx.x * FF_WAVES + wid;
        if (first < n_tiles) prefetch_tile<WIDTH>(pf_base, first, grad, fb, inputs, num_layers, layer_stride, rows, in_dim, in_planar, lane, n, h);
    }
    for (uint32_t tile = blockIdx.x * FF_WAVES + wid; tile < n_tiles; tile += tile_step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's operands have landed in buffer `cur`
        const unsigned char* tb = pf_base + (size_t)cur * tile_frags * 1024;
        if (pf_depth > 1 && diag != 1u) {
            cur ^= 1u;
            if (tile + tile_step < n_tiles)
                prefetch_tile<WIDTH>(pf_base + (size_t)cur * tile_frags * 1024, tile + tile_step, grad, fb, inputs, num_layers, layer_stride,
                                     rows, in_dim, in_planar, lane, n, h);
        }
        if (diag == 2u) continue;
        const half8_t* tfrag = reinterpret_cast<const half8_t*>(tb) + lane;
        // ---- output layer -------------------------------------------------------------------
        const half8_t dy = tfrag[0];
        half8_t a_prev[NKB];  // post-activations of the layer below the one being differentiated
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) a_prev[kb] = tfrag[(1 + (num_layers - 1) * NKB + kb) * 64];
        half8_t aT[NIB][2], zT[NIB][2];
        transpose_hidden<WIDTH>(a_prev, sel, aT);
        {
            // dW_out [16(+16 zero) x WIDTH] += dY^T . A_{L-1}
            half8_t yT[2];
            pack_transposed(mfma(dy, sel.out, zero16()), yT);
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int jb = 0; jb < NIB; jb++) gw_out[jb] = mfma(yT[g], aT[jb][g], gw_out[jb]);
        }
        // dH_{L-1}^T = W_out^T . dY^T   (K = 16: one k block)
        float16_t acc[NIB];
#pragma unroll
        for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(img_out[ib * 64], dy, zero16());

        // ---- hidden layers, top down --------------------------------------------------------
        half8_t dz[NKB];
#pragma unroll
        for (int li = 0; li < NHM; li++) {  // li-th hidden matmul from the top: layer index num_layers-1-li
            // activation transfer with the stored post-activations, round to fp16
            activation_transfer<WIDTH, RELU>(act, acc, a_prev, dz);
            const uint32_t lay = num_layers - 1 - li;  // dz = dL/d(pre-activation of hidden layer `lay`)
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) a_prev[kb] = tfrag[(1 + (lay - 1) * NKB + kb) * 64];
            transpose_hidden<WIDTH>(dz, sel, zT);
            transpose_hidden<WIDTH>(a_prev, sel, aT);
            // dW_lay [WIDTH x WIDTH] += dZ_lay^T . A_{lay-1}
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                    for (int jb = 0; jb < NIB; jb++) gw_hid[li][ib][jb] = mfma(zT[ib][g], aT[jb][g], gw_hid[li][ib][jb]);
            // dH_{lay-1}^T = W_lay^T . dZ_lay^T
            const half8_t* wl = img_hid + (size_t)li * NIB * NKB * 64;
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) {
                acc[ib] = zero16();
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) acc[ib] = mfma(wl[(ib * NKB + kb) * 64], dz[kb], acc[ib]);
            }
        }
        // ---- input layer --------------------------------------------------------------------
        activation_transfer<WIDTH, RELU>(act, acc, a_prev, dz);
        transpose_hidden<WIDTH>(dz, sel, zT);
        {
            // X^T blocks of 32 input features (natural feature order), dW_in [WIDTH x in] += dZ_0^T . X
#pragma unroll
            for (int jb = 0; jb < IN_JB; jb++) {
                float16_t t = zero16();
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint32_t kb = 2 * jb + e;
                    if (kb < in_kb) t = mfma(tile_x(tb + (size_t)(1 + num_layers * NKB) * 1024, kb, in_planar, lane), sel.nat[e], t);
                }
                half8_t xT[2];
                pack_transposed(t, xT);
#pragma unroll
                for (int g = 0; g < 2; g++)
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++) gw_in[ib][jb] = mfma(zT[ib][g], xT[g], gw_in[ib][jb]);
            }
        }
        if (with_dx) {
            // dX^T [in x samples] = W_in^T . dZ_0^T ; rows beyond in_dim are zero weights and are not stored
#pragma unroll
            for (int ib = 0; ib < IN_JB; ib++) {
                float16_t dx = zero16();
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) dx = mfma(img_in[(ib * NKB + kb) * 64], dz[kb], dx);
                const size_t srow = (size_t)tile * FF_TILE + n;
                half_t* grow = grad_inputs + srow * in_dim;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t f0 = 32 * ib + 8 * q + 4 * h;
                    if (f0 < in_dim) {
                        if (dx_planar) {  // [in_dim/2][rows][2]: the layout grid_encode_backward consumes
                            const half2_t lo = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1]}, hi = {(half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                            *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2) * rows + srow) * 2) = lo;
                            *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2 + 1) * rows + srow) * 2) = hi;
                        } else {
                            half4_t v = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1], (half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                            *reinterpret_cast<half4_t*>(grow + f0) = v;
                        }
                    }
                }
            }
        }
        if (pf_depth == 1 && tile + tile_step < n_tiles) {
            // single buffer (the weight image + two stages per wave do not fit 160 KiB): the next tile can only be requested once
            // this one has been consumed -- no overlap, but every tile is loaded
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prefetch_tile<WIDTH>(pf_base, tile + tile_step, grad, fb, inputs, num_layers, layer_stride, rows, in_dim, in_planar, lane, n, h);
        }
    }

    // ---- combine the four waves' partial weight gradients in LDS, emit one slab per workgroup ----
    __syncthreads();  // every wave is done with the weight image and the stages
    float* red = reinterpret_cast<float*>(smem);
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    for (uint32_t i = threadIdx.x; i < n_params; i += FF_THREADS) red[i] = 0.0f;
    __syncthreads();
    // accumulator (row = output feature o, col = input feature i = lane&31)
    // every (row, col) of a gradient matrix belongs to exactly one lane of a wave, so the waves add their
    // partial sums one after the other without atomics: the summation order is fixed => bit-reproducible.
    // (the sixteen partial sums of a block are read together, then added and written: `red[..] += a[r]` in a loop is sixteen dependent LDS
    // round trips, the compiler cannot see that the addresses differ)
    auto flush = [&](const float16_t& a, uint32_t base, uint32_t ld, int ib, int jb, uint32_t rows, uint32_t cols) {
        float t[16];
        const uint32_t i = 32 * jb + n;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t o = (uint32_t)acc_row(ib, h, r);
            t[r] = (o < rows && i < cols) ? red[base + o * ld + i] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t o = (uint32_t)acc_row(ib, h, r);
            if (o < rows && i < cols) red[base + o * ld + i] = t[r] + a[r];
        }
    };
    for (int turn = 0; turn < FF_WAVES; turn++) {
        if (wid == turn) {
#pragma unroll
            for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                for (int jb = 0; jb < IN_JB; jb++) flush(gw_in[ib][jb], 0, in_dim, ib, jb, WIDTH, in_dim);
#pragma unroll
            for (int li = 0; li < NHM; li++) {
                const uint32_t lay = num_layers - 1 - li;
                const uint32_t base = WIDTH * in_dim + (lay - 1) * WIDTH * WIDTH;
#pragma unroll
                for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                    for (int jb = 0; jb < NIB; jb++) flush(gw_hid[li][ib][jb], base, WIDTH, ib, jb, WIDTH, WIDTH);
            }
            const uint32_t base = WIDTH * in_dim + (num_layers - 1) * WIDTH * WIDTH;
#pragma unroll
            for (int jb = 0; jb < NIB; jb++) flush(gw_out[jb], base, WIDTH, 0, jb, 16, WIDTH);
        }
        __syncthreads();
    }
    if (grad_weights_direct) {
        for (uint32_t i = threadIdx.x; i < n_params; i += FF_THREADS) grad_weights_direct[i] = (half_t)red[i];
    } else {
        float* slab = slabs + (size_t)blockIdx.x * n_params;
        for (uint32_t i = threadIdx.x; i < n_params; i += FF_THREADS) slab[i] = red[i];
    }
}

// ------------------------------------------------------------------------------------------------
// backward, PAIRED variant (networks with 2 or 3 layers: the instant-ngp ones).
//
// What bounds the single-wave kernel above is not memory but its own instruction stream: with every weight-gradient accumulator
// of the network in one wave's registers (> 256), only one wave fits a SIMD -- nothing to interleave with -- and the compiler puts
// every MFMA destination into accumulation registers, so each result that VALU code consumes costs 16 register copies (a quarter of
// the kernel's instructions).  Here two sibling waves share one stream of tiles and SPLIT the accumulators:
//   role 0 (waves 0..3): dW of the output layer and of the top hidden matmul -- needs the dgrad chain only down to there;
//   role 1 (waves 4..7): the full dgrad chain, dW of the remaining hidden matmul (3-layer nets) and of the input layer, dL/dx.
// Each role fits the 256-register budget of two waves per SIMD: MFMA results land in ordinary vector registers (no copies), and the
// two waves of a SIMD hide each other's MFMA / LDS latencies.  The duplicated part of the dgrad chain is ~10 MFMAs per tile.
// Role 0 issues the global->LDS DMA of the pair's tile buffers; one workgroup barrier per tile hands the landed tile to role 1 and
// frees the other buffer for the next prefetch.  The four pairs of a workgroup share the weight image.
// ------------------------------------------------------------------------------------------------
constexpr int FP_PAIRS = 4;
constexpr int FP_THREADS = 2 * FP_PAIRS * 64;

// Optional epilogue of the colour head of the fused network (ngp_network_backward_color): instead of storing dL/d(colour input) [M,32]
// for a separate kernel to reshuffle, role 1 writes the sigma net's output gradient grad_h16 [M,16] itself -- column 0 =
// to_half(density_scale * grad_sigma * exp(clamp(h0, -15, 15))) (trunc_exp backward, activation.py:12-17), columns 1..15 = the input
// gradient of features 16..30 (the geometric features; the SH block 0..15 has no gradient to carry, feature 31 is padding):
// pipeline.hip's k_mid_backward, bit for bit, without its launch and without the [M,32] round trip.
struct MidEpilogue {
    const float* grad_sigma;  // [M] fp32; NULL = epilogue off (the plain dL/dx store)
    const half_t* h16;        // [M,16] fp16, the sigma net's output
    half_t* grad_h16;         // [M,16] fp16, out
    float density_scale;
};

// RECOMP (round 4, fused training step only): the hidden activations are not read from a forward buffer -- the forward pass did not store
// them -- but recomputed from the tile's inputs X with the forward image of the same weights: the same MFMA sequence, the same ReLU and
// fp16 rounding as k_network_forward / k_ffmlp_forward, so bit for bit what the forward buffer would have held.  A 64-wide layer is
// 128 B per sample written by the forward pass and read again here; recomputing it is 8 MFMAs and ~50 vector instructions per 32 samples.
// The tile DMA then carries dY and X only (3 KiB instead of 11-15), both roles run the forward chain themselves: role 0 keeps the two
// layers it needs in registers, role 1 parks all of them in a private LDS scratch (each lane re-reads only what it wrote).
template <int WIDTH, int NHM, typename Sink>
__device__ __forceinline__ void recompute_hidden(const half8_t* __restrict__ fimg_lane, uint32_t in_kb, const unsigned char* xb, bool in_planar,
                                                 int lane, Sink&& sink) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    float16_t acc[NIB];
#pragma unroll
    for (int ib = 0; ib < NIB; ib++) acc[ib] = zero16();
    for (uint32_t kb = 0; kb < in_kb; kb++) {
        const half8_t x = tile_x(xb, kb, in_planar, lane);
#pragma unroll
        for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(fimg_lane[(ib * in_kb + kb) * 64], x, acc[ib]);
    }
    const half8_t* fhid = fimg_lane + (size_t)NIB * in_kb * 64;
    half8_t hid[NKB];
#pragma unroll
    for (int l = 0; l <= NHM; l++) {
        pack_hidden_relu<WIDTH>(acc, hid);
        sink(l, hid);
        if (l == NHM) break;
        const half8_t* wl = fhid + (size_t)l * NIB * NKB * 64;
#pragma unroll
        for (int ib = 0; ib < NIB; ib++) {
            acc[ib] = zero16();
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) acc[ib] = mfma(wl[(ib * NKB + kb) * 64], hid[kb], acc[ib]);
        }
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WIDTH, int IN_JB, int NHM /* 1 or 2 */, bool RELU, bool RECOMP = false>
__global__ __launch_bounds__(FP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ffmlp_backward_paired(const half_t* __restrict__ grad, const half_t* __restrict__ inputs, const half_t* __restrict__ weights,
                             const half_t* __restrict__ forward_buffer, uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                             bool with_dx, half_t* __restrict__ grad_inputs, float* __restrict__ slabs,
                             half_t* __restrict__ grad_weights_direct, bool in_planar, bool dx_planar, uint32_t pf_depth, MidEpilogue mid) {
    static_assert(NHM == 1 || NHM == 2, "the paired backward covers 2- and 3-layer networks");
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img = reinterpret_cast<half8_t*>(smem);
    const uint32_t nfrag = bwd_frag_count<WIDTH>(in_dim, num_layers, with_dx);
    const uint32_t in_kb = in_dim / 16;
    // RECOMP: the forward image (input layer + hidden matmuls, no output layer) sits behind the backward image
    const uint32_t ffrag = RECOMP ? NIB * in_kb + (num_layers - 1) * NIB * NKB : 0u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int pair = wid & (FP_PAIRS - 1), role = wid / FP_PAIRS;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;
    const half8_t* fb = reinterpret_cast<const half8_t*>(forward_buffer);
    const size_t rows = (size_t)n_tiles * FF_TILE;
    const uint32_t act_layers = RECOMP ? 0u : num_layers;  // layers of stored activations in a tile buffer
    const uint32_t tile_frags = 1 + act_layers * NKB + in_kb;
    unsigned char* pf_base = smem + (size_t)(nfrag + ffrag) * 1024 + (size_t)pair * pf_depth * tile_frags * 1024;
    // RECOMP: role 1's recomputed activations of hidden layers 0 .. NHM-1, [layer][kb] fragments behind all tile buffers
    half8_t* scratch = reinterpret_cast<half8_t*>(smem + (size_t)(nfrag + ffrag) * 1024 + (size_t)FP_PAIRS * pf_depth * tile_frags * 1024 +
                                                  (size_t)pair * NHM * NKB * 1024) + lane;
    const uint32_t base_step = gridDim.x * FP_PAIRS;
    uint32_t cur = 0;
    // DMA instructions per tile (dY, the stored activations, X as 16-byte fragments or four 4-byte rows each): the counted wait below names
    // how many YOUNGER ones may still be in flight, as an immediate -- the counts of the shapes that run three deep are enumerated
    const uint32_t tile_loads = tile_dma_count<WIDTH>(act_layers, in_dim, in_planar);
    auto prefetch = [&](uint32_t buffer, uint32_t tile) {
        const uint32_t issued = prefetch_tile<WIDTH>(pf_base + (size_t)buffer * tile_frags * 1024, tile, grad, fb, inputs, act_layers, layer_stride,
                                                     rows, in_dim, in_planar, lane, n, h);
        NGP_BOUNDS(issued == tile_loads);   // the counted vmcnt wait and the prefetch agree on the number of loads per tile
        (void)issued;
    };
    // pf_depth buffers per pair: tiles of the next pf_depth - 1 rounds are in flight while one is worked on.  The first requests leave BEFORE
    // the weight images are built (the tile buffers are a region of their own): their round trip runs under the build instead of behind it.
    if (role == 0) {
        for (uint32_t d = 0; d + 1 < (pf_depth > 1 ? pf_depth : 2u); d++)
            if (blockIdx.x * FP_PAIRS + d * base_step + pair < n_tiles) prefetch(d, blockIdx.x * FP_PAIRS + d * base_step + pair);
    }
    build_backward_image<WIDTH>(img, weights, in_dim, num_layers, with_dx);
    if (RECOMP) build_forward_image<WIDTH>(img + (size_t)nfrag * 64, weights, in_dim, num_layers, 0, ffrag);
    const Selectors sel = make_selectors(n, h);
    __syncthreads();

    const half8_t* img_out = img + lane;
    const half8_t* img_hid = img_out + (size_t)NIB * 64;
    const half8_t* img_in = img_hid + (size_t)(num_layers - 1) * NIB * NKB * 64;
    const half8_t* fimg = img + (size_t)nfrag * 64 + lane;
    static_assert(tile_dma_count<64>(0, 32, false) == 3u && tile_dma_count<64>(0, 32, true) == 9u && tile_dma_count<64>(2, 32, false) == 11u &&
                  tile_dma_count<64>(2, 32, true) == 17u, "the enumerated vmcnt immediates below are the DMA counts of the three-deep shapes");
    const bool deep = pf_depth == 3 && (tile_loads == 3u || tile_loads == 9u || tile_loads == 11u || tile_loads == 17u);
    // per tile round: the landed tile is handed over (barrier), the free buffer is refilled; returns the tile's buffer
    auto next_tile_buffer = [&](uint32_t base) -> const unsigned char* {
        if (role == 0) {
            // a tile that does not exist was not requested: its slot in the queue is missing, so only the oldest request may be waited for
            // by count while a full set of younger ones exists
            if (deep && base + base_step + pair < n_tiles) {
                switch (tile_loads) {
                    case 3u: wait_vmcnt<3>(); break;     // recomputing, 32 row-major inputs
                    case 9u: wait_vmcnt<9>(); break;     // recomputing, 32 planar inputs
                    case 11u: wait_vmcnt<11>(); break;   // two stored layers, 32 row-major inputs
                    default: wait_vmcnt<17>(); break;    // two stored layers, 32 planar inputs (the sigma network of the training step)
                }
            } else {
                wait_vmcnt<0>();
            }
        }
        __syncthreads();
        const unsigned char* tb = pf_base + (size_t)cur * tile_frags * 1024;
        if (pf_depth > 1) {
            const uint32_t ahead = pf_depth - 1;
            const uint32_t fill = cur == 0 ? pf_depth - 1 : cur - 1;   // the buffer of the round before this one: both roles are done with it
            cur = cur + 1 == pf_depth ? 0u : cur + 1;
            if (role == 0 && base + ahead * base_step + pair < n_tiles) prefetch(fill, base + ahead * base_step + pair);
        }
        return tb;
    };
    auto end_of_round = [&](uint32_t base) {  // single buffer: refill only after both roles are done with it
        if (pf_depth == 1) {
            __syncthreads();
            if (role == 0 && base + base_step + pair < n_tiles) prefetch(0, base + base_step + pair);
        }
    };
    float* red = reinterpret_cast<float*>(smem);
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    // (round 4) the sixteen partial sums of a block are READ together, then added and written: written as `red[..] += a[r]` the compiler
    // has to order every read behind the previous write (it cannot see that the sixteen addresses differ), and the epilogue became
    // sixteen dependent LDS round trips per block and wave -- with one wave flushing at a time, ~20 of the kernel's 47 us.
    auto flush = [&](const float16_t& a, uint32_t base, uint32_t ld, int ib, int jb, uint32_t rws, uint32_t cols) {
        float t[16];
        const uint32_t i = 32 * jb + n;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t o = (uint32_t)acc_row(ib, h, r);
            t[r] = (o < rws && i < cols) ? red[base + o * ld + i] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t o = (uint32_t)acc_row(ib, h, r);
            if (o < rws && i < cols) red[base + o * ld + i] = t[r] + a[r];
        }
    };
    auto begin_reduction = [&]() {
        __syncthreads();  // every wave is done with the weight image and the stages
        for (uint32_t i = threadIdx.x; i < n_params; i += FP_THREADS) red[i] = 0.0f;
        __syncthreads();
    };
    const uint32_t base_out = WIDTH * in_dim + (num_layers - 1) * WIDTH * WIDTH;
    const uint32_t base_top = WIDTH * in_dim + (num_layers - 2) * WIDTH * WIDTH;  // hidden matmul into the last hidden layer

    if (role == 0) {
        // ---- dW_out and dW of the top hidden matmul ----
        float16_t gw_out[NIB], gw_top[NIB][NIB];
#pragma unroll
        for (int jb = 0; jb < NIB; jb++) gw_out[jb] = zero16();
#pragma unroll
        for (int ib = 0; ib < NIB; ib++)
#pragma unroll
            for (int jb = 0; jb < NIB; jb++) gw_top[ib][jb] = zero16();
        for (uint32_t base = blockIdx.x * FP_PAIRS; base < n_tiles; base += base_step) {
            const unsigned char* tb = next_tile_buffer(base);
            if (base + pair < n_tiles) {
                const half8_t* tfrag = reinterpret_cast<const half8_t*>(tb) + lane;
                const half8_t dy = tfrag[0];
                half8_t a_top[NKB], a_below[NKB];
                if constexpr (RECOMP) {
                    recompute_hidden<WIDTH, NHM>(fimg, in_kb, tb + 1024, in_planar, lane, [&](int l, const half8_t (&hid)[NKB]) {
                        if (l == NHM) {
#pragma unroll
                            for (int kb = 0; kb < NKB; kb++) a_top[kb] = hid[kb];
                        } else if (l == NHM - 1) {
#pragma unroll
                            for (int kb = 0; kb < NKB; kb++) a_below[kb] = hid[kb];
                        }
                    });
                } else {
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) a_top[kb] = tfrag[(1 + NHM * NKB + kb) * 64];
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) a_below[kb] = tfrag[(1 + (NHM - 1) * NKB + kb) * 64];
                }
                half8_t aT[NIB][2], zT[NIB][2];
                transpose_hidden<WIDTH>(a_top, sel, aT);
                {
                    half8_t yT[2];
                    pack_transposed(mfma(dy, sel.out, zero16()), yT);
#pragma unroll
                    for (int g = 0; g < 2; g++)
#pragma unroll
                        for (int jb = 0; jb < NIB; jb++) gw_out[jb] = mfma(yT[g], aT[jb][g], gw_out[jb]);
                }
                float16_t acc[NIB];
#pragma unroll
                for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(img_out[ib * 64], dy, zero16());
                half8_t dz[NKB];
                activation_transfer<WIDTH, RELU>(act, acc, a_top, dz);
                transpose_hidden<WIDTH>(dz, sel, zT);
                transpose_hidden<WIDTH>(a_below, sel, aT);
#pragma unroll
                for (int g = 0; g < 2; g++)
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                        for (int jb = 0; jb < NIB; jb++) gw_top[ib][jb] = mfma(zT[ib][g], aT[jb][g], gw_top[ib][jb]);
            }
            end_of_round(base);
        }
        begin_reduction();
        // the waves of a role add into the same entries, one after the other (a fixed order: deterministic); the two roles own disjoint
        // parts of the slab and take their turns at the same time
        for (int turn = 0; turn < FP_PAIRS; turn++) {
            if (pair == turn) {
#pragma unroll
                for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                    for (int jb = 0; jb < NIB; jb++) flush(gw_top[ib][jb], base_top, WIDTH, ib, jb, WIDTH, WIDTH);
#pragma unroll
                for (int jb = 0; jb < NIB; jb++) flush(gw_out[jb], base_out, WIDTH, 0, jb, 16, WIDTH);
            }
            __syncthreads();
        }
    } else {
        // ---- the full dgrad chain, dW of the lower hidden matmul (3-layer nets) and of the input layer, dL/dx ----
        float16_t gw_in[NIB][IN_JB], gw_low[NHM == 2 ? NIB : 1][NHM == 2 ? NIB : 1];
#pragma unroll
        for (int ib = 0; ib < NIB; ib++)
#pragma unroll
            for (int jb = 0; jb < IN_JB; jb++) gw_in[ib][jb] = zero16();
        if constexpr (NHM == 2) {
#pragma unroll
            for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                for (int jb = 0; jb < NIB; jb++) gw_low[ib][jb] = zero16();
        }
        for (uint32_t base = blockIdx.x * FP_PAIRS; base < n_tiles; base += base_step) {
            const unsigned char* tb = next_tile_buffer(base);
            const uint32_t tile = base + pair;
            if (tile < n_tiles) {
                const half8_t* tfrag = reinterpret_cast<const half8_t*>(tb) + lane;
                const half8_t dy = tfrag[0];
                half8_t a_prev[NKB];
                // post-activations of hidden layer l: the tile buffer, or this wave's recomputed copy
                const half8_t* afrag = RECOMP ? scratch : tfrag + 64;
                if constexpr (RECOMP) {
                    recompute_hidden<WIDTH, NHM>(fimg, in_kb, tb + 1024, in_planar, lane, [&](int l, const half8_t (&hid)[NKB]) {
                        if (l == NHM) {
#pragma unroll
                            for (int kb = 0; kb < NKB; kb++) a_prev[kb] = hid[kb];
                        } else {
#pragma unroll
                            for (int kb = 0; kb < NKB; kb++) scratch[(l * NKB + kb) * 64] = hid[kb];
                        }
                    });
                    // the lower layers come back from LDS when they are needed: forwarding the stored registers would keep 16-32 more of
                    // them alive through the dgrad chain, which has none to spare
                    asm volatile("" ::: "memory");
                } else {
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) a_prev[kb] = afrag[(NHM * NKB + kb) * 64];
                }
                float16_t acc[NIB];
#pragma unroll
                for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(img_out[ib * 64], dy, zero16());
                half8_t dz[NKB];
                half8_t aT[NIB][2], zT[NIB][2];
                // top hidden matmul: dgrad only (its dW belongs to role 0)
                activation_transfer<WIDTH, RELU>(act, acc, a_prev, dz);
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) a_prev[kb] = afrag[((NHM - 1) * NKB + kb) * 64];
#pragma unroll
                for (int ib = 0; ib < NIB; ib++) {
                    acc[ib] = zero16();
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) acc[ib] = mfma(img_hid[(ib * NKB + kb) * 64], dz[kb], acc[ib]);
                }
                if constexpr (NHM == 2) {  // lower hidden matmul: dW and dgrad
                    activation_transfer<WIDTH, RELU>(act, acc, a_prev, dz);
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) a_prev[kb] = afrag[kb * 64];
                    transpose_hidden<WIDTH>(dz, sel, zT);
                    transpose_hidden<WIDTH>(a_prev, sel, aT);
#pragma unroll
                    for (int g = 0; g < 2; g++)
#pragma unroll
                        for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                            for (int jb = 0; jb < NIB; jb++) gw_low[ib][jb] = mfma(zT[ib][g], aT[jb][g], gw_low[ib][jb]);
                    const half8_t* wl = img_hid + (size_t)NIB * NKB * 64;
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++) {
                        acc[ib] = zero16();
#pragma unroll
                        for (int kb = 0; kb < NKB; kb++) acc[ib] = mfma(wl[(ib * NKB + kb) * 64], dz[kb], acc[ib]);
                    }
                }
                // input layer
                activation_transfer<WIDTH, RELU>(act, acc, a_prev, dz);
                transpose_hidden<WIDTH>(dz, sel, zT);
#pragma unroll
                for (int jb = 0; jb < IN_JB; jb++) {
                    float16_t t = zero16();
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const uint32_t kb = 2 * jb + e;
                        if (kb < in_kb) t = mfma(tile_x(tb + (size_t)(1 + act_layers * NKB) * 1024, kb, in_planar, lane), sel.nat[e], t);
                    }
                    half8_t xT[2];
                    pack_transposed(t, xT);
#pragma unroll
                    for (int g = 0; g < 2; g++)
#pragma unroll
                        for (int ib = 0; ib < NIB; ib++) gw_in[ib][jb] = mfma(zT[ib][g], xT[g], gw_in[ib][jb]);
                }
                if (with_dx && mid.grad_sigma) {
                    if constexpr (IN_JB == 1) {
                        // features 16..31 of this lane's sample: q = 2 -> 16 + 4h .. 19 + 4h, q = 3 -> 24 + 4h .. 27 + 4h; the output row is
                        // [gs, f16 .. f30] in four 8-byte pieces -- h = 0 stores pieces 0 and 2, h = 1 pieces 1 and 3; each needs one
                        // value of the sibling lane (f19 / f23 and f27): lane ^ 32
                        float16_t dx = zero16();
#pragma unroll
                        for (int kb = 0; kb < NKB; kb++) dx = mfma(img_in[kb * 64], dz[kb], dx);
                        const size_t srow = (size_t)tile * FF_TILE + n;
                        half_t v2[4], v3[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) { v2[j] = (half_t)dx[8 + j]; v3[j] = (half_t)dx[12 + j]; }
                        const float sib2 = __shfl_xor((float)v2[3], 32, 64), sib3 = __shfl_xor((float)v3[3], 32, 64);
                        half4_t a, b;
                        if (h == 0) {
                            const float x0 = (float)mid.h16[srow * 16];
                            const float gs = (mid.density_scale * mid.grad_sigma[srow]) * expf(fminf(15.0f, fmaxf(-15.0f, x0)));
                            a = half4_t{to_half_rne(gs), v2[0], v2[1], v2[2]};          // columns 0..3
                            b = half4_t{(half_t)sib2, v3[0], v3[1], v3[2]};             // columns 8..11 (f23 from the sibling)
                        } else {
                            a = half4_t{(half_t)sib2, v2[0], v2[1], v2[2]};             // columns 4..7 (f19 from the sibling)
                            b = half4_t{(half_t)sib3, v3[0], v3[1], v3[2]};             // columns 12..15 (f27 from the sibling)
                        }
                        half_t* grow = mid.grad_h16 + srow * 16 + 4 * h;
                        *reinterpret_cast<half4_t*>(grow) = a;
                        *reinterpret_cast<half4_t*>(grow + 8) = b;
                    }
                } else if (with_dx) {
#pragma unroll
                    for (int ib = 0; ib < IN_JB; ib++) {
                        float16_t dx = zero16();
#pragma unroll
                        for (int kb = 0; kb < NKB; kb++) dx = mfma(img_in[(ib * NKB + kb) * 64], dz[kb], dx);
                        const size_t srow = (size_t)tile * FF_TILE + n;
                        half_t* grow = grad_inputs + srow * in_dim;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t f0 = 32 * ib + 8 * q + 4 * h;
                            if (f0 < in_dim) {
                                if (dx_planar) {
                                    const half2_t lo = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1]}, hi = {(half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                                    *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2) * rows + srow) * 2) = lo;
                                    *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2 + 1) * rows + srow) * 2) = hi;
                                } else {
                                    half4_t v = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1], (half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                                    *reinterpret_cast<half4_t*>(grow + f0) = v;
                                }
                            }
                        }
                    }
                }
            }
            end_of_round(base);
        }
        begin_reduction();
        for (int turn = 0; turn < FP_PAIRS; turn++) {
            if (pair == turn) {
#pragma unroll
                for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                    for (int jb = 0; jb < IN_JB; jb++) flush(gw_in[ib][jb], 0, in_dim, ib, jb, WIDTH, in_dim);
                if constexpr (NHM == 2) {
                    const uint32_t base_low = WIDTH * in_dim;  // hidden matmul into hidden layer 1
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++)
#pragma unroll
                        for (int jb = 0; jb < NIB; jb++) flush(gw_low[ib][jb], base_low, WIDTH, ib, jb, WIDTH, WIDTH);
                }
            }
            __syncthreads();
        }
    }
    if (grad_weights_direct) {
        for (uint32_t i = threadIdx.x; i < n_params; i += FP_THREADS) grad_weights_direct[i] = (half_t)red[i];
    } else {
        float* slab = slabs + (size_t)blockIdx.x * n_params;
        for (uint32_t i = threadIdx.x; i < n_params; i += FP_THREADS) slab[i] = red[i];
    }
}


// ------------------------------------------------------------------------------------------------
// LAYERED kernels: every shape the reference's API accepts beyond the register-resident fast paths above -- hidden_dim 16 / 128 /
// 256, more than 4 hidden layers, wide inputs (ffmlp.py:112-115, ffmlp.cu:543-556,653-658,830-835).
// A 256-wide layer is 128 KiB of fp16 weights: only ONE matmul's fragment image fits the 160 KiB LDS.  So the loops are turned inside
// out: a workgroup walks the network matmul by matmul, keeps that matmul's image in LDS, and pushes all of ITS tiles through it; the
// activations between matmuls travel through the caller's buffers in the private fragment order (forward_buffer in training -- it
// has to be written anyway --, the reference's inference_buffer [B, hidden] in place otherwise; backward_buffer holds the dZ of every
// layer, exactly what the reference keeps there).  A wave re-reads only fragments it wrote itself, a __threadfence() between passes
// orders them.  The weight gradients are their own kernel (k_ffmlp_wgrad): one workgroup per (matmul, 32-row block, 8 column
// blocks, sample chunk), operands transposed on the matrix core as in the fast kernel, per-chunk fp32 slabs in a caller-provided
// workspace summed in a fixed order (k_ffmlp_reduce_slabs) -- or, without a workspace, one chunk and a direct store.  Deterministic.
// ------------------------------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ void fwd_matmul_range(uint32_t m, uint32_t in_dim, uint32_t num_layers, uint32_t& first, uint32_t& count) {
    constexpr uint32_t NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const uint32_t in_kb = in_dim / 16;
    if (m == 0) { first = 0; count = NIB * in_kb; }
    else if (m < num_layers) { first = NIB * in_kb + (m - 1) * NIB * NKB; count = NIB * NKB; }
    else { first = NIB * in_kb + (num_layers - 1) * NIB * NKB; count = NKB; }
}

template <int WIDTH, bool TRAIN>
__global__ __launch_bounds__(FF_THREADS) void k_ffmlp_forward_layered(const half_t* __restrict__ inputs, const half_t* __restrict__ weights,
                                                                      half_t* __restrict__ buffer, half_t* __restrict__ outputs, uint32_t n_tiles,
                                                                      uint32_t in_dim, uint32_t num_layers, uint32_t act, uint32_t out_act,
                                                                      bool in_planar) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img = reinterpret_cast<half8_t*>(smem);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const uint32_t in_kb = in_dim / 16;
    const size_t rows = (size_t)n_tiles * FF_TILE;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;  // half8 units
    half8_t* buf = reinterpret_cast<half8_t*>(buffer);
    const half8_t* a = img + lane;
    for (uint32_t m = 0; m <= num_layers; m++) {
        uint32_t first, count;
        fwd_matmul_range<WIDTH>(m, in_dim, num_layers, first, count);
        __syncthreads();  // the previous matmul's readers are done with the image
        build_forward_image<WIDTH>(img, weights, in_dim, num_layers, first, count);
        __syncthreads();
        for (uint32_t tile = blockIdx.x * FF_WAVES + wid; tile < n_tiles; tile += gridDim.x * FF_WAVES) {
                const half8_t* src = buf + (TRAIN ? (size_t)(m ? m - 1 : 0) * layer_stride : 0) + (size_t)tile * NKB * 64 + lane;  // m >= 1 only
            if (m == num_layers) {  // output layer: one 32-row block, rows 0..15 real
                float16_t o = zero16();
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) o = mfma(a[kb * 64], src[kb * 64], o);
                half4_t lo, hi;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    lo[c] = (half_t)act_forward(out_act, o[c]);
                    hi[c] = (half_t)act_forward(out_act, o[4 + c]);
                }
                half_t* orow = outputs + ((size_t)tile * FF_TILE + n) * 16 + 4 * h;
                *reinterpret_cast<half4_t*>(orow) = lo;
                *reinterpret_cast<half4_t*>(orow + 8) = hi;
                continue;
            }
            float16_t acc[NIB];
#pragma unroll
            for (int ib = 0; ib < NIB; ib++) acc[ib] = zero16();
            if (m == 0) {
                const size_t srow = (size_t)tile * FF_TILE + n;
                for (uint32_t kb = 0; kb < in_kb; kb++) {
                    const half8_t x = load_features8(inputs, in_planar, rows, srow, in_dim, 16 * kb + 8 * h);
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(a[(ib * in_kb + kb) * 64], x, acc[ib]);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) {
                    const half8_t hk = src[kb * 64];
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(a[(ib * NKB + kb) * 64], hk, acc[ib]);
                }
            }
            half8_t* dst = buf + (TRAIN ? (size_t)m * layer_stride : 0) + (size_t)tile * NKB * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) {
                half8_t f;
#pragma unroll
                for (int j = 0; j < 8; j++) f[j] = (half_t)act_forward(act, acc[kb >> 1][(kb & 1) * 8 + j]);
                dst[kb * 64] = f;
            }
        }
        __threadfence();
    }
}

template <int WIDTH, bool RELU>
__global__ __launch_bounds__(FF_THREADS) void k_ffmlp_dgrad_layered(const half_t* __restrict__ grad, const half_t* __restrict__ weights,
                                                                    const half_t* __restrict__ forward_buffer, half_t* __restrict__ backward_buffer,
                                                                    uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, uint32_t act, bool with_dx,
                                                                    half_t* __restrict__ grad_inputs, bool dx_planar) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8_t* img = reinterpret_cast<half8_t*>(smem);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const size_t rows = (size_t)n_tiles * FF_TILE;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;
    const half8_t* fb = reinterpret_cast<const half8_t*>(forward_buffer);
    half8_t* bb = reinterpret_cast<half8_t*>(backward_buffer);
    const half8_t* a = img + lane;
    const uint32_t in_jb = (in_dim + 31) / 32;
    const uint32_t passes = num_layers + (with_dx ? 1u : 0u);
    // pass 0: W_out^T . dY -> dZ of the top hidden layer; pass p (1..nl-1): W_h[nl-p]^T . dZ_{nl-p} -> dZ_{nl-1-p}; pass nl: W_in^T . dZ_0 -> dX
    for (uint32_t p = 0; p < passes; p++) {
        uint32_t first, count;
        if (p == 0) { first = 0; count = NIB; }
        else if (p < num_layers) { first = NIB + (p - 1) * NIB * NKB; count = NIB * NKB; }
        else { first = NIB + (num_layers - 1) * NIB * NKB; count = in_jb * NKB; }
        __syncthreads();
        build_backward_image<WIDTH>(img, weights, in_dim, num_layers, with_dx, first, count);
        __syncthreads();
        for (uint32_t tile = blockIdx.x * FF_WAVES + wid; tile < n_tiles; tile += gridDim.x * FF_WAVES) {
            const size_t srow = (size_t)tile * FF_TILE + n;
            const half8_t* src = bb + (size_t)(p ? num_layers - p : 0) * layer_stride + (size_t)tile * NKB * 64 + lane;  // dZ of layer nl-p (p >= 1)
            if (p == num_layers) {
                for (uint32_t ib = 0; ib < in_jb; ib++) {
                    float16_t dx = zero16();
#pragma unroll
                    for (int kb = 0; kb < NKB; kb++) dx = mfma(a[(ib * NKB + kb) * 64], src[kb * 64], dx);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t f0 = 32 * ib + 8 * q + 4 * h;
                        if (f0 < in_dim) {
                            if (dx_planar) {
                                const half2_t lo = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1]}, hi = {(half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                                *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2) * rows + srow) * 2) = lo;
                                *reinterpret_cast<half2_t*>(grad_inputs + ((size_t)(f0 / 2 + 1) * rows + srow) * 2) = hi;
                            } else {
                                half4_t v = {(half_t)dx[4 * q], (half_t)dx[4 * q + 1], (half_t)dx[4 * q + 2], (half_t)dx[4 * q + 3]};
                                *reinterpret_cast<half4_t*>(grad_inputs + srow * in_dim + f0) = v;
                            }
                        }
                    }
                }
                continue;
            }
            float16_t acc[NIB];
            if (p == 0) {
                const half8_t dy = *reinterpret_cast<const half8_t*>(grad + srow * 16 + 8 * h);
#pragma unroll
                for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(a[ib * 64], dy, zero16());
            } else {
#pragma unroll
                for (int ib = 0; ib < NIB; ib++) acc[ib] = zero16();
#pragma unroll
                for (int kb = 0; kb < NKB; kb++) {
                    const half8_t d = src[kb * 64];
#pragma unroll
                    for (int ib = 0; ib < NIB; ib++) acc[ib] = mfma(a[(ib * NKB + kb) * 64], d, acc[ib]);
                }
            }
            const uint32_t target = num_layers - 1 - p;
            const half8_t* post = fb + (size_t)target * layer_stride + (size_t)tile * NKB * 64 + lane;
            half8_t* dst = bb + (size_t)target * layer_stride + (size_t)tile * NKB * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < NKB; kb++) {
                const half8_t y = post[kb * 64];
                half8_t dz;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float g = acc[kb >> 1][(kb & 1) * 8 + j];
                    dz[j] = RELU ? ((float)y[j] > 0.0f ? (half_t)g : (half_t)0.0f) : (half_t)(g * act_backward_factor_slow(act, (float)y[j]));
                }
                dst[kb * 64] = dz;
            }
        }
        __threadfence();
    }
}

// weight gradients of the layered path.  job = (matmul m, 32-row block ib, group of WG_NJB column blocks); blockIdx.x = sample chunk.
constexpr int WG_NJB = 8;
__host__ __device__ inline uint32_t wgrad_jobs_of(uint32_t row_blocks, uint32_t col_blocks) { return row_blocks * ((col_blocks + WG_NJB - 1) / WG_NJB); }

template <int WIDTH>
__global__ __launch_bounds__(FF_THREADS) void k_ffmlp_wgrad(const half_t* __restrict__ grad, const half_t* __restrict__ inputs,
                                                            const half_t* __restrict__ forward_buffer, const half_t* __restrict__ backward_buffer,
                                                            uint32_t n_tiles, uint32_t in_dim, uint32_t num_layers, bool in_planar,
                                                            uint32_t tiles_per_chunk, float* __restrict__ slabs,
                                                            half_t* __restrict__ grad_weights_direct) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    __shared__ float red[32][WG_NJB * 32];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const Selectors sel = make_selectors(n, h);
    const uint32_t in_kb = in_dim / 16, in_jb = (in_dim + 31) / 32;
    // decode the job
    uint32_t job = blockIdx.y, m = 0, row_blocks = NIB, col_blocks = in_jb;
    for (;; m++) {
        row_blocks = m == num_layers ? 1u : (uint32_t)NIB;
        col_blocks = m == 0 ? in_jb : (uint32_t)NIB;
        const uint32_t jobs = wgrad_jobs_of(row_blocks, col_blocks);
        if (job < jobs) break;
        job -= jobs;
    }
    const uint32_t groups = (col_blocks + WG_NJB - 1) / WG_NJB;
    const uint32_t ib = job / groups, jg = job % groups;
    const size_t rows = (size_t)n_tiles * FF_TILE;
    const size_t layer_stride = (size_t)n_tiles * NKB * 64;
    const half8_t* fb = reinterpret_cast<const half8_t*>(forward_buffer);
    const half8_t* bb = reinterpret_cast<const half8_t*>(backward_buffer);

    float16_t acc[WG_NJB];
#pragma unroll
    for (int j = 0; j < WG_NJB; j++) acc[j] = zero16();
    const uint32_t t0 = blockIdx.x * tiles_per_chunk, t1 = t0 + tiles_per_chunk < n_tiles ? t0 + tiles_per_chunk : n_tiles;
    for (uint32_t tile = t0 + wid; tile < t1; tile += FF_WAVES) {
        const size_t srow = (size_t)tile * FF_TILE + n;
        half8_t zT[2];
        if (m == num_layers) {
            const half8_t dy = *reinterpret_cast<const half8_t*>(grad + srow * 16 + 8 * h);
            pack_transposed(mfma(dy, sel.out, zero16()), zT);
        } else {
            const half8_t* z = bb + (size_t)m * layer_stride + (size_t)tile * NKB * 64 + lane;
            float16_t t = mfma(z[(2 * ib) * 64], sel.hid[0], zero16());
            if (2 * ib + 1 < (uint32_t)NKB) t = mfma(z[(2 * ib + 1) * 64], sel.hid[1], t);
            pack_transposed(t, zT);
        }
#pragma unroll
        for (int j = 0; j < WG_NJB; j++) {
            const uint32_t jb = jg * WG_NJB + j;
            if (jb < col_blocks) {
                float16_t t = zero16();
                if (m == 0) {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const uint32_t kb = 2 * jb + e;
                        if (kb < in_kb) t = mfma(load_features8(inputs, in_planar, rows, srow, in_dim, 16 * kb + 8 * h), sel.nat[e], t);
                    }
                } else {
                    const half8_t* av = fb + (size_t)(m - 1) * layer_stride + (size_t)tile * NKB * 64 + lane;
                    t = mfma(av[(2 * jb) * 64], sel.hid[0], t);
                    if (2 * jb + 1 < (uint32_t)NKB) t = mfma(av[(2 * jb + 1) * 64], sel.hid[1], t);
                }
                half8_t aT[2];
                pack_transposed(t, aT);
                acc[j] = mfma(zT[0], aT[0], acc[j]);
                acc[j] = mfma(zT[1], aT[1], acc[j]);
            }
        }
    }
    // combine the four waves in a fixed order, then store this job's rectangle of the parameter vector
    for (uint32_t i = threadIdx.x; i < 32 * WG_NJB * 32; i += FF_THREADS) (&red[0][0])[i] = 0.0f;
    __syncthreads();
    for (int turn = 0; turn < FF_WAVES; turn++) {
        if (wid == turn) {
#pragma unroll
            for (int j = 0; j < WG_NJB; j++) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; r++) t[r] = red[acc_row(0, h, r)][32 * j + n];
#pragma unroll
                for (int r = 0; r < 16; r++) red[acc_row(0, h, r)][32 * j + n] = t[r] + acc[j][r];
            }
        }
        __syncthreads();
    }
    const uint32_t n_rows = m == num_layers ? 16u : (uint32_t)WIDTH, n_cols = m == 0 ? in_dim : (uint32_t)WIDTH;
    const uint32_t base = m == 0 ? 0u : (uint32_t)WIDTH * in_dim + (m - 1) * (uint32_t)WIDTH * WIDTH;
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    for (uint32_t i = threadIdx.x; i < 32 * WG_NJB * 32; i += FF_THREADS) {
        const uint32_t r = i / (WG_NJB * 32), c = i % (WG_NJB * 32);
        const uint32_t o = 32 * ib + r, col = jg * WG_NJB * 32 + c;
        if (o < n_rows && col < n_cols) {
            const uint32_t idx = base + o * n_cols + col;
            if (grad_weights_direct) grad_weights_direct[idx] = (half_t)red[r][c];
            else slabs[(size_t)blockIdx.x * n_params + idx] = red[r][c];
        }
    }
}

// sum the per-workgroup slabs in a fixed order and round once to fp16.
// One workgroup = 64 consecutive parameters x 16 slab groups: thread (g, i) adds slabs g, g+16, g+32, ... of parameter i
// (coalesced 256-byte rows), the 16 partial sums are combined in LDS in ascending g -- a fixed summation tree, so the
// result is bit-reproducible -- and rounded once.
__global__ __launch_bounds__(RS_PARAMS * RS_GROUPS) void k_ffmlp_reduce_slabs(const float* __restrict__ slabs, uint32_t n_slabs,
                                                                               uint32_t n_params, half_t* __restrict__ grad_weights) {
    __shared__ float part[RS_GROUPS][RS_PARAMS];
    const uint32_t li = threadIdx.x & (RS_PARAMS - 1), g = threadIdx.x / RS_PARAMS;
    const uint32_t i = blockIdx.x * RS_PARAMS + li;
    float s = 0.0f;
    if (i < n_params)
        for (uint32_t k = g; k < n_slabs; k += RS_GROUPS) s += slabs[(size_t)k * n_params + i];
    part[g][li] = s;
    __syncthreads();
    if (g == 0 && i < n_params) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < RS_GROUPS; q++) t += part[q][li];
        grad_weights[i] = (half_t)t;
    }
}

// the same for TWO slab sets in one launch (the two MLPs of the fused network): workgroups [0, blocks_a) take set a, the rest set b
// (common.h: slab_reduce_block -- shared with the grid encoder's slice accumulate, which can carry these blocks in its own grid)
__global__ __launch_bounds__(RS_PARAMS * RS_GROUPS) void k_ffmlp_reduce_slabs_pair(SlabSets sets, float* __restrict__ found_inf) {
    __shared__ float part[RS_GROUPS][RS_PARAMS];
    slab_reduce_block(sets, blockIdx.x, part, found_inf);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DeviceInfo {
    int cus = 0;
    bool ok = false;
};
static DeviceInfo device_info() {
    static thread_local DeviceInfo info;
    if (!info.ok) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
                info.cus = p.multiProcessorCount;
                info.ok = true;
            }
        }
        if (!info.ok) info.cus = 256;
    }
    return info;
}

static int check_ff_args(const char* fn, uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t hidden, uint32_t num_layers) {
    // the reference's own checks (ffmlp.py:112-115, ffmlp.cu:543-556,653-658)
    NGP_REQUIRE(hidden == 16 || hidden == 32 || hidden == 64 || hidden == 128 || hidden == 256, NGP_ERR_INVALID,
                "%s: hidden_dim should in [16, 32, 64, 128, 256], but got %u", fn, hidden);
    NGP_REQUIRE(in_dim > 0 && in_dim % 16 == 0, NGP_ERR_INVALID, "%s: input_dim should be 16 * m (m > 0), but got %u", fn, in_dim);
    NGP_REQUIRE(out_dim == 16, NGP_ERR_INVALID, "%s: output_dim must be padded to 16 by the caller (got %u)", fn, out_dim);
    NGP_REQUIRE(num_layers >= 2, NGP_ERR_INVALID, "%s: num_layers should be larger than 2 (3 matmuls), but got %u", fn, num_layers);
    NGP_REQUIRE(B % 128 == 0, NGP_ERR_INVALID, "%s: batch size must be 128 * m, but got %u", fn, B);
    return NGP_OK;
}

static int raise_lds(const void* kern, size_t lds, const char* what) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit: %s", what, hipGetErrorString(e));
    }
    return NGP_OK;
}

template <int WIDTH>
static size_t forward_image_bytes(uint32_t in_dim, uint32_t num_layers) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    return (size_t)(NIB * (in_dim / 16) + (num_layers - 1) * NIB * NKB + NKB) * 1024;
}

// layered forward: one matmul image at a time (any width / depth whose single largest matmul fits the LDS)
template <int WIDTH, bool TRAIN>
static int launch_forward_layered(const void* inputs, const void* weights, uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                                  uint32_t out_act, void* buffer, void* outputs, uint32_t flags, hipStream_t st) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const uint32_t n_tiles = B / FF_TILE;
    uint32_t frags = NIB * (in_dim / 16);
    if (frags < (uint32_t)(NIB * NKB)) frags = NIB * NKB;
    const size_t lds = (size_t)frags * 1024;
    NGP_REQUIRE(lds <= 152 * 1024, NGP_ERR_INVALID, "ffmlp: one %u x %u layer (%zu B) exceeds the LDS of a CU", (unsigned)WIDTH, in_dim, lds);
    NGP_REQUIRE(buffer, NGP_ERR_INVALID, "ffmlp: this network shape needs the %s the reference passes", TRAIN ? "forward_buffer" : "inference_buffer [B, hidden]");
    auto kern = k_ffmlp_forward_layered<WIDTH, TRAIN>;
    int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, "ffmlp");
    if (rc) return rc;
    const uint32_t per_cu = (uint32_t)((160 * 1024) / (lds + 1024));
    uint32_t blocks = (uint32_t)device_info().cus * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
    const uint32_t need = cdiv(n_tiles, FF_WAVES);
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(FF_THREADS), lds, st, (const half_t*)inputs, (const half_t*)weights, (half_t*)buffer,
                       (half_t*)outputs, n_tiles, in_dim, num_layers, act, out_act, (flags & NGP_FF_INPUT_PLANAR) != 0);
    return check_launch(TRAIN ? "ffmlp_forward(layered)" : "ffmlp_inference(layered)");
}

template <int WIDTH, bool TRAIN>
static int launch_forward(const void* inputs, const void* weights, uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                          uint32_t out_act, void* fwd, void* outputs, uint32_t flags, hipStream_t st) {
    const uint32_t n_tiles = B / FF_TILE;
    const size_t lds = forward_image_bytes<WIDTH>(in_dim, num_layers);
    // register-resident kernel: the whole network's fragment image in LDS (widths up to 128); everything else walks the network
    // matmul by matmul (k_ffmlp_forward_layered)
    if (WIDTH > 128 || lds > 152 * 1024 || (flags & NGP_FF_LAYERED))
        return launch_forward_layered<WIDTH, TRAIN>(inputs, weights, B, in_dim, num_layers, act, out_act, fwd, outputs, flags, st);
    // (256-wide layers never get here: the register-resident kernel is not even instantiated for them -- it would need 128 accumulator
    // and 64 operand registers per lane and spilled 29-176 of them)
    if constexpr (WIDTH > 128) {
        return NGP_ERR_INVALID;
    } else {
    // the specialisation without the other activations' code is a fifth of the size (instruction fetch at kernel start matters for a
    // 25 us kernel)
    const bool plain = act == ACT_RELU && out_act == ACT_NONE;
    const void* kern;
    if constexpr (WIDTH > 64) kern = plain ? reinterpret_cast<const void*>(k_ffmlp_forward_wide<WIDTH, TRAIN, true>)
                                           : reinterpret_cast<const void*>(k_ffmlp_forward_wide<WIDTH, TRAIN, false>);
    else kern = plain ? reinterpret_cast<const void*>(k_ffmlp_forward<WIDTH, TRAIN, true>) : reinterpret_cast<const void*>(k_ffmlp_forward<WIDTH, TRAIN, false>);
    int rc = raise_lds(kern, lds, "ffmlp");
    if (rc) return rc;
    const uint32_t per_cu = (uint32_t)((160 * 1024) / (lds + 1024));
    uint32_t blocks = (uint32_t)device_info().cus * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    const uint32_t need = cdiv(n_tiles, FF_WAVES);
    if (blocks > need) blocks = need;
    const half_t* a0 = (const half_t*)inputs;
    const half_t* a1 = (const half_t*)weights;
    half_t* a2 = (half_t*)fwd;
    half_t* a3 = (half_t*)outputs;
    uint32_t a4 = n_tiles, a5 = in_dim, a6 = num_layers, a7 = act, a8 = out_act;
    bool a9 = (flags & NGP_FF_INPUT_PLANAR) != 0;
    void* args[] = {&a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &a8, &a9};
    hipError_t e = hipLaunchKernel(kern, dim3(blocks), dim3(FF_THREADS), args, lds, st);
    NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "ffmlp: kernel launch failed: %s", hipGetErrorString(e));
    return check_launch(TRAIN ? "ffmlp_forward" : "ffmlp_inference");
    }
}

// layered backward: dgrad chain (dZ of every layer into backward_buffer, dL/dx) + the weight-gradient kernel
constexpr uint32_t WG_MAX_CHUNKS = 64;

template <int WIDTH>
static int launch_backward_layered(const void* grad, const void* inputs, const void* weights, const void* fwd, uint32_t B, uint32_t in_dim,
                                   uint32_t num_layers, uint32_t act, bool with_dx, void* backward_buffer, void* grad_inputs,
                                   void* grad_weights, uint32_t flags, void* workspace, size_t workspace_bytes, hipStream_t st) {
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const bool in_planar = (flags & NGP_FF_INPUT_PLANAR) != 0, dx_planar = (flags & NGP_FF_DX_PLANAR) != 0;
    const uint32_t n_tiles = B / FF_TILE;
    const uint32_t in_jb = (in_dim + 31) / 32;
    uint32_t frags = NIB * NKB;
    if (with_dx && frags < in_jb * NKB) frags = in_jb * NKB;
    const size_t lds = (size_t)frags * 1024;
    NGP_REQUIRE(lds <= 152 * 1024, NGP_ERR_INVALID, "ffmlp_backward: one %u x %u layer (%zu B) exceeds the LDS of a CU", (unsigned)WIDTH, in_dim, lds);
    const void* dk = act == ACT_RELU ? reinterpret_cast<const void*>(k_ffmlp_dgrad_layered<WIDTH, true>)
                                     : reinterpret_cast<const void*>(k_ffmlp_dgrad_layered<WIDTH, false>);
    int rc = raise_lds(dk, lds, "ffmlp_backward");
    if (rc) return rc;
    const uint32_t per_cu = (uint32_t)((160 * 1024) / (lds + 1024));
    uint32_t blocks = (uint32_t)device_info().cus * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
    const uint32_t need = cdiv(n_tiles, FF_WAVES);
    if (blocks > need) blocks = need;
    if (act == ACT_RELU)
        hipLaunchKernelGGL((k_ffmlp_dgrad_layered<WIDTH, true>), dim3(blocks), dim3(FF_THREADS), lds, st, (const half_t*)grad, (const half_t*)weights,
                           (const half_t*)fwd, (half_t*)backward_buffer, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs, dx_planar);
    else
        hipLaunchKernelGGL((k_ffmlp_dgrad_layered<WIDTH, false>), dim3(blocks), dim3(FF_THREADS), lds, st, (const half_t*)grad, (const half_t*)weights,
                           (const half_t*)fwd, (half_t*)backward_buffer, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs, dx_planar);
    rc = check_launch("ffmlp_backward(dgrad)");
    if (rc) return rc;
    // weight gradients
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    const uint32_t jobs = wgrad_jobs_of(NIB, in_jb) + (num_layers - 1) * wgrad_jobs_of(NIB, NIB) + wgrad_jobs_of(1, NIB);
    uint32_t chunks = 1;
    if (workspace) {
        chunks = (uint32_t)(workspace_bytes / ((size_t)n_params * 4));
        const uint32_t want = cdiv((uint32_t)device_info().cus * 4u, jobs);  // ~4 workgroups per CU in flight
        if (chunks > want) chunks = want;
        if (chunks > WG_MAX_CHUNKS) chunks = WG_MAX_CHUNKS;
        const uint32_t by_tiles = cdiv(n_tiles, 2 * FF_WAVES);               // at least two rounds of tiles per wave
        if (chunks > by_tiles) chunks = by_tiles;
        if (chunks < 1) chunks = 1;
    }
    const uint32_t tiles_per_chunk = cdiv(n_tiles, chunks);
    chunks = cdiv(n_tiles, tiles_per_chunk);
    const bool direct = chunks == 1;
    hipLaunchKernelGGL(k_ffmlp_wgrad<WIDTH>, dim3(chunks, jobs), dim3(FF_THREADS), 0, st, (const half_t*)grad, (const half_t*)inputs, (const half_t*)fwd,
                       (const half_t*)backward_buffer, n_tiles, in_dim, num_layers, in_planar, tiles_per_chunk, direct ? (float*)nullptr : (float*)workspace,
                       direct ? (half_t*)grad_weights : (half_t*)nullptr);
    rc = check_launch("ffmlp_backward(wgrad)");
    if (rc || direct) return rc;
    hipLaunchKernelGGL(k_ffmlp_reduce_slabs, dim3(cdiv(n_params, RS_PARAMS)), dim3(RS_PARAMS * RS_GROUPS), 0, st, (const float*)workspace, chunks,
                       n_params, (half_t*)grad_weights);
    return check_launch("ffmlp_backward(reduce)");
}

// workgroups of the register-resident backward = fp32 weight-gradient slabs it leaves in backward_buffer ([num_layers, B, hidden] fp16)
template <int WIDTH>
static uint32_t backward_slab_count(uint32_t B, uint32_t in_dim, uint32_t num_layers) {
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    const size_t buf_bytes = (size_t)num_layers * B * WIDTH * sizeof(half_t);
    uint32_t blocks = (uint32_t)device_info().cus;
    const uint32_t need = cdiv(B / FF_TILE, FF_WAVES);
    if (blocks > need) blocks = need;
    const size_t fit = buf_bytes / ((size_t)n_params * 4);
    if (blocks > fit) blocks = (uint32_t)fit;
    return blocks;
}

template <int WIDTH, int IN_JB, int NHM, bool RELU>
static int launch_backward_t(const void* grad, const void* inputs, const void* weights, const void* fwd, uint32_t B, uint32_t in_dim,
                           uint32_t num_layers, uint32_t act, bool with_dx, void* backward_buffer, void* grad_inputs,
                           void* grad_weights, uint32_t flags, hipStream_t st, const MidEpilogue* mid_in) {
    const bool in_planar = (flags & NGP_FF_INPUT_PLANAR) != 0, dx_planar = (flags & NGP_FF_DX_PLANAR) != 0;
    const bool defer = (flags & NGP_FF_DEFER_REDUCE) != 0;
    const MidEpilogue mid = mid_in ? *mid_in : MidEpilogue{nullptr, nullptr, nullptr, 0.0f};
    constexpr int NIB = Shape<WIDTH>::NIB, NKB = Shape<WIDTH>::NKB;
    const uint32_t n_tiles = B / FF_TILE;
    const uint32_t nfrag = NIB + (num_layers - 1) * NIB * NKB + (with_dx ? ((in_dim + 31) / 32) * NKB : 0);
    const uint32_t n_params = ff_param_count(in_dim, WIDTH, num_layers);
    // weight image + per-wave tile buffers (double-buffered when both fit the 160 KiB LDS of a CU)
    const size_t tile_bytes = (size_t)(1 + num_layers * NKB + in_dim / 16) * 1024;
    uint32_t pf_depth = 2;
    if ((size_t)nfrag * 1024 + 2 * FF_WAVES * tile_bytes > 160 * 1024) pf_depth = 1;
    // (three tile buffers for the two-layer networks, which would fit: measured SLOWER, sigma-net backward 32.6 -> 34.1 us; the kernel's wait
    // counts cover the shapes should that change -- the recomputing variant, whose tiles are 3 KiB, does run three deep)
    size_t lds = (size_t)nfrag * 1024 + (size_t)pf_depth * FF_WAVES * tile_bytes;
    if (lds < (size_t)n_params * 4) lds = (size_t)n_params * 4;
    NGP_REQUIRE(lds <= 160 * 1024, NGP_ERR_INVALID, "ffmlp_backward: LDS need (%zu B) exceeds 160 KiB", lds);
    // one fp32 slab per workgroup lives in the caller's backward_buffer ([num_layers, B, hidden] fp16)
    const uint32_t blocks = backward_slab_count<WIDTH>(B, in_dim, num_layers);
    if (flags & NGP_FF_RECOMPUTE) {
        // no stored activations: the paired kernel recomputes them from the inputs (64-wide ReLU networks with 32 inputs, 2 or 3 layers)
        if constexpr (WIDTH == 64 && IN_JB == 1 && RELU && (NHM == 1 || NHM == 2)) {
            NGP_REQUIRE(in_dim == 32 && !(flags & NGP_FF_SINGLE_WAVE), NGP_ERR_INVALID, "ffmlp_backward: NGP_FF_RECOMPUTE needs input_dim 32 and the paired kernel");
            const uint32_t ffrag = NIB * (in_dim / 16) + (num_layers - 1) * NIB * NKB;
            const size_t tile_b = (size_t)(1 + in_dim / 16) * 1024;
            const uint32_t depth = 3;
            size_t lds_r = (size_t)(nfrag + ffrag) * 1024 + (size_t)depth * FP_PAIRS * tile_b + (size_t)FP_PAIRS * NHM * NKB * 1024;
            if (lds_r < (size_t)n_params * 4) lds_r = (size_t)n_params * 4;
            auto pk = k_ffmlp_backward_paired<WIDTH, IN_JB, NHM, RELU, true>;
            if (lds_r > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
                NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "ffmlp_backward: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            }
            const bool direct = blocks <= 1;
            hipLaunchKernelGGL(pk, dim3(direct ? 1 : blocks), dim3(FP_THREADS), lds_r, st, (const half_t*)grad, (const half_t*)inputs,
                               (const half_t*)weights, (const half_t*)nullptr, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs,
                               direct ? (float*)nullptr : (float*)backward_buffer, direct ? (half_t*)grad_weights : (half_t*)nullptr, in_planar,
                               dx_planar, depth, mid);
            int rc = check_launch("ffmlp_backward");
            if (rc || direct || defer) return rc;
            hipLaunchKernelGGL(k_ffmlp_reduce_slabs, dim3(cdiv(n_params, RS_PARAMS)), dim3(RS_PARAMS * RS_GROUPS), 0, st,
                               (const float*)backward_buffer, blocks, n_params, (half_t*)grad_weights);
            return check_launch("ffmlp_backward(reduce)");
        } else {
            NGP_REQUIRE(false, NGP_ERR_INVALID, "ffmlp_backward: NGP_FF_RECOMPUTE serves 64-wide ReLU networks with 32 inputs and 2 or 3 layers");
        }
    }
    if constexpr (NHM == 1 || NHM == 2) {
        // 2- and 3-layer networks: two sibling waves per tile stream split the weight-gradient accumulators (see the kernel)
        if (!(flags & NGP_FF_SINGLE_WAVE)) {
            auto pk = k_ffmlp_backward_paired<WIDTH, IN_JB, NHM, RELU>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "ffmlp_backward: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            }
            const bool direct = blocks <= 1;
            hipLaunchKernelGGL(pk, dim3(direct ? 1 : blocks), dim3(FP_THREADS), lds, st, (const half_t*)grad, (const half_t*)inputs,
                               (const half_t*)weights, (const half_t*)fwd, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs,
                               direct ? (float*)nullptr : (float*)backward_buffer, direct ? (half_t*)grad_weights : (half_t*)nullptr, in_planar,
                               dx_planar, pf_depth, mid);
            int rc = check_launch("ffmlp_backward");
            if (rc || direct || defer) return rc;  // deferred: the caller sums the slabs (ngp_ffmlp_reduce_slabs_pair)
            hipLaunchKernelGGL(k_ffmlp_reduce_slabs, dim3(cdiv(n_params, RS_PARAMS)), dim3(RS_PARAMS * RS_GROUPS), 0, st,
                               (const float*)backward_buffer, blocks, n_params, (half_t*)grad_weights);
            return check_launch("ffmlp_backward(reduce)");
        }
    }
    NGP_REQUIRE(!mid.grad_sigma && !defer, NGP_ERR_INVALID, "ffmlp_backward: this extension needs the paired kernel (2 or 3 layers)");
    auto kern = k_ffmlp_backward<WIDTH, IN_JB, NHM, RELU>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "ffmlp_backward: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    if (blocks <= 1) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(FF_THREADS), lds, st, (const half_t*)grad, (const half_t*)inputs, (const half_t*)weights,
                           (const half_t*)fwd, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs, (float*)nullptr,
                           (half_t*)grad_weights, in_planar, dx_planar, pf_depth);
        return check_launch("ffmlp_backward");
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(FF_THREADS), lds, st, (const half_t*)grad, (const half_t*)inputs, (const half_t*)weights,
                       (const half_t*)fwd, n_tiles, in_dim, num_layers, act, with_dx, (half_t*)grad_inputs, (float*)backward_buffer,
                       (half_t*)nullptr, in_planar, dx_planar, pf_depth);
    int rc = check_launch("ffmlp_backward");
    if (rc) return rc;
    hipLaunchKernelGGL(k_ffmlp_reduce_slabs, dim3(cdiv(n_params, RS_PARAMS)), dim3(RS_PARAMS * RS_GROUPS), 0, st, (const float*)backward_buffer, blocks, n_params,
                       (half_t*)grad_weights);
    return check_launch("ffmlp_backward(reduce)");
}

template <int WIDTH, int IN_JB, int NHM>
static int launch_backward(const void* grad, const void* inputs, const void* weights, const void* fwd, uint32_t B, uint32_t in_dim,
                           uint32_t num_layers, uint32_t act, bool with_dx, void* backward_buffer, void* grad_inputs,
                           void* grad_weights, uint32_t flags, hipStream_t st, const MidEpilogue* mid = nullptr) {
    if (act == ACT_RELU)
        return launch_backward_t<WIDTH, IN_JB, NHM, true>(grad, inputs, weights, fwd, B, in_dim, num_layers, act, with_dx, backward_buffer,
                                                          grad_inputs, grad_weights, flags, st, mid);
    return launch_backward_t<WIDTH, IN_JB, NHM, false>(grad, inputs, weights, fwd, B, in_dim, num_layers, act, with_dx, backward_buffer,
                                                       grad_inputs, grad_weights, flags, st, mid);
}

}  // namespace ngp

using namespace ngp;

#define FF_WIDTHS(CALL)                  \
    switch (hidden_dim) {                \
        case 16: return CALL(16);        \
        case 32: return CALL(32);        \
        case 64: return CALL(64);        \
        case 128: return CALL(128);      \
        default: return CALL(256);       \
    }

extern "C" int ngp_ffmlp_forward_ex(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                    uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                    void* forward_buffer, void* outputs, uint32_t flags, ngp_stream_t stream) {
    int rc = check_ff_args("ffmlp_forward", B, input_dim, output_dim, hidden_dim, num_layers);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && weights && forward_buffer && outputs, NGP_ERR_INVALID, "ffmlp_forward: NULL tensor");
    hipStream_t st = as_stream(stream);
#define FF_FWD(W) launch_forward<W, true>(inputs, weights, B, input_dim, num_layers, activation, output_activation, forward_buffer, outputs, flags, st)
    FF_WIDTHS(FF_FWD)
#undef FF_FWD
}

extern "C" int ngp_ffmlp_inference_ex(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                      void* inference_buffer, void* outputs, uint32_t flags, ngp_stream_t stream) {
    // inference_buffer [B, hidden] (the reference's scratch, ffmlp.cu:673-709) is only touched by the layered kernel (256-wide layers
    // and networks whose whole fragment image does not fit the LDS); the register-resident kernel leaves it alone and accepts NULL
    int rc = check_ff_args("ffmlp_inference", B, input_dim, output_dim, hidden_dim, num_layers);
    if (rc) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && weights && outputs, NGP_ERR_INVALID, "ffmlp_inference: NULL tensor");
    hipStream_t st = as_stream(stream);
#define FF_INF(W) launch_forward<W, false>(inputs, weights, B, input_dim, num_layers, activation, output_activation, inference_buffer, outputs, flags, st)
    FF_WIDTHS(FF_INF)
#undef FF_INF
}

extern "C" size_t ngp_ffmlp_backward_workspace_bytes(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    const bool fast = (hidden_dim == 32 || hidden_dim == 64) && num_layers <= 4 && input_dim <= 64;
    if (fast || B == 0) return 0;
    return (size_t)WG_MAX_CHUNKS * ff_param_count(input_dim, hidden_dim, num_layers) * sizeof(float);
}

extern "C" int ngp_ffmlp_backward_ws(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                                     uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                     uint32_t output_activation, int calc_grad_inputs, void* backward_buffer, void* grad_inputs,
                                     void* grad_weights, uint32_t flags, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    (void)output_activation;  // the reference discards it as well (ffmlp.cu:780)
    int rc = check_ff_args("ffmlp_backward", B, input_dim, output_dim, hidden_dim, num_layers);
    if (rc) return rc;
    if (B == 0) {
        // an empty batch: no kernel runs, but the caller's weight gradient must still be what the sum over zero samples is (callers hand
        // over uninitialised memory: every other path OVERWRITES it) -- ADVICE r5.  (With NGP_FF_DEFER_REDUCE the later reduction writes it.)
        if (grad_weights && !(flags & NGP_FF_DEFER_REDUCE)) {
            const size_t n_w = (size_t)hidden_dim * ((size_t)input_dim + (size_t)hidden_dim * (num_layers - 1) + (size_t)output_dim);
            hipError_t e = hipMemsetAsync(grad_weights, 0, n_w * sizeof(half_t), as_stream(stream));
            NGP_REQUIRE(e == hipSuccess, NGP_ERR_LAUNCH, "ffmlp_backward: hipMemsetAsync failed: %s", hipGetErrorString(e));
        }
        return NGP_OK;
    }
    NGP_REQUIRE(grad && inputs && weights && (forward_buffer || (flags & NGP_FF_RECOMPUTE)) && backward_buffer && grad_weights, NGP_ERR_INVALID,
                "ffmlp_backward: NULL tensor");
    NGP_REQUIRE(!calc_grad_inputs || grad_inputs, NGP_ERR_INVALID, "ffmlp_backward: grad_inputs is NULL but calc_grad_inputs is set");
    hipStream_t st = as_stream(stream);
    const bool dx = calc_grad_inputs != 0;
    const bool fast = (hidden_dim == 32 || hidden_dim == 64) && num_layers <= 4 && input_dim <= 64 && !(flags & NGP_FF_LAYERED);
    NGP_REQUIRE(fast || !(flags & NGP_FF_RECOMPUTE), NGP_ERR_INVALID, "ffmlp_backward: NGP_FF_RECOMPUTE serves 64-wide ReLU networks with 32 inputs and 2 or 3 layers");
    NGP_REQUIRE(fast || !(flags & NGP_FF_DEFER_REDUCE), NGP_ERR_INVALID, "ffmlp_backward: NGP_FF_DEFER_REDUCE needs a 2- or 3-layer network of width 32 / 64");
    if (!fast) {
#define FF_LAY(W) launch_backward_layered<W>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, activation, dx, backward_buffer, grad_inputs, grad_weights, flags, workspace, workspace_bytes, st)
        FF_WIDTHS(FF_LAY)
#undef FF_LAY
    }
    const uint32_t in_jb = (input_dim + 31) / 32;
#define FF_BWD(W, J, N) \
    return launch_backward<W, J, N>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, activation, dx, backward_buffer, grad_inputs, grad_weights, flags, st)
#define FF_BWD_N(W, J)                      \
    switch (num_layers - 1) {               \
        case 1: FF_BWD(W, J, 1);            \
        case 2: FF_BWD(W, J, 2);            \
        default: FF_BWD(W, J, 3);           \
    }
    if (hidden_dim == 64) {
        if (in_jb == 1) { FF_BWD_N(64, 1) }
        FF_BWD_N(64, 2)
    }
    if (in_jb == 1) { FF_BWD_N(32, 1) }
    FF_BWD_N(32, 2)
#undef FF_BWD_N
#undef FF_BWD
}

extern "C" int ngp_ffmlp_backward_ex(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                                     uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                     uint32_t output_activation, int calc_grad_inputs, void* backward_buffer, void* grad_inputs,
                                     void* grad_weights, uint32_t flags, ngp_stream_t stream) {
    return ngp_ffmlp_backward_ws(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                 output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, flags, nullptr, 0, stream);
}


extern "C" uint32_t ngp_ffmlp_backward_slab_count(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    if ((hidden_dim != 32 && hidden_dim != 64) || num_layers < 2 || num_layers > 3 || input_dim > 64 || B == 0) return 0;
    const uint32_t blocks = hidden_dim == 64 ? backward_slab_count<64>(B, input_dim, num_layers) : backward_slab_count<32>(B, input_dim, num_layers);
    return blocks <= 1 ? 0u : blocks;  // one workgroup stores the gradients directly: nothing left to sum
}

extern "C" int ngp_ffmlp_reduce_slabs_pair(const void* slabs_a, uint32_t n_slabs_a, uint32_t n_params_a, void* grad_weights_a,
                                           const void* slabs_b, uint32_t n_slabs_b, uint32_t n_params_b, void* grad_weights_b,
                                           float* found_inf, ngp_stream_t stream) {
    // a set without slabs (its backward stored the gradients directly) still gets workgroups when found_inf asks for the sweep
    const uint32_t blocks_a = (n_slabs_a || found_inf) && n_params_a ? cdiv(n_params_a, RS_PARAMS) : 0u;
    const uint32_t blocks_b = (n_slabs_b || found_inf) && n_params_b ? cdiv(n_params_b, RS_PARAMS) : 0u;
    if (blocks_a + blocks_b == 0) return NGP_OK;
    NGP_REQUIRE((!blocks_a || ((slabs_a || !n_slabs_a) && grad_weights_a)) && (!blocks_b || ((slabs_b || !n_slabs_b) && grad_weights_b)),
                NGP_ERR_INVALID, "ffmlp_reduce_slabs_pair: NULL tensor");
    SlabSets sets;
    sets.slabs[0] = (const float*)slabs_a; sets.n_slabs[0] = n_slabs_a; sets.n_params[0] = n_params_a; sets.grad_weights[0] = (half_t*)grad_weights_a;
    sets.slabs[1] = (const float*)slabs_b; sets.n_slabs[1] = n_slabs_b; sets.n_params[1] = n_params_b; sets.grad_weights[1] = (half_t*)grad_weights_b;
    sets.blocks[0] = blocks_a; sets.blocks[1] = blocks_b;
    hipLaunchKernelGGL(k_ffmlp_reduce_slabs_pair, dim3(blocks_a + blocks_b), dim3(RS_PARAMS * RS_GROUPS), 0, as_stream(stream), sets, found_inf);
    return check_launch("ffmlp_reduce_slabs_pair");
}

extern "C" int ngp_network_backward_color(const void* grad_out16, const void* color_in, const void* w_color, const void* forward_buffer_color,
                                          uint32_t M, uint32_t num_layers_color, void* backward_buffer, const float* grad_sigma,
                                          const void* h16, float density_scale, void* grad_h16, void* grad_w_color, uint32_t flags,
                                          ngp_stream_t stream) {
    NGP_REQUIRE(num_layers_color == 2 || num_layers_color == 3, NGP_ERR_INVALID,
                "network_backward_color: 2 or 3 layers (got %u); use ngp_ffmlp_backward_ex + ngp_pipeline_mid_backward", num_layers_color);
    NGP_REQUIRE(!(flags & ~(NGP_FF_DEFER_REDUCE | NGP_FF_RECOMPUTE)), NGP_ERR_INVALID,
                "network_backward_color: only NGP_FF_DEFER_REDUCE and NGP_FF_RECOMPUTE are accepted");
    int rc = check_ff_args("network_backward_color", M, 32, 16, 64, num_layers_color);
    if (rc) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(grad_out16 && color_in && w_color && (forward_buffer_color || (flags & NGP_FF_RECOMPUTE)) && backward_buffer && grad_sigma && h16 &&
                    grad_h16 && grad_w_color,
                NGP_ERR_INVALID, "network_backward_color: NULL tensor");
    const MidEpilogue mid{grad_sigma, (const half_t*)h16, (half_t*)grad_h16, density_scale};
    if (num_layers_color == 2)
        return launch_backward<64, 1, 1>(grad_out16, color_in, w_color, forward_buffer_color, M, 32, 2, ACT_RELU, true, backward_buffer, nullptr,
                                         grad_w_color, flags, as_stream(stream), &mid);
    return launch_backward<64, 1, 2>(grad_out16, color_in, w_color, forward_buffer_color, M, 32, 3, ACT_RELU, true, backward_buffer, nullptr,
                                     grad_w_color, flags, as_stream(stream), &mid);
}

extern "C" int ngp_network_forward(const void* enc, const float* dirs, uint32_t M, uint32_t M_valid, const void* w_sigma, const void* w_color,
                                   uint32_t num_layers_sigma, uint32_t num_layers_color, float density_scale, int training,
                                   void* forward_buffer_sigma, void* h16, float* sigma, void* color_in, void* forward_buffer_color,
                                   float* rgb, uint32_t flags, ngp_stream_t stream) {
    return ngp_network_forward_rows(enc, dirs, M, M_valid, w_sigma, w_color, num_layers_sigma, num_layers_color, density_scale, training,
                                    forward_buffer_sigma, h16, sigma, color_in, forward_buffer_color, rgb, flags, nullptr, stream);
}

extern "C" int ngp_network_forward_rows(const void* enc, const float* dirs, uint32_t M, uint32_t M_valid, const void* w_sigma, const void* w_color,
                                        uint32_t num_layers_sigma, uint32_t num_layers_color, float density_scale, int training,
                                        void* forward_buffer_sigma, void* h16, float* sigma, void* color_in, void* forward_buffer_color,
                                        float* rgb, uint32_t flags, const uint32_t* rows_dev, ngp_stream_t stream) {
    NGP_REQUIRE(M % 128 == 0, NGP_ERR_INVALID, "network_forward: sample count must be 128 * m, but got %u", M);
    NGP_REQUIRE(num_layers_sigma >= 2 && num_layers_color >= 2, NGP_ERR_INVALID, "network_forward: num_layers should be larger than 2");
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(enc && dirs && w_sigma && w_color && sigma && rgb, NGP_ERR_INVALID, "network_forward: NULL tensor");
    NGP_REQUIRE(!training || (h16 && color_in && (forward_buffer_sigma != nullptr) == (forward_buffer_color != nullptr)), NGP_ERR_INVALID,
                "network_forward: the training variant needs h16 and color_in, and both forward buffers or (NGP_FF_RECOMPUTE backward) neither");
    const size_t lds = forward_image_bytes<64>(32, num_layers_sigma) + forward_image_bytes<64>(32, num_layers_color);
    NGP_REQUIRE(lds <= 152 * 1024, NGP_ERR_INVALID, "network_forward: weights (%zu B) exceed the LDS of a CU", lds);
#ifndef NGP_NETFWD_TRAIN_WAVES
#define NGP_NETFWD_TRAIN_WAVES 4
#endif
#ifndef NGP_NETFWD_INFER_WAVES
#define NGP_NETFWD_INFER_WAVES 4
#endif
#ifndef NGP_NETFWD_TRAIN_TILES
#define NGP_NETFWD_TRAIN_TILES 1
#endif
#ifndef NGP_NETFWD_INFER_TILES
#define NGP_NETFWD_INFER_TILES 1
#endif
    constexpr int TW = NGP_NETFWD_TRAIN_WAVES, IW = NGP_NETFWD_INFER_WAVES, TT = NGP_NETFWD_TRAIN_TILES, IT = NGP_NETFWD_INFER_TILES;
    const void* kern = training ? reinterpret_cast<const void*>(k_network_forward<true, TW, TT>) : reinterpret_cast<const void*>(k_network_forward<false, IW, IT>);
    int rc = raise_lds(kern, lds, "network_forward");
    if (rc) return rc;
    const uint32_t n_tiles = M / FF_TILE;
    const uint32_t waves = training ? TW : IW, nt = training ? TT : IT;
    // workgroups per CU: what the LDS holds, at most 16 waves per CU (four per SIMD at one tile per wave iteration; 12 at two: 166 registers)
    const uint32_t by_lds = (uint32_t)((160 * 1024) / (lds + 1024)), by_waves = (nt == 1 ? 16u : 12u) / waves;
#ifndef NGP_NETFWD_PER_CU
#define NGP_NETFWD_PER_CU 4u   // cap; 1 / 2 / 3 measured slower for training (60 / 55 / 56 vs 55 us) and inference (26 / 21 / 20 vs 20 us) at 4 waves
#endif
    uint32_t per_cu = by_lds < by_waves ? by_lds : by_waves;
    per_cu = per_cu < 1 ? 1 : (per_cu > NGP_NETFWD_PER_CU ? NGP_NETFWD_PER_CU : per_cu);
    uint32_t blocks = (uint32_t)device_info().cus * per_cu;
    const uint32_t need = cdiv(cdiv(n_tiles, nt), waves);
    if (blocks > need) blocks = need;
    hipStream_t st = as_stream(stream);
    const bool planar = (flags & NGP_FF_INPUT_PLANAR) != 0;
    if (training)
        hipLaunchKernelGGL((k_network_forward<true, TW, TT>), dim3(blocks), dim3(TW * 64), lds, st, (const half_t*)enc, planar, dirs, M_valid, (const half_t*)w_sigma,
                           (const half_t*)w_color, (half_t*)forward_buffer_sigma, (half_t*)h16, sigma, (half_t*)color_in, (half_t*)forward_buffer_color, rgb,
                           n_tiles, num_layers_sigma, num_layers_color, density_scale, rows_dev);
    else
        hipLaunchKernelGGL((k_network_forward<false, IW, IT>), dim3(blocks), dim3(IW * 64), lds, st, (const half_t*)enc, planar, dirs, M_valid, (const half_t*)w_sigma,
                           (const half_t*)w_color, (half_t*)nullptr, (half_t*)nullptr, sigma, (half_t*)nullptr, (half_t*)nullptr, rgb, n_tiles,
                           num_layers_sigma, num_layers_color, density_scale, rows_dev);
    return check_launch("network_forward");
}

extern "C" int ngp_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                 uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                 void* forward_buffer, void* outputs, ngp_stream_t stream) {
    return ngp_ffmlp_forward_ex(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                                forward_buffer, outputs, 0u, stream);
}
extern "C" int ngp_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                   uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                   void* inference_buffer, void* outputs, ngp_stream_t stream) {
    return ngp_ffmlp_inference_ex(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                                  inference_buffer, outputs, 0u, stream);
}
extern "C" int ngp_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                                  uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                  uint32_t output_activation, int calc_grad_inputs, void* backward_buffer, void* grad_inputs,
                                  void* grad_weights, ngp_stream_t stream) {
    return ngp_ffmlp_backward_ex(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                 output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, 0u, stream);
}

static size_t g_splitk_request = 0;
extern "C" int ngp_allocate_splitk(size_t size) {
    g_splitk_request = size;
    return NGP_OK;
}
extern "C" int ngp_free_splitk(void) {
    g_splitk_request = 0;
    return NGP_OK;
}
