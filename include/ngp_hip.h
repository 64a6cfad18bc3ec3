/*
 * ngp_hip.h -- C ABI of libngp_hip.so: the MI355X (gfx950) implementation of torch-ngp's instant-ngp
 * hot path (gridencoder, shencoder, raymarching, ffmlp).
 *
 * This is the drop-in boundary.  Each entry point replaces one callable of the reference's pybind
 * `_backend` modules (cited per function, paths relative to the reference checkout) and keeps its
 * argument order and meaning; the differences forced by a C ABI are uniform:
 *   - at::Tensor arguments become raw DEVICE pointers (the caller owns every buffer, exactly as in
 *     the reference where the Python wrapper allocates all outputs and workspaces);
 *   - at::optional<at::Tensor> becomes a pointer that may be NULL;
 *   - the element type that the reference dispatches on (AT_DISPATCH_FLOATING_TYPES_AND_HALF) is an
 *     explicit `dtype` code; fp64 is not provided;
 *   - a trailing `stream` (a hipStream_t passed as void*; NULL = the legacy default stream the
 *     reference launches on);
 *   - every function returns 0 on success and a non-zero NGP_ERR_* code on failure, with a
 *     human-readable message available from ngp_last_error() (the Python binding turns it into the
 *     RuntimeError the reference's TORCH_CHECK / std::runtime_error would have produced).
 * No function allocates device memory, synchronises the device, or touches the host copy of any
 * buffer.  All launches are asynchronous on `stream`.
 *
 * Contracts kept from the reference: buffers the reference requires pre-zeroed stay caller-zeroed
 * (xyzs/dirs/deltas, grad_embeddings, grad_inputs of the grid and SH backward, grad_sigmas/grad_rgbs);
 * `counter` is read-modify-written.  (ffmlp_backward OVERWRITES grad_weights: no pre-zeroing needed.)
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ngp_stream_t; /* hipStream_t */

enum {
    NGP_OK = 0,
    NGP_ERR_INVALID = 1, /* bad argument (unsupported D/C/hidden_dim, NULL pointer, misaligned size) */
    NGP_ERR_LAUNCH = 2,  /* HIP reported a launch error */
    NGP_ERR_DEVICE = 3   /* no usable gfx950 device / runtime error */
};

enum { NGP_F32 = 0, NGP_F16 = 1 };

#define NGP_MAX_LEVELS 32 /* grid encoder: L <= 32 */

/* message of the last failing call on the calling thread ("" if none) */
const char* ngp_last_error(void);
/* ABI version of this header (bumped on any signature change) */
int ngp_abi_version(void);
/* gfx architecture string the library was compiled for ("gfx950") */
const char* ngp_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * gridencoder      (reference: gridencoder/src/gridencoder.h:12-15, bindings.cpp:5-9)
 * --------------------------------------------------------------------------------------------- */

/* replaces grid_encode_forward (gridencoder.cu:448-471).
 * inputs [B,D] fp32 in [0,1]; embeddings [sO,C] dtype; offsets [L+1] int32 (device);
 * outputs [L,B,C] dtype; dy_dx [B,L*D*C] dtype or NULL.  D in {2,3,4,5}, C in {1,2,4,8}, L <= 32.
 * gridtype 0 = hash, 1 = tiled; interp 0 = linear, 1 = smoothstep. */
int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                            uint32_t gridtype, int align_corners, uint32_t interp, int dtype, ngp_stream_t stream);

/* replaces grid_encode_backward (gridencoder.cu:473-503).
 * grad [L,B,C] dtype; grad_embeddings [sO,C] dtype, pre-zeroed, accumulated with hardware atomics
 * (packed fp16 when dtype is F16 and C is even, as the reference); dy_dx / grad_inputs [B,D] dtype
 * both NULL or both non-NULL. */
int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                             uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                             uint32_t interp, int dtype, ngp_stream_t stream);

/* replaces grad_total_variation (gridencoder.cu:639-645): inputs [B,D] dtype in [0,1]; adds into grad [sO,C]. */
int ngp_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                             float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, int dtype, ngp_stream_t stream);

/* Diagnostic (no reference counterpart): the per-corner table entry index (before *C) the kernels
 * use, [L,B,2^D] uint32, 0xFFFFFFFF for out-of-range points.  Lets the parity suite check grid
 * indexing bit-for-bit. */
int ngp_grid_corner_indices(const float* inputs, const int32_t* offsets, uint32_t* indices, uint32_t B, uint32_t D,
                            uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                            ngp_stream_t stream);

/* The per-level (scale, resolution) table every grid kernel uses (host computation, no device work):
 * scale_l = fma(exp2((float)l*S), H, -1) in fp32 with a correctly rounded exp2; resolution_l = ceil(scale_l)+1.
 * Restates gridencoder.cu:137-139 with a reproducible exp2. */
int ngp_grid_level_table(uint32_t L, float S, uint32_t H, float* scale_out, uint32_t* resolution_out);

/* ---------------------------------------------------------------------------------------------
 * shencoder        (reference: shencoder/src/shencoder.h:9-10, bindings.cpp:5-8)
 * --------------------------------------------------------------------------------------------- */

/* replaces sh_encode_forward (shencoder.cu:400-417): inputs [B,3]; outputs [B,C*C]; dy_dx [B,3*C*C] or NULL;
 * D must be 3, C (number of bands) in 1..8. */
int ngp_sh_encode_forward(const void* inputs, void* outputs, uint32_t B, uint32_t D, uint32_t C, void* dy_dx,
                          int dtype, ngp_stream_t stream);
/* replaces sh_encode_backward (shencoder.cu:419-439): grad_inputs[b,d] += sum_ch grad[b,ch]*dy_dx[b,d,ch] */
int ngp_sh_encode_backward(const void* grad, const void* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const void* dy_dx, void* grad_inputs, int dtype, ngp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * raymarching      (reference: raymarching/src/raymarching.h:7-18, bindings.cpp:5-19)
 * All floating tensors are fp32 (the reference's wrappers force it with custom_fwd(cast_inputs=float32)).
 * --------------------------------------------------------------------------------------------- */

/* replaces near_far_from_aabb (raymarching.cu:148-156) */
int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                           float* nears, float* fars, ngp_stream_t stream);
/* replaces sph_from_ray (raymarching.cu:201-209) */
int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     ngp_stream_t stream);
/* replaces morton3D / morton3D_invert (raymarching.cu:229-232, 257-260) */
int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, ngp_stream_t stream);
int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, ngp_stream_t stream);
/* replaces packbits (raymarching.cu:292-300): N = number of output bytes, grid has 8*N floats */
int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, ngp_stream_t stream);
/* extension: the threshold is min(density_thresh, *thresh_cap) with thresh_cap a DEVICE scalar (may be NULL) -- the occupancy refresh
 * packs against min(mean_density, density_thresh) (renderer.py:527-529) without reading the mean back to the host */
int ngp_packbits_ex(const float* grid, uint32_t N, float density_thresh, const float* thresh_cap, uint8_t* bitfield,
                    ngp_stream_t stream);

/* replaces march_rays_train (raymarching.cu:482-490).
 * Sample slots are handed out by a deterministic prefix sum in ray order (rays[n] = (n, offset_n,
 * count_n)), which is the allocation a sequential execution of the reference kernel produces;
 * counter[0] += total samples, counter[1] += N.  `workspace` must hold
 * ngp_march_rays_train_workspace_bytes(N) bytes (contents undefined on entry). */
size_t ngp_march_rays_train_workspace_bytes(uint32_t N);
int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                         const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays,
                         int32_t* counter, const float* noises, void* workspace, ngp_stream_t stream);

/* replaces composite_rays_train_forward / _backward (raymarching.cu:580-588, 685-693) */
int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     float* weights_sum, float* depth, float* image, ngp_stream_t stream);
int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float T_thresh, float* grad_sigmas, float* grad_rgbs, ngp_stream_t stream);

/* replaces march_rays / composite_rays (raymarching.cu:808-815, 908-914) */
int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                   float* dirs, float* deltas, const float* noises, ngp_stream_t stream);
int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                       float* depth, float* image, ngp_stream_t stream);

/* Extension (no reference counterpart): stream compaction of the alive list on the
 * device, replacing `rays_alive[rays_alive >= 0]` + its host sync in the caller's render loop.
 * out_alive receives the surviving ids in order, *out_count (device int32) their number. */
size_t ngp_compact_rays_workspace_bytes(uint32_t n_alive);
int ngp_compact_rays(const int32_t* rays_alive, uint32_t n_alive, int32_t* out_alive, int32_t* out_count,
                     void* workspace, ngp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ffmlp            (reference: ffmlp/src/ffmlp.h:8-14, bindings.cpp:5-11)
 * All tensors fp16.  weights: flat [hidden*in] + (num_layers-1) x [hidden*hidden] + [output_dim*hidden],
 * each matrix row-major [out,in].  B must be a multiple of 128 (the wrapper pads), hidden_dim in
 * {16,32,64,128,256}, input_dim % 16 == 0, output_dim == 16 (padded by the wrapper), num_layers >= 2 -- the reference's own
 * constraints (ffmlp.py:112-115, ffmlp.cu:543-556,653-658).  Register-resident kernels serve hidden_dim <= 128 forward and the
 * instant-ngp shapes backward (hidden 32/64, <= 4 hidden layers, input_dim <= 64); every other shape runs the layered kernels
 * (one matmul's weights in LDS at a time).  The only shapes refused (NGP_ERR_INVALID) are those whose single largest
 * layer exceeds the LDS: hidden_dim * max(input_dim, hidden_dim) * 2 bytes > 152 KiB, i.e. input_dim > 304 with 256-wide layers.
 * activation ids: 0 ReLU, 1 Exp, 2 Sine, 3 Sigmoid, 4 Squareplus, 5 Softplus, 6 None (utils.h:29-37).
 * --------------------------------------------------------------------------------------------- */

/* replaces ffmlp_forward (ffmlp.cu:635-671): forward_buffer [num_layers,B,hidden] receives the hidden
 * post-activations in a layout private to this library (only ngp_ffmlp_backward reads it). */
int ngp_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                      void* forward_buffer, void* outputs, ngp_stream_t stream);
/* replaces ffmlp_inference (ffmlp.cu:673-709): inference_buffer [B,hidden] is scratch (used by the layered kernel only) */
int ngp_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                        uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                        uint32_t output_activation, void* inference_buffer, void* outputs, ngp_stream_t stream);
/* replaces ffmlp_backward (ffmlp.cu:749-895): grad [B,output_dim]; backward_buffer [num_layers,B,hidden]
 * scratch; grad_inputs [B,input_dim] written iff calc_grad_inputs; grad_weights flat fp16: OVERWRITTEN with the
 * fp32-accumulated batch sums rounded once to fp16 whatever it held (the reference's wrapper zero-fills it,
 * ffmlp.py:66-71: not needed here) -- also for an empty batch (B == 0: zeros). */
int ngp_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                       uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                       void* grad_inputs, void* grad_weights, ngp_stream_t stream);
/* replace allocate_splitk / free_splitk (ffmlp.cu:721-740).  The reference creates size streams+events
 * for its split-K weight-gradient GEMMs; this library reduces weight gradients inside the backward
 * kernel, so these only record the request (kept so that FFMLP.__init__ runs unchanged). */
int ngp_allocate_splitk(size_t size);
int ngp_free_splitk(void);

/* ---------------------------------------------------------------------------------------------
 * Fused sample pipeline -- EXTENSIONS beyond the reference surface.
 * The reference wrappers glue the native ops with permute/cat/cast/elementwise PyTorch kernels
 * (grid.py:57,75,149; ffmlp.py:157-165; network_ff.py:51-123).  These entry points let the mirror in
 * torch-ngp_amd/ run the same arithmetic with the same rounding points and no glue copies; the
 * reference-contract functions above are thin wrappers over them (flags = 0, bound = 0).
 * --------------------------------------------------------------------------------------------- */
/* bound > 0: inputs are world coordinates in [-bound, bound], mapped in-kernel as (x + bound) * (1/(2 bound))
 * (what GridEncoder.forward does with two PyTorch kernels, grid.py:149); bound == 0: unit inputs. */
int ngp_grid_encode_forward_ex(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                               ngp_stream_t stream);
int ngp_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                uint32_t interp, int dtype, float bound, ngp_stream_t stream);
/* grid_encode_forward_ex with relative per-level costs from the caller (HOST array of L positive floats, e.g. the expected fraction of
 * consecutive samples that change cell at each level): the launch's per-XCD work lists are balanced by them -- whole levels stay on XCD
 * (level mod 8), the tail of the most loaded XCDs' last level moves to the least loaded ones.  Scheduling only: the arithmetic per
 * (level, point) and the results are identical.  level_cost_host == NULL: ngp_grid_encode_forward_ex (every level costs the same). */
int ngp_grid_encode_forward_sched(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                  uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                  uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                  const float* level_cost_host, ngp_stream_t stream);

/* ngp_grid_encode_forward_sched for a DOUBLE-BUFFERED fp16 table (ngp_table_adam_t): the kernel reads *parity (a device float: the
 * optimizer's state[5]) at entry and gathers from `embeddings` when it is 0, from `embeddings_alt` otherwise -- the selection is made on the
 * device, so a captured HIP graph follows the commit / skip decisions of the steps replayed before it.  embeddings_alt == NULL or parity ==
 * NULL: ngp_grid_encode_forward_sched.  dy_dx must be NULL (the inference / fused-training forward).
 * rows_dev (optional device word, independent of the table selection): only the first min(B, *rows_dev) points are encoded -- the launch is sized
 * for B, workgroups behind the device-side count leave at once (the eval loop's emitted-row count, ngp_march_rays_dev_rows). */
int ngp_grid_encode_forward_sel(const float* inputs, const void* embeddings, const void* embeddings_alt, const float* parity,
                                const uint32_t* rows_dev, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                const float* level_cost_host, ngp_stream_t stream);

/* diagnostic: the per-XCD work lists ngp_grid_encode_forward_sched would launch for L levels of `tiles` tiles each (host computation):
 * 8 x 8 segments (level, first tile, cumulative slot end; level 0xffff = unused); returns the slots of the longest list (0 on bad arguments) */
uint32_t ngp_grid_forward_work_lists(uint32_t L, uint32_t tiles, const float* level_cost_host, uint16_t* level_out, uint32_t* tile0_out,
                                     uint32_t* end_out);

/* grid_encode_backward_ex with a caller-provided workspace: fp16 tables with C = 2 and D <= 3 (the instant-ngp configuration) then
 * run EVERY level WITHOUT memory-side atomics -- contributions are sorted by table slice (12-byte pair units = 6 bytes per record,
 * coalesced stores) and each slice is summed exactly in a 64-bit fixed-point LDS accumulator, rounded once and added to grad_embeddings
 * by the workgroup that owns it.  The result is the exact sum of the fp16 contributions rounded once (the reference's atomics round after every
 * add, in an undefined order) and is bit-reproducible; entries that receive a non-finite contribution become NaN.
 * offsets_host: a HOST copy of `offsets` (L + 1 values; the level sizes steer the plan).  workspace: device memory of at least
 * ngp_grid_backward_workspace_bytes(...) bytes, contents irrelevant.  offsets_host == NULL or workspace == NULL -> the atomic path
 * (= ngp_grid_encode_backward_ex).  Small batches and other dtypes/shapes use the atomic path as well (workspace_bytes() == 0). */
size_t ngp_grid_backward_workspace_bytes(const int32_t* offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                         uint32_t gridtype, int align_corners, int dtype);
/* ngp_grid_encode_backward_ws that also does the optimizer's non-finite sweep over the gradient table where the values are produced:
 * found_inf (optional device float) is set to 1 when a gradient entry this call wrote is not finite (needs offsets_host; levels that
 * went through atomics are swept by one extra launch).  With it, ngp_optim_adam_step_ex can be called without its CHECK phase. */
int ngp_grid_encode_backward_checked(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                     void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                     const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                     int dtype, float bound, const int32_t* offsets_host, void* workspace, size_t workspace_bytes,
                                     float* found_inf, ngp_stream_t stream);

/* ngp_grid_encode_backward_checked that also CARRIES the deferred slab reduction of two FFMLP backward launches (NGP_FF_DEFER_REDUCE; same
 * arguments and results as ngp_ffmlp_reduce_slabs_pair, found_inf shared) in its own last launch instead of a launch of ~270 small blocks
 * behind it: the two are independent, the reduction fills slots the last table slices leave idle (training step: -1 launch, ~6 us).
 * slab_sets == NULL: plain ngp_grid_encode_backward_checked.  When this call has no such launch (few samples, no workspace, B == 0) the
 * reduction is launched on its own: the reduced gradients are there when the call returns either way (stream order). */
/* Adam on the hash table INSIDE the slice accumulate of the grid backward (round 6).  torch.optim.Adam + GradScaler.step
 * (nerf/utils.py:751-753) need the global "any gradient non-finite?" verdict before the first weight moves; the accumulate produces the final
 * gradient of a 4096-entry slice in LDS while most of the chip waits on its record walk, and a separate 28 B/parameter Adam sweep over the
 * 12 M-entry table follows.  Here the flush of every slice applies Adam at once -- SPECULATIVELY: it reads the parameters, moments of buffer
 * set state[5] (the parity: 0 or 1) and writes the updated parameters, moments and fp16 shadow into the OTHER set.  The commit
 * (ngp_optim_adam_small_commit / NGP_OPT_PHASE_FLIP) flips the parity only when the step stands; a skipped step leaves the current set
 * untouched -- GradScaler's semantics, exactly.  Readers of the fp16 table select the current set at kernel entry
 * (ngp_grid_encode_forward_sel).  Needs overwrite_table with every level on the record-sort path (else NGP_ERR_INVALID: nothing was
 * launched); the fp16 gradient is rounded exactly as when it is stored, so the result equals ngp_optim_adam_step_ex on the stored gradient
 * bit for bit.  The DENSE levels at the start of the table (their entries are dealt round-robin to the accumulate's workgroups: 8-byte
 * accesses 1 KiB apart would be all the flush could do for them) are left out: their gradient is stored to grad_embeddings as usual and
 * ngp_optim_adam_small_commit sweeps that prefix contiguously (same double buffer, verdict already known) -- ngp_grid_table_adam_prefix()
 * entries.  Behind the prefix grad_embeddings is not written (it may be NULL when the prefix is empty). */
typedef struct ngp_table_adam {
    float* param[2]; float* exp_avg[2]; float* exp_avg_sq[2]; void* param_fp16[2];   /* [n_entries, C] each; set index = parity */
    const float* state;          /* the optimizer's scalars (ngp_optim_adam_step: state[0] scale, [3] step count, [4] lr multiplier, [5] parity) */
    float lr, beta1, beta2, eps;
} ngp_table_adam_t;
/* entries (whole levels) at the start of the table that a backward with table_adam leaves to ngp_optim_adam_small_commit; 0xffffffff: this
 * batch / table shape cannot carry the sweep at all (the backward would refuse table_adam).  Host computation, same plan as the backward. */
uint32_t ngp_grid_table_adam_prefix(const int32_t* offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    uint32_t gridtype, int align_corners, int dtype);
typedef struct ngp_slab_sets {
    const void* slabs_a; uint32_t n_slabs_a, n_params_a; void* grad_weights_a;
    const void* slabs_b; uint32_t n_slabs_b, n_params_b; void* grad_weights_b;
    /* optional third carried job: loss[0] = sum(ray_err[0 .. n_rays)) / (3 n_rays), the loss VALUE ngp_composite_train_loss_backward would
     * have summed itself (call it with loss = NULL: its workgroups then skip the ticket round trip); same routine, same bits.  loss = NULL: none */
    const float* ray_err; uint32_t n_rays; float* loss;
    /* != 0: grad_embeddings receives this call's gradient instead of having it ADDED (every entry is written, zeros included, nothing is
     * read): for a caller whose optimizer then does not zero the buffer (ngp_optim_adam_step*: grad_is_half & 2).  Same bits as adding into
     * a zeroed buffer.  Calls that cannot write every entry from the sort (levels on the atomic path) zero the table first. */
    uint32_t overwrite_table;
    /* optional (NULL: none): the table's Adam sweep rides in the accumulate's flush, see ngp_table_adam_t */
    const ngp_table_adam_t* table_adam;
} ngp_slab_sets_t;
int ngp_grid_encode_backward_checked_slabs(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                           void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                           const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                           int dtype, float bound, const int32_t* offsets_host, void* workspace, size_t workspace_bytes,
                                           float* found_inf, const ngp_slab_sets_t* slab_sets, ngp_stream_t stream);
int ngp_grid_encode_backward_ws(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                int dtype, float bound, const int32_t* offsets_host, void* workspace, size_t workspace_bytes,
                                ngp_stream_t stream);

/* The apply half of the occupancy-grid refresh (NeRFRenderer.update_extra_state, nerf/renderer.py:515-529) in three launches instead of
 * ~15 PyTorch ones: `tmp_grid[cas, indices] = sigmas` (scatter of density_scale * sigmas), the EMA-max update
 * `grid = max(grid * decay, tmp)` where both sides are >= 0 (one update per distinct cell), `mean(clamp(grid, 0))` (double, fixed order)
 * and packbits against min(density_thresh, mean) -- nothing read back.  cells [n] int64: cascade * H^3 + morton index of each queried
 * point; scratch [n_cells] fp32: -1 everywhere before the first call, left that way; workspace: ngp_density_grid_update_workspace_bytes,
 * zeroed before the first call. */
size_t ngp_density_grid_update_workspace_bytes(uint32_t n_cells);
int ngp_density_grid_update(const float* sigmas, const int64_t* cells, uint32_t n, float density_scale, float decay, float* density_grid,
                            uint32_t n_cells, float* scratch, float density_thresh, float* mean_out, uint8_t* bitfield, void* workspace,
                            ngp_stream_t stream);

/* flags of the ffmlp *_ex entry points */
#define NGP_FF_INPUT_PLANAR 1u /* inputs are [input_dim/2][B][2] fp16 planes = the grid encoder's [L,B,C=2] output */
#define NGP_FF_DX_PLANAR 2u    /* grad_inputs is written in that planar layout = what grid_encode_backward reads */
#define NGP_FF_LAYERED 4u      /* testing: force the layered kernels on shapes the register-resident ones would serve */
#define NGP_FF_SINGLE_WAVE 8u  /* testing: backward with one wave per tile stream instead of the paired kernel (2- and 3-layer nets) */
#define NGP_FF_DEFER_REDUCE 16u /* backward of a 2- / 3-layer net: leave the per-workgroup fp32 weight-gradient slabs in backward_buffer; the
                                 * caller sums them later (ngp_ffmlp_reduce_slabs_pair), e.g. two networks' slabs in one launch */
#define NGP_FF_RECOMPUTE 32u   /* backward of a 64-wide ReLU net with 32 inputs and 2 / 3 layers: forward_buffer is not read (may be NULL); the
                                 * hidden activations are recomputed from `inputs` -- bit for bit what the forward pass would have stored.
                                 * ngp_network_forward(training) with both forward buffers NULL is the matching forward: it does not store them */
int ngp_ffmlp_forward_ex(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                         uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                         void* forward_buffer, void* outputs, uint32_t flags, ngp_stream_t stream);
int ngp_ffmlp_inference_ex(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                           uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                           uint32_t output_activation, void* inference_buffer, void* outputs, uint32_t flags,
                           ngp_stream_t stream);
int ngp_ffmlp_backward_ex(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                          uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                          uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                          void* grad_inputs, void* grad_weights, uint32_t flags, ngp_stream_t stream);

/* ffmlp_backward with a caller-provided workspace (ngp_ffmlp_backward_workspace_bytes(...) bytes; 0 for the shapes served by the
 * register-resident kernels): the layered path splits the batch reduction of the weight gradients over sample chunks and keeps one fp32
 * slab per chunk there.  workspace == NULL: one chunk (same results to fp32 summation order, less parallelism). */
size_t ngp_ffmlp_backward_workspace_bytes(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);
int ngp_ffmlp_backward_ws(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                          uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                          uint32_t output_activation, int calc_grad_inputs, void* backward_buffer, void* grad_inputs, void* grad_weights,
                          uint32_t flags, void* workspace, size_t workspace_bytes, ngp_stream_t stream);

/* The whole network of nerf/network_ff.py:51-74 behind the encoder in ONE launch (64-wide ReLU networks with 32 inputs, the instant-ngp
 * configuration): sigma FFMLP -> trunc_exp / SH(4) / feature shuffle -> colour FFMLP -> sigmoid.  enc: the encoder output, [M,32] fp16
 * row-major or (flags & NGP_FF_INPUT_PLANAR) the encoder's own [16][M][2] layout; dirs [M_valid,3] fp32 (rows >= M_valid use dir = 0);
 * -> sigma [M] fp32 (= density_scale * exp(h0)), rgb [M,3] fp32.  training != 0 also writes what ngp_ffmlp_backward* and the
 * ngp_pipeline_*_backward kernels read: forward_buffer_sigma [nl_s,M,64], h16 [M,16], color_in [M,32], forward_buffer_color [nl_c,M,64]
 * (all fp16, the forward buffers in this library's private layout; both forward buffers NULL: not stored, for a backward that runs with
 * NGP_FF_RECOMPUTE).  Bit-identical to the sequence ngp_ffmlp_forward_ex ->
 * ngp_pipeline_mid_forward -> ngp_ffmlp_forward_ex -> ngp_pipeline_rgb_forward it replaces. */
int ngp_network_forward(const void* enc, const float* dirs, uint32_t M, uint32_t M_valid, const void* w_sigma, const void* w_color,
                        uint32_t num_layers_sigma, uint32_t num_layers_color, float density_scale, int training,
                        void* forward_buffer_sigma, void* h16, float* sigma, void* color_in, void* forward_buffer_color, float* rgb,
                        uint32_t flags, ngp_stream_t stream);

/* ngp_network_forward whose launch is sized for M rows but evaluates only the first ceil(*rows_dev / 32) tiles (rows_dev: optional device word, the
 * eval loop's emitted-row count; NULL: ngp_network_forward).  Buffer strides follow M. */
int ngp_network_forward_rows(const void* enc, const float* dirs, uint32_t M, uint32_t M_valid, const void* w_sigma, const void* w_color,
                             uint32_t num_layers_sigma, uint32_t num_layers_color, float density_scale, int training,
                             void* forward_buffer_sigma, void* h16, float* sigma, void* color_in, void* forward_buffer_color, float* rgb,
                             uint32_t flags, const uint32_t* rows_dev, ngp_stream_t stream);

/* Extensions for the fused training iteration (the backward of network_ff.py:40-74):
 *  - ngp_network_backward_color = ngp_ffmlp_backward_ex of the colour MLP (32 -> 64 x (n-1) -> 16, ReLU, n = 2 or 3) whose input-gradient
 *    epilogue writes the sigma net's output gradient grad_h16 [M,16] directly (what ngp_pipeline_mid_backward would assemble from
 *    grad_sigma [M], h16 [M,16] and dL/d(colour input)[:,16:31]); same bits, one launch and a [M,32] round trip less.
 *    flags: 0, NGP_FF_DEFER_REDUCE, NGP_FF_RECOMPUTE (forward_buffer_color may then be NULL).
 *  - ngp_ffmlp_backward_slab_count: how many fp32 slabs [n_params] a deferred backward of that shape leaves at the start of its
 *    backward_buffer (0: the gradients were stored directly, nothing to sum).
 *  - ngp_ffmlp_reduce_slabs_pair: sums two slab sets (n_slabs x n_params fp32 each, either may be empty) into fp16 weight
 *    gradients in ONE launch, in the fixed order of the single-set reduction (deterministic, same bits).  found_inf (optional device
 *    float): set to 1 when a resulting gradient is not finite -- the optimizer's non-finite sweep (ngp_optim_adam_step_ex, phase
 *    CHECK) done by the producer; a set with n_slabs = 0 (gradients already stored) is then only swept. */
int ngp_network_backward_color(const void* grad_out16, const void* color_in, const void* w_color, const void* forward_buffer_color, uint32_t M,
                               uint32_t num_layers_color, void* backward_buffer, const float* grad_sigma, const void* h16,
                               float density_scale, void* grad_h16, void* grad_w_color, uint32_t flags, ngp_stream_t stream);
uint32_t ngp_ffmlp_backward_slab_count(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);
int ngp_ffmlp_reduce_slabs_pair(const void* slabs_a, uint32_t n_slabs_a, uint32_t n_params_a, void* grad_weights_a, const void* slabs_b,
                                uint32_t n_slabs_b, uint32_t n_params_b, void* grad_weights_b, float* found_inf, ngp_stream_t stream);

/* network_ff.py:55-72 between the two MLPs: h16 [M,16] fp16 (sigma-net output), dirs [M_valid,3] fp32 ->
 * sigma [M] fp32 = exp(h[:,0]) (trunc_exp), color_in [M,32] fp16 = [SH deg 4 | h[:,1:16] | 0]; rows >= M_valid use dir = 0 */
int ngp_pipeline_mid_forward(const void* h16, const float* dirs, float* sigma, void* color_in, uint32_t M, uint32_t M_valid,
                             float density_scale, ngp_stream_t stream);
/* rgb [M,3] fp32 = fp16-rounded sigmoid of the colour net's first three outputs (network_ff.py:72) */
int ngp_pipeline_rgb_forward(const void* out16, float* rgb, uint32_t M, ngp_stream_t stream);
/* grad_out16 [M,16] fp16: columns 0..2 = grad_rgb * y (1 - y), the rest 0 */
int ngp_pipeline_rgb_backward(const float* grad_rgb, const float* rgb, void* grad_out16, uint32_t M, ngp_stream_t stream);
/* grad_h16 [M,16] fp16: column 0 = grad_sigma * exp(clamp(h0, -15, 15)) (activation.py:12-17), columns 1..15 =
 * grad_color_in[:,16:31] */
int ngp_pipeline_mid_backward(const float* grad_sigma, const void* h16, const void* grad_color_in, void* grad_h16, uint32_t M,
                              float density_scale, ngp_stream_t stream);

/* The Trainer's loss (nerf/utils.py:516,557: MSELoss(reduction='none')(pred, gt).mean(-1).mean()) and its gradient times the loss scale
 * in one launch: loss[0] = mean((image - target)^2) over n values, grad_image[i] = (2/n * (image[i] - target[i])) * loss_scale[0]
 * (loss_scale: device scalar, NULL = 1).  Deterministic (fixed-order reduction).  n must be > 0. */
int ngp_pipeline_mse_loss(const float* image, const float* target, uint32_t n, const float* loss_scale, float* loss,
                          float* grad_image, ngp_stream_t stream);

/* nn.Linear stack <-> the flat fp16 weight vector of the fused MLP (the reference's non---ff model, nerf/network.py:32-63 + 155-215, runs its
 * bias-free Linear / ReLU stacks on the ffmlp kernels: torch-ngp_amd/nerf/network.py).  weights[l]: fp32 [out_l, in_l] row major, layer 0
 * [hidden, n_in], layers 1 .. depth-2 [hidden, hidden], layer depth-1 [n_out, hidden] (n_out <= 16).  flat (ngp_linear_stack_flat_size
 * elements): half(W_0) with its columns zero-padded to a multiple of 16 | an exact identity [hidden, hidden] when `identity` (a stack with ONE
 * hidden layer: the kernels want two) | half(W_1 .. W_{depth-2}) | half(W_{depth-1}) zero-padded to 16 rows -- the layout of
 * ngp_ffmlp_forward's `weights`.  unpack_grad: the flat fp16 weight gradient back into fp32 tensors of the layers' shapes (every element
 * written; the padding's and the identity's gradient are dropped).  One launch each (PyTorch's pad / eye / cat and their autograd: ~8). */
#define NGP_LINEAR_STACK_MAX 8
uint32_t ngp_linear_stack_flat_size(uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity);
int ngp_linear_stack_pack(const float* const* weights, uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity,
                          void* flat_fp16, ngp_stream_t stream);
int ngp_linear_stack_unpack_grad(const void* grad_flat_fp16, uint32_t depth, uint32_t n_in, uint32_t hidden, uint32_t n_out, int identity,
                                 float* const* grads, ngp_stream_t stream);

/* dst [dst_rows, dst_cols] fp16 = src [src_rows, src_cols] (rows src_row_stride elements apart) in its top-left corner, zeros elsewhere: the row /
 * column padding ngp_ffmlp_forward wants (rows to a multiple of 128, columns to 16), one launch. */
int ngp_pad_2d_fp16(const void* src, uint32_t src_rows, uint32_t src_cols, uint32_t src_row_stride, void* dst, uint32_t dst_rows,
                    uint32_t dst_cols, ngp_stream_t stream);

/* Data path: the arithmetic of get_rays (nerf/utils.py:53-137).  poses [B,4,4] row-major camera-to-world; pixel
 * indices inds [B,N] int64 (inds_batch_stride = N) or [N] shared by every pose (inds_batch_stride = 0) or NULL (pixel n = n: a full
 * H x W frame with N = H * W); pixel p is (column p % W, row p / W), sampled at its centre.  rays_o, rays_d [B,N,3] fp32:
 * rays_d = normalize(((col + 0.5 - cx) / fx, (row + 0.5 - cy) / fy, 1)) . R^T,  rays_o = camera position. */
int ngp_rays_from_pixels(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t W, const int64_t* inds,
                         uint32_t inds_batch_stride, uint32_t N, float* rays_o, float* rays_d, ngp_stream_t stream);

/* march_rays_train with two conveniences for the fused renderer: the counter may be reset in-kernel and the sample rows no ray
 * writes are zeroed in-kernel (the reference contract keeps both with the caller: counter.zero_(), torch.zeros buffers). */
#define NGP_MARCH_RESET_COUNTER 1u
#define NGP_MARCH_ZERO_TAIL 2u
#define NGP_MARCH_SCAN_LAUNCH 8u /* testing: keep the separate scan launch between the passes (default with NGP_MARCH_RESET_COUNTER and
                                  * N <= 8192 rays: the write pass hands out the sample slots itself -- same slots, one launch less) */
#define NGP_MARCH_NOISE_FROM_SEED 4u /* `noises` points to ONE device uint32 (a seed that the caller changes from step to step) instead of N
                                      * floats: ray n starts at near + dt * u(n, seed), u a counter-based uniform draw in [0, 1) */
int ngp_march_rays_train_ex(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                            const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                            const float* noises, void* workspace, uint32_t flags, ngp_stream_t stream);
/* the same with near_far_from_aabb (raymarching.cu:92-145) folded into the first pass: nears / fars [N] are OUTPUTS, computed from the
 * box aabb [6] and min_near with that function's arithmetic (one launch less per training iteration) */
int ngp_march_rays_train_aabb(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                              uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* aabb, float min_near,
                              float* nears, float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                              const float* noises, void* workspace, uint32_t flags, ngp_stream_t stream);

/* march_rays (inference) that zeroes every sample slot it does not fill: the unused tail of each ray's n_step slots and the rows
 * between n_alive*n_step and zero_rows (>= n_alive*n_step), so the buffers need no memset; noises may be NULL (= no perturbation). */
int ngp_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                      const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                      const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                      const float* noises, uint32_t zero_rows, ngp_stream_t stream);

/* On-device inference loop: march_rays / composite_rays / alive-list compaction of NeRFRenderer.run_cuda's eval
 * branch (renderer.py:322-367) with the loop state on the DEVICE: state = int32[2] {n_alive, samples marched per ray so far}.  The host
 * launches for an upper bound `alive_bound` of the alive count (the value it last read back) and may issue many iterations between
 * read-backs; the kernels take the true count from `state`, derive n_step = max(min(n_total / n_alive, cap), 1) from it (renderer.py:349:
 * cap = 8 = n_step_cap 0; a caller may raise the cap for the tail of a frame, where a handful of surviving rays would otherwise need one
 * launch set per 8 samples -- a ray's samples and their compositing order do not depend on the chunking, the SAME n_total and n_step_cap
 * must go to the three calls of one iteration), and do nothing for lanes beyond it.  march: sample rows [n_alive * n_step, rows) are
 * zero-filled (rows >= min(n_total, cap * alive_bound) always suffices); noises may be NULL.  compact: writes the surviving ids in order to out_alive and the next iteration's state to
 * out_state (count forced to 0 once max_steps samples were marched); workspace: ngp_compact_rays_workspace_bytes(alive_bound).
 * Slot layout, n_step sequence and compaction order equal the host-driven loop's, so do the results. */
int ngp_march_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                       const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                       const float* noises, uint32_t rows, ngp_stream_t stream);
/* ngp_march_rays_dev that also PUBLISHES the rows that can carry a sample in this iteration (rows_used: device word = min(rows, n_alive * n_step
 * padded by the marchers' rule); only those rows are zero-filled) for consumers that are launched for a stale bound and stop at the device-side
 * count (ngp_grid_encode_forward_sel, ngp_network_forward_rows).  rows_used == NULL: ngp_march_rays_dev. */
int ngp_march_rays_dev_rows(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, const int32_t* rays_alive, const float* rays_t,
                            const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                            const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                            const float* noises, uint32_t rows, uint32_t* rows_used, ngp_stream_t stream);
int ngp_composite_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, float T_thresh, int32_t* rays_alive, float* rays_t,
                           const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                           ngp_stream_t stream);
int ngp_compact_rays_dev(const int32_t* state, uint32_t alive_bound, uint32_t n_total, uint32_t n_step_cap, uint32_t max_steps, const int32_t* rays_alive,
                         int32_t* out_alive, int32_t* out_state, void* workspace, ngp_stream_t stream);

/* n iterations of the eval loop (renderer.py:341-367: march_rays -> network -> composite_rays -> compaction of the alive list) issued by ONE
 * call -- EXTENSION: ngp_march_rays_dev -> ngp_grid_encode_forward_sched -> ngp_network_forward (inference, density scale
 * folded) -> ngp_composite_rays_dev -> ngp_compact_rays_dev per iteration, on the ping-pong alive lists / device state of the *_dev entry
 * points (iteration i works on state[(first_cur + i) & 1] and leaves the compacted list in the other one).  Launches are sized by `lanes`
 * and `rows` (host values: the caller's last known alive count); rows beyond the device-side count are zero rows, lanes beyond it idle.
 * `noises` (may be NULL) is handed to the first iteration of the call only.  Same kernels, arguments and order as the per-stage calls: the
 * image is the same, bit for bit.  Sample buffers xyzs / dirs [rows,3], deltas [rows,2], sigmas [rows], rgbs [rows,3] fp32 and enc
 * [L,rows,2] fp16 are caller-provided scratch; the network tensors are the fp16 copies (embeddings [n,2], w_sigma, w_color flat). */
typedef struct ngp_render_loop {
    int32_t* state;                 /* [2][2] {alive rays, steps marched} */
    int32_t* alive[2];
    float* rays_t; const float* rays_o; const float* rays_d; const float* nears; const float* fars; const uint8_t* grid; const float* noises;
    float* xyzs; float* dirs; float* deltas; void* enc; float* sigmas; float* rgbs;
    const void* embeddings; const int32_t* offsets; const float* level_cost_host; const void* w_sigma; const void* w_color;
    float* weights_sum; float* depth; float* image; void* compact_workspace;
    uint32_t* rows_used;            /* optional device word: the encoder / network launches stop at the rows the march of the same iteration emitted */
    uint32_t lanes, rows, n_total, n_step_cap, max_steps, cascade, grid_size;
    uint32_t L, H, gridtype, interp, num_layers_sigma, num_layers_color;
    int32_t align_corners;
    float bound, dt_gamma, T_thresh, S, density_scale;
} ngp_render_loop_t;
int ngp_render_iterations_dev(const ngp_render_loop_t* args, uint32_t n_iter, uint32_t first_cur, ngp_stream_t stream);

/* Empty-ray culling for the inference loop -- EXTENSION (no reference counterpart: renderer.py:330-333 starts every frame with all N rays
 * alive).  ngp_coarse_occupancy: per cascade a (H/4)^3 byte grid (x fastest), 1 when any voxel of the cell's 4^3 block or of one of its 26
 * neighbouring blocks is occupied (ngp_coarse_occupancy_bytes(C, H) bytes).  ngp_cull_rays: rays_alive[n] = n, or -1 for a ray whose segment
 * [near, far] provably (conservatively: half-cell sampling of the dilated grid, every cascade the marcher could select) meets no occupied
 * voxel -- such a ray would emit no sample, so dropping it from the initial alive list (ngp_compact_rays) leaves the image bit-identical. */
size_t ngp_coarse_occupancy_bytes(uint32_t C, uint32_t H);
int ngp_coarse_occupancy(const uint8_t* grid, uint32_t C, uint32_t H, uint8_t* coarse, ngp_stream_t stream);
int ngp_cull_rays(const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N, float bound, uint32_t C,
                  uint32_t H, const uint8_t* coarse, int32_t* rays_alive, ngp_stream_t stream);

/* composite_rays_train with NeRFRenderer.run_cuda's epilogue fused (renderer.py:316-318):
 *   image_out = image + (1 - weights_sum) * bg,  depth_out = clamp(depth - nears, 0) / (fars - nears)
 * bg_mode 0: off (= the reference op), 1: scalar background bg_scalar, 2: per-ray background bg [N,3].
 * The backward takes the gradient of image_out (and optionally of weights_sum, may be NULL when bg_mode != 0).
 * rows_used (backward, may be NULL): device scalar = number of sample rows the marcher handed out (first word of the
 * march_rays_train workspace).  When given, grad_sigmas / grad_rgbs may arrive uninitialised: the kernel zeroes every row it does not
 * write a gradient to (rows behind a ray's early termination, rows >= *rows_used); when NULL the caller pre-zeroes them (the
 * reference contract). */
int ngp_composite_rays_train_forward_ex(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                        uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth, float* image,
                                        int bg_mode, float bg_scalar, const float* bg, const float* nears, const float* fars,
                                        float* image_out, float* depth_out, ngp_stream_t stream);
int ngp_composite_rays_train_backward_ex(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                         const float* rgbs, const float* deltas, const int32_t* rays,
                                         const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                         float T_thresh, float* grad_sigmas, float* grad_rgbs, int bg_mode, float bg_scalar,
                                         const float* bg, const uint32_t* rows_used, ngp_stream_t stream);

/* The image-space middle of one TRAINING iteration in one launch (optional extension; what the four calls
 * ngp_composite_rays_train_forward_ex -> ngp_pipeline_mse_loss -> ngp_composite_rays_train_backward_ex -> ngp_pipeline_rgb_backward
 * compute, expression for expression: raymarching.cu:501-577 forward, renderer.py:316-318 finish, nerf/utils.py:516,557 loss,
 * raymarching.cu:602-682 backward, the sigmoid of network_ff.py:72 backward).  One wavefront per ray.
 *   in : sigmas [M], rgbs [M,3], deltas [M,2] fp32, rays [N,3] int32 (index, offset, count), bg as in the _ex calls (bg_mode 1 or 2),
 *        nears / fars [N], target [N,3] fp32, loss_scale (device scalar, NULL = 1)
 *   out: weights_sum [N], image_out [N,3], depth_out [N] (finished), loss [1] = mean((image_out - target)^2) (deterministic),
 *        grad_sigmas [M] fp32, grad_out16 [M,16] fp16 (columns 0..2 = dL/d(colour-net output) * loss_scale, rest 0) -- both may arrive
 *        uninitialised, every row is written (zeros where no gradient flows)
 *   loss may be NULL: the sum is then left to the caller (ngp_grid_encode_backward_checked_slabs carries it); ray_err [N] holds the per-ray squared errors.
 *   ray_err [N] fp32: scratch.  march_workspace: the workspace ngp_march_rays_train_ex filled for these rays, all
 *        ngp_march_rays_train_workspace_bytes(N) bytes of it (word 0 = rows handed out, word 1 and the 32 words at its end, 128 bytes
 *        apart = tickets that call leaves at 0 and this one returns to 0).  march_workspace_bytes: the size of that buffer -- a call with fewer
 *        than ngp_march_rays_train_workspace_bytes(N) bytes is refused (the group tickets sit at its end). */
int ngp_composite_train_loss_backward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M,
                                      uint32_t N, float T_thresh, int bg_mode, float bg_scalar, const float* bg, const float* nears,
                                      const float* fars, const float* target, const float* loss_scale, float* weights_sum,
                                      float* image_out, float* depth_out, float* loss, float* ray_err, float* grad_sigmas,
                                      void* grad_out16, void* march_workspace, size_t march_workspace_bytes, ngp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * freqencoder       (reference: freqencoder/src/freqencoder.h:6-10, bindings.cpp:5-8) -- SURVEY.md 8(f).3, fp32 only.
 * outputs [B,C], C = D + 2*D*deg: x | per frequency f: sin(2^f x_d) for all d, then cos(2^f x_d) for all d.
 * --------------------------------------------------------------------------------------------- */
/* replaces freq_encode_forward (freqencoder.cu:97-111) */
int ngp_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                            ngp_stream_t stream);
/* replaces freq_encode_backward (freqencoder.cu:114-131): grad [B,C], outputs [B,C] (saved), grad_inputs [B,D] overwritten */
int ngp_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float* grad_inputs, ngp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimizer + loss-scaling step -- EXTENSION: replaces torch.optim.Adam + GradScaler.step/update
 * (main_nerf.py:132, nerf/utils.py:557-560) for up to 8 tensors per call.  grads[k] holds the loss-scaled gradient, fp16 when
 * grad_is_half[k] & 1 (the buffer grid_encode_backward / ffmlp_backward wrote) else fp32, and is ZEROED by the call -- unless
 * grad_is_half[k] & 2: the producer of that buffer overwrites ALL of it every step (ngp_grid_encode_backward_checked_slabs with
 * overwrite_table), so it is left as it is (and a skipped step does not touch the tensor at all); params_fp16[k]
 * (optional, may be NULL per tensor or as a whole) receives the fp16 copy of the updated weights.
 * state = device float[8], 16-byte aligned: {loss scale, growth tracker, found_inf, Adam step count, lr multiplier, table parity (ngp_table_adam_t), ticket of
 * ngp_optim_adam_small_commit (0 between launches), scale_dead (sticky: set by COMMIT once the loss scale has underflowed -- 1 / scale not
 * finite -- after which every step is skipped: the run is dead and says so)};
 * no host sync.
 * grad_mult: extra factor on the gradients (1 / world_size after a SUM all-reduce).  A step over more than 8 tensors uses
 * ngp_optim_adam_step_ex below (phases), which keeps "a non-finite gradient anywhere skips the whole step" across calls; for
 * compatibility growth_interval < 0 here still means "CHECK + UPDATE of this chunk, no COMMIT".
 * A loss scale that has underflowed (1 / scale not finite in fp32: 0 or a denormal, after a long run of overflowing steps) counts as an
 * overflow of its own: UPDATE skips, COMMIT backs off -- GradScaler's unscale-then-check order, so that a zero gradient is never
 * multiplied by 1 / 0 into the weights. */
int ngp_optim_adam_step(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                        void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                        float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor, float growth_interval,
                        float* state, ngp_stream_t stream);

/* The same step in phases, for parameter sets of more than 8 tensors and for the data-parallel sharded update: CHECK sweeps the
 * gradients of this call's tensors into found_inf, UPDATE applies Adam (or skips) to them, COMMIT updates the loss scale and the step
 * count once (count may be 0 for a COMMIT-only call).  A caller with several chunks issues CHECK for every chunk, then UPDATE for every
 * chunk, then COMMIT -- a non-finite gradient anywhere then skips the step everywhere, as GradScaler.step does.
 * ema / ema_one_minus_decay (optional: NULL / 0): exponential moving average of the parameters in the same sweep,
 * ema[k] -= ema_one_minus_decay * (ema[k] - params[k]) on the updated parameters -- torch_ema's ExponentialMovingAverage.update()
 * (the reference Trainer's `ema`, nerf/utils.py:388-391,760-761,891-892); the caller computes the effective decay
 * min(decay, (1 + num_updates) / (10 + num_updates)). */
#define NGP_OPT_PHASE_CHECK 1u
#define NGP_OPT_PHASE_UPDATE 2u
#define NGP_OPT_PHASE_COMMIT 4u
/* with COMMIT: a step that stands also flips state[5], the parity of a speculatively updated table (ngp_table_adam_t) */
#define NGP_OPT_PHASE_FLIP 8u
int ngp_optim_adam_step_ex(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                           void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                           float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor, float growth_interval,
                           float* state, float* const* ema, float ema_one_minus_decay, uint32_t phases, ngp_stream_t stream);
/* The rest of a step whose hash table was updated inside the grid backward (ngp_table_adam_t): Adam (same rule, same arithmetic, the
 * skip-as-a-whole verdict read from state[2]) on what is left -- the few SMALL tensors (the two MLPs' weights, updated in place) and,
 * table != NULL, the first table_prefix_params parameters of the double-buffered table (the dense levels: read from buffer set state[5]
 * with the stored fp16 gradient table_grad_fp16, written to the other set; nothing is written when the step is skipped) -- followed by
 * the COMMIT of the loss scale / step count and, flip_parity != 0, the parity flip that makes the speculatively written table current.
 * ONE launch of a few workgroups; the last one to finish (a ticket in state[6]) commits.  count may be 0.  At most 8 tensors and 4 M
 * parameters in total; table_prefix_params a multiple of 4. */
struct ngp_table_adam;
int ngp_optim_adam_small_commit(int count, const uint64_t* n, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                                void* const* grads, void* const* params_fp16, const int* grad_is_half, const float* lr, float beta1,
                                float beta2, float eps, float grad_mult, float growth_factor, float backoff_factor, float growth_interval,
                                float* state, int flip_parity, const struct ngp_table_adam* table, const void* table_grad_fp16,
                                uint64_t table_prefix_params, ngp_stream_t stream);
/* Data-parallel sharded update (one process per GPU, reduce-scatter -> Adam on 1/world -> all-gather; the reference's dormant DDP hook,
 * nerf/utils.py:364-366, all-reduces everything): the global "skip this step" verdict without a collective of its own.
 * ngp_optim_poison_shards: when state[2] (found_inf of THIS rank's local gradient) is set, writes NaN into element 0 of each of the `shards`
 * shards (of `payload` fp16 elements) of the flat gradient, BEFORE the reduce-scatter; ngp_optim_shard_verdict: AFTER it, sets state[2] when
 * element 0 of this rank's averaged shard is not finite, and (flat_grad_fp16 != NULL) takes the poison out of the flat buffer again -- a
 * poisoned element inside padding would otherwise survive when the producers overwrite and nobody zeroes.  Both are single tiny launches,
 * graph-capturable. */
int ngp_optim_poison_shards(void* flat_grad_fp16, uint32_t shards, uint64_t payload, const float* state, ngp_stream_t stream);
int ngp_optim_shard_verdict(const void* shard_grad_fp16, float* state, void* flat_grad_fp16, uint32_t shards, uint64_t payload,
                            ngp_stream_t stream);
/* torch_ema's update() on its own (the Trainer calls it once per epoch): ema[k] -= one_minus_decay * (ema[k] - params[k]), up to 8
 * tensors per call */
int ngp_optim_ema_update(int count, const uint64_t* n, float* const* params, float* const* ema, float one_minus_decay, ngp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
