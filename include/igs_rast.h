/*
 * igs_rast.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the hot path of asd56585452/IGS: the entry points are what the
 * reference's torch glue binds to (DGR = submodules/RaDe-GS/submodules/diff-gaussian-rasterization):
 *
 *   igs_rast_forward       <-> CudaRasterizer::Rasterizer::forward   (DGR/cuda_rasterizer/rasterizer.h:31-65,
 *                              called from RasterizeGaussiansCUDA, DGR/rasterize_points.cu:98-130)
 *   igs_rast_backward      <-> CudaRasterizer::Rasterizer::backward  (rasterizer.h:67-112,
 *                              called from RasterizeGaussiansBackwardCUDA, rasterize_points.cu:197-242)
 *   igs_rast_mark_visible  <-> CudaRasterizer::Rasterizer::markVisible (rasterizer.h:24-29, rasterize_points.cu:259-263)
 *
 * Conventions (identical to the reference unless stated):
 *   - plain device pointers to contiguous fp32 / int32 data, row-major; no torch types;
 *   - optional inputs are signalled by NULL (the reference relies on data_ptr()==nullptr of empty tensors);
 *   - viewmatrix / projmatrix are the TRANSPOSED 4x4 matrices the callers build (row-vector convention);
 *   - the three scratch buffers are opaque byte buffers grown through callbacks; they must stay alive and
 *     unmodified until the matching backward call, and P, R (= the value forward returned), W, H must be the same;
 *   - `stream` is a hipStream_t (pass the framework's current stream; NULL = default stream).  All work is
 *     enqueued on it.  The host wait is bounded: IGS_RAST_WAIT_TIMEOUT_S seconds (default 10) without the status, a stream
 *     error, or a drained stream without it return IGS_RAST_E_HIP.  forward performs ONE host wait (for the 12-byte frame status {num_rendered, slab overflow, prefilter
 *     flag} that the blend kernel posts into pinned host memory; the reference reads its count back at
 *     rasterizer_impl.cu:354); backward performs none;
 *   - threading: the library keeps a little state PER HOST THREAD (the pinned status slot, the per-tile slab size that worked
 *     for the last frames, the pending frame of igs_rast_forward_async, the last error text); calls from different threads
 *     do not interfere, a frame started on one thread must be finished / differentiated on the same thread.  The optional
 *     stage profiler (igs_rast_profile_*) is process-wide and meant for single-threaded benchmarking;
 *   - image outputs of forward must be zero-filled by the caller when P == 0 (nothing is launched, as in
 *     rasterize_points.cu:90); for P > 0 every pixel of every output is written;
 *   - any of backward's seven upstream image gradients may be NULL, meaning all zeros (the output did not take part in
 *     the loss); geometry branches without any upstream gradient are skipped, the result is the same;
 *   - backward writes every element of its eight outputs (no pre-zeroing needed) and uses a caller-provided
 *     workspace of igs_rast_backward_workspace_bytes(P) bytes (contents undefined on entry).
 *
 * Return values: forward returns num_rendered (>= 0) or a negative IGS_RAST_E_* code; the others return 0 or a
 * negative code.  igs_rast_last_error() returns a static, thread-local description of the last failure.
 */
#ifndef IGS_RAST_H
#define IGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGS_RAST_VERSION 4

#define IGS_RAST_E_INVALID   (-1)   /* bad argument (NULL required pointer, negative size, ...) */
#define IGS_RAST_E_HIP       (-2)   /* a HIP runtime call or kernel launch failed */
#define IGS_RAST_E_ALLOC     (-3)   /* a scratch callback returned NULL */
#define IGS_RAST_E_PREFILTER (-4)   /* `prefiltered` set but a point was culled (the reference __trap()s, auxiliary.h:172-176) */
#define IGS_RAST_E_CHANNELS  (-5)   /* reserved: non-RGB without precomputed colours (rasterizer_impl.cu:308-311) */

#define IGS_RAST_E_RETRY     (-6)   /* igs_rast_forward_finish: a tile overflowed its instance slab: redo the frame */

/* Scratch growth callback: make the buffer at least `bytes` long and return its device address
 * (the reference's std::function<char*(size_t)> resizeFunctional, rasterize_points.cu:27-33). */
typedef char* (*igs_rast_alloc_fn)(void* user, size_t bytes);

int igs_rast_version(void);
const char* igs_rast_last_error(void);

int igs_rast_forward(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user,
    igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M,
    const float* background,              /* [3] */
    int width, int height,
    const float* means3D,                 /* [P,3] */
    const float* shs,                     /* [P,M,3] or NULL */
    const float* colors_precomp,          /* [P,3] or NULL */
    const float* opacities,               /* [P] */
    const float* scales,                  /* [P,3] or NULL */
    float scale_modifier,
    const float* rotations,               /* [P,4] (w,x,y,z) or NULL */
    const float* cov3D_precomp,           /* [P,6] or NULL */
    const float* viewmatrix,              /* [16] */
    const float* projmatrix,              /* [16] */
    const float* cam_pos,                 /* [3] */
    float tan_fovx, float tan_fovy,
    float kernel_size,
    int prefiltered,
    float* out_color,                     /* [3,H,W] */
    float* out_coord,                     /* [3,H,W] */
    float* out_mcoord,                    /* [3,H,W] */
    float* out_depth,                     /* [1,H,W] */
    float* out_mdepth,                    /* [1,H,W] */
    float* out_alpha,                     /* [1,H,W] */
    float* out_normal,                    /* [3,H,W] */
    int* radii,                           /* [P] */
    int require_coord, int require_depth,
    int debug);                           /* debug != 0: synchronise and check after every launch (auxiliary.h:404-411) */

/* Asynchronous pair (no reference counterpart; used by the native refine step so the GPU never waits for the host):
 * igs_rast_forward_async = igs_rast_forward without the final wait for the instance count; it returns INT_MAX ("not known
 * yet"), which igs_rast_backward accepts as R.  igs_rast_forward_finish() waits for the 12-byte
 * read-back (done right after the tile scan) and returns the true num_rendered, or IGS_RAST_E_RETRY when the guessed
 * per-tile instance slabs were too small: whatever was enqueued on top of the frame must then be discarded and the frame
 * redone with igs_rast_forward. */
int igs_rast_forward_async(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug);
int igs_rast_forward_finish(void);

/* Forward for stream capture (hipGraph / torch.cuda.graph; no reference counterpart -- the reference's forward reads its
 * instance count back in the middle, rasterizer_impl.cu:354, and cannot be captured): the launches of igs_rast_forward with no
 * host-side wait and no pending latch; returns INT_MAX like igs_rast_forward_async.  Nothing it does is illegal on a capturing
 * stream PROVIDED one ordinary igs_rast_forward has run on this host thread and device before (the pinned status slot is
 * allocated then) and the scratch callbacks do not allocate illegally (PyTorch's graph-pool allocations are fine).
 * Every replay posts {num_rendered, overflow, prefilter flag} into the thread's status slot; igs_rast_last_status() returns
 * them once the caller has synchronised the stream.  overflow != 0 means the per-tile instance slabs baked into the capture
 * were too small for that replay: its results are invalid, the slab hint has been raised, capture again.
 * igs_rast_last_status() checks the slot's sequence word against the number baked into the calling thread's last
 * igs_rast_forward_nowait: before the first replay has run (or when an eager forward has posted since) it returns
 * IGS_RAST_E_RETRY instead of an older frame's numbers. */
int igs_rast_forward_nowait(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug);
int igs_rast_last_status(int* num_rendered, unsigned* overflow, unsigned* prefilter_flag);
/* The same without the sequence check, for callers who replay SEVERAL captured graphs in turn (only the last capture's number is
 * remembered) and know that a forward has run since: what the last forward that EXECUTED on this thread's slot posted. */
int igs_rast_last_posted_status(int* num_rendered, unsigned* overflow, unsigned* prefilter_flag);

/* One-shot promise for the NEXT igs_rast_forward / _async / _nowait of the calling thread: the image buffer its callback will hand out
 * was zero-filled when it was allocated and has been used by this library only since.  Every slab-binned forward leaves the binning
 * counters in the image buffer zeroed behind it, so such a forward needs no zero-fill launch (igs_refine_step_args::scratch_clean is
 * the same promise for the fused step).  A broken promise costs a wrong or redone frame or an error code, never a fault. */
void igs_rast_hint_scratch_clean(int on);

/* Binning scratch tuning (no reference counterpart).  The default path gives every 16x16 tile a slab of `slots_per_tile`
 * instance slots (12 bytes each) in binningBuffer; a frame in which some tile needs more is redone automatically with larger
 * slabs (and, beyond 16384 per tile, with the global radix sort), and the size then sticks for the calling thread.  Setting the
 * hint up front avoids that first redo; 0 restores the default (1024).  get returns the current value. */
void igs_rast_set_slab_hint(unsigned slots_per_tile);
unsigned igs_rast_get_slab_hint(void);

size_t igs_rast_backward_workspace_bytes(int P);

int igs_rast_backward(
    void* stream,
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* alphas,                  /* forward's out_alpha */
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    float kernel_size,
    const int* radii,
    const float* normalmap,               /* forward's out_normal */
    const char* geom_buffer,
    const char* binning_buffer,
    const char* image_buffer,
    const float* dL_dpix,                 /* [3,H,W] */
    const float* dL_dpix_coord,           /* [3,H,W] */
    const float* dL_dpix_mcoord,          /* [3,H,W] */
    const float* dL_dpix_depth,           /* [1,H,W] */
    const float* dL_dpix_mdepth,          /* [1,H,W] */
    const float* dL_dalphas,              /* [1,H,W] */
    const float* dL_dpixel_normals,       /* [3,H,W] */
    void* workspace,
    float* dL_dmean2D,                    /* [P,3]  (z = sum |.|, the GOF densification statistic) */
    float* dL_dcolor,                     /* [P,3] */
    float* dL_dopacity,                   /* [P] */
    float* dL_dmean3D,                    /* [P,3] */
    float* dL_dcov3D,                     /* [P,6] */
    float* dL_dsh,                        /* [P,M,3] (ignored when M == 0) */
    float* dL_dscale,                     /* [P,3] */
    float* dL_drot,                       /* [P,4] */
    int require_coord, int require_depth,
    int debug);

int igs_rast_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                          const float* projmatrix, uint8_t* present /* [P], 0/1 */);

/* ---- introspection for tests and the roofline harness (no reference counterpart) ---- */

/* Copies per-stage scratch contents into caller-provided DEVICE arrays (any may be NULL) so that the HIP
 * stages can be compared one by one with the CPU oracle's intermediates:
 *   rec32     [P,32] the packed per-Gaussian record (layout in igs_amd/csrc/common.h)
 *   tiles     [P]    tiles touched
 *   point_list[R]    sorted Gaussian ids
 *   ranges    [T,2]  per-tile [start,end)
 *   n_contrib [2,H,W] last / median contributor counts */
int igs_rast_debug_dump(void* stream, int P, int R, int width, int height,
                        const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                        float* rec32, uint32_t* tiles, uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib);

/* Optional per-stage timing with HIP events recorded on the caller's stream (used by bench.py for the roofline line).
 * Stage order of the arrays (IGS_RAST_NSTAGES entries): preprocess, depth_sort, scan, emit, tile_sort, ranges,
 * blend_fwd, memset, blend_bwd, geom_bwd.  igs_rast_profile_read synchronises on the last recorded event; r_sum is the
 * sum of num_rendered over the forward calls seen, calls their number.
 * igs_rast_profile_enable(N): 0 = off, N > 0 = mark every N-th frame (forward call / refine step); an event record costs a
 * few microseconds of stream time, so marking every frame slows a 0.35 ms refine step by about 10 %, every 8th by about 1 %.
 * In igs_refine_step the geom_bwd stage includes the activation backward and the Adam update. */
#define IGS_RAST_NSTAGES 10
/* Which backward tile-blend instance the last igs_rast_backward / igs_refine_step of this process launched (any thread): bit 0 coord, bit 1
 * depth, bit 2 normal gradients present, bit 3 the |screen-space gradient| moment; -1 = none.  (The reference instantiates from
 * require_coord / require_depth alone, backward.cu:1153-1160; here branches whose upstream gradients are all NULL are left out.) */
/* Options of the NEXT igs_rast_backward of this host thread (extension; one-shot, consumed by that call): the reference's argument
 * list has no room for them and stays as it is.
 *  nan_report != 0: the reference's Python backward ends with seven `assert not torch.isnan(grad).any()` -- seven reductions and host
 *   syncs (DGR/diff_gaussian_rasterization_rade/__init__.py:156-162).  The per-Gaussian kernel instead tests every element it writes of
 *   dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D, dL_dsh, dL_dscale, dL_drot (not dL_dcov3D: the reference does not look at it either)
 *   and, if it wrote a NaN, says so in one word of pinned host memory; an event is recorded behind the kernel.  igs_rast_nan_report_wait()
 *   blocks until that event has completed and returns 1 (a NaN was written), 0 (none) or a negative error code.  Bounded like the forward's status wait (IGS_RAST_WAIT_TIMEOUT_S).
 *  clamp_grads > 0: dL_dmean3D, dL_dsh, dL_dopacity, dL_dscale, dL_drot are clamped to +-clamp_grads as they are written (what the clamp
 *   package does with five torch.clamp calls afterwards, DGRC/diff_gaussian_rasterization_rade_clamp/__init__.py:156-162); a NaN stays
 *   a NaN for the report, as it does through torch.clamp. */
void igs_rast_next_backward_options(int nan_report, float clamp_grads);
int igs_rast_nan_report_wait(void);
/* The same wait from ANOTHER host thread or at a later time (PyTorch runs a Function's backward on its own worker thread and the
 * assert is better raised once the whole backward pass has been enqueued): right after the igs_rast_backward that was asked for a report,
 * on the thread that called it, igs_rast_nan_report_handle() returns a ticket for the verdict (valid for the life of the process; the
 * last 256 reports of a thread stay readable) and its sequence number, instead of waiting;
 * igs_rast_nan_report_wait_at(word, seq) then blocks (bounded by IGS_RAST_WAIT_TIMEOUT_S) and returns 1 / 0 / a negative code. */
int igs_rast_nan_report_handle(const void** ticket, unsigned* seq);
int igs_rast_nan_report_wait_at(const void* ticket, unsigned seq);
int igs_rast_last_backward_instance(void);
/* Test hook: overwrites the LDS of every CU with NaN bit patterns (a kernel that reads LDS it never wrote then fails small parity
 * tests instead of passing on a fresh device's zeros). */
int igs_rast_debug_poison_lds(void* stream);
int igs_rast_profile_enable(int on);
int igs_rast_profile_read(double* ms_sum, long long* count, double* r_sum, long long* calls, int reset);

/* ---- refine-loop helpers ("next" rows of SURVEY.md 8f: fused loss / fused multi-group Adam) ---- */

/* One Adam step over a flat fp32 parameter span, torch.optim.Adam semantics without weight decay / amsgrad, as
 * built by GaussianModel.load_fromstream (igs/models/gaussian_model.py:295-348: Adam(lr=0, eps=1e-15), per-group lr).
 * bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t). */
int igs_adam_step(void* stream, size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt);

/* The same for up to 8 parameter groups in ONE launch: group k covers param[offset[k] .. offset[k]+count[k]) with lr[k]. */
int igs_adam_step_groups(void* stream, int ngroups, const size_t* offset, const size_t* count, const float* lr, float* param,
                         const float* grad, float* exp_avg, float* exp_avg_sq, float beta1, float beta2, float eps,
                         float bias_correction1, float bias_correction2_sqrt);

/* The same for up to 8 SEPARATE tensors in one launch (each parameter of the reference's optimiser is its own nn.Parameter with its own
 * gradient and state tensors, torch.optim.Adam keeps one step count per parameter: gaussian_model.py:303-348): arrays of `ntensors`
 * device pointers / element counts / learning rates / bias corrections in HOST memory, read before the call returns. */
int igs_adam_step_multi(void* stream, int ntensors, float* const* param, const float* const* grad, float* const* exp_avg,
                        float* const* exp_avg_sq, const size_t* count, const float* lr, const float* bias_correction1,
                        const float* bias_correction2_sqrt, float beta1, float beta2, float eps);

/* The same with the step counts in DEVICE memory (step[k]: one float per tensor = the number of COMPLETED steps, the layout of
 * torch.optim.Adam(capturable=True)): every workgroup computes its bias corrections from step + 1, the workgroup that finishes last
 * advances the counts.  Nothing of the call depends on host state that changes from step to step, so it can be captured into a
 * hipGraph and replayed.  done_scratch: igs_adam_step_multi_dev_scratch_words() 32-bit words, zero before the first call and zero
 * again after every call (a buffer the caller zero-fills once and keeps; calls that share it must be ordered on one stream). */
size_t igs_adam_step_multi_dev_scratch_words(void);
int igs_adam_step_multi_dev(void* stream, int ntensors, float* const* param, const float* const* grad, float* const* exp_avg,
                            float* const* exp_avg_sq, const size_t* count, const float* lr, float* const* step, unsigned* done_scratch,
                            float beta1, float beta2, float eps);

/* ---- one whole refine iteration on one view, single GPU --------------------------------------------------------------
 * Native form of the body of the reference's per-frame refine loop (infer_batch.py:279-324 with the L1 photometric loss,
 * igs/utils/loss_utils.py:17; activations of igs/models/gaussian_model.py:90-127; torch.optim.Adam of :295-348):
 *   opacity = sigmoid, scale = exp, rotation = normalize  ->  render  ->  loss = loss_weight * mean|color - gt|
 *   ->  backward through rasterizer and activations  ->  Adam update of the five parameter groups, in place.
 * Everything is enqueued on `stream` without a host wait; the gradients never reach HBM (the per-Gaussian backward kernel
 * applies the update itself).  `param` / `exp_avg` / `exp_avg_sq` are flat fp32 buffers holding the five groups at the given
 * float offsets: xyz [P][3], rotation [P][4] (raw quaternion), shs [P][M][3], opacity [P] (logit), scale [P][3] (log).
 * Multi-GPU runs need the gradients for the exchange: with `grad_out` set the same launches end in the flat gradient
 * instead of the update (then all-reduce it and call igs_adam_step_groups -- or, with `color_grad_out` set as well, gather the
 * per-view colour gradients, all-reduce only the 11 small-group floats, and let igs_adam_sh_from_view_colors update the SH
 * coefficients: the SH span of `grad_out` is then left untouched).
 * Returns num_rendered or a negative error code. */
typedef struct igs_refine_step_args {
    void* stream;
    igs_rast_alloc_fn geometry_buffer; void* geometry_user;
    igs_rast_alloc_fn binning_buffer;  void* binning_user;
    igs_rast_alloc_fn image_buffer;    void* image_user;
    void* workspace;                          /* igs_rast_backward_workspace_bytes(P) */
    int P, D, M, width, height;
    const float* background;                  /* [3] device */
    float *param, *exp_avg, *exp_avg_sq;      /* flat optimiser state, device */
    float* grad_out;                          /* NULL: apply Adam in place.  Non-NULL (multi-GPU): write the flat gradient of the raw
                                                 leaves here (same offsets) and leave param / exp_avg / exp_avg_sq untouched */
    size_t off_xyz, off_rot, off_sh, off_opacity, off_scale;
    float lr_xyz, lr_rot, lr_sh, lr_opacity, lr_scale;
    float beta1, beta2, eps;
    int step;                                 /* 1-based number of this update (bias correction) */
    const float *viewmatrix, *projmatrix, *cam_pos;   /* device */
    float tan_fovx, tan_fovy;
    const float* gt;                          /* [3][H][W] device */
    float loss_weight;                        /* loss = loss_weight * ((1 - lambda_dssim) * mean|color - gt| + lambda_dssim * (1 - mean SSIM)) */
    float lambda_dssim;                       /* 0: pure L1 (fused into the blend backward); the reference uses 0.2 (loss_utils.py:34-63) */
    float lambda_depth_normal;                /* > 0 adds loss_weight * lambda_depth_normal * depth-normal regulariser (RaDe-GS train.py:143-160;
                                                 needs require_depth): the backward then runs its <depth, normal> instance */
    float depth_ratio;                        /* weight of the median-depth term of the regulariser (0 = the reference's 0.6) */
    void* loss_scratch;                       /* lambda_dssim > 0 or lambda_depth_normal > 0: igs_refine_loss_scratch_bytes(width, height) bytes */
    float* out_images;                        /* [15][H][W]: color 3 | coord 3 | mcoord 3 | depth 1 | mdepth 1 | alpha 1 | normal 3 */
    int* radii;                               /* [P] */
    float* dL_dmean2D;                        /* [P][3] view-space gradient (densification statistic) or NULL = not wanted: the
                                                 blend backward then leaves out the |gradient| sum nothing else reads */
    float* loss_out;                          /* device, 1 float: the loss value, or NULL */
    int require_coord, require_depth;
    float clamp_grads;                        /* > 0: the rasterizer's gradients w.r.t. means3D / sh / opacities / scales / rotations are clamped to
                                                 +-clamp_grads before they go on (diff_gaussian_rasterization_rade_clamp, 15); 0: off */
    float* color_grad_out;                    /* [P][3] or NULL: dL/d(colour) of this view per Gaussian (clamped channels and unseen Gaussians
                                                 zero) -- what the ranks of a multi-GPU step gather instead of all-reducing dL/dSH */
    int scratch_clean;                        /* != 0: the caller vouches that the image buffer the callback hands out was zero-filled when
                                                 it was allocated and has since been used by this library only (every slab-binned forward
                                                 leaves the binning counters in it zeroed again): the per-frame zero-fill launch is skipped.
                                                 0: no assumption (a fresh, uninitialised buffer is fine) */
    float* gt_stats;                          /* lambda_dssim > 0, optional (NULL = off): igs_ssim_gt_stats_bytes(width, height) bytes that belong to
                                                 THIS ground-truth image.  loss_utils.py:34-63 blurs gt and gt^2 again in every iteration; with a
                                                 buffer the step that finds gt_stats_valid == 0 stores the two maps in it and every later step on the
                                                 same gt (gt_stats_valid != 0) reads them instead of recomputing them -- same values, 3 blurs for 5 */
    int gt_stats_valid;
    void* color_ready_event;                  /* optional hipEvent_t (with grad_out and color_grad_out, multi-GPU): color_grad_out is then written
                                                 right after the blend backward -- one kernel before the step ends -- and this event is recorded on
                                                 `stream` behind it, so that the caller can start the all-gather of the colour gradients on another
                                                 stream underneath the per-Gaussian kernel.  NULL: written by the last kernel of the step.
                                                 Queue the wait on the event only AFTER igs_refine_step has returned: in the rare frame that is
                                                 redone with larger slabs the event is recorded twice, and only the second record is behind valid data */
} igs_refine_step_args;
int igs_refine_step(const igs_refine_step_args* args);
size_t igs_refine_step_args_size(void);       /* sizeof(igs_refine_step_args) of the loaded library: bindings check it before the first call */

/* Multi-GPU refine step (extension; views are sharded over the ranks): dL/dSH of the step = sum over its views of
 * basis(direction_v) x dL/dcolour_v (backward.cu:21-140, the W(k, b) rows of the SH backward).  Every rank all-gathers the
 * 3-float colour gradients of all views (`color_grad_out` of igs_refine_step: 12 bytes per Gaussian and view instead of a
 * 192-byte SH gradient in an all-reduce) and rebuilds the sum itself, views in the order given -- identical bits on every rank.
 *   campos [n_views][3] in HOST memory (read before the call returns; n_views <= 64), color_grads [n_views][P][3] and means3D on the
 *   device, dL_dsh [P][M][3] (overwritten), clamp_grads as in igs_refine_step. */
int igs_sh_grad_from_view_colors(void* stream, int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                 const float* color_grads, float clamp_grads, float* dL_dsh);
/* The same sum applied directly as the Adam update of the SH coefficients (torch.optim.Adam semantics as igs_adam_step_groups:
 * lr / bias_correction1, sqrt(v) / bias_correction2_sqrt + eps); param_sh / exp_avg_sh / exp_avg_sq_sh = the [P][M][3] SH spans of
 * the optimiser state.  The rebuilt gradient is never written to memory. */
int igs_adam_sh_from_view_colors(void* stream, int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                 const float* color_grads, float clamp_grads, float* param_sh, float* exp_avg_sh, float* exp_avg_sq_sh,
                                 float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt);
/* The whole optimiser step of an N > 1 rank in one launch, after the exchange: the SH update as igs_adam_sh_from_view_colors plus the
 * four small groups (xyz, rotation, opacity, scale) from their all-reduced gradients in the flat `grad` buffer (same arithmetic as
 * igs_adam_step_groups).  Flat buffers and float offsets as in igs_refine_step_args; M = 0 skips the SH part. */
int igs_adam_exchange_step(void* stream, int P, int D, int M, int n_views, const float* campos, const float* color_grads,
                           float clamp_grads, float* param, float* exp_avg, float* exp_avg_sq, const float* grad,
                           size_t off_xyz, size_t off_rot, size_t off_sh, size_t off_opacity, size_t off_scale,
                           float lr_xyz, float lr_rot, float lr_sh, float lr_opacity, float lr_scale,
                           float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt);
size_t igs_refine_loss_scratch_bytes(int width, int height);

/* Photometric loss of the refine loop, forward + backward in two launches (igs/utils/loss_utils.py:17-63; infer_batch.py:300-306):
 *   loss = weight * ((1 - lambda_dssim) * mean|pred - gt| + lambda_dssim * (1 - mean SSIM(pred, gt))),   grad = dloss/dpred.
 * SSIM: 11x11 Gaussian window (sigma 1.5), zero padding, C1 = 1e-4, C2 = 9e-4, mean over all elements.  `scratch` holds
 * igs_ssim_l1_scratch_bytes(width, height) bytes.  `sums` (device, 2048 floats, or NULL) receives the 64 SSIM-sum shards at
 * [16*s] and the 64 L1-sum shards at [1024 + 16*s]; the caller adds them up and forms the loss value. */
size_t igs_ssim_l1_scratch_bytes(int width, int height);
int igs_ssim_l1_loss_fwd_bwd(void* stream, int width, int height, const float* pred, const float* gt, float lambda_dssim, float weight,
                             void* scratch, float* grad, float* sums);
/* The same with a cache of the two statistics that depend on the ground truth alone, blur(gt) and blur(gt^2) (the reference blurs them
 * again in every iteration; gt does not change while a frame is refined): `gt_stats` = igs_ssim_gt_stats_bytes(width, height) bytes
 * owned by the caller, one buffer per ground-truth image.  gt_stats_valid == 0: computed as usual AND stored; != 0: read (3 blurs
 * instead of 5).  Same values either way.  gt_stats == NULL: igs_ssim_l1_loss_fwd_bwd. */
size_t igs_ssim_gt_stats_bytes(int width, int height);
int igs_ssim_l1_loss_fwd_bwd_cached(void* stream, int width, int height, const float* pred, const float* gt, float lambda_dssim, float weight,
                                    void* scratch, float* grad, float* sums, float* gt_stats, int gt_stats_valid);
/* The SSIM term alone with its VALUE finished on the device: mean_out[0] = mean SSIM(pred, gt) over all 3 * width * height elements,
 * grad = d(mean SSIM)/d pred; same scratch as igs_ssim_l1_loss_fwd_bwd, two launches, nothing for the host to add up
 * (igs_amd.losses.ssim; loss_utils.py:34-63). */
int igs_ssim_mean_fwd_bwd(void* stream, int width, int height, const float* pred, const float* gt, void* scratch, float* grad,
                          float* mean_out);

/* RaDe-GS depth-normal consistency regulariser, value and gradients in one launch
 * (submodules/RaDe-GS/utils/graphics_utils.py:97-126, train.py:143-160):
 *   loss = weight * ((1 - depth_ratio) * mean(1 - normal . n(depth)) + depth_ratio * mean(1 - normal . n(mdepth))),
 * n(d) = normalised cross product of the central differences of the back-projected depth map (zero on the image border).
 * Writes dloss/ddepth [H][W], dloss/dmdepth [H][W], dloss/dnormal [3][H][W]; the loss value is left as 64 partial sums at
 * loss_shards[16*s] (1024 floats, zero-filled here). */
int igs_depth_normal_loss_fwd_bwd(void* stream, int width, int height, float tan_fovx, float tan_fovy, const float* depth,
                                  const float* mdepth, const float* normal, float weight, float depth_ratio, float* g_depth,
                                  float* g_mdepth, float* g_normal, float* loss_shards);

/* Fused L1 loss forward + backward (igs/utils/loss_utils.py:17-18): grad[i] = sign(pred[i] - gt[i]) * scale, and
 * sum |pred - gt| is accumulated into 64 shards loss_sum[16*s], s = 0..63 (1024 floats, zeroed by the caller, summed by
 * the caller: same-address atomics would serialise). */
int igs_l1_loss_fwd_bwd(void* stream, size_t n, const float* pred, const float* gt, float* grad, float* loss_sum, float scale);
/* The same with the VALUE finished on the device, in one launch: mean_out[0] = mean |pred - gt|, grad[i] = sign(pred[i] - gt[i]) / n.
 * partials: 1024 floats of scratch; counter: 33 * 64 words (2112) that are zero on entry and zero again on exit (a buffer the caller zero-fills once
 * and keeps; calls that share it must be ordered on one stream). */
int igs_l1_mean_fwd_bwd(void* stream, size_t n, const float* pred, const float* gt, float* grad, float* mean_out, float* partials,
                        unsigned* counter);

/* On-disk format <-> parameter store on the device (extension).  The host reads / writes the file; these do the
 * per-Gaussian re-layout and arithmetic of igs/models/gs.py:400-462 (load_ply) and :297-343 (save_ply) in one pass each.
 *   igs_ply_to_params: `table` = the PLY's vertex element as [P][stride] float32 on the device; `cols` (HOST memory, n_cols = 15 + 3 (K - 1)
 *     entries) = the column of x y z | f_dc_0..2 | f_rest_0 .. f_rest_{3(K-1)-1} | opacity | scale_0..2 | rot_0..3 | filter_3D (-1: absent).
 *     Writes xyz [P,3], rotation [P,4], shs [P,K,3] (the file is channel-major), opacity [P] (logit), scaling [P,3] (log); with a filter_3D
 *     column the Mip-Splatting filter is folded in exactly as gs.py:480-490 + inverse_sigmoid / log (:451-455).
 *   igs_params_to_ply: the [P][14 + 3 K] table save_ply writes (normals zero, f_dc / f_rest channel-major). */
int igs_ply_to_params(void* stream, int P, const float* table, int stride, const int* cols, int n_cols, int K,
                      float* xyz, float* rotation, float* shs, float* opacity, float* scaling);
int igs_params_to_ply(void* stream, int P, int K, const float* xyz, const float* rotation, const float* shs, const float* opacity,
                      const float* scaling, float* table);

/* Morton (Z-order) permutation of the Gaussians' positions (extension, no reference counterpart; used by the refine loop's store so
 * that consecutive Gaussians project to neighbouring tiles -- the binning stage then reserves instance slots once per (workgroup,
 * tile) instead of once per instance).  perm[i] = index of the Gaussian that comes i-th; ties keep their order (stable radix sort
 * of the library itself, no PyTorch / rocPRIM sort on the stream path).  lohi: {lo.xyz, hi.xyz} of the positions, 6 floats in
 * DEVICE memory; bits: 1..10 per axis; scratch: igs_morton_order_scratch_bytes(P) bytes. */
size_t igs_morton_order_scratch_bytes(int P);

/* Test support: the per-tile sort of the slab binning (the replacement of cub::DeviceRadixSort::SortPairs + identifyTileRanges,
 * rasterizer_impl.cu:373-391, for instances that were binned per tile) on caller-made slabs.  tile_count[T]: instances per tile (reset
 * to zero by the call); pairs[T * slab]: depth bits << 32 | Gaussian id; out: point_list[T * slab] = the ids of every tile's slab in
 * ascending key order, ranges[2 T] = {t * slab, t * slab + count} (empty for a tile whose count exceeds `slab`), stats[4]: [1] = the
 * largest such count.  ids are clamped to P - 1.  All pointers DEVICE memory. */
int igs_debug_tile_sort(void* stream, int T, unsigned int* tile_count, const unsigned long long* pairs, unsigned int* point_list,
                        unsigned int* ranges, int slab, unsigned int* stats, int P);
int igs_morton_order(void* stream, int P, const float* xyz, const float* lohi, int bits, void* scratch, int* perm);

/* Densification support (igs/models/gaussian_model.py:586-663,865-868; driven by infer_batch.py:308-321).
 * igs_densify_stats: per-step statistics of add_densification_stats + the max_radii2D update, for Gaussians with radii > 0:
 *   grad_accum += ||dL_dmean2D[:2]||, denom += 1, max_radii = max(max_radii, radii).
 * igs_densify_remap: rebuilds the flat optimiser state (param / exp_avg / exp_avg_sq, five groups at off_*[5] = xyz, rotation,
 *   shs, opacity, scaling) after clone / split / prune in one pass: new Gaussian i copies old Gaussian src[i]; fresh[i] != 0
 *   zeroes its Adam moments; ovr[i] >= 0 takes position and log-scale from row ovr[i] of ovr_xyz / ovr_scale (split children).
 *   The selection itself (masks, top-k, sampling) is host logic: igs_amd/densify.py. */
int igs_densify_stats(void* stream, int P, const float* dL_dmean2D, const int* radii, float* grad_accum, float* denom, float* max_radii);
int igs_densify_remap(void* stream, int P_new, int M, const int* src, const int* fresh, const int* ovr, const float* ovr_xyz,
                      const float* ovr_scale, const float* param_old, const float* exp_avg_old, const float* exp_avg_sq_old,
                      const size_t* off_old, float* param_new, float* exp_avg_new, float* exp_avg_sq_new, const size_t* off_new);

/* Fused activations applied outside the rasterizer (igs/models/gaussian_model.py:90-127): opacity = sigmoid(logit),
 * scale = exp(log_scale), rotation = F.normalize(rot) (eps 1e-12); and their backward. */
int igs_activate_fwd(void* stream, int P, const float* logit, const float* log_scale, const float* rot, float* opacity,
                     float* scale, float* rot_n);
int igs_activate_bwd(void* stream, int P, const float* opacity, const float* scale, const float* rot, const float* d_opacity,
                     const float* d_scale, const float* d_rot, float* g_logit, float* g_log_scale, float* g_rot);

#ifdef __cplusplus
}
#endif
#endif
