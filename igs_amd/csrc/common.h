// common.h -- shared declarations of the MI355X (gfx950) rasterizer library.
// Private scratch layouts, kernel launch entry points, small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define TILE 16              // reference config.h: BLOCK_X = BLOCK_Y = 16
#define TILE_PIX 256
#define CHUNK 256            // instances staged per LDS round

// ---------------------------------------------------------------------------------------------
// Packed per-Gaussian record: ONE 128-byte line per Gaussian (the blend kernels gather whole lines).
//   float4 q0 = { xy.x, xy.y, conic.x, conic.y }
//   float4 q1 = { conic.z, opacity*coef, r, g }
//   float4 q2 = { b, ts, ray_plane.x, ray_plane.y }
//   float4 q3 = { view_point.x, .y, .z, normal.x }
//   float4 q4 = { camera_plane[0..3] }
//   float4 q5 = { camera_plane[4], camera_plane[5], normal.y, normal.z }
//   float4 q6 = { cov3D[0..3] }
//   float4 q7 = { cov3D[4], cov3D[5], bits(clamped mask), bits(depth key) }
// (reference GeometryState, rasterizer_impl.h:29-48, holds these as ~13 separate arrays)
// ---------------------------------------------------------------------------------------------
#define REC_F 32
#define REC_BLEND_F 24       // floats staged into LDS by the blend kernels

// Per-Gaussian gradient accumulator written by the blend backward (one 128-byte line, fp32 atomics):
//   0..2  dL_dcolor            3..5  dL_dview_point      6..11 dL_dcamera_plane (d/fx, d/fy folded in)
//   12    dL_dts               13,14 dL_dray_plane       15..17 dL_dnormal
//   18,19 dL_dmean2D.xy (NDC-scaled)  20 dL_dmean2D.z (abs sum)
//   21..23 dL_dconic (x, y, w)  24 dL_dopacity (before coef)
#define GACC_F 32
#define GACC_COMPACT_F 16              // row stride of the colour-only instance's 10 moments: one 64-byte atomic request, two Gaussians per line
enum { GA_COLOR = 0, GA_VP = 3, GA_CP = 6, GA_TS = 12, GA_RP = 13, GA_NRM = 15, GA_M2D = 18, GA_M2DZ = 20,
       GA_CONIC = 21, GA_OPA = 24, GA_USED = 25 };

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define COUNTER_SHARDS 64                  // instance-count shards (counters[16*(1+s)]); counters[1] = prefilter flag
#define COUNTER_SHARD_STRIDE 16            // uint32 per shard = one 64-byte line
#define SORT_ITEMS 8                       // elements per thread per sort sub-tile
#define SORT_TILE (256 * SORT_ITEMS)       // 2048 elements per block sub-tile
#define SORT_MAX_PASSES 5                  // histogram tables kept per sort (32-bit keys: 4 passes)
#define SORT_MAX_BLOCKS 256                // one block per CU

struct GeomLayout {          // sizes in bytes, offsets from a 256-byte aligned base
    size_t rec, tiles, keys_a, keys_b, vals_a, vals_b, hist, blocksum, counters, planes, total;
    __host__ explicit GeomLayout(size_t P) {
        size_t o = 0;
        rec = o;      o += align_up(P * REC_F * 4, 256);
        tiles = o;    o += align_up(P * 4, 256);
        keys_a = o;   o += align_up(P * 4, 256);
        keys_b = o;   o += align_up(P * 4, 256);
        vals_a = o;   o += align_up(P * 4, 256);     // after the depth sort: Gaussian ids in depth order
        vals_b = o;   o += align_up(P * 4, 256);
        hist = o;     o += align_up((size_t)SORT_MAX_PASSES * 256 * SORT_MAX_BLOCKS * 4, 256);
        blocksum = o; o += align_up((P / 256 + 2) * 4, 256);   // per-256 block instance counts / offsets (depth order)
        counters = o; o += align_up((COUNTER_SHARDS + 1) * COUNTER_SHARD_STRIDE * 4, 256);   // [1] prefilter flag, [16*(1+s)] count shard s
        planes = o;   o += align_up(P * 12 * 4, 256);          // PlaneCache (geom_math.h): Sigma^-1 per visible Gaussian, kept on request for the backward
        total = o + 256;
    }
};
struct BinLayout {
    size_t point_list, keys_a, keys_b, vals_b, hist, total;
    __host__ explicit BinLayout(size_t R) {
        size_t o = 0;
        point_list = o; o += align_up(R * 4, 256);   // final sorted Gaussian ids (kept for backward)
        keys_a = o;     o += align_up(R * 4, 256);
        keys_b = o;     o += align_up(R * 4, 256);
        vals_b = o;     o += align_up(R * 4, 256);
        hist = o;       o += align_up((size_t)SORT_MAX_PASSES * 256 * SORT_MAX_BLOCKS * 4, 256);
        total = o + 256;
    }
};
#define TILE_SORT_SMALL 2048                // tiles up to this many instances are sorted with 16 KB of LDS
#define TILE_SORT_BIG 16384                 // ... up to this many with 128 KB; denser tiles fall back to the global radix path
#define TILE_SORT_WAVE 1024                 // ... and tiles up to this many by a single wave without workgroup barriers
// Slab binning: every tile owns a fixed-capacity slab of `slab` instance slots, so an instance can be dropped into its tile
// with ONE atomic (slot = count++) by the kernel that creates it -- no count pass, no scan, no separate scatter.
struct SlabLayout {
    size_t point_list, pairs, total;
    __host__ SlabLayout(size_t T, size_t slab) {
        size_t o = 0;
        point_list = o; o += align_up(T * slab * 4, 256);   // sorted ids, tile t at [t*slab, t*slab+n) (same offset as BinLayout::point_list)
        pairs = o;      o += align_up(T * slab * 8, 256);   // unsorted (depth bits << 32 | id)
        total = o + 256;
    }
};
#define LOAD_CLASSES 64                     // tiles are ordered by load class for the backward blend: heaviest classes are dispatched first
struct ImgLayout {
    size_t ranges, n_contrib, accum_coord, accum_depth, normal_length, tile_count, stats, counters, zero_end, tile_order, total;
    __host__ ImgLayout(size_t HW, size_t T) {
        size_t o = 0;
        ranges = o;        o += align_up(T * 8, 256);
        n_contrib = o;     o += align_up(HW * 8, 256);
        accum_coord = o;   o += align_up(HW * 12, 256);
        accum_depth = o;   o += align_up(HW * 4, 256);
        normal_length = o; o += align_up(HW * 4, 256);
        tile_count = o;    o += align_up(T * 4, 256);      // slab binning: instances per tile (fill cursor)
        stats = o;         o += 256;                       // ... [0] R, [1] largest tile that overflowed its slab (0 = none), [2] prefilter flag
        counters = o;      o += (COUNTER_SHARDS + 1) * COUNTER_SHARD_STRIDE * 4;   // ... instance-count shards; these three are zeroed by one fill
        zero_end = o;
        tile_order = o;    o += align_up((8 * T + 16) * 4, 256);  // tile ids, heaviest load class first, then [T] = "use it" (blend_fwd's workgroup 0);
                                                                  // or one tile id per workgroup of the fused blend kernel (its grid has < 8 T + 8 of them)
        total = o + 256;
    }
};
static inline char* align_ptr(const char* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

struct FwdParams {
    int P, D, M, W, H, gx, gy;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    float scale_modifier, tan_fovx, tan_fovy, fx, fy, kernel_size;
    int prefiltered;
    const float *view, *proj, *campos;      // device pointers (transposed 4x4 matrices, camera centre)
    uint32_t* zero_stats = nullptr;     // slab binning on a buffer whose counters are known clean: workgroup 0 zeroes the four status words here
    int zero_gacc_stride = GACC_F;      // floats per accumulator row to zero (GACC_COMPACT_F when the colour-only backward will run)
    float* zero_gacc; float* zero_loss; float* zero_loss2;     // refine step: backward accumulators / loss shards to zero-fill on the side (NULL = no)
    int raw_activations;                    // refine step: opacities / scales / rotations are the raw optimiser leaves
                                            // (sigmoid / exp / normalize applied here: gaussian_model.py:90-127)
    float* plane_cache = nullptr; uint32_t plane_tag = 0;      // refine step with a plane / depth / normal gradient to come: keep Sigma^-1 (geom_math.h: PlaneCache)
};

// ---- launchers (each returns hipError_t of the launch) ----
hipError_t launch_preprocess_fwd(hipStream_t s, const FwdParams& p, float* rec, uint32_t* tiles, uint32_t* depth_keys,
                                 uint32_t* ident, int* radii, uint32_t* counters, uint32_t* hist0, uint32_t per_block,
                                 uint32_t* tile_count, uint64_t* pairs, uint32_t slab);
void sort_geometry(uint32_t n, uint32_t* nb, uint32_t* per);
hipError_t launch_tile_sort(hipStream_t s, uint32_t T, uint32_t* tile_count, const uint64_t* pairs, uint32_t* point_list,
                            uint32_t* ranges, uint32_t slab, uint32_t* stats, uint32_t* counters, uint32_t P,
                            uint32_t* step_order = nullptr, uint32_t gx = 0, uint32_t gy = 0);
hipError_t launch_compact_lists(hipStream_t s, uint32_t T, const uint32_t* ranges_in, const uint32_t* list_in, uint32_t* ranges_out,
                                uint32_t* list_out, uint32_t out_capacity);
hipError_t launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* view, uint8_t* present);

// stable LSD radix sort of (key,value) pairs on key bits [bit_lo, bit_hi); result ends in *out_keys/*out_vals
// (ping-pong between a and b).  `hist` holds SORT_MAX_PASSES tables of 256*SORT_MAX_BLOCKS uint32, zeroed by the caller
// except table 0, which the producer of keys_a filled with the first pass's per-block digit counts.
hipError_t radix_sort_pairs(hipStream_t s, uint32_t n, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                            uint32_t* hist, int bit_lo, int bit_hi, uint32_t** out_keys, uint32_t** out_vals);
size_t morton_scratch_bytes(int P);
hipError_t launch_morton_order(hipStream_t s, int P, const float* xyz, const float* lohi, int bits, void* scratch, int* perm);
hipError_t launch_count_sorted(hipStream_t s, int P, const uint32_t* order, const uint32_t* tiles, uint32_t* blocksum);
hipError_t launch_scan_blocksums(hipStream_t s, int nblocks, uint32_t* blocksum);
hipError_t launch_emit_instances(hipStream_t s, int P, int gx, int gy, const uint32_t* order, const uint32_t* tiles,
                                 const uint32_t* blocksum, const float* rec, const int* radii, uint32_t* tile_keys,
                                 uint32_t* vals, uint32_t* hist0, uint32_t per_block, uint32_t mask0);
hipError_t launch_tile_ranges(hipStream_t s, uint32_t R, const uint32_t* tile_keys, uint32_t* ranges);

struct BlendFwdArgs {
    int W, H, gx, gy;
    float fx, fy; const float* bg;
    const uint32_t* ranges; const uint32_t* point_list; const float* rec; const float* colors_precomp;
    float *out_color, *out_coord, *out_mcoord, *out_depth, *out_mdepth, *out_alpha, *out_normal;
    uint32_t* n_contrib; float *accum_coord, *accum_depth, *normal_length;
    // slab binning: workgroup 0 forwards {R, overflow, prefilter flag} to host-visible memory (no copy kernels on the stream)
    const uint32_t* stats_src; const uint32_t* flag_src; uint32_t* host_dst; uint32_t host_seq;      // host_dst[3] = host_seq, written last
    int skip_bwd_state = 0;      // igs_refine_step with a colour-only loss: the backward instance that will run reads none of accum_coord /
                                 // accum_depth / normal_length / the median index -- do not write them (24 of 88 bytes per pixel)
    uint32_t* tile_order = nullptr;            // [T] out: tile ids by descending load class, built by workgroup 0 on the side (for the backward)
    // fused step (blend_step.hip) only: step_order[b] = the tile of workgroup b (tile_sort's extra workgroup wrote it: light tiles last);
    // reset_cursors[T] = the slab binning's fill cursors, which tile_sort then leaves for this kernel to zero (one word per tile)
    const uint32_t* step_order = nullptr; uint32_t* reset_cursors = nullptr;
};
hipError_t launch_blend_fwd(hipStream_t s, const BlendFwdArgs& a, bool coord, bool depth);

struct BlendBwdArgs {
    int W, H, gx, gy;
    float fx, fy; const float* bg;
    const uint32_t* ranges; const uint32_t* point_list; const float* rec; const float* colors_precomp;
    const float *alphas, *normalmap, *accum_coord, *accum_depth, *normal_length; const uint32_t* n_contrib;
    const float *dL_dpix, *dL_dcoord, *dL_dmcoord, *dL_ddepth, *dL_dmdepth, *dL_dalpha, *dL_dnormal;
    float* gacc;
    // refine step: L1 loss fused in -- dL_dpix = l1_scale * sign(l1_color - l1_gt), sum |l1_color - l1_gt| -> 64 shards l1_loss[16*s]
    const float *l1_color, *l1_gt; float l1_scale; float* l1_loss;
    // 0: nobody reads dL_dmean2D.z (the absolute screen-space gradient sum the densification statistics use): the colour-only
    // instance then drops that moment (its |.| terms, one LDS row, one atomic lane per row)
    int want_absgrad = 1;
    const uint32_t* tile_order = nullptr;      // [T] tile ids, heaviest first (the forward's blend kernel wrote them); NULL: plain XCD-aware order
};
hipError_t launch_blend_bwd(hipStream_t s, const BlendBwdArgs& a, bool coord, bool depth, bool* compact_layout, int* instance_bits = nullptr);
// forward + colour-only backward (L1 fused in) of every tile in one kernel (blend_step.hip): igs_refine_step with the L1 loss
hipError_t launch_blend_step(hipStream_t s, const BlendFwdArgs& f, const BlendBwdArgs& b, bool coord, bool depth, int* instance_bits = nullptr);

struct GeomBwdArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *scales, *rotations, *cov3D_precomp; const int* radii;
    float scale_modifier, tan_fovx, tan_fovy, fx, fy, kernel_size;
    const float *view, *proj, *campos;
    const float* rec; const float* gacc;
    int gacc_compact;                       // gacc rows hold {colour 3, Q0, Qx, Qy, Qxx, Qxy, Qyy, Z} in slots 0..9 (colour-only blend instance)
    float *dL_dmean2D, *dL_dcolor, *dL_dopacity, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    // NaN report (igs_rast_next_backward_options): a thread that writes a NaN stores nan_seq into *nan_host (pinned host memory, device
    // address); the host looks at the word after an event recorded behind the kernel.  NULL = no report.
    uint32_t* nan_host = nullptr; uint32_t nan_seq = 0;
    float clamp = 0.f;          // > 0 (unfused kernel): dL/d(means3D, sh, opacity, scale, rotation) clamped to +-clamp as they are written (clamp package)
    const float* plane_cache = nullptr; uint32_t plane_tag = 0;      // what the forward of this frame kept (NULL = nothing: run the eigen-solver)
};
hipError_t launch_geom_bwd(hipStream_t s, const GeomBwdArgs& a);

// Single-GPU refine step: the per-Gaussian backward continues through the activations (sigmoid / exp / normalize) and applies
// the Adam update in place, so no gradient array ever reaches HBM.  means3D / shs / scales / rotations of GeomBwdArgs then
// point at the RAW leaves inside `param`; only dL_dmean2D (may be NULL) is still written.
struct RefineFuse {
    float *param, *exp_avg, *exp_avg_sq;                                   // flat optimiser state
    float* grad_out;                                                       // non-NULL: write the flat gradient (same offsets) INSTEAD of applying Adam
    size_t off_xyz, off_rot, off_sh, off_opacity, off_scale;               // group offsets (floats) into the three buffers
    float lr_xyz, lr_rot, lr_sh, lr_opacity, lr_scale;                     // lr / bias_correction1 per group
    float b1, b2, eps, inv_sqrt_bc2;
    float clamp;                                                           // > 0: clamp dL/d(means3D, sh, opacity, scale, rotation) to +-clamp (clamp variant)
    const uint32_t *guard_overflow, *guard_prefilter;                      // nonzero = the frame is invalid: touch nothing
    // loss_out[0] = loss_bias + sum_k loss_scale_k * sum(loss_shards_k)   (64 shards each, 16 floats apart; NULL = absent)
    const float* loss_shards; const float* loss_shards2; const float* loss_shards3; float* loss_out;
    float loss_scale, loss_scale2, loss_scale3, loss_bias;
    int prezeroed;                                                         // the accumulators were zero-filled by the forward
    int blend_done = 0;                                                    // the blend backward already ran inside the forward's tile kernel (blend_step.hip)
    uint32_t plane_tag = 0;                                                // nonzero: the forward of this step kept Sigma^-1 under this tag (PlaneCache)
    // N > 1: the view's colour gradients are final as soon as the blend backward is done -- backward_impl extracts them into color_out
    // right there and records `color_event` on the stream, so that the ranks' all-gather can run underneath the per-Gaussian kernel
    void* color_event = nullptr;
    int colors_extracted = 0;                                              // (set by backward_impl for the per-Gaussian kernel: color_out is already written)
    float* color_out = nullptr;                                            // [P][3] non-NULL: also write dL/d(colour) of this view (clamped channels and
                                                                           // invisible Gaussians zero): what the N > 1 exchange gathers instead of dL/dSH
};
hipError_t launch_geom_bwd_adam(hipStream_t s, const GeomBwdArgs& a, const RefineFuse& f);
hipError_t launch_extract_view_colors(hipStream_t s, int P, const int* radii, const float* rec, const float* gacc, int gacc_compact, bool have_sh, float* color_out);
#define IGS_MAX_EXCHANGE_VIEWS 64
hipError_t launch_sh_grad_views(hipStream_t s, int P, int D, int M, int V, const float* means3D, const float* campos_host, const float* gc,
                                float clamp, float* dsh_out);
// flat optimiser state + flat gradient and the float offsets of {xyz, rotation, opacity, scale} in them (lr already divided by bias_correction1)
struct SmallGroupsAdam { float *param, *exp_avg, *exp_avg_sq; const float* grad; size_t off[4]; float lr_over_bc1[4]; };
hipError_t launch_sh_adam_views(hipStream_t s, int P, int D, int M, int V, const float* means3D, const float* campos_host, const float* gc,
                                float clamp, float* param_sh, float* exp_avg_sh, float* exp_avg_sq_sh, float lr_over_bc1, float b1, float b2,
                                float eps, float inv_sqrt_bc2, const SmallGroupsAdam* sm = nullptr);
hipError_t launch_depth_normal(hipStream_t s, int W, int H, float fx, float fy, const float* depth, const float* mdepth, const float* normal,
                               float weight, float depth_ratio, float* g_depth, float* g_mdepth, float* g_normal, float* loss_shards);
// 0.8 L1 + 0.2 (1 - SSIM)-style loss, forward + backward (loss_ops.hip); scratch: igs_ssim_l1_scratch_bytes
// (optional `dn`: the depth-normal regulariser of the same step, run inside the SSIM gradient's launch: loss_ops.hip)
struct DepthNormalJob { float fx, fy; const float *depth, *mdepth, *normal; float weight, depth_ratio; float *g_depth, *g_mdepth, *g_normal, *loss_shards; };
hipError_t launch_ssim_l1(hipStream_t s, int W, int H, const float* pred, const float* gt, float lambda_dssim, float weight,
                          void* scratch, float* grad, bool zero_shards, float* gt_stats = nullptr, bool gt_stats_valid = false,
                          const DepthNormalJob* dn = nullptr, float* ssim_mean_out = nullptr);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// activations of the refine loop (igs/models/gaussian_model.py:90-127), shared by every kernel that applies them
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_exp(float x) { return expf(x); }
__device__ __forceinline__ float act_inv_norm4(float q0, float q1, float q2, float q3) {          // F.normalize eps
    return 1.0f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
}
// torch.optim.Adam (single-tensor form) on one scalar; lr = lr / bias_correction1
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps, float inv_sqrt_bc2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= lr * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
}
struct M3 { float m[3][3]; };     // standard row-major [row][col]

// The helpers below feed discrete decisions of the per-Gaussian stage (near-plane cull, radius, tile rectangle, depth key):
// every multiply and add is rounded separately, like the oracle's plain C (`#pragma clang fp contract(off)` travels with the
// instructions when the function is inlined; a pragma placed after this header's inclusion would not reach them).

__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
#pragma clang fp contract(off)
    M3 R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return R;
}
__device__ __forceinline__ M3 m3_T(const M3& A) {
    M3 R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R.m[i][j] = A.m[j][i];
    return R;
}
__device__ __forceinline__ float3 m3_vec(const M3& A, float3 v) {
#pragma clang fp contract(off)
    return make_float3(A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
                       A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
                       A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z);
}
__device__ __forceinline__ float3 m3T_vec(const M3& A, float3 v) {   // A^T v
#pragma clang fp contract(off)
    return make_float3(A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
                       A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
                       A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z);
}
__device__ __forceinline__ float dot3(float3 a, float3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ float3 operator*(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }

// the reference's transformPoint4x3 / 4x4 on the TRANSPOSED matrices the callers pass (auxiliary.h:74-93)
__device__ __forceinline__ float3 xform4x3(float3 p, const float* m) {
#pragma clang fp contract(off)
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float* m) {
#pragma clang fp contract(off)
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// tile rectangle of a splat (auxiliary.h:62-72); float->int conversions saturate on the GPU
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
    float r = (float)max_radius;
    x0 = min(gx, max(0, (int)((px - r) / (float)TILE)));
    y0 = min(gy, max(0, (int)((py - r) / (float)TILE)));
    x1 = min(gx, max(0, (int)((px + r + (float)TILE - 1.0f) / (float)TILE)));
    y1 = min(gy, max(0, (int)((py + r + (float)TILE - 1.0f) / (float)TILE)));
}
// XCD-aware tile assignment.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD group; speed only,
// never correctness).  XCD group x renders the tile ROWS x, x+8, x+16, ...: horizontally adjacent tiles (which share most
// of their splats) run on one L2, and every XCD samples the whole image height, so a dense band of the image does not
// land on a single XCD.  The grid is 8 * ceil(gy/8) * gx workgroups; the few that map past the last row exit at once.
__device__ __forceinline__ bool tile_for_block(uint32_t b, uint32_t gx, uint32_t gy, uint32_t& tile) {
    const uint32_t xcd = b & 7u, k = b >> 3;
    const uint32_t row = xcd + 8u * (k / gx), col = k % gx;
    tile = row * gx + col;
    return row < gy;
}
static inline uint32_t tile_grid_blocks(uint32_t gx, uint32_t gy) { return 8u * ((gy + 7u) / 8u) * gx; }
// The fused blend kernel's dispatch order (sort.hip: build_step_order): usable when tile_sort's one-wave-per-tile kernel runs (slabs
// of at most 1024 slots) and an XCD group has at most 64 * STEP_ORDER_CHUNKS workgroups.
#define STEP_ORDER_CHUNKS 16
// slabs up to this size: every tile is sorted by one wave in registers and no second launch follows (sort.hip: launch_tile_sort) -- the
// only configuration in which the tile sort also writes the fused blend kernel's dispatch order, so both places read THIS constant
#ifndef TILE_SORT_WAVE_ALONE
#define TILE_SORT_WAVE_ALONE 1024
#endif
static inline bool step_order_usable(uint32_t gx, uint32_t gy, uint32_t slab) { return slab <= (uint32_t)TILE_SORT_WAVE_ALONE && ((gy + 7u) / 8u) * gx <= 64u * STEP_ORDER_CHUNKS; }

// Longest-first dispatch of the BACKWARD blend.  A tile's blend time is proportional to its instance count, which ranges from 0 to
// several times the mean; workgroups are dispatched in block order, and with ~2.7 generations of resident workgroups a heavy tile
// that starts late runs on alone while the rest of the chip idles.  Workgroup 0 of the forward blend kernel -- which has the final
// ranges in front of it and 60 us of kernel around it -- orders the tiles by load class (counting sort in LDS, build_tile_order) and
// block b of the backward takes order[b].  Measured (same-box A/B, round 2): backward 214.6 -> 194.5 us on the dense diagnostic
// scene (mean 430 instances per tile), no change on the bench scene (mean 95: its idle slots come from the unequal quads INSIDE a
// workgroup, not from a tail).  The forward itself keeps the plain order: heaviest-first separates its compute-heavy tiles from the
// store-heavy near-empty ones (every tile writes 15 floats per pixel) and the stores no longer hide under the blending: 62 -> 83 us.
// (First version: the tile sort filed the tiles with one returning global atomic each -- 5440 atomics on a handful of words
// serialise at ~12 ns each: tile_sort 17 -> 58 us.)  The order gives up the XCD-aware placement (neighbouring tiles on one L2): on
// the bench scene the backward then fetched 190 MB from HBM instead of 117 (rocprofv3 FETCH_SIZE) for no gain, so it is used only
// where it pays -- when the tiles are heavy enough for their blend time to outweigh the fixed per-workgroup work: mean load >=
// LOAD_ORDER_MIN_MEAN instances per tile; order[T] tells the backward.
#define LOAD_ORDER_MIN_MEAN 192u
__device__ __forceinline__ uint32_t load_class(uint32_t n) {            // 0 = empty ... 63 = heaviest; ~13 % of load per class
    if (n == 0u) return 0u;
    const uint32_t c = 1u + (uint32_t)(5.5f * __log2f((float)n));
    return c > (uint32_t)(LOAD_CLASSES - 1) ? (uint32_t)(LOAD_CLASSES - 1) : c;
}
// called by ALL 256 threads of one workgroup; `hist` = 2 * LOAD_CLASSES words of LDS
__device__ __forceinline__ void build_tile_order(const uint32_t* __restrict__ ranges, uint32_t T, uint32_t* __restrict__ order, uint32_t* hist) {
    const uint32_t tid = threadIdx.x;
    if (tid < 2 * LOAD_CLASSES) hist[tid] = 0u;
    __syncthreads();
    uint32_t mine = 0u;
    for (uint32_t t = tid; t < T; t += 256) { const uint32_t n = ranges[2 * t + 1] - ranges[2 * t]; mine += n; atomicAdd(&hist[load_class(n)], 1u); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_down((int)mine, off, 64);
    if ((tid & 63u) == 0u) atomicAdd(&hist[LOAD_CLASSES], mine);             // (slot of class 0's start: rewritten below)
    __syncthreads();
    const uint32_t total_load = hist[LOAD_CLASSES];
    __syncthreads();
    if (tid == 0) order[T] = (total_load >= LOAD_ORDER_MIN_MEAN * T) ? 1u : 0u;
    if (tid < 64) {                                                      // start of every class in descending order (one wave)
        const uint32_t c = hist[LOAD_CLASSES - 1 - tid];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64); if (tid >= (uint32_t)off) incl += v; }
        hist[LOAD_CLASSES + (LOAD_CLASSES - 1 - tid)] = incl - c;
    }
    __syncthreads();
    for (uint32_t t = tid; t < T; t += 256) order[atomicAdd(&hist[LOAD_CLASSES + load_class(ranges[2 * t + 1] - ranges[2 * t])], 1u)] = t;
}

// XCD-aware bijective remap (contiguous bands; kept for comparison): blocks b, b+8, .. share an XCD (round-robin dispatch), give each XCD a contiguous band of tiles
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t q = n / 8, r = n % 8, xcd = b % 8, k = b / 8;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
#endif

// Adam moments are read once and written once per step and there are 189 MB of them per step: streamed past the caches with the
// non-temporal policy (GEOM_NT: 1 = moments, 2 = the parameters' stores too) so that they do not evict what the next kernels re-read
#ifndef GEOM_NT
#define GEOM_NT 1          // (same-box A/B, round 2: 0.2798 -> 0.2738 ms/step; 2 = no further gain)
#endif
#ifdef __HIPCC__
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_moment(const float4* p)
{
#if GEOM_NT >= 1
    const v4f_t v = __builtin_nontemporal_load((const v4f_t*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_moment(float4* p, const float4& x)
{
#if GEOM_NT >= 1
    v4f_t v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, (v4f_t*)p);
#else
    *p = x;
#endif
}
__device__ __forceinline__ void st_param(float4* p, const float4& x)
{
#if GEOM_NT >= 2
    v4f_t v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, (v4f_t*)p);
#else
    *p = x;
#endif
}
#endif

// Zero-fill of a 4-byte-aligned span by a kernel of this library (refine_ops.hip) instead of hipMemsetAsync: the same cost on the
// stream (the runtime's memset is a fill kernel too), and a plain kernel node when the stream is being captured into a hipGraph --
// captured memset nodes misbehaved on this runtime (round 2: stale counters in the replay, a crash at hipStreamEndCapture).
// `extra` / `extra_words`: a second range of at most 256 words zeroed by the same launch.
hipError_t zero_fill_async(hipStream_t s, void* p, size_t bytes, void* extra = nullptr, int extra_words = 0);

