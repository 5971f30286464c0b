// loss_ops.hip -- the photometric loss of the refine loop, forward + backward in two launches, gfx950.
//   loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))        (infer_batch.py:300-306, configs: lambda_dssim 0.2)
// SSIM as igs/utils/loss_utils.py:34-63: 11x11 Gaussian window (sigma 1.5) = outer product of a normalised 1-D kernel, grouped
// conv with ZERO padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over every element.  The reference runs 5 convolutions forward and
// their autograd graph backward ("second-largest cost after the blend backward"); here
//   ssim_stats_kernel : per 32x32 tile and channel, a 42x42 halo of x and y in LDS, separable blur of {x, y, xx, yy, xy},
//                       the SSIM value (summed into 64 shards) and its three partial derivatives w.r.t. the blurred
//                       {x, xx, xy} at that pixel -> three maps;
//   ssim_grad_kernel  : the adjoint: the same separable blur of the three maps (the window is symmetric), combined with x and y
//                       into dL/dx, plus the L1 term and its sum.
// With independent variables m1 = blur(x), e2 = blur(xx), e12 = blur(xy) (m2, blur(yy) constant w.r.t. x):
//   A = 2 m1 m2 + C1,  B = 2 (e12 - m1 m2) + C2,  C = m1^2 + m2^2 + C1,  D = (e2 - m1^2) + s2 + C2,   ssim = A B / (C D)
//   d/dm1  = 2 m2 (B - A) / (C D) - 2 m1 A B (D - C) / (C D)^2,   d/de2 = -A B / (C D^2),   d/de12 = 2 A / (C D)
//   dL/dx(p) = sum_q w(q - p) g(q) d/dm1(q) + 2 x(p) sum_q w(q - p) g(q) d/de2(q) + y(p) sum_q w(q - p) g(q) d/de12(q)
#include "common.h"
#include "../../include/igs_rast.h"
#include <math.h>

#define SSIM_R 5                       // window radius
#define SSIM_T 32                      // tile edge (outputs per workgroup: 32 x 32)
#define SSIM_H (SSIM_T + 2 * SSIM_R)   // halo edge (42): 1.72x the tile area (2.6x at 16 x 16)
#define SSIM_HS 44                     // floats per halo row (16-byte aligned rows)
#define SSIM_BS 36                     // floats per row of the horizontally blurred arrays
// LDS banks: in the horizontal pass consecutive lanes take consecutive ROWS and move float4s; with row strides of 44 and 36
// floats (11 and 9 float4s, both odd) 16 consecutive rows start in 16 different 4-bank groups, so the b128 reads and writes
// are conflict-free (one output column group per lane and scalar accesses put 64 lanes on 16 banks: 4-way conflicts, 2.5x slower)
// Both passes are register-blocked: a work item produces 4 adjacent outputs from 14 loaded inputs (3x fewer LDS reads than one
// output per thread); the vertical pass has exactly 32 columns x 8 row groups = 256 work items.

struct SsimWin { float g[2 * SSIM_R + 1]; };

// XCD-aware tile assignment of the image-space loss kernels.  Workgroups are dealt round-robin over the 8 XCDs (b % 8), each with its
// own L2; a tile's halo is the interior of its neighbours, so neighbours should run on the SAME L2: XCD x takes a contiguous BAND of tile
// rows (rows [x * rpb, (x + 1) * rpb) of every plane), row-major inside the band.  Launched as a 1-D grid of 8 * planes * rpb * gx
// workgroups; the few that map past the last row exit at once (whole workgroups, before any barrier).  Speed only, never correctness.
__device__ __forceinline__ bool band_tile(unsigned b, unsigned gx, unsigned gy, unsigned planes, unsigned& col, unsigned& row, unsigned& plane)
{
    const unsigned rpb = (gy + 7u) / 8u;
    const unsigned xcd = b & 7u, k = b >> 3;
    const unsigned per = rpb * gx;
    plane = k / per;
    const unsigned rem = k - plane * per;
    row = xcd * rpb + rem / gx;
    col = rem % gx;
    return plane < planes && row < gy;
}
static inline unsigned band_grid(unsigned gx, unsigned gy, unsigned planes) { return 8u * planes * ((gy + 7u) / 8u) * gx; }

static SsimWin make_window()
{
    // loss_utils.py:21-24: gauss = Tensor([exp(-(x - 5)^2 / (2 * 1.5^2))]) ; gauss / gauss.sum()   (float32 tensor arithmetic)
    SsimWin w; float sum = 0.f;
    for (int i = 0; i <= 2 * SSIM_R; i++) { w.g[i] = (float)exp(-(double)((i - SSIM_R) * (i - SSIM_R)) / (2.0 * 1.5 * 1.5)); sum += w.g[i]; }
    for (int i = 0; i <= 2 * SSIM_R; i++) w.g[i] = w.g[i] / sum;
    return w;
}

// GT: what happens to the two statistics that depend on the ground truth alone, blur(y) and blur(y y) -- loss_utils.py:34-63
// recomputes them in every iteration although the ground truth of a view does not change while a frame is refined (50 iterations,
// 10 views: every view comes back five times):
//   GT_INLINE  compute them here (5 blurs);     GT_FILL   compute them and store them in `ystats` [2][3][H][W] for the next visits;
//   GT_CACHED  read them from `ystats` (3 blurs: x, x x, x y; two fifths of the multiply-adds of both passes and 12 KB of LDS less,
//              4 workgroups per CU instead of 3).  The stored values are the ones GT_INLINE computes (same code, same order).
enum { GT_INLINE = 0, GT_FILL = 1, GT_CACHED = 2 };
#ifdef SSIM_WPE
#define SSIM_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SSIM_WPE)))
#else
#define SSIM_WPE_ATTR
#endif
template <int GT>
__global__ void __launch_bounds__(256) SSIM_WPE_ATTR
ssim_stats_kernel(const SsimWin win, int W, int H, const float* __restrict__ x, const float* __restrict__ y,
                  float* __restrict__ maps, float* __restrict__ ssim_sum, float* __restrict__ ystats, float* __restrict__ zero1k)
{
    // (workgroup 0, on the side: 1024 floats somebody behind this kernel wants zeroed -- the depth-normal regulariser's loss shards in
    //  igs_refine_step: one launch less per step)
    if (blockIdx.x == 0 && zero1k) ((float4*)zero1k)[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NST = GT == GT_CACHED ? 3 : 5;          // blurred planes: x, (y), x x, (y y), x y
    // LDS: the halo of x and y, and the horizontally blurred planes.  IN PLACE (round 4): blur(x) of a halo row goes back into the row it
    // came from (columns 0..31 of sx[r]), blur(y) -- or, with the ground-truth statistics cached, blur(x y) -- into sy[r]; only the
    // remaining planes have arrays of their own.  Every thread first computes its eight outputs of every plane from the rows in
    // registers, then ONE more barrier, then the stores: 20.8 KB per workgroup instead of 32.9 (cached) / 32.9 instead of 45 --
    // the kernel is bound by the latency of its phases at 3-4 workgroups per CU, not by arithmetic, and LDS was what
    // capped the workgroups per CU.
    constexpr int NEXTRA = NST - 2;                       // planes with storage of their own: x x (cached) | x x, y y, x y
    __shared__ __attribute__((aligned(16))) float sx[SSIM_H][SSIM_HS], sy[SSIM_H][SSIM_HS];
    __shared__ __attribute__((aligned(16))) float hbx[NEXTRA][SSIM_H][SSIM_BS];
    unsigned bx, by, bz;
    if (!band_tile(blockIdx.x, (unsigned)(W + SSIM_T - 1) / SSIM_T, (unsigned)(H + SSIM_T - 1) / SSIM_T, 3u, bx, by, bz)) return;
    const int tid = threadIdx.x, c = (int)bz;
    const int x0 = (int)bx * SSIM_T - SSIM_R, y0 = (int)by * SSIM_T - SSIM_R;
    const size_t HW = (size_t)W * H;
    const float* xc = x + c * HW; const float* yc = y + c * HW;
    {
        // halo load: every global load of the tile is issued before the first one is waited for (7 independent round trips
        // instead of 7 dependent ones -- the kernel is otherwise bound by exactly that latency at 3 workgroups per CU)
        constexpr int NIT = (SSIM_H * SSIM_H + 255) / 256;
        float tx[NIT], ty[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * 256;
            const int r = i / SSIM_H, q = i - r * SSIM_H;
            const int gx = x0 + q, gy = y0 + r;
            const bool in = i < SSIM_H * SSIM_H && gx >= 0 && gx < W && gy >= 0 && gy < H;      // zero padding (F.conv2d padding=5)
            const size_t o = in ? (size_t)gy * W + gx : 0;
            const float a = xc[o], b = yc[o];
            tx[it] = in ? a : 0.f; ty[it] = in ? b : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * 256;
            if (i < SSIM_H * SSIM_H) { const int r = i / SSIM_H, q = i - r * SSIM_H; sx[r][q] = tx[it]; sy[r][q] = ty[it]; }
        }
    }
    __syncthreads();
    // horizontal: work item = (halo row r, group of 8 output columns): 42 x 4 = 168 items, one pass of the 256 threads;
    // consecutive lanes = consecutive rows
    const bool hitem = tid < SSIM_H * (SSIM_T / 8);
    const int hqg = tid / SSIM_H, hr = hitem ? tid - hqg * SSIM_H : 0, hq = hitem ? hqg * 8 : 0;
    float o0[8], o1[8], o2[8], o3[8], o4[8];              // blur of x, y, x x, y y, x y at the item's eight columns
    if (hitem) {
        float u[20], v[20], uu[18], vv[18], uv[18];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float4 a = *(const float4*)&sx[hr][hq + 4 * k], b = *(const float4*)&sy[hr][hq + 4 * k];
            u[4 * k] = a.x; u[4 * k + 1] = a.y; u[4 * k + 2] = a.z; u[4 * k + 3] = a.w;
            v[4 * k] = b.x; v[4 * k + 1] = b.y; v[4 * k + 2] = b.z; v[4 * k + 3] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 18; k++) { uu[k] = u[k] * u[k]; vv[k] = v[k] * v[k]; uv[k] = u[k] * v[k]; }
#pragma unroll
        for (int o = 0; o < 8; o++) {
            o0[o] = 0.f; o1[o] = 0.f; o2[o] = 0.f; o3[o] = 0.f; o4[o] = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * SSIM_R; k++) {
                const float w = win.g[k]; const int j = o + k;
                o0[o] += w * u[j]; o2[o] += w * uu[j]; o4[o] += w * uv[j];
                if constexpr (GT != GT_CACHED) { o1[o] += w * v[j]; o3[o] += w * vv[j]; }
            }
        }
    }
    __syncthreads();                                      // every thread has its rows in registers: the halo rows may be overwritten
    // plane -> storage: blur(x) in sx; cached: blur(x y) in sy, blur(x x) in hbx[0]; otherwise blur(y) in sy, x x / y y / x y in hbx[0..2]
    if (hitem) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            *(float4*)&sx[hr][hq + 4 * h] = make_float4(o0[4 * h], o0[4 * h + 1], o0[4 * h + 2], o0[4 * h + 3]);
            *(float4*)&hbx[0][hr][hq + 4 * h] = make_float4(o2[4 * h], o2[4 * h + 1], o2[4 * h + 2], o2[4 * h + 3]);
            if constexpr (GT == GT_CACHED) {
                *(float4*)&sy[hr][hq + 4 * h] = make_float4(o4[4 * h], o4[4 * h + 1], o4[4 * h + 2], o4[4 * h + 3]);
            } else {
                *(float4*)&sy[hr][hq + 4 * h] = make_float4(o1[4 * h], o1[4 * h + 1], o1[4 * h + 2], o1[4 * h + 3]);
                *(float4*)&hbx[1][hr][hq + 4 * h] = make_float4(o3[4 * h], o3[4 * h + 1], o3[4 * h + 2], o3[4 * h + 3]);
                *(float4*)&hbx[2][hr][hq + 4 * h] = make_float4(o4[4 * h], o4[4 * h + 1], o4[4 * h + 2], o4[4 * h + 3]);
            }
        }
    }
    __syncthreads();
    // vertical: work item = (column lx, group of 4 output rows)
    const int lx = tid & 31, ly = (tid >> 5) * 4;
    const int px = (int)bx * SSIM_T + lx;
    float val = 0.f;
    float m1[4] = { 0, 0, 0, 0 }, m2[4] = { 0, 0, 0, 0 }, e1[4] = { 0, 0, 0, 0 }, e2[4] = { 0, 0, 0, 0 }, e12[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const float h0 = sx[ly + k][lx], h2 = hbx[0][ly + k][lx];
        float h1 = 0.f, h3 = 0.f, h4;
        if constexpr (GT != GT_CACHED) { h1 = sy[ly + k][lx]; h3 = hbx[1][ly + k][lx]; h4 = hbx[2][ly + k][lx]; }
        else h4 = sy[ly + k][lx];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (k - o >= 0 && k - o <= 2 * SSIM_R) {
                const float w = win.g[k - o];
                m1[o] += w * h0; e1[o] += w * h2; e12[o] += w * h4;
                if constexpr (GT != GT_CACHED) { m2[o] += w * h1; e2[o] += w * h3; }
            }
        }
    }
    float* mc = maps + (size_t)c * 3 * HW;
    float* ys = ystats ? ystats + (size_t)c * HW : nullptr;      // blur(y) of channel c; blur(y y) at + 3 HW
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int py = (int)by * SSIM_T + ly + o;
        if (px < W && py < H) {
            if constexpr (GT == GT_CACHED) { m2[o] = ys[(size_t)py * W + px]; e2[o] = ys[3 * HW + (size_t)py * W + px]; }
            if constexpr (GT == GT_FILL) { ys[(size_t)py * W + px] = m2[o]; ys[3 * HW + (size_t)py * W + px] = e2[o]; }
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float m1s = m1[o] * m1[o], m2s = m2[o] * m2[o], m12 = m1[o] * m2[o];
            const float s1 = e1[o] - m1s, s2 = e2[o] - m2s, s12 = e12[o] - m12;
            const float A = 2.f * m12 + C1, B = 2.f * s12 + C2, Cc = m1s + m2s + C1, D = s1 + s2 + C2;
            const float inv_cd = 1.f / (Cc * D);
            val += A * B * inv_cd;
            const size_t off = (size_t)py * W + px;
            mc[off] = 2.f * m2[o] * (B - A) * inv_cd - 2.f * m1[o] * A * B * (D - Cc) * inv_cd * inv_cd;     // d/d blur(x)
            mc[HW + off] = -A * B * inv_cd / D;                                                              // d/d blur(xx)
            mc[2 * HW + off] = 2.f * A * inv_cd;                                                             // d/d blur(xy)
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    if ((tid & 63) == 0 && val != 0.f)
        atomicAdd(&ssim_sum[16 * ((blockIdx.x * 4u + (unsigned)(tid >> 6)) & 63u)], val);
}

typedef float SsimGradLds[3][SSIM_H][SSIM_HS];
// `b`: the workgroup's index among the ssim_grad workgroups (XCD-band mapping, band_tile); `sm`: 22 KB of LDS
__device__ __forceinline__ void
ssim_grad_body(SsimGradLds& sm, const unsigned b, const SsimWin& win, int W, int H, const float* __restrict__ x, const float* __restrict__ y,
               const float* __restrict__ maps, float c_ssim, float c_l1, float* __restrict__ grad, float* __restrict__ l1_sum,
               const float* __restrict__ ssim_shards = nullptr, float* __restrict__ ssim_mean_out = nullptr)
{
    // (workgroup 0, on the side: mean SSIM for a caller who wants the VALUE on the device -- igs_ssim_mean_fwd_bwd.  The 64 shards were
    //  finished by ssim_stats_kernel, the launch in front of this one; fixed order of summation)
    if (b == 0u && ssim_mean_out && threadIdx.x < 64) {
        float v = ssim_shards[16 * threadIdx.x];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) ssim_mean_out[0] = v / (3.f * (float)W * (float)H);
    }
    // (in place, as ssim_stats_kernel: the horizontally blurred row replaces columns 0..31 of the halo row it was computed from, behind
    //  one extra barrier: 22 KB of LDS per workgroup instead of 40 -- six workgroups per CU instead of three)
    unsigned bx, by, bz;
    if (!band_tile(b, (unsigned)(W + SSIM_T - 1) / SSIM_T, (unsigned)(H + SSIM_T - 1) / SSIM_T, 3u, bx, by, bz)) return;
    const int tid = threadIdx.x, c = (int)bz;
    const int x0 = (int)bx * SSIM_T - SSIM_R, y0 = (int)by * SSIM_T - SSIM_R;
    const size_t HW = (size_t)W * H;
    const float* mc = maps + (size_t)c * 3 * HW;
    {
        constexpr int NIT = (SSIM_H * SSIM_H + 255) / 256;
        float t0[NIT], t1[NIT], t2[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * 256;
            const int r = i / SSIM_H, q = i - r * SSIM_H;
            const int gx = x0 + q, gy = y0 + r;
            const bool in = i < SSIM_H * SSIM_H && gx >= 0 && gx < W && gy >= 0 && gy < H;      // no SSIM value exists outside the image
            const size_t o = in ? (size_t)gy * W + gx : 0;
            const float a = mc[o], b = mc[HW + o], d = mc[2 * HW + o];
            t0[it] = in ? a : 0.f; t1[it] = in ? b : 0.f; t2[it] = in ? d : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * 256;
            if (i < SSIM_H * SSIM_H) { const int r = i / SSIM_H, q = i - r * SSIM_H; sm[0][r][q] = t0[it]; sm[1][r][q] = t1[it]; sm[2][r][q] = t2[it]; }
        }
    }
    __syncthreads();
    const bool hitem = tid < SSIM_H * (SSIM_T / 8);
    const int hqg = tid / SSIM_H, hr = hitem ? tid - hqg * SSIM_H : 0, hq = hitem ? hqg * 8 : 0;
    float hacc[3][8];
    if (hitem) {
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float u[20];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float4 a = *(const float4*)&sm[m][hr][hq + 4 * k];
                u[4 * k] = a.x; u[4 * k + 1] = a.y; u[4 * k + 2] = a.z; u[4 * k + 3] = a.w;
            }
#pragma unroll
            for (int o = 0; o < 8; o++) {
                hacc[m][o] = 0.f;
#pragma unroll
                for (int k = 0; k <= 2 * SSIM_R; k++) hacc[m][o] += win.g[k] * u[o + k];
            }
        }
    }
    __syncthreads();                                      // all reads of the halo rows are done: overwrite them
    if (hitem) {
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                *(float4*)&sm[m][hr][hq + 4 * h] = make_float4(hacc[m][4 * h], hacc[m][4 * h + 1], hacc[m][4 * h + 2], hacc[m][4 * h + 3]);
    }
    __syncthreads();
    const int lx = tid & 31, ly = (tid >> 5) * 4;
    const int px = (int)bx * SSIM_T + lx;
    float b0[4] = { 0, 0, 0, 0 }, b1[4] = { 0, 0, 0, 0 }, b2[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const float h0 = sm[0][ly + k][lx], h1 = sm[1][ly + k][lx], h2 = sm[2][ly + k][lx];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (k - o >= 0 && k - o <= 2 * SSIM_R) { const float w = win.g[k - o]; b0[o] += w * h0; b1[o] += w * h1; b2[o] += w * h2; }
        }
    }
    float l1 = 0.f;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int py = (int)by * SSIM_T + ly + o;
        if (px < W && py < H) {
            const size_t off = (size_t)c * HW + (size_t)py * W + px;
            const float xv = x[off], yv = y[off], d = xv - yv;
            l1 += fabsf(d);
            const float g_l1 = d > 0.f ? c_l1 : (d < 0.f ? -c_l1 : 0.f);
            grad[off] = c_ssim * (b0[o] + 2.f * xv * b1[o] + yv * b2[o]) + g_l1;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l1 += __shfl_down(l1, off, 64);
    if ((tid & 63) == 0 && l1 != 0.f)
        atomicAdd(&l1_sum[16 * ((b * 4u + (unsigned)(tid >> 6)) & 63u)], l1);
}
__global__ void __launch_bounds__(256)
ssim_grad_kernel(const SsimWin win, int W, int H, const float* __restrict__ x, const float* __restrict__ y,
                 const float* __restrict__ maps, float c_ssim, float c_l1, float* __restrict__ grad, float* __restrict__ l1_sum,
                 const float* __restrict__ ssim_shards, float* __restrict__ ssim_mean_out)
{
    __shared__ __attribute__((aligned(16))) SsimGradLds sm;
    ssim_grad_body(sm, blockIdx.x, win, W, H, x, y, maps, c_ssim, c_l1, grad, l1_sum, ssim_shards, ssim_mean_out);
}

// scratch = { maps [3 channels][3][H][W] | 64 SSIM-sum shards | 64 L1-sum shards } (shards 16 floats apart)
extern "C" size_t igs_ssim_l1_scratch_bytes(int width, int height)
{
    return (size_t)9 * (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0) * 4 + 2 * 4096 + 256;
}
static inline float* scratch_shards(void* scratch, int width, int height)
{
    return (float*)((char*)scratch + (((size_t)9 * width * height * 4 + 255) & ~(size_t)255));
}

// gt_stats (may be NULL): [2][3][H][W] floats of the caller, one buffer per ground-truth image; gt_stats_valid: it holds blur(gt),
// blur(gt gt) of THIS ground truth already (an earlier call with the same buffer and gt_stats_valid = 0 filled it)
extern "C" size_t igs_ssim_gt_stats_bytes(int width, int height)
{
    return (size_t)6 * (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0) * 4;
}

static int ssim_l1_impl(void* stream, int width, int height, const float* pred, const float* gt, float lambda_dssim,
                        float weight, void* scratch, float* grad, float* sums, float* gt_stats, int gt_stats_valid)
{
    if (width <= 0 || height <= 0) return 0;
    if (!pred || !gt || !scratch || !grad) return IGS_RAST_E_INVALID;
    if (launch_ssim_l1((hipStream_t)stream, width, height, pred, gt, lambda_dssim, weight, scratch, grad, true, gt_stats, gt_stats_valid != 0) != hipSuccess)
        return IGS_RAST_E_HIP;
    if (sums) {
        // sums[0..1023] = SSIM shards, sums[1024..2047] = L1 shards (the caller adds up [16*s])
        if (hipMemcpyAsync(sums, scratch_shards(scratch, width, height), 2 * 4096, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return IGS_RAST_E_HIP;
    }
    return 0;
}
extern "C" int igs_ssim_l1_loss_fwd_bwd(void* stream, int width, int height, const float* pred, const float* gt, float lambda_dssim,
                                        float weight, void* scratch, float* grad, float* sums)
{
    return ssim_l1_impl(stream, width, height, pred, gt, lambda_dssim, weight, scratch, grad, sums, nullptr, 0);
}
// mean SSIM(pred, gt) finished ON THE DEVICE (mean_out[0]) and grad = d(mean SSIM)/d pred (weight -1 of the loss 1 - mean SSIM), in the same two launches: what
// igs_amd.losses.ssim needs -- no copy of the shards, no reduction kernel and no division behind them (three small launches per call)
extern "C" int igs_ssim_mean_fwd_bwd(void* stream, int width, int height, const float* pred, const float* gt, void* scratch, float* grad,
                                     float* mean_out)
{
    if (width <= 0 || height <= 0) return IGS_RAST_E_INVALID;
    if (!pred || !gt || !scratch || !grad || !mean_out) return IGS_RAST_E_INVALID;
    if (launch_ssim_l1((hipStream_t)stream, width, height, pred, gt, 1.0f, -1.0f, scratch, grad, true, nullptr, false, nullptr, mean_out) != hipSuccess)
        return IGS_RAST_E_HIP;
    return 0;
}
extern "C" int igs_ssim_l1_loss_fwd_bwd_cached(void* stream, int width, int height, const float* pred, const float* gt, float lambda_dssim,
                                               float weight, void* scratch, float* grad, float* sums, float* gt_stats, int gt_stats_valid)
{
    return ssim_l1_impl(stream, width, height, pred, gt, lambda_dssim, weight, scratch, grad, sums, gt_stats, gt_stats_valid);
}

// ---- RaDe-GS depth-normal consistency regulariser, forward + backward in one launch ------------------------------------------
// submodules/RaDe-GS/utils/graphics_utils.py:97-126 (depths_double_to_points, point_double_to_normal) and train.py:143-160:
//   points_m(p) = depth_m(p) * ray(p),  ray = ((x + 0.5 - W/2) / fx, (y + 0.5 - H/2) / fy, 1)        m = expected, median depth
//   n_m(p) = normalize(cross(points_m(p + row) - points_m(p - row), points_m(p + col) - points_m(p - col)))   interior p, else 0
//   loss = (1 - ratio) * mean_p(1 - rendered_normal . n_0) + ratio * mean_p(1 - rendered_normal . n_1),  ratio = 0.6
// One workgroup per DN_T x DN_T tile (14 x 14): the two depth maps with a 2-pixel halo in LDS, the gradients w.r.t. the two difference
// vectors of every pixel of the tile + 1 ring in LDS, then each pixel GATHERS what its four neighbours owe it:
//   dL/dpoints(q) = ga(q - row) - ga(q + row) + gb(q - col) - gb(q + col),   dL/ddepth(q) = ray(q) . dL/dpoints(q).
struct DnArgs {
    int W, H; float fx, fy, inv_fx, inv_fy;
    const float *depth, *mdepth, *normal;          // [H][W], [H][W], [3][H][W]
    float s0, s1;                                   // weight * lambda * (1 - ratio) / (H W), weight * lambda * ratio / (H W)
    float *g_depth, *g_mdepth, *g_normal;           // outputs
    float* loss_sum;                                // 64 shards, 16 floats apart: sum of s_m * (1 - rn . n_m)
};
// Tile = 14 x 14 pixels (round 4; was 16 x 16): the gradient stage works on the tile + 1 ring = 16 x 16 = exactly the 256 threads of the
// workgroup in ONE pass (18 x 18 = 324 items took two passes of the whole body with 68 threads in the second), the ray components are
// multiplied by 1 / fx, 1 / fy instead of divided, and the normalisation uses v_rsq_f32 (1 ulp) -- the kernel issued ~600 instructions
// per wave for 20 bytes per pixel each way.
#define DN_T 14
#define DN_LDS_FLOATS (2 * (DN_T + 4) * (DN_T + 5) + 3 * (DN_T + 2) * (DN_T + 3) + 2 * (DN_T + 2) * (DN_T + 2) * 6)
// `b`: the workgroup's index among the depth-normal workgroups; `lds`: DN_LDS_FLOATS floats
__device__ __forceinline__ void depth_normal_body(float* __restrict__ lds, const unsigned b, const DnArgs& a)
{
    float (*dep)[DN_T + 4][DN_T + 5] = (float (*)[DN_T + 4][DN_T + 5])lds;
    float (*rn)[DN_T + 2][DN_T + 3] = (float (*)[DN_T + 2][DN_T + 3])(lds + 2 * (DN_T + 4) * (DN_T + 5));
    float (*G)[DN_T + 2][DN_T + 2][6] = (float (*)[DN_T + 2][DN_T + 2][6])(lds + 2 * (DN_T + 4) * (DN_T + 5) + 3 * (DN_T + 2) * (DN_T + 3));
    unsigned bx, by, bz;
    if (!band_tile(b, (unsigned)(a.W + DN_T - 1) / DN_T, (unsigned)(a.H + DN_T - 1) / DN_T, 1u, bx, by, bz)) return;
    const int tid = threadIdx.x;
    const int tx0 = (int)bx * DN_T, ty0 = (int)by * DN_T;
    const size_t HW = (size_t)a.W * a.H;
    for (int i = tid; i < (DN_T + 4) * (DN_T + 4); i += 256) {
        const int r = i / (DN_T + 4), c = i - r * (DN_T + 4);
        const int gx = tx0 - 2 + c, gy = ty0 - 2 + r;
        const bool in = gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
        const size_t o = in ? (size_t)gy * a.W + gx : 0;
        const float d0 = a.depth[o], d1 = a.mdepth[o];
        dep[0][r][c] = in ? d0 : 0.f; dep[1][r][c] = in ? d1 : 0.f;
    }
    const int r = tid >> 4, c = tid & 15;                      // this thread's pixel of the tile + 1 ring: (ty0 - 1 + r, tx0 - 1 + c)
    const int gx = tx0 - 1 + c, gy = ty0 - 1 + r;
    {
        const bool in = gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
        const size_t o = in ? (size_t)gy * a.W + gx : 0;
        const float n0 = a.normal[o], n1 = a.normal[HW + o], n2 = a.normal[2 * HW + o];
        rn[0][r][c] = in ? n0 : 0.f; rn[1][r][c] = in ? n1 : 0.f; rn[2][r][c] = in ? n2 : 0.f;
    }
    __syncthreads();
    float lsum = 0.f;
    {
        const bool interior = gx >= 1 && gx <= a.W - 2 && gy >= 1 && gy <= a.H - 2;
        const bool mine = r >= 1 && r <= DN_T && c >= 1 && c <= DN_T && gx < a.W && gy < a.H;      // p belongs to this tile
        const float cx = 0.5f - a.W * 0.5f, cy = 0.5f - a.H * 0.5f;
        const float rxm = ((float)gx + cx) * a.inv_fx, rym = ((float)gy + cy) * a.inv_fy;
        const float rxl = ((float)(gx - 1) + cx) * a.inv_fx, rxr = ((float)(gx + 1) + cx) * a.inv_fx;
        const float ryu = ((float)(gy - 1) + cy) * a.inv_fy, ryd = ((float)(gy + 1) + cy) * a.inv_fy;
        const float q0 = rn[0][r][c], q1 = rn[1][r][c], q2 = rn[2][r][c];
        float gn0 = 0.f, gn1 = 0.f, gn2 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const float s = m == 0 ? a.s0 : a.s1;
            float* g = G[m][r][c];
            float ga0 = 0, ga1 = 0, ga2 = 0, gb0 = 0, gb1 = 0, gb2 = 0, dotn = 0.f;
            if (interior) {
                const float dd = dep[m][r + 2][c + 1], du = dep[m][r][c + 1], dr = dep[m][r + 1][c + 2], dl = dep[m][r + 1][c];
                // dx = P(row + 1) - P(row - 1), dy = P(col + 1) - P(col - 1)
                const float ax = (dd - du) * rxm, ay = dd * ryd - du * ryu, az = dd - du;
                const float bx_ = dr * rxr - dl * rxl, by_ = (dr - dl) * rym, bz_ = dr - dl;
                const float c0 = ay * bz_ - az * by_, c1 = az * bx_ - ax * bz_, c2 = ax * by_ - ay * bx_;
                const float len2 = c0 * c0 + c1 * c1 + c2 * c2;
                const bool tiny = !(len2 > 1e-24f);                                  // len <= 1e-12: F.normalize divides by its eps
                const float inv = tiny ? 1e12f : __builtin_amdgcn_rsqf(len2);
                const float n0 = c0 * inv, n1 = c1 * inv, n2 = c2 * inv;
                dotn = q0 * n0 + q1 * n1 + q2 * n2;
                if (mine) { gn0 -= s * n0; gn1 -= s * n1; gn2 -= s * n2; }
                // backward of normalize (for len > eps) and of the cross product; dL/dn = -s * rendered_normal
                const float h0 = -s * q0, h1 = -s * q1, h2 = -s * q2;
                const float hn = h0 * n0 + h1 * n1 + h2 * n2;
                const float k = tiny ? 0.f : inv;
                const float e0 = (h0 - n0 * hn) * k, e1 = (h1 - n1 * hn) * k, e2 = (h2 - n2 * hn) * k;
                ga0 = by_ * e2 - bz_ * e1; ga1 = bz_ * e0 - bx_ * e2; ga2 = bx_ * e1 - by_ * e0;        // b x e
                gb0 = e1 * az - e2 * ay; gb1 = e2 * ax - e0 * az; gb2 = e0 * ay - e1 * ax;              // e x a
            }
            g[0] = ga0; g[1] = ga1; g[2] = ga2; g[3] = gb0; g[4] = gb1; g[5] = gb2;
            if (mine) lsum += s * (1.0f - dotn);
        }
        if (mine) {
            const size_t o = (size_t)gy * a.W + gx;
            a.g_normal[o] = gn0; a.g_normal[HW + o] = gn1; a.g_normal[2 * HW + o] = gn2;
        }
    }
    __syncthreads();
    if (tid < DN_T * DN_T) {
        const int ly = tid / DN_T, lx = tid - ly * DN_T;
        const int qx = tx0 + lx, qy = ty0 + ly;
        if (qx < a.W && qy < a.H) {
            const int rr = ly + 1, cc = lx + 1;                  // q in the (DN_T + 2)^2 array
            const float rx = ((float)qx + 0.5f - a.W * 0.5f) * a.inv_fx, ry = ((float)qy + 0.5f - a.H * 0.5f) * a.inv_fy;
            float out[2];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const float* up = G[m][rr - 1][cc]; const float* dn = G[m][rr + 1][cc];
                const float* lf = G[m][rr][cc - 1]; const float* rt = G[m][rr][cc + 1];
                const float p0 = up[0] - dn[0] + lf[3] - rt[3];
                const float p1 = up[1] - dn[1] + lf[4] - rt[4];
                const float p2 = up[2] - dn[2] + lf[5] - rt[5];
                out[m] = rx * p0 + ry * p1 + p2;
            }
            const size_t o = (size_t)qy * a.W + qx;
            a.g_depth[o] = out[0]; a.g_mdepth[o] = out[1];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if ((tid & 63) == 0 && lsum != 0.f) atomicAdd(&a.loss_sum[16 * ((b * 4u + (unsigned)(tid >> 6)) & 63u)], lsum);
}
__global__ void __launch_bounds__(256)
depth_normal_kernel(const DnArgs a)
{
    __shared__ float lds[DN_LDS_FLOATS];
    depth_normal_body(lds, blockIdx.x, a);
}

// ssim_grad AND depth_normal in ONE launch (igs_refine_step with both losses, BASELINE configs[4]): the two kernels are independent (one
// reads the colour image's statistics maps, the other depth / median depth / normal), each leaves half the chip's bandwidth unused on its
// own (3.7 and 2.2 TB/s of compulsory traffic), and two kernels on one stream never overlap.  Workgroups come in groups of 8 (one per XCD,
// so that band_tile keeps its meaning); the groups of the two roles are interleaved in proportion (Bresenham), so both kinds are resident
// on every CU from the first moment to the last.
struct MixArgs { unsigned groups_ssim, groups_dn; };
__global__ void __launch_bounds__(256)
ssim_grad_dn_kernel(const SsimWin win, int W, int H, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ maps,
                    float c_ssim, float c_l1, float* __restrict__ grad, float* __restrict__ l1_sum, const DnArgs dn, const MixArgs mix)
{
    __shared__ __attribute__((aligned(16))) SsimGradLds sm;
    static_assert(sizeof(SsimGradLds) >= DN_LDS_FLOATS * sizeof(float), "the depth-normal stage must fit the SSIM stage's LDS");
    const unsigned g = blockIdx.x >> 3, xcd = blockIdx.x & 7u, total = mix.groups_ssim + mix.groups_dn;
    // groups [0, total): group g is an SSIM group iff floor((g + 1) S / total) > floor(g S / total); `before` SSIM groups precede it
    const unsigned before = (unsigned)(((unsigned long long)g * mix.groups_ssim) / total);
    const bool is_ssim = (unsigned)(((unsigned long long)(g + 1) * mix.groups_ssim) / total) > before;
    if (is_ssim) ssim_grad_body(sm, before * 8u + xcd, win, W, H, x, y, maps, c_ssim, c_l1, grad, l1_sum);
    else depth_normal_body(&sm[0][0][0], (g - before) * 8u + xcd, dn);
}

static DnArgs make_dn_args(int W, int H, float fx, float fy, const float* depth, const float* mdepth, const float* normal, float weight,
                           float depth_ratio, float* g_depth, float* g_mdepth, float* g_normal, float* loss_shards)
{
    DnArgs a;
    a.W = W; a.H = H; a.fx = fx; a.fy = fy; a.inv_fx = 1.0f / fx; a.inv_fy = 1.0f / fy; a.depth = depth; a.mdepth = mdepth; a.normal = normal;
    const float n = (float)W * (float)H;
    a.s0 = weight * (1.f - depth_ratio) / n; a.s1 = weight * depth_ratio / n;
    a.g_depth = g_depth; a.g_mdepth = g_mdepth; a.g_normal = g_normal; a.loss_sum = loss_shards;
    return a;
}

// `dn` (igs_refine_step with both losses): the depth-normal regulariser rides in the same launch as ssim_grad, and its loss shards are
// zeroed by workgroup 0 of ssim_stats
hipError_t launch_ssim_l1(hipStream_t s, int W, int H, const float* pred, const float* gt, float lambda_dssim, float weight,
                          void* scratch, float* grad, bool zero_shards, float* gt_stats, bool gt_stats_valid, const DepthNormalJob* dn,
                          float* ssim_mean_out)
{
    static const SsimWin win = make_window();
    float* maps = (float*)scratch;
    float* shards = scratch_shards(scratch, W, H);
    if (zero_shards) {
        const hipError_t e = zero_fill_async(s, shards, 2 * 4096);
        if (e != hipSuccess) return e;
    }
    const unsigned n_ssim = band_grid((unsigned)(W + SSIM_T - 1) / SSIM_T, (unsigned)(H + SSIM_T - 1) / SSIM_T, 3u);
    const dim3 grid(n_ssim), block(256);
    const float n = 3.f * (float)W * (float)H;
    float* z = dn ? dn->loss_shards : nullptr;
    if (!gt_stats) hipLaunchKernelGGL(ssim_stats_kernel<GT_INLINE>, grid, block, 0, s, win, W, H, pred, gt, maps, shards, (float*)nullptr, z);
    else if (!gt_stats_valid) hipLaunchKernelGGL(ssim_stats_kernel<GT_FILL>, grid, block, 0, s, win, W, H, pred, gt, maps, shards, gt_stats, z);
    else hipLaunchKernelGGL(ssim_stats_kernel<GT_CACHED>, grid, block, 0, s, win, W, H, pred, gt, maps, shards, gt_stats, z);
    if (!dn) {
        hipLaunchKernelGGL(ssim_grad_kernel, grid, block, 0, s, win, W, H, pred, gt, maps, -lambda_dssim * weight / n,
                           (1.f - lambda_dssim) * weight / n, grad, shards + 1024, (const float*)shards, ssim_mean_out);
    } else {
        const DnArgs da = make_dn_args(W, H, dn->fx, dn->fy, dn->depth, dn->mdepth, dn->normal, dn->weight, dn->depth_ratio, dn->g_depth,
                                       dn->g_mdepth, dn->g_normal, dn->loss_shards);
        const unsigned n_dn = band_grid((unsigned)(W + DN_T - 1) / DN_T, (unsigned)(H + DN_T - 1) / DN_T, 1u);
        MixArgs mix; mix.groups_ssim = n_ssim / 8u; mix.groups_dn = n_dn / 8u;
        hipLaunchKernelGGL(ssim_grad_dn_kernel, dim3(n_ssim + n_dn), block, 0, s, win, W, H, pred, gt, maps, -lambda_dssim * weight / n,
                           (1.f - lambda_dssim) * weight / n, grad, shards + 1024, da, mix);
    }
    return hipGetLastError();
}

hipError_t launch_depth_normal(hipStream_t s, int W, int H, float fx, float fy, const float* depth, const float* mdepth, const float* normal,
                               float weight, float depth_ratio, float* g_depth, float* g_mdepth, float* g_normal, float* loss_shards)
{
    const DnArgs a = make_dn_args(W, H, fx, fy, depth, mdepth, normal, weight, depth_ratio, g_depth, g_mdepth, g_normal, loss_shards);
    hipLaunchKernelGGL(depth_normal_kernel, dim3(band_grid((unsigned)(W + DN_T - 1) / DN_T, (unsigned)(H + DN_T - 1) / DN_T, 1u)), dim3(256), 0, s, a);
    return hipGetLastError();
}

extern "C" int igs_depth_normal_loss_fwd_bwd(void* stream, int width, int height, float tan_fovx, float tan_fovy, const float* depth,
                                             const float* mdepth, const float* normal, float weight, float depth_ratio, float* g_depth,
                                             float* g_mdepth, float* g_normal, float* loss_shards)
{
    if (width <= 0 || height <= 0) return 0;
    if (!depth || !mdepth || !normal || !g_depth || !g_mdepth || !g_normal || !loss_shards) return IGS_RAST_E_INVALID;
    if (zero_fill_async((hipStream_t)stream, loss_shards, 4096) != hipSuccess) return IGS_RAST_E_HIP;
    // fx = W / (2 tan(FoVx / 2)) as graphics_utils.py:99-100
    const float fx = width / (2.0f * tan_fovx), fy = height / (2.0f * tan_fovy);
    return launch_depth_normal((hipStream_t)stream, width, height, fx, fy, depth, mdepth, normal, weight, depth_ratio, g_depth, g_mdepth,
                               g_normal, loss_shards) == hipSuccess ? 0 : IGS_RAST_E_HIP;
}
