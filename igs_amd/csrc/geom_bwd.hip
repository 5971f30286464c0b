// geom_bwd.hip -- per-Gaussian backward (one thread per Gaussian), gfx950.
// Fuses what the reference runs as two kernels plus 14 zero-filled temporaries:
//   computeCov2DCUDA (DGR/cuda_rasterizer/backward.cu:145-488), preprocessCUDA backward (:560-628) with the SH
//   backward (:21-140) and computeCov3D backward (:492-555).
// Input is the 128-byte raw-moment accumulator line written by blend_bwd.hip; every output element is written
// (zeros for culled Gaussians), so the caller needs no memsets.
//
// Quirks of the reference reproduced on purpose:
//  * computeCov2DCUDA is handed dL_dconic in its `conic_opacity` parameter (rasterizer_impl.cu:569), so what it calls
//    combined_opacity is dL_dconic.w;
//  * the conic backward adds kernel_size to a and c, the forward conic does not (backward.cu:377-379);
//  * no quaternion-normalisation backward (backward.cu:554).
#include "geom_math.h"

__constant__ float BSH_C0 = 0.28209479177387814f;
__constant__ float BSH_C1 = 0.4886025119029199f;
__constant__ float BSH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                 0.5462742152960396f };
__constant__ float BSH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv) {   // auxiliary.h:123-133
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

// SH backward (backward.cu:21-140): writes dL_dsh[M][3] (all M rows; rows beyond the active degree are zero) and
// returns the contribution to dL_dmean through the view direction.
__device__ __forceinline__ float3 sh_backward(int deg, int M, const float* __restrict__ sh, float3 dir_orig, uint32_t clamped,
                                              float3 dL_dcolor, float* __restrict__ dsh)
{
    const float len = sqrtf(dot3(dir_orig, dir_orig));
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float3 g = make_float3((clamped & 1u) ? 0.f : dL_dcolor.x, (clamped & 2u) ? 0.f : dL_dcolor.y, (clamped & 4u) ? 0.f : dL_dcolor.z);
    auto L = [&](int k) { return make_float3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
    auto W = [&](int k, float b) { dsh[3 * k] = b * g.x; dsh[3 * k + 1] = b * g.y; dsh[3 * k + 2] = b * g.z; };
    float3 dx = make_float3(0, 0, 0), dy = make_float3(0, 0, 0), dz = make_float3(0, 0, 0);
    W(0, BSH_C0);
    if (deg > 0) {
        W(1, -BSH_C1 * y); W(2, BSH_C1 * z); W(3, -BSH_C1 * x);
        dx = L(3) * (-BSH_C1); dy = L(1) * (-BSH_C1); dz = L(2) * BSH_C1;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            W(4, BSH_C2[0] * xy); W(5, BSH_C2[1] * yz); W(6, BSH_C2[2] * (2.f * zz - xx - yy)); W(7, BSH_C2[3] * xz);
            W(8, BSH_C2[4] * (xx - yy));
            dx = dx + (L(4) * (BSH_C2[0] * y) + L(6) * (BSH_C2[2] * 2.f * -x) + L(7) * (BSH_C2[3] * z) + L(8) * (BSH_C2[4] * 2.f * x));
            dy = dy + (L(4) * (BSH_C2[0] * x) + L(5) * (BSH_C2[1] * z) + L(6) * (BSH_C2[2] * 2.f * -y) + L(8) * (BSH_C2[4] * 2.f * -y));
            dz = dz + (L(5) * (BSH_C2[1] * y) + L(6) * (BSH_C2[2] * 2.f * 2.f * z) + L(7) * (BSH_C2[3] * x));
            if (deg > 2) {
                W(9, BSH_C3[0] * y * (3.f * xx - yy)); W(10, BSH_C3[1] * xy * z); W(11, BSH_C3[2] * y * (4.f * zz - xx - yy));
                W(12, BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)); W(13, BSH_C3[4] * x * (4.f * zz - xx - yy));
                W(14, BSH_C3[5] * z * (xx - yy)); W(15, BSH_C3[6] * x * (xx - 3.f * yy));
                dx = dx + (L(9) * (BSH_C3[0] * 3.f * 2.f * xy) + L(10) * (BSH_C3[1] * yz) + L(11) * (BSH_C3[2] * -2.f * xy)
                           + L(12) * (BSH_C3[3] * -3.f * 2.f * xz) + L(13) * (BSH_C3[4] * (-3.f * xx + 4.f * zz - yy))
                           + L(14) * (BSH_C3[5] * 2.f * xz) + L(15) * (BSH_C3[6] * 3.f * (xx - yy)));
                dy = dy + (L(9) * (BSH_C3[0] * 3.f * (xx - yy)) + L(10) * (BSH_C3[1] * xz)
                           + L(11) * (BSH_C3[2] * (-3.f * yy + 4.f * zz - xx)) + L(12) * (BSH_C3[3] * -3.f * 2.f * yz)
                           + L(13) * (BSH_C3[4] * -2.f * xy) + L(14) * (BSH_C3[5] * -2.f * yz) + L(15) * (BSH_C3[6] * -3.f * 2.f * xy));
                dz = dz + (L(10) * (BSH_C3[1] * xy) + L(11) * (BSH_C3[2] * 4.f * 2.f * yz) + L(12) * (BSH_C3[3] * 3.f * (2.f * zz - xx - yy))
                           + L(13) * (BSH_C3[4] * 4.f * 2.f * xz) + L(14) * (BSH_C3[5] * (xx - yy)));
            }
        }
    }
    const int used = (deg + 1) * (deg + 1);
    for (int k = used; k < M; k++) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
    const float3 dL_ddir = make_float3(dot3(dx, g), dot3(dy, g), dot3(dz, g));
    return dnormvdv(dir_orig, dL_ddir);
}

// N > 1 exchange: dL/dSH of a step = sum over the views of basis(direction_v) x dL/dcolour_v.  The ranks
// gather the 3-float colour gradients of every view (12 bytes per Gaussian and view) instead of all-reducing the 48-float SH
// gradients, and every rank rebuilds the sum here, views in rank order -- the same bits everywhere.  The basis expressions
// and the order of operations are those of sh_backward above (W(k, b): b * g, then clamp for the clamp variant, then the sum).
struct ShViewCams { float pos[3 * IGS_MAX_EXCHANGE_VIEWS]; };
struct ShAdam { float *param, *exp_avg, *exp_avg_sq; float lr_over_bc1, b1, b2, eps, inv_sqrt_bc2;     // SH spans ([P][M][3]) of the optimiser state
                // optional: the four small groups (xyz 3 | rotation 4 | opacity 1 | scale 3 floats per Gaussian) updated by the same
                // launch from their (already all-reduced) gradients -- one kernel per N > 1 step instead of two
                float *sm_param = nullptr, *sm_exp_avg = nullptr, *sm_exp_avg_sq = nullptr; const float* sm_grad = nullptr;
                size_t sm_off[4] = {0, 0, 0, 0}; float sm_lr_over_bc1[4] = {0, 0, 0, 0}; };
template <bool ADAM>
__global__ void __launch_bounds__(128)
sh_grad_views_kernel(int P, int D, int M, int V, const float* __restrict__ means3D, const ShViewCams cams,
                     const float* __restrict__ gc, float clamp, float* __restrict__ dsh_out, const ShAdam ad)
{
    const float* campos = cams.pos;
    const int idx = blockIdx.x * 128 + threadIdx.x;
    const bool live = idx < P;
    if (!ADAM && !live) return;                       // (the ADAM variant has a workgroup barrier further down)
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    const size_t ci = live ? (size_t)idx : 0;
    const float3 mean = make_float3(means3D[3 * ci], means3D[3 * ci + 1], means3D[3 * ci + 2]);
    for (int v = 0; v < (live ? V : 0); v++) {
        const float* gp = gc + ((size_t)v * P + idx) * 3;
        const float3 g = make_float3(gp[0], gp[1], gp[2]);
        if (g.x == 0.f && g.y == 0.f && g.z == 0.f) continue;          // not seen by this view: exact zeros
        const float3 dir = mean - make_float3(campos[3 * v], campos[3 * v + 1], campos[3 * v + 2]);
        const float len = sqrtf(dot3(dir, dir));
        const float x = dir.x / len, y = dir.y / len, z = dir.z / len;
        float b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = 0.f;
        b[0] = BSH_C0;
        if (D > 0) {
            b[1] = -BSH_C1 * y; b[2] = BSH_C1 * z; b[3] = -BSH_C1 * x;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                b[4] = BSH_C2[0] * xy; b[5] = BSH_C2[1] * yz; b[6] = BSH_C2[2] * (2.f * zz - xx - yy); b[7] = BSH_C2[3] * xz;
                b[8] = BSH_C2[4] * (xx - yy);
                if (D > 2) {
                    b[9] = BSH_C3[0] * y * (3.f * xx - yy); b[10] = BSH_C3[1] * xy * z; b[11] = BSH_C3[2] * y * (4.f * zz - xx - yy);
                    b[12] = BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); b[13] = BSH_C3[4] * x * (4.f * zz - xx - yy);
                    b[14] = BSH_C3[5] * z * (xx - yy); b[15] = BSH_C3[6] * x * (xx - 3.f * yy);
                }
            }
        }
        const int used = (D + 1) * (D + 1);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < used && k < M) {
                float t0 = b[k] * g.x, t1 = b[k] * g.y, t2 = b[k] * g.z;
                if (clamp > 0.f) { t0 = fminf(fmaxf(t0, -clamp), clamp); t1 = fminf(fmaxf(t1, -clamp), clamp); t2 = fminf(fmaxf(t2, -clamp), clamp); }
                acc[3 * k] += t0; acc[3 * k + 1] += t1; acc[3 * k + 2] += t2;
            }
        }
    }
    if constexpr (ADAM) {
        // the Adam update of the workgroup's 128 Gaussians right here (the rebuilt gradient never goes to HBM): rows through LDS
        // (49-float padded), then one coalesced float4 sweep over the contiguous 128 x 3M span of param / exp_avg / exp_avg_sq
        __shared__ float rows[128 * 49];
        const int F = 3 * M;
#pragma unroll
        for (int k = 0; k < 48; k++)
            if (k < F) rows[threadIdx.x * 49 + k] = acc[k];
        __syncthreads();
        const int g0 = blockIdx.x * 128, ng = min(128, P - g0);
        const size_t base = (size_t)g0 * F;
        const int total = ng * F;
        const bool al = ((F & 3) == 0) && ((((uintptr_t)(ad.param + base)) | ((uintptr_t)(ad.exp_avg + base)) | ((uintptr_t)(ad.exp_avg_sq + base))) & 15) == 0;
        const int total4 = al ? total >> 2 : 0;
        float4* P4p = (float4*)(ad.param + base); float4* M4p = (float4*)(ad.exp_avg + base); float4* V4p = (float4*)(ad.exp_avg_sq + base);
        for (int i0 = threadIdx.x; i0 < total4; i0 += 2 * 128) {
            const int i1 = i0 + 128;
            const bool two = i1 < total4;
            float4 Pa = P4p[i0], Ma = ld_moment(M4p + i0), Va = ld_moment(V4p + i0), Pb, Mb, Vb;
            if (two) { Pb = P4p[i1]; Mb = ld_moment(M4p + i1); Vb = ld_moment(V4p + i1); }
            {
                const int f = 4 * i0, g = f / F, k = f - g * F;
                const float* sp = rows + g * 49 + k;
                adam_update(Pa.x, Ma.x, Va.x, sp[0], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pa.y, Ma.y, Va.y, sp[1], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pa.z, Ma.z, Va.z, sp[2], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pa.w, Ma.w, Va.w, sp[3], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                P4p[i0] = Pa; st_moment(M4p + i0, Ma); st_moment(V4p + i0, Va);
            }
            if (two) {
                const int f = 4 * i1, g = f / F, k = f - g * F;
                const float* sp = rows + g * 49 + k;
                adam_update(Pb.x, Mb.x, Vb.x, sp[0], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pb.y, Mb.y, Vb.y, sp[1], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pb.z, Mb.z, Vb.z, sp[2], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                adam_update(Pb.w, Mb.w, Vb.w, sp[3], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                P4p[i1] = Pb; st_moment(M4p + i1, Mb); st_moment(V4p + i1, Vb);
            }
        }
        for (int e = 4 * total4 + threadIdx.x; e < total; e += 128) {
            const int g = e / F, k = e - g * F;
            float pp = ad.param[base + e], mm = ad.exp_avg[base + e], vv = ad.exp_avg_sq[base + e];
            adam_update(pp, mm, vv, rows[g * 49 + k], ad.lr_over_bc1, ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
            ad.param[base + e] = pp; ad.exp_avg[base + e] = mm; ad.exp_avg_sq[base + e] = vv;
        }
        if (ad.sm_param && live) {
            // this Gaussian's 11 small parameters (its position was read above, before it moves)
            constexpr int KS[4] = { 3, 4, 1, 3 };
#pragma unroll
            for (int gk = 0; gk < 4; gk++) {
                const size_t o = ad.sm_off[gk] + (size_t)idx * KS[gk];
#pragma unroll
                for (int c = 0; c < KS[gk]; c++) {
                    float pp = ad.sm_param[o + c], mm = ad.sm_exp_avg[o + c], vv = ad.sm_exp_avg_sq[o + c];
                    adam_update(pp, mm, vv, ad.sm_grad[o + c], ad.sm_lr_over_bc1[gk], ad.b1, ad.b2, ad.eps, ad.inv_sqrt_bc2);
                    ad.sm_param[o + c] = pp; ad.sm_exp_avg[o + c] = mm; ad.sm_exp_avg_sq[o + c] = vv;
                }
            }
        }
        return;
    }
    float* dst = dsh_out + (size_t)idx * M * 3;
    if (M == 16 && (((uintptr_t)dsh_out) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) ((float4*)dst)[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < M) { dst[3 * k] = acc[3 * k]; dst[3 * k + 1] = acc[3 * k + 1]; dst[3 * k + 2] = acc[3 * k + 2]; }
}

hipError_t launch_sh_grad_views(hipStream_t s, int P, int D, int M, int V, const float* means3D, const float* campos_host, const float* gc,
                                float clamp, float* dsh_out)
{
    ShViewCams cams;                                  // camera centres travel in the kernel arguments: no device buffer, no copy
    for (int i = 0; i < 3 * V; i++) cams.pos[i] = campos_host[i];
    hipLaunchKernelGGL(sh_grad_views_kernel<false>, dim3((P + 127) / 128), dim3(128), 0, s, P, D, M, V, means3D, cams, gc, clamp, dsh_out, ShAdam());
    return hipGetLastError();
}

hipError_t launch_sh_adam_views(hipStream_t s, int P, int D, int M, int V, const float* means3D, const float* campos_host, const float* gc,
                                float clamp, float* param_sh, float* exp_avg_sh, float* exp_avg_sq_sh, float lr_over_bc1, float b1, float b2,
                                float eps, float inv_sqrt_bc2, const SmallGroupsAdam* sm)
{
    ShViewCams cams;
    for (int i = 0; i < 3 * V; i++) cams.pos[i] = campos_host[i];
    ShAdam ad; ad.param = param_sh; ad.exp_avg = exp_avg_sh; ad.exp_avg_sq = exp_avg_sq_sh;
    ad.lr_over_bc1 = lr_over_bc1; ad.b1 = b1; ad.b2 = b2; ad.eps = eps; ad.inv_sqrt_bc2 = inv_sqrt_bc2;
    if (sm) {
        ad.sm_param = sm->param; ad.sm_exp_avg = sm->exp_avg; ad.sm_exp_avg_sq = sm->exp_avg_sq; ad.sm_grad = sm->grad;
        for (int k = 0; k < 4; k++) { ad.sm_off[k] = sm->off[k]; ad.sm_lr_over_bc1[k] = sm->lr_over_bc1[k]; }
    }
    hipLaunchKernelGGL(sh_grad_views_kernel<true>, dim3((P + 127) / 128), dim3(128), 0, s, P, D, M, V, means3D, cams, gc, clamp, (float*)nullptr, ad);
    return hipGetLastError();
}

struct GBArgs { GeomBwdArgs a; RefineFuse f; };

#ifndef GEOM_WAVES_PER_EU
#define GEOM_WAVES_PER_EU 2
#endif
#ifdef GEOM_TIMELINE
// debug build only (tools/debug/geom_timeline.py): per workgroup, the 100 MHz clock at the marks of its life
#define GEOM_TL_MARKS 8
#define GEOM_TL_WGS 8192
__device__ unsigned long long g_geom_tl[GEOM_TL_WGS * GEOM_TL_MARKS];
extern "C" int igs_debug_geom_timeline(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_geom_tl), (size_t)n * 8);
}
#define GTL(k) do { if (threadIdx.x == 0 && blockIdx.x < GEOM_TL_WGS) g_geom_tl[blockIdx.x * GEOM_TL_MARKS + (k)] = wall_clock64(); } while (0)
#else
#define GTL(k) do { } while (0)
#endif
// COLOUR_ONLY: the blend backward ran its colour-only instance (compact accumulator rows): every plane / depth / normal moment is an
// exact zero for every Gaussian, so the plane-fit backward and its eigen-decomposition are not even compiled in -- the general instance
// needs 256 VGPRs + 86 spilled at 2 waves per SIMD for a block this one never executes (4 spilled here; 73.0 -> 70.6 us).
template <bool FUSED, int NT, bool COLOUR_ONLY>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(GEOM_WAVES_PER_EU)))
geom_bwd_kernel(const GBArgs args)
{
    const GeomBwdArgs& a = args.a;
    const RefineFuse& fz = args.f;
    if constexpr (FUSED) {
        if ((fz.guard_overflow && *fz.guard_overflow) || (fz.guard_prefilter && *fz.guard_prefilter)) return;   // frame to be redone
        if (blockIdx.x == 0 && threadIdx.x < 64 && fz.loss_out) {
            float v = fz.loss_scale * fz.loss_shards[16 * threadIdx.x];
            if (fz.loss_shards2) v += fz.loss_scale2 * fz.loss_shards2[16 * threadIdx.x];
            if (fz.loss_shards3) v += fz.loss_scale3 * fz.loss_shards3[16 * threadIdx.x];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (threadIdx.x == 0) fz.loss_out[0] = fz.loss_bias + v;
        }
    }
    // a workgroup walks groups of NT Gaussians (grid-stride): with fewer workgroups than groups, the workgroups of a CU drift
    // apart after their first group, so the latency-bound per-Gaussian phase of one overlaps the streaming Adam phase of another
    GTL(0);
    const int ngroups = (a.P + NT - 1) / NT;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int idx = grp * NT + threadIdx.x;
    // dL_dsh rows are staged in LDS ((3M+1)-float padded rows) and written out with coalesced stores at the end
    extern __shared__ __attribute__((aligned(16))) float dsh_lds[];
    const int F = 3 * a.M, FS = F + 1;
    const bool active = idx < a.P;

    float3 o_m2d = make_float3(0, 0, 0), o_color = make_float3(0, 0, 0), o_mean = make_float3(0, 0, 0), o_scale = make_float3(0, 0, 0);
    float o_opacity = 0.f, o_cov[6] = { 0, 0, 0, 0, 0, 0 };
    float4 o_rot = make_float4(0, 0, 0, 0);
    float* dsh = a.M ? dsh_lds + threadIdx.x * FS : nullptr;
    bool sh_written = false;
    bool found_nan = false;          // (!FUSED, a.nan_host: any NaN among the gradients the reference asserts on, __init__.py:156-162)
    __shared__ float small_lds[FUSED ? NT * 12 : 1];      // xyz 3 | rotation 4 | opacity 1 | scale 3 gradients of every thread

    if (active && a.radii[idx] > 0) {
        const float4* R4 = (const float4*)(a.rec + (size_t)idx * REC_F);
        const float4* G4 = (const float4*)(a.gacc + (size_t)idx * ((COLOUR_ONLY || a.gacc_compact) ? GACC_COMPACT_F : GACC_F));
        const float4 r0 = R4[0], r1 = R4[1], r2 = R4[2], r4 = R4[4], r5 = R4[5], r6 = R4[6], r7 = R4[7];
        PlaneCacheEntry pce; pce.have = false;
        if (!COLOUR_ONLY && a.plane_cache) pce = plane_cache_fetch(a.plane_cache + (size_t)idx * PLANE_CACHE_F);      // (kernel-uniform condition)
        float4 g0, g1, g2, g3, g4, g5, g6;
        if (COLOUR_ONLY || a.gacc_compact) {
            // colour-only blend instance: {c0 c1 c2 Q0} {Qx Qy Qxx Qxy} {Qyy Z - -}; every plane / depth / normal moment is zero
            const float4 c0 = G4[0], c1 = G4[1], c2 = G4[2];
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            g0 = make_float4(c0.x, c0.y, c0.z, 0.f); g1 = z; g2 = z; g3 = z;
            g4 = make_float4(0.f, 0.f, c0.w, c1.x); g5 = make_float4(c1.y, c1.z, c1.w, c2.x); g6 = make_float4(c2.y, 0.f, 0.f, 0.f);
        } else {
            g0 = G4[0]; g1 = G4[1]; g2 = G4[2]; g3 = G4[3]; g4 = G4[4]; g5 = G4[5]; g6 = G4[6];
        }
        GTL(1);
        // ---- unpack raw moments ----
        o_color = make_float3(g0.x, g0.y, g0.z);
        const float Sv0 = g0.w, Sv1 = g1.x, Sv2 = g1.y;
        const float Sx0 = g1.z, Sx1 = g1.w, Sx2 = g2.x, Sy0 = g2.y, Sy1 = g2.z, Sy2 = g2.w;
        const float St = g3.x, Stx = g3.y, Sty = g3.z;
        const float3 dL_dnormal = make_float3(g3.w, g4.x, g4.y);
        const float Q0 = g4.z, Qx = g4.w, Qy = g5.x, Qxx = g5.y, Qxy = g5.z, Qyy = g5.w, Z = g6.x;
        const float conx = r0.z, cony = r0.w, conz = r1.x, opac = r1.y;
        const float cpl[6] = { r4.x, r4.y, r4.z, r4.w, r5.x, r5.y };
        const float rplx = r2.z, rply = r2.w;
        // ---- what the reference accumulated with atomics (backward.cu:878-1013) ----
        const float halfW = 0.5f * a.W, halfH = 0.5f * a.H;                                   // ddelx_dx, ddely_dy (backward.cu:792-793)
        o_m2d.x = (-(conx * Qx + cony * Qy) + cpl[0] * Sv0 + cpl[2] * Sv1 + cpl[4] * Sv2 + rplx * St) * halfW;
        o_m2d.y = (-(conz * Qy + cony * Qx) + cpl[1] * Sv0 + cpl[3] * Sv1 + cpl[5] * Sv2 + rply * St) * halfH;
        o_m2d.z = Z;
        const float dLc_x = -0.5f * Qxx, dLc_y = -0.5f * Qxy, dLc_z = -0.5f * Qyy;         // dL_dconic .x .y .w
        const float c0x = Sx0 / a.fx, c0y = Sy0 / a.fy, c1x = Sx1 / a.fx, c1y = Sy1 / a.fy, c2x = Sx2 / a.fx, c2y = Sy2 / a.fy;
        const float drx = Stx / a.fx, dry = Sty / a.fy;
        const float dL_dts = St;
        float dL_dopacity = (opac != 0.f) ? Q0 / opac : 0.f;

        const float3 mean = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        float cov3D[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov3D[k] = a.cov3D_precomp[6 * (size_t)idx + k];
        } else { cov3D[0] = r6.x; cov3D[1] = r6.y; cov3D[2] = r6.z; cov3D[3] = r6.w; cov3D[4] = r7.x; cov3D[5] = r7.y; }
        const uint32_t clamped = __float_as_uint(r7.z);

        // ================= computeCov2DCUDA (backward.cu:145-488) =================
        // With no plane / depth / normal gradient arriving (e.g. a colour-only loss) every term of the plane-fit backward is an
        // exact zero: skip the eigen-decomposition and that whole block.
        const bool need_planes = !COLOUR_ONLY && ((Sv0 != 0.f) | (Sv1 != 0.f) | (Sv2 != 0.f) | (Sx0 != 0.f) | (Sx1 != 0.f) | (Sx2 != 0.f)
                                 | (Sy0 != 0.f) | (Sy1 != 0.f) | (Sy2 != 0.f) | (St != 0.f) | (Stx != 0.f) | (Sty != 0.f)
                                 | (dL_dnormal.x != 0.f) | (dL_dnormal.y != 0.f) | (dL_dnormal.z != 0.f));
        Cov2DCtx c;
        cov2d_ctx(c, mean, cov3D, a.view, a.fx, a.fy, a.tan_fovx, a.tan_fovy, a.kernel_size, need_planes,
                  &pce, a.plane_tag);
        const float3 t = c.t;
        const float u = c.txtz, v = c.tytz, u2 = u * u, v2 = v * v, uv = u * v;
        const float combined_opacity = dLc_z;          // sic: dL_dconic.w, see header
        float dVs[6] = { 0, 0, 0, 0, 0, 0 };            // symmetric sums of dL_dVrk: [00, 01+10, 02+20, 11, 12+21, 22]
        float DN[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };   // dL_dnJ as a math matrix: DN[r][c] = dL_dcnv[r] * rnv[c]
        float plane0 = 0.f, plane1 = 0.f, dL_du = 0.f, dL_dv = 0.f, dL_dl = 0.f, l = 1.f, nl = 1.f;
        if (!c.degenerate) {
            const float vb = dot3(c.uvh_m, c.uvh), vbn = dot3(c.uvh_mn, c.uvh);
            l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
            const float clamp_vb = fmaxf(vb, 0.0000001f), clamp_vbn = fmaxf(vbn, 0.0000001f);
            nl = u2 + v2 + 1;
            const float factor_normal = l / nl;
            const float3 m = make_float3(c.uvh_mn.x / clamp_vbn, c.uvh_mn.y / clamp_vbn, c.uvh_mn.z / clamp_vbn);   // uvh_m_vb
            plane0 = (v2 + 1) * m.x + (-uv) * m.y + (-u) * m.z;
            plane1 = (-uv) * m.x + (u2 + 1) * m.y + (-v) * m.z;
            const float cp0x = (-(v2 + 1) * t.z + plane0 * t.x) / nl, cp0y = (uv * t.z + plane1 * t.x) / nl;
            const float cp1x = (uv * t.z + plane0 * t.y) / nl,        cp1y = (-(u2 + 1) * t.z + plane1 * t.y) / nl;
            const float cp2x = (t.x + plane0 * t.z) / nl,             cp2y = (t.y + plane1 * t.z) / nl;
            const float rpx = plane0 * factor_normal, rpy = plane1 * factor_normal;
            const float3 rnv = make_float3(-plane0 * factor_normal, -plane1 * factor_normal, -1.f);
            // N (rows): (1/z, 0, x/l), (0, 1/z, y/l), (-x/z^2, -y/z^2, z/l)
            const float n00 = 1 / t.z, n02 = t.x / l, n11 = 1 / t.z, n12 = t.y / l;
            const float n20 = -(t.x) / (t.z * t.z), n21 = -(t.y) / (t.z * t.z), n22 = t.z / l;
            const float3 cnv = make_float3(n00 * rnv.x + 0.0f * rnv.y + n02 * rnv.z, 0.0f * rnv.x + n11 * rnv.y + n12 * rnv.z,
                                           n20 * rnv.x + n21 * rnv.y + n22 * rnv.z);
            const float lv = sqrtf(dot3(cnv, cnv));
            const float3 nv = cnv * (1.0f / lv);
            const float3 dLn_lv = make_float3(dL_dnormal.x / lv, dL_dnormal.y / lv, dL_dnormal.z / lv);
            const float3 dL_dcnv = dLn_lv - nv * dot3(nv, dLn_lv);
            // N^T * dL_dcnv
            const float3 dL_drnv = make_float3(n00 * dL_dcnv.x + 0.0f * dL_dcnv.y + n20 * dL_dcnv.z,
                                               0.0f * dL_dcnv.x + n11 * dL_dcnv.y + n21 * dL_dcnv.z,
                                               n02 * dL_dcnv.x + n12 * dL_dcnv.y + n22 * dL_dcnv.z);
            const float cn[3] = { dL_dcnv.x, dL_dcnv.y, dL_dcnv.z }, rn[3] = { rnv.x, rnv.y, rnv.z };
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int q = 0; q < 3; q++) DN[r][q] = cn[r] * rn[q];
            dL_dl = (-plane0 * dL_drnv.x - plane1 * dL_drnv.y + plane0 * drx + plane1 * dry) / nl;
            const float dpx = (t.x * c0x + t.y * c1x + t.z * c2x - l * dL_drnv.x + drx * l) / nl;
            const float dpy = (t.x * c0y + t.y * c1y + t.z * c2y - l * dL_drnv.y + dry * l) / nl;
            const float dL_dnl = (-c0x * cp0x - c0y * cp0y - c1x * cp1x - c1y * cp1y - c2x * cp2x - c2y * cp2y
                                  - dL_drnv.x * rnv.x - dL_drnv.y * rnv.y - drx * rpx - dry * rpy) / nl;
            const float tmp = dpx * plane0 + dpy * plane1;
            const float3 Wu = m3T_vec(c.Rwc, c.uvh);                      // W * uvh = Rwc^T uvh
            // Ni^T * (dpx, dpy, 0):  Ni rows (v2+1,-uv,-u), (-uv,u2+1,-v), (0,0,0)
            const float3 NiT_dpa = make_float3((v2 + 1) * dpx + (-uv) * dpy, (-uv) * dpx + (u2 + 1) * dpy, (-u) * dpx + (-v) * dpy);
            if (c.well) {
                const float3 av = m3_vec(c.Vinv, Wu);
                const float3 inner = Wu * (-tmp) + m3T_vec(c.Rwc, NiT_dpa);
                M3 Vd;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int q = 0; q < 3; q++) Vd.m[r][q] = c.Vinv.m[r][q] / clamp_vb;
                const float3 bv = m3_vec(Vd, inner);
                const float A_[3] = { av.x, av.y, av.z }, B_[3] = { bv.x, bv.y, bv.z };
                dVs[0] = -(A_[0] * B_[0]); dVs[3] = -(A_[1] * B_[1]); dVs[5] = -(A_[2] * B_[2]);
                dVs[1] = -(A_[1] * B_[0]) + -(A_[0] * B_[1]);
                dVs[2] = -(A_[2] * B_[0]) + -(A_[0] * B_[2]);
                dVs[4] = -(A_[2] * B_[1]) + -(A_[1] * B_[2]);
            } else {
                const float dL_dvb = -tmp / clamp_vb;
                const float3 qv = make_float3((v2 + 1) * (dpx / clamp_vb) + (-uv) * (dpy / clamp_vb),
                                              (-uv) * (dpx / clamp_vb) + (u2 + 1) * (dpy / clamp_vb),
                                              (-u) * (dpx / clamp_vb) + (-v) * (dpy / clamp_vb));
                const float3 rv = Wu * dL_dvb + m3T_vec(c.Rwc, qv);       // dVi = Wu rv^T
                // (dVi + dVi^T) emin = Wu (rv.emin) + rv (Wu.emin)
                const float3 dLv = Wu * dot3(rv, c.emin) + rv * dot3(Wu, c.emin);
                const float evj[3] = { c.ev0, c.ev1, c.ev2 };
                const float3 vcj[3] = { c.vc0, c.vc1, c.vc2 };
                float acc[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
                const float em[3] = { c.emin.x, c.emin.y, c.emin.z };
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (j != c.min_id) {
                        const float sc = dot3(vcj[j], dLv) / fminf(c.evmin - evj[j], -0.0000001f);
                        const float vj[3] = { vcj[j].x * sc, vcj[j].y * sc, vcj[j].z * sc };
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int q = 0; q < 3; q++) acc[r][q] += vj[r] * em[q];
                    }
                }
                dVs[0] = acc[0][0]; dVs[3] = acc[1][1]; dVs[5] = acc[2][2];
                dVs[1] = acc[1][0] + acc[0][1]; dVs[2] = acc[2][0] + acc[0][2]; dVs[4] = acc[2][1] + acc[1][2];
            }
            // dL_duvh = 2(-tmp) m + (Cinv/clamp_vb) Ni^T dpa
            M3 Cd;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int q = 0; q < 3; q++) Cd.m[r][q] = c.Cinv.m[r][q] / clamp_vb;
            const float3 dL_duvh = m * (2 * (-tmp)) + m3_vec(Cd, NiT_dpa);
            // DI = dpa m^T
            const float DI01 = dpx * m.y, DI10 = dpy * m.x, DI11 = dpy * m.y, DI02 = dpx * m.z, DI00 = dpx * m.x, DI12 = dpy * m.z;
            dL_du = dL_dnl * 2 * u + dL_duvh.x + (DI10 + DI01) * (-v) + 2 * DI11 * u - DI02
                    + (c0y * t.y + c1x * t.y + c1y * (-2 * t.x)) / nl;
            dL_dv = dL_dnl * 2 * v + dL_duvh.y + (DI10 + DI01) * (-u) + 2 * DI00 * v - DI12
                    + (c0x * (-2 * t.y) + c0y * t.x + c1x * t.x) / nl;
        }
        // backward.cu:367-375 (double literals)
        const float coef = c.coef, det_0 = c.det0, det_1 = c.det1, ks = a.kernel_size;
        const float opacity = (float)((double)combined_opacity / ((double)coef + 1e-6));
        const float dL_dcoef = dL_dopacity * opacity;
        const float dL_dsqrtcoef = (float)((double)dL_dcoef * 0.5 * 1. / ((double)coef + 1e-6));
        const float dL_ddet0 = (float)((double)dL_dsqrtcoef / ((double)det_1 + 1e-6));
        const float dL_ddet1 = (float)((double)(dL_dsqrtcoef * det_0) * (double)(-1.f / ((double)(det_1 * det_1) + 1e-6)));
        const float dcoef_da = dL_ddet0 * c.cov2[2] + dL_ddet1 * (c.cov2[2] + ks);
        const float dcoef_db = (float)((double)dL_ddet0 * (-2. * (double)c.cov2[1]) + (double)dL_ddet1 * (-2. * (double)c.cov2[1]));
        const float dcoef_dc = dL_ddet0 * c.cov2[0] + dL_ddet1 * (c.cov2[0] + ks);
        const float ca = c.cov2[0] + ks, cb = c.cov2[1], cc = c.cov2[2] + ks;
        const float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define T_(c_, r_) c.A[c_][r_]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dLc_x + 2 * cb * cc * dLc_y + (denom - ca * cc) * dLc_z);
            dL_dc = denom2inv * (-ca * ca * dLc_z + 2 * ca * cb * dLc_y + (denom - ca * cc) * dLc_x);
            dL_db = denom2inv * 2 * (cb * cc * dLc_x - (denom + 2 * cb * cb) * dLc_y + ca * cb * dLc_z);
            if (c.coef_zero) {
                dL_dopacity = 0;
            } else {
                dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db;
                dL_dopacity = dL_dopacity * coef;
            }
            o_cov[0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
            o_cov[3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
            o_cov[5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
            o_cov[1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 1) * dL_dc;
            o_cov[2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 2) * dL_dc;
            o_cov[4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db + 2 * T_(1, 1) * T_(1, 2) * dL_dc;
        }
        o_cov[0] += dVs[0]; o_cov[3] += dVs[3]; o_cov[5] += dVs[5];
        o_cov[1] += dVs[1]; o_cov[2] += dVs[2]; o_cov[4] += dVs[4];
        o_opacity = dL_dopacity;
#define V_(c_, r_) c.Sigma.m[c_][r_]
        const float dL_dT00 = 2 * (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_da + (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_db;
        const float dL_dT01 = 2 * (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_da + (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_db;
        const float dL_dT02 = 2 * (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_da + (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_db;
        const float dL_dT10 = 2 * (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_dc + (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_db;
        const float dL_dT11 = 2 * (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_dc + (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_db;
        const float dL_dT12 = 2 * (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_dc + (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_db;
#undef V_
#undef T_
        // glm W[c][r] = Rwc[c][r]
        const float dL_dJ00 = c.Rwc.m[0][0] * dL_dT00 + c.Rwc.m[0][1] * dL_dT01 + c.Rwc.m[0][2] * dL_dT02;
        const float dL_dJ02 = c.Rwc.m[2][0] * dL_dT00 + c.Rwc.m[2][1] * dL_dT01 + c.Rwc.m[2][2] * dL_dT02;
        const float dL_dJ11 = c.Rwc.m[1][0] * dL_dT10 + c.Rwc.m[1][1] * dL_dT11 + c.Rwc.m[1][2] * dL_dT12;
        const float dL_dJ12 = c.Rwc.m[2][0] * dL_dT10 + c.Rwc.m[2][1] * dL_dT11 + c.Rwc.m[2][2] * dL_dT12;
        const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float l3 = l * l * l;
        // glm dL_dnJ[a][b] = DN[b][a]
        const float dL_dtx = c.xgm * (-a.fx * tz2 * dL_dJ02 + dL_du * tz
                                      - DN[2][0] * tz2 + DN[0][2] * (1 / l - t.x * t.x / l3) + DN[1][2] * (-t.x * t.y / l3) + DN[2][2] * (-t.x * t.z / l3)
                                      + (c0x * plane0 + c0y * plane1 + c2x) / nl
                                      + dL_dl * t.x / l);
        const float dL_dty = c.ygm * (-a.fy * tz2 * dL_dJ12 + dL_dv * tz
                                      - DN[2][1] * tz2 + DN[0][2] * (-t.x * t.y / l3) + DN[1][2] * (1 / l - t.y * t.y / l3) + DN[2][2] * (-t.y * t.z / l3)
                                      + (c1x * plane0 + c1y * plane1 + c2y) / nl
                                      + dL_dl * t.y / l);
        const float dL_dtz = -a.fx * tz2 * dL_dJ00 - a.fy * tz2 * dL_dJ11 + (2 * a.fx * t.x) * tz3 * dL_dJ02 + (2 * a.fy * t.y) * tz3 * dL_dJ12
                             - (dL_du * t.x + dL_dv * t.y) * tz2
                             + (DN[0][0] + DN[1][1]) * (-tz2) + DN[2][0] * (2 * t.x * tz3) + DN[2][1] * (2 * t.y * tz3)
                             + (DN[0][2] * t.x + DN[1][2] * t.y) * (-t.z / l3) + DN[2][2] * (1 / l - t.z * t.z / l3)
                             + (c0x * (-(v2 + 1)) + c0y * uv + c1x * uv + c1y * (-(u2 + 1)) + c2x * plane0 + c2y * plane1) / nl
                             + dL_dl * t.z / l;
        // transformVec4x3Transpose (auxiliary.h:105-113)
        o_mean = make_float3(a.view[0] * dL_dtx + a.view[1] * dL_dty + a.view[2] * dL_dtz,
                             a.view[4] * dL_dtx + a.view[5] * dL_dty + a.view[6] * dL_dtz,
                             a.view[8] * dL_dtx + a.view[9] * dL_dty + a.view[10] * dL_dtz);

        // ================= preprocessCUDA backward (backward.cu:560-628) =================
        {
            const float4 m_hom = xform4x4(mean, a.proj);
            const float m_w = 1.0f / (m_hom.w + 0.0000001f);
            const float* pr = a.proj;
            const float mul1 = (pr[0] * mean.x + pr[4] * mean.y + pr[8] * mean.z + pr[12]) * m_w * m_w;
            const float mul2 = (pr[1] * mean.x + pr[5] * mean.y + pr[9] * mean.z + pr[13]) * m_w * m_w;
            const float gx = o_m2d.x, gy = o_m2d.y;
            const float d1x = (pr[0] * m_w - pr[3] * mul1) * gx + (pr[1] * m_w - pr[3] * mul2) * gy;
            const float d1y = (pr[4] * m_w - pr[7] * mul1) * gx + (pr[5] * m_w - pr[7] * mul2) * gy;
            const float d1z = (pr[8] * m_w - pr[11] * mul1) * gx + (pr[9] * m_w - pr[11] * mul2) * gy;
            const float3 mv = xform4x3(mean, a.view);
            const float tt = sqrtf(mv.x * mv.x + mv.y * mv.y + mv.z * mv.z);
            const float3 q = make_float3(Sv0 + mv.x / tt * dL_dts, Sv1 + mv.y / tt * dL_dts, Sv2 + mv.z / tt * dL_dts);
            const float3 d2 = make_float3(a.view[0] * q.x + a.view[1] * q.y + a.view[2] * q.z,
                                          a.view[4] * q.x + a.view[5] * q.y + a.view[6] * q.z,
                                          a.view[8] * q.x + a.view[9] * q.y + a.view[10] * q.z);
            o_mean.x += d1x + d2.x; o_mean.y += d1y + d2.y; o_mean.z += d1z + d2.z;
        }
        GTL(2);
        if (a.shs) {
            const float3 dir_orig = mean - make_float3(a.campos[0], a.campos[1], a.campos[2]);
            const float3 dm = sh_backward(a.D, a.M, a.shs + (size_t)idx * a.M * 3, dir_orig, clamped, o_color, dsh);
            o_mean = o_mean + dm;
            sh_written = true;
            if constexpr (FUSED) {
                if (fz.color_out) {
                    fz.color_out[3 * (size_t)idx] = (clamped & 1u) ? 0.f : o_color.x;
                    fz.color_out[3 * (size_t)idx + 1] = (clamped & 2u) ? 0.f : o_color.y;
                    fz.color_out[3 * (size_t)idx + 2] = (clamped & 4u) ? 0.f : o_color.z;
                }
            }
        }
        if (a.scales) {       // computeCov3D backward (backward.cu:492-555)
            float3 sc = make_float3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
            float4 qr = make_float4(a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2], a.rotations[4 * idx + 3]);
            if constexpr (FUSED) {                   // raw leaves -> activated values, exactly as the forward pass did
                sc = make_float3(act_exp(sc.x), act_exp(sc.y), act_exp(sc.z));
                const float inv = act_inv_norm4(qr.x, qr.y, qr.z, qr.w);
                qr = make_float4(qr.x * inv, qr.y * inv, qr.z * inv, qr.w * inv);
            }
            const float r = qr.x, x = qr.y, y = qr.z, z = qr.w;
            const M3 Rq = quat_rot(qr);
            const float s[3] = { a.scale_modifier * sc.x, a.scale_modifier * sc.y, a.scale_modifier * sc.z };
            float dS[3][3];
            dS[0][0] = o_cov[0]; dS[0][1] = 0.5f * o_cov[1]; dS[0][2] = 0.5f * o_cov[2];
            dS[1][0] = 0.5f * o_cov[1]; dS[1][1] = o_cov[3]; dS[1][2] = 0.5f * o_cov[4];
            dS[2][0] = 0.5f * o_cov[2]; dS[2][1] = 0.5f * o_cov[4]; dS[2][2] = o_cov[5];
            // M = S Rq^T : M[k][j] = s_k Rq[j][k];  dM = (2 M) dSigma
            float dM[3][3];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int b = 0; b < 3; b++)
                    dM[k][b] = (2.0f * (s[k] * Rq.m[0][k])) * dS[0][b] + (2.0f * (s[k] * Rq.m[1][k])) * dS[1][b] + (2.0f * (s[k] * Rq.m[2][k])) * dS[2][b];
            const float ds0 = Rq.m[0][0] * dM[0][0] + Rq.m[1][0] * dM[0][1] + Rq.m[2][0] * dM[0][2];
            const float ds1 = Rq.m[0][1] * dM[1][0] + Rq.m[1][1] * dM[1][1] + Rq.m[2][1] * dM[1][2];
            const float ds2 = Rq.m[0][2] * dM[2][0] + Rq.m[1][2] * dM[2][1] + Rq.m[2][2] * dM[2][2];
            o_scale = make_float3(ds0, ds1, ds2);
            // X[a][b] = glm dL_dMt[a][b] after the column scaling = s_a * dM[a][b]
#define X_(a_, b_) (dM[a_][b_] * s[a_])
            o_rot.x = 2 * z * (X_(0, 1) - X_(1, 0)) + 2 * y * (X_(2, 0) - X_(0, 2)) + 2 * x * (X_(1, 2) - X_(2, 1));
            o_rot.y = 2 * y * (X_(1, 0) + X_(0, 1)) + 2 * z * (X_(2, 0) + X_(0, 2)) + 2 * r * (X_(1, 2) - X_(2, 1)) - 4 * x * (X_(2, 2) + X_(1, 1));
            o_rot.z = 2 * x * (X_(1, 0) + X_(0, 1)) + 2 * r * (X_(2, 0) - X_(0, 2)) + 2 * z * (X_(1, 2) + X_(2, 1)) - 4 * y * (X_(2, 2) + X_(0, 0));
            o_rot.w = 2 * r * (X_(0, 1) - X_(1, 0)) + 2 * x * (X_(2, 0) + X_(0, 2)) + 2 * y * (X_(1, 2) + X_(2, 1)) - 4 * z * (X_(1, 1) + X_(0, 0));
#undef X_
        }
    }
    if constexpr (FUSED) {
        if (active && fz.color_out && !sh_written) {
            fz.color_out[3 * (size_t)idx] = 0.f; fz.color_out[3 * (size_t)idx + 1] = 0.f; fz.color_out[3 * (size_t)idx + 2] = 0.f;
        }
    }
    if (active) {
        if (!FUSED || a.dL_dmean2D) {
            a.dL_dmean2D[3 * idx] = o_m2d.x; a.dL_dmean2D[3 * idx + 1] = o_m2d.y; a.dL_dmean2D[3 * idx + 2] = o_m2d.z;
        }
        if constexpr (!FUSED) {
            if (a.clamp > 0.f) {        // diff_gaussian_rasterization_rade_clamp/__init__.py:156-162 (means2D, colours, cov3D are not clamped)
                const float c = a.clamp;
                auto cl = [c](float x) { return x < -c ? -c : (x > c ? c : x); };      // (a NaN stays a NaN, as through torch.clamp)
                o_mean = make_float3(cl(o_mean.x), cl(o_mean.y), cl(o_mean.z));
                o_opacity = cl(o_opacity);
                o_scale = make_float3(cl(o_scale.x), cl(o_scale.y), cl(o_scale.z));
                o_rot = make_float4(cl(o_rot.x), cl(o_rot.y), cl(o_rot.z), cl(o_rot.w));
                if (dsh && sh_written) for (int k = 0; k < F; k++) dsh[k] = cl(dsh[k]);
            }
            a.dL_dcolor[3 * idx] = o_color.x; a.dL_dcolor[3 * idx + 1] = o_color.y; a.dL_dcolor[3 * idx + 2] = o_color.z;
            a.dL_dopacity[idx] = o_opacity;
            a.dL_dmean3D[3 * idx] = o_mean.x; a.dL_dmean3D[3 * idx + 1] = o_mean.y; a.dL_dmean3D[3 * idx + 2] = o_mean.z;
#pragma unroll
            for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = o_cov[k];
            a.dL_dscale[3 * idx] = o_scale.x; a.dL_dscale[3 * idx + 1] = o_scale.y; a.dL_dscale[3 * idx + 2] = o_scale.z;
            a.dL_drot[4 * idx] = o_rot.x; a.dL_drot[4 * idx + 1] = o_rot.y; a.dL_drot[4 * idx + 2] = o_rot.z; a.dL_drot[4 * idx + 3] = o_rot.w;
            if (a.nan_host) {
                // x != x for every element of means2D, colors, opacity, means3D, scales, rotations (not cov3D: the reference does not look at it)
                const float chk[17] = { o_m2d.x, o_m2d.y, o_m2d.z, o_color.x, o_color.y, o_color.z, o_opacity, o_mean.x, o_mean.y, o_mean.z,
                                        o_scale.x, o_scale.y, o_scale.z, o_rot.x, o_rot.y, o_rot.z, o_rot.w };
#pragma unroll
                for (int k = 0; k < 17; k++) found_nan |= chk[k] != chk[k];
            }
        } else {
            if (fz.clamp > 0.f) {       // diff_gaussian_rasterization_rade_clamp/__init__.py:156-162 (means2D is not clamped)
                const float c = fz.clamp;
                auto cl = [c](float x) { return x < -c ? -c : (x > c ? c : x); };      // (a NaN stays a NaN, as through torch.clamp)
                o_mean = make_float3(cl(o_mean.x), cl(o_mean.y), cl(o_mean.z));
                o_opacity = cl(o_opacity);
                o_scale = make_float3(cl(o_scale.x), cl(o_scale.y), cl(o_scale.z));
                o_rot = make_float4(cl(o_rot.x), cl(o_rot.y), cl(o_rot.z), cl(o_rot.w));
                if (dsh && sh_written) for (int k = 0; k < F; k++) dsh[k] = cl(dsh[k]);
            }
            // ---- activation backward (gaussian_model.py:90-127) for this Gaussian's 11 small parameters ----
            const float g_xyz[3] = { o_mean.x, o_mean.y, o_mean.z };
            const float s_op = act_sigmoid(fz.param[fz.off_opacity + idx]);
            const float g_logit = o_opacity * s_op * (1.0f - s_op);
            float g_ls[3], g_rot[4];
            {
                const float* ls = fz.param + fz.off_scale + 3 * (size_t)idx;
                const float go[3] = { o_scale.x, o_scale.y, o_scale.z };
#pragma unroll
                for (int k = 0; k < 3; k++) g_ls[k] = go[k] * act_exp(ls[k]);
                const float* rq = fz.param + fz.off_rot + 4 * (size_t)idx;
                const float q0 = rq[0], q1 = rq[1], q2 = rq[2], q3 = rq[3];
                const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
                const float inv = 1.0f / nrm;
                const float n0 = q0 * inv, n1 = q1 * inv, n2 = q2 * inv, n3 = q3 * inv;
                const float dt = n0 * o_rot.x + n1 * o_rot.y + n2 * o_rot.z + n3 * o_rot.w;
                g_rot[0] = (o_rot.x - n0 * dt) * inv; g_rot[1] = (o_rot.y - n1 * dt) * inv;
                g_rot[2] = (o_rot.z - n2 * dt) * inv; g_rot[3] = (o_rot.w - n3 * dt) * inv;
            }
            // staged in LDS: the update below walks each parameter group's contiguous span of this workgroup with coalesced
            // accesses (per-thread 4-byte updates at strides of 12 / 16 bytes cost 30 us for 11 of the 59 parameters)
            float* sg = small_lds + threadIdx.x * 12;
            sg[0] = g_xyz[0]; sg[1] = g_xyz[1]; sg[2] = g_xyz[2];
            sg[3] = g_rot[0]; sg[4] = g_rot[1]; sg[5] = g_rot[2]; sg[6] = g_rot[3];
            sg[7] = g_logit; sg[8] = g_ls[0]; sg[9] = g_ls[1]; sg[10] = g_ls[2];
        }
        if (dsh && !sh_written) for (int k = 0; k < F; k++) dsh[k] = 0.f;
    }
    GTL(3);
    if (a.M) {
        __syncthreads();
        const int g0 = grp * NT;
        const int ng = min(NT, a.P - g0);
        bool fast_done = false;
        if constexpr (FUSED) {
            // ---- fast path of the in-place update (every span 16-byte aligned, no gradient output): all loads of a batch are
            //      issued before the first result is needed.  With one load-compute-store round trip per item the kernel ran at
            //      the latency of ~16 dependent HBM round trips per wave instead of at bandwidth.
            const size_t bx = fz.off_xyz + (size_t)g0 * 3, br = fz.off_rot + (size_t)g0 * 4, bo = fz.off_opacity + (size_t)g0,
                         bs = fz.off_scale + (size_t)g0 * 3, bh = fz.off_sh + (size_t)g0 * F;
            const uintptr_t pa = (uintptr_t)fz.param, ma = (uintptr_t)fz.exp_avg, va = (uintptr_t)fz.exp_avg_sq;
            const bool al = !fz.grad_out && ((F & 3) == 0) && (((pa | ma | va) & 15) == 0) && (((bx | br | bo | bs | bh) & 3) == 0)
                            && ((ng & 3) == 0);
            if (al) {
                fast_done = true;
                auto adam4 = [&](float4& P4, float4& M4, float4& V4, float g0_, float g1_, float g2_, float g3_, float lr) {
                    adam_update(P4.x, M4.x, V4.x, g0_, lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.y, M4.y, V4.y, g1_, lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.z, M4.z, V4.z, g2_, lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.w, M4.w, V4.w, g3_, lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                };
                // -- the four small groups: one float4 item per thread and group at most (NT threads, <= 4 NT floats per group)
                const size_t sb[4] = { bx, br, bo, bs };
                const int sk[4] = { 3, 4, 1, 3 }, sfirst[4] = { 0, 3, 7, 8 };
                const float slr[4] = { fz.lr_xyz, fz.lr_rot, fz.lr_opacity, fz.lr_scale };
                float4 SP[4], SM[4], SV[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool ok = (int)threadIdx.x < (ng * sk[q]) / 4;
                    const size_t o = sb[q] / 4 + threadIdx.x;
                    if (ok) { SP[q] = ((const float4*)fz.param)[o]; SM[q] = ld_moment((const float4*)fz.exp_avg + o); SV[q] = ld_moment((const float4*)fz.exp_avg_sq + o); }
                }
                // -- first batch of the SH span goes out before the small groups are computed
                const int total4 = (ng * F) >> 2;
                constexpr int UB = 2;
                float4 HP[UB], HM[UB], HV[UB];
                auto sh_load = [&](int batch) {
#pragma unroll
                    for (int u = 0; u < UB; u++) {
                        const int i = (int)threadIdx.x + (batch * UB + u) * NT;
                        if (i < total4) {
                            const size_t o = bh / 4 + i;
                            HP[u] = ((const float4*)fz.param)[o]; HM[u] = ld_moment((const float4*)fz.exp_avg + o); HV[u] = ld_moment((const float4*)fz.exp_avg_sq + o);
                        }
                    }
                };
                int row_g = (4 * (int)threadIdx.x) / F, row_k = 4 * (int)threadIdx.x - row_g * F;
                const int row_dg = (4 * NT) / F, row_dk = 4 * NT - row_dg * F;
                auto sh_finish = [&](int batch) {
#pragma unroll
                    for (int u = 0; u < UB; u++) {
                        const int i = (int)threadIdx.x + (batch * UB + u) * NT;
                        if (i < total4) {
                            const float* sp = dsh_lds + row_g * FS + row_k;       // float 4 i of the span = Gaussian row_g, coefficient row_k
                            adam4(HP[u], HM[u], HV[u], sp[0], sp[1], sp[2], sp[3], fz.lr_sh);
                            const size_t o = bh / 4 + i;
                            st_param((float4*)fz.param + o, HP[u]); st_moment((float4*)fz.exp_avg + o, HM[u]); st_moment((float4*)fz.exp_avg_sq + o, HV[u]);
                        }
                        row_g += row_dg; row_k += row_dk;                         // (items are visited in increasing i: no division per item)
                        if (row_k >= F) { row_k -= F; row_g++; }
                    }
                };
                const int nbatch = (total4 + UB * NT - 1) / (UB * NT);
                if (nbatch > 0) sh_load(0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if ((int)threadIdx.x < (ng * sk[q]) / 4) {
                        float g[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) { const int e = 4 * (int)threadIdx.x + c, gl = e / sk[q]; g[c] = small_lds[gl * 12 + sfirst[q] + (e - gl * sk[q])]; }
                        adam4(SP[q], SM[q], SV[q], g[0], g[1], g[2], g[3], slr[q]);
                        const size_t o = sb[q] / 4 + threadIdx.x;
                        st_param((float4*)fz.param + o, SP[q]); st_moment((float4*)fz.exp_avg + o, SM[q]); st_moment((float4*)fz.exp_avg_sq + o, SV[q]);
                    }
                }
                GTL(4);
                for (int b = 0; b < nbatch; b++) {
                    sh_finish(b);
                    if (b + 1 < nbatch) sh_load(b + 1);
                }
            }
        }
        if (!fast_done) {
        if constexpr (FUSED) {
            // the four small groups: element e of the workgroup's span of group (off, k) belongs to Gaussian e / k, component e % k
            auto small_group = [&](size_t off, int k, int first, float lr) {
                const size_t base = off + (size_t)g0 * k;
                const int n = ng * k;
                float* P = fz.grad_out ? fz.grad_out + base : fz.param + base;
                float* Mo = fz.exp_avg + base; float* Vo = fz.exp_avg_sq + base;
                const bool al = (((uintptr_t)P | (uintptr_t)Mo | (uintptr_t)Vo) & 15) == 0;
                const int n4 = al ? n >> 2 : 0;
                for (int i = threadIdx.x; i < n4; i += NT) {
                    float g[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int e = 4 * i + q, gl = e / k; g[q] = small_lds[gl * 12 + first + (e - gl * k)]; }
                    if (fz.grad_out) { ((float4*)P)[i] = make_float4(g[0], g[1], g[2], g[3]); continue; }      // multi-GPU: gradient for the all-reduce
                    float4 P4 = ((float4*)P)[i], M4 = ((float4*)Mo)[i], V4 = ((float4*)Vo)[i];
                    adam_update(P4.x, M4.x, V4.x, g[0], lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.y, M4.y, V4.y, g[1], lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.z, M4.z, V4.z, g[2], lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.w, M4.w, V4.w, g[3], lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    ((float4*)P)[i] = P4; ((float4*)Mo)[i] = M4; ((float4*)Vo)[i] = V4;
                }
                for (int e = 4 * n4 + threadIdx.x; e < n; e += NT) {
                    const int gl = e / k;
                    const float g = small_lds[gl * 12 + first + (e - gl * k)];
                    if (fz.grad_out) { P[e] = g; continue; }
                    float p = P[e], m = Mo[e], v = Vo[e];
                    adam_update(p, m, v, g, lr, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    P[e] = p; Mo[e] = m; Vo[e] = v;
                }
            };
            small_group(fz.off_xyz, 3, 0, fz.lr_xyz);
            small_group(fz.off_rot, 4, 3, fz.lr_rot);
            small_group(fz.off_opacity, 1, 7, fz.lr_opacity);
            small_group(fz.off_scale, 3, 8, fz.lr_scale);
        }
        // N > 1 colour exchange: the SH gradient of the step is rebuilt from the gathered colour gradients, this view's is not needed
        const bool skip_sh = FUSED && fz.grad_out && (fz.color_out || fz.colors_extracted);
        if (!skip_sh) {
        const int total = ng * F;
        float* dst = FUSED ? fz.param + fz.off_sh + (size_t)g0 * F : a.dL_dsh + (size_t)g0 * F;
        float* dst_m = FUSED ? fz.exp_avg + fz.off_sh + (size_t)g0 * F : nullptr;
        float* dst_v = FUSED ? fz.exp_avg_sq + fz.off_sh + (size_t)g0 * F : nullptr;
        bool aligned = ((F & 3) == 0) && ((((uintptr_t)dst) & 15) == 0);
        if constexpr (FUSED) aligned = aligned && (((((uintptr_t)dst_m) | ((uintptr_t)dst_v)) & 15) == 0)
                                       && (!fz.grad_out || (((uintptr_t)(fz.grad_out + fz.off_sh + (size_t)g0 * F)) & 15) == 0);
        if (aligned) {
            const int total4 = total >> 2;
            int f = (int)threadIdx.x * 4;
            int g = f / F, k = f - g * F;
            const int dg = (4 * NT) / F, dk = (4 * NT) - dg * F;
            for (int i = threadIdx.x; i < total4; i += NT) {
                const float* sp = dsh_lds + g * FS + k;
                if (FUSED && fz.grad_out) {
                    ((float4*)(fz.grad_out + fz.off_sh + (size_t)g0 * F))[i] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                } else if constexpr (FUSED) {
                    // Adam on the SH coefficients of the workgroup's NT Gaussians: one contiguous, coalesced span of each buffer
                    float4 P4 = ((float4*)dst)[i], M4 = ((float4*)dst_m)[i], V4 = ((float4*)dst_v)[i];
                    adam_update(P4.x, M4.x, V4.x, sp[0], fz.lr_sh, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.y, M4.y, V4.y, sp[1], fz.lr_sh, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.z, M4.z, V4.z, sp[2], fz.lr_sh, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    adam_update(P4.w, M4.w, V4.w, sp[3], fz.lr_sh, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    ((float4*)dst)[i] = P4; ((float4*)dst_m)[i] = M4; ((float4*)dst_v)[i] = V4;
                } else {
                    ((float4*)dst)[i] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                    if (a.nan_host) found_nan |= (sp[0] != sp[0]) | (sp[1] != sp[1]) | (sp[2] != sp[2]) | (sp[3] != sp[3]);
                }
                g += dg; k += dk;
                if (k >= F) { k -= F; g++; }
            }
        } else {
            int f = (int)threadIdx.x;
            int g = f / F, k = f - g * F;
            const int dg = NT / F, dk = NT - dg * F;
            for (int i = threadIdx.x; i < total; i += NT) {
                if (FUSED && fz.grad_out) {
                    fz.grad_out[fz.off_sh + (size_t)g0 * F + i] = dsh_lds[g * FS + k];
                } else if constexpr (FUSED) {
                    float p = dst[i], m = dst_m[i], v = dst_v[i];
                    adam_update(p, m, v, dsh_lds[g * FS + k], fz.lr_sh, fz.b1, fz.b2, fz.eps, fz.inv_sqrt_bc2);
                    dst[i] = p; dst_m[i] = m; dst_v[i] = v;
                } else {
                    dst[i] = dsh_lds[g * FS + k];
                    if (a.nan_host) found_nan |= dst[i] != dst[i];
                }
                g += dg; k += dk;
                if (k >= F) { k -= F; g++; }
            }
        }
        }      // !skip_sh
        }      // !fast_done
    }
    GTL(5);
    if constexpr (!FUSED) {
        // a NaN was written: say so in the caller's verdict word (pinned host memory; the host reads it once an event recorded behind
        // this kernel has completed -- the end of the kernel releases the store system-wide).  No counters: a "last workgroup" scheme
        // (__threadfence() + a done-counter per workgroup) was measured at +37 us for this kernel -- the fence is an L2 write-back on
        // this GPU, and the kernel has just stored 70 MB through that L2 (refine_ops.hip: l1_mean_kernel).
        if (a.nan_host && found_nan) __atomic_store_n(a.nan_host, a.nan_seq, __ATOMIC_RELAXED);
    }
    if (grp + (int)gridDim.x < ngroups) __syncthreads();          // the LDS rows are reused by the next group
    }
}

hipError_t launch_geom_bwd(hipStream_t s, const GeomBwdArgs& a)
{
    GBArgs g; g.a = a; g.f = RefineFuse();
    const size_t lds = a.M ? (size_t)256 * (3 * a.M + 1) * sizeof(float) : 0;
    hipLaunchKernelGGL((geom_bwd_kernel<false, 256, false>), dim3((a.P + 255) / 256), dim3(256), lds, s, g);
    return hipGetLastError();
}
// The fused kernel alternates a latency-bound phase (per-Gaussian math) with a bandwidth-bound one (Adam over the SH span):
// small workgroups put more of them on a CU (LDS: 49 floats per thread), so the two phases of different workgroups overlap.
#ifndef GEOM_ADAM_THREADS
#define GEOM_ADAM_THREADS 64
#endif
#ifndef GEOM_ADAM_BLOCKS
#define GEOM_ADAM_BLOCKS (1 << 30)     // (capping the grid was measured: at 12 waves per CU by LDS every group is resident at once anyway, fewer workgroups only lose occupancy)
#endif
hipError_t launch_geom_bwd_adam(hipStream_t s, const GeomBwdArgs& a, const RefineFuse& f)
{
    GBArgs g; g.a = a; g.f = f;
    constexpr int NT = GEOM_ADAM_THREADS;
    const size_t lds = a.M ? (size_t)NT * (3 * a.M + 1) * sizeof(float) : 0;
    int blocks = (a.P + NT - 1) / NT;
    const int cap = GEOM_ADAM_BLOCKS;
    if (blocks > cap) blocks = cap;
    if (a.gacc_compact) hipLaunchKernelGGL((geom_bwd_kernel<true, NT, true>), dim3(blocks), dim3(NT), lds, s, g);
    else                hipLaunchKernelGGL((geom_bwd_kernel<true, NT, false>), dim3(blocks), dim3(NT), lds, s, g);
    return hipGetLastError();
}

// dL/d(colour) of this view per Gaussian, straight from the blend backward's accumulator rows (moments 0..2), with the semantics of the
// per-Gaussian kernel's `color_out`: channels the SH evaluation clamped at zero (forward.cu:71-73 -> backward.cu:36-40) and
// Gaussians the view does not see give zero.  Lets the N > 1 exchange start one kernel earlier (RefineFuse::color_event).
__global__ void __launch_bounds__(256)
extract_view_colors_kernel(int P, const int* __restrict__ radii, const float* __restrict__ rec, const float* __restrict__ gacc,
                           int stride, int have_sh, float* __restrict__ color_out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    float3 c = make_float3(0.f, 0.f, 0.f);
    if (have_sh && radii[idx] > 0) {
        const float4 g = *(const float4*)(gacc + (size_t)idx * stride);
        const uint32_t clamped = __float_as_uint(rec[(size_t)idx * REC_F + 30]);      // record word 7.z (preprocess.hip)
        c = make_float3((clamped & 1u) ? 0.f : g.x, (clamped & 2u) ? 0.f : g.y, (clamped & 4u) ? 0.f : g.z);
    }
    color_out[3 * (size_t)idx] = c.x; color_out[3 * (size_t)idx + 1] = c.y; color_out[3 * (size_t)idx + 2] = c.z;
}
hipError_t launch_extract_view_colors(hipStream_t s, int P, const int* radii, const float* rec, const float* gacc, int gacc_compact, bool have_sh,
                                      float* color_out)
{
    hipLaunchKernelGGL(extract_view_colors_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, rec, gacc,
                       gacc_compact ? GACC_COMPACT_F : GACC_F, have_sh ? 1 : 0, color_out);
    return hipGetLastError();
}
