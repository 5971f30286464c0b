// geom_math.h -- per-Gaussian geometry shared by the forward preprocess and the backward kernels (device only).
// Restates the arithmetic of DGR/cuda_rasterizer/forward.cu:77-304 and auxiliary.h:189-401 in standard
// row-major matrix notation (the reference uses glm column-major literals).
#pragma once
#include "common.h"

// Per-Gaussian arithmetic decides discrete things (near-plane cull, splat radius, tile rectangle, the depth sort key):
// keep every multiply and add separately rounded, like the oracle's plain C, so those decisions agree bit for bit far
// more often than with fused multiply-adds.  (One thread per Gaussian: not where the time goes.)
#pragma clang fp contract(off)

__device__ __forceinline__ bool near0(float x) { return fabsf(x) <= 0.0000001f; }

__device__ __forceinline__ float hyp(float a, float b) {   // auxiliary.h:200-214
    float absa = fabsf(a), absb = fabsf(b);
    if (absa > absb) { absb /= absa; absb *= absb; return absa * sqrtf(1.0f + absb); }
    if (near0(absb)) return 0.0f;
    absa /= absb; absa *= absa;
    return absb * sqrtf(1.0f + absa);
}

// 3x3 symmetric eigen-decomposition: Householder tridiagonalisation + implicit-shift QL, absolute 1e-7 thresholds,
// at most 30 sweeps per eigenvalue (auxiliary.h:217-401).  S is symmetric, row-major.  On return val[i] are the
// eigenvalues and vc_i the matching eigenvectors.  Returns false if QL did not converge.
// Everything is written for D = 3 with compile-time indices so that a[][] / d[] / e[] stay in registers.
__device__ __forceinline__ bool eig_sym3(const M3& S, float& val0, float& val1, float& val2, float3& vc0, float3& vc1, float3& vc2) {
    float a00 = S.m[0][0], a01 = S.m[0][1], a02 = S.m[0][2];
    float a10 = S.m[1][0], a11 = S.m[1][1], a12 = S.m[1][2];
    float a20 = S.m[2][0], a21 = S.m[2][1], a22 = S.m[2][2];
    float d0, d1, d2, e0 = 0.f, e1 = 0.f, e2 = 0.f;

    // ---- Householder step for row 2 (the only row with l > 1) ----
    {
        float h = 0.f, scale = fabsf(a20) + fabsf(a21);
        if (near0(scale)) {
            e2 = a21;
        } else {
            a20 /= scale; h += a20 * a20;
            a21 /= scale; h += a21 * a21;
            float f = a21;
            float g = (f >= 0) ? -sqrtf(h) : sqrtf(h);
            e2 = scale * g;
            h -= f * g;
            a21 = f - g;
            f = 0.f;
            // j = 0
            a02 = a20 / h;
            g = a00 * a20;            // k <= j
            g += a10 * a21;           // k = j+1 .. l-1
            e0 = g / h;
            f += e0 * a20;
            // j = 1
            a12 = a21 / h;
            g = a10 * a20;
            g += a11 * a21;
            e1 = g / h;
            f += e1 * a21;
            float hh = f / (h + h);
            // j = 0
            f = a20; g = e0 - hh * f; e0 = g;
            a00 -= (f * e0 + g * a20);
            // j = 1
            f = a21; g = e1 - hh * f; e1 = g;
            a10 -= (f * e0 + g * a20);
            a11 -= (f * e1 + g * a21);
        }
        d2 = h;
    }
    // row 1: l == 1
    e1 = a10;
    d1 = 0.f;
    d0 = 0.f; e0 = 0.f;
    // ---- accumulate the transformation ----
    // i = 0: l = 0, d0 == 0 -> nothing
    d0 = a00; a00 = 1.f;
    // i = 1: l = 1, d1 == 0 -> skip the update
    d1 = a11; a11 = 1.f; a01 = 0.f; a10 = 0.f;
    // i = 2: l = 2
    if (!near0(d2)) {
        // j = 0
        float g = a20 * a00 + a21 * a10;
        a00 -= g * a02; a10 -= g * a12;
        // j = 1
        g = a20 * a01 + a21 * a11;
        a01 -= g * a02; a11 -= g * a12;
    }
    d2 = a22; a22 = 1.f; a02 = 0.f; a20 = 0.f; a12 = 0.f; a21 = 0.f;

    // ---- QL ----
    e0 = e1; e1 = e2; e2 = 0.f;
    // helpers on named registers (columns c and c+1 of a)
#define ROT_COLS(C0, C1)                                                          \
    { float f_;                                                                     \
      f_ = a0##C1; a0##C1 = s * a0##C0 + c * f_; a0##C0 = c * a0##C0 - s * f_;      \
      f_ = a1##C1; a1##C1 = s * a1##C0 + c * f_; a1##C0 = c * a1##C0 - s * f_;      \
      f_ = a2##C1; a2##C1 = s * a2##C0 + c * f_; a2##C0 = c * a2##C0 - s * f_; }
    // l = 0
    for (int iter = 0;;) {
        int m = 0;
        if (!near0(fabsf(e0))) { m = 1; if (!near0(fabsf(e1))) m = 2; }
        if (m == 0) break;
        if (iter++ == 30) return false;
        float g = (d1 - d0) / (2 * e0);
        float r = hyp(g, 1.0f);
        g = (m == 2 ? d2 : d1) - d0 + e0 / (g + (g >= 0 ? fabsf(r) : -fabsf(r)));
        float s = 1.f, c = 1.f, p = 0.f;
        bool brk = false; int ii = m - 1;
        if (m == 2) {      // i = 1
            float f = s * e1, b = c * e1;
            e2 = r = hyp(f, g);
            if (near0(r)) { d2 -= p; e2 = 0.f; brk = true; ii = 1; }
            else {
                s = f / r; c = g / r;
                g = d2 - p;
                r = (d1 - g) * s + 2 * c * b;
                p = s * r; d2 = g + p;
                g = c * r - b;
                ROT_COLS(1, 2)
                ii = 0;
            }
        }
        if (!brk) {        // i = 0
            float f = s * e0, b = c * e0;
            e1 = r = hyp(f, g);
            if (near0(r)) { d1 -= p; if (m == 2) e2 = 0.f; else e1 = 0.f; brk = true; ii = 0; }
            else {
                s = f / r; c = g / r;
                g = d1 - p;
                r = (d0 - g) * s + 2 * c * b;
                p = s * r; d1 = g + p;
                g = c * r - b;
                ROT_COLS(0, 1)
                ii = -1;
            }
        }
        if (near0(r) && ii >= 0) continue;
        d0 -= p; e0 = g; if (m == 2) e2 = 0.f; else e1 = 0.f;
    }
    // l = 1
    for (int iter = 0;;) {
        int m = 1;
        if (!near0(fabsf(e1))) m = 2;
        if (m == 1) break;
        if (iter++ == 30) return false;
        float g = (d2 - d1) / (2 * e1);
        float r = hyp(g, 1.0f);
        g = d2 - d1 + e1 / (g + (g >= 0 ? fabsf(r) : -fabsf(r)));
        float s = 1.f, c = 1.f, p = 0.f;
        // i = 1
        float f = s * e1, b = c * e1;
        e2 = r = hyp(f, g);
        if (near0(r)) { d2 -= p; e2 = 0.f; continue; }
        s = f / r; c = g / r;
        g = d2 - p;
        r = (d1 - g) * s + 2 * c * b;
        p = s * r; d2 = g + p;
        g = c * r - b;
        ROT_COLS(1, 2)
        d1 -= p; e1 = g; e2 = 0.f;
    }
#undef ROT_COLS
    val0 = d0; val1 = d1; val2 = d2;
    vc0 = make_float3(a00, a10, a20);      // eigenvector i = column i of a
    vc1 = make_float3(a01, a11, a21);
    vc2 = make_float3(a02, a12, a22);
    return true;
}

// standard rotation matrix of the (w,x,y,z) quaternion, NOT normalised (forward.cu:279-290)
__device__ __forceinline__ M3 quat_rot(float4 q) {
    float r = q.x, x = q.y, y = q.z, z = q.w;
    M3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// Sigma = R diag(s^2) R^T, packed (xx,xy,xz,yy,yz,zz)  (forward.cu:270-304)
__device__ __forceinline__ void cov3d_from_scale_rot(float3 s, float mod, float4 q, float cov[6]) {
    M3 R = quat_rot(q);
    float sc[3] = { mod * s.x, mod * s.y, mod * s.z };
    float Mm[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mm[i][k] = sc[k] * R.m[i][k];
    cov[0] = Mm[0][0] * Mm[0][0] + Mm[0][1] * Mm[0][1] + Mm[0][2] * Mm[0][2];
    cov[1] = Mm[0][0] * Mm[1][0] + Mm[0][1] * Mm[1][1] + Mm[0][2] * Mm[1][2];
    cov[2] = Mm[0][0] * Mm[2][0] + Mm[0][1] * Mm[2][1] + Mm[0][2] * Mm[2][2];
    cov[3] = Mm[1][0] * Mm[1][0] + Mm[1][1] * Mm[1][1] + Mm[1][2] * Mm[1][2];
    cov[4] = Mm[1][0] * Mm[2][0] + Mm[1][1] * Mm[2][1] + Mm[1][2] * Mm[2][2];
    cov[5] = Mm[2][0] * Mm[2][0] + Mm[2][1] * Mm[2][1] + Mm[2][2] * Mm[2][2];
}

// Everything the EWA projection / RaDe-GS plane fit needs, forward and backward alike.
struct Cov2DCtx {
    float3 t;               // view-space mean with clamped x,y (forward.cu:85-94)
    float txtz, tytz;       // clamped t.x/t.z, t.y/t.z
    float xgm, ygm;         // 0 where the clamp was active (backward.cu:191-192)
    M3 Rwc;                 // world->camera rotation (rows)
    float A[2][3];          // J * Rwc (2x3): cov2D = A Sigma A^T ; equals glm T[c][r]
    M3 Sigma;
    float cov2[3];          // a, b, c
    float det0, det1, coef; // forward.cu:119-124 (coef NOT zeroed here)
    bool coef_zero;
    float ev0, ev1, ev2; float3 vc0, vc1, vc2;   // named scalars: arrays here end up dynamically indexed in scratch
    bool eig_ok; int min_id; bool well;
    float evmin; float3 emin;
    M3 Vinv;                // Sigma^-1 (or e_min e_min^T)
    M3 Cinv;                // Rwc Vinv Rwc^T
    float3 uvh, uvh_m, uvh_mn;
    bool degenerate;        // isnan(uvh_mn.x) || !eig_ok
};

// The plane fit of cov2d_ctx in two parts, so that the backward can take Sigma^-1 from what the forward kept (PlaneCache) instead of
// running the eigen-solver again.
// part 1: eigen-decomposition of Sigma -> eigenvalues / vectors, the smallest pair, Sigma^-1 (or e_min e_min^T)
__device__ __forceinline__ void cov2d_planes_eig(Cov2DCtx& c) {
    // locals (separate allocas), not struct fields: a select between adjacent fields gets turned into a
    // dynamically indexed scratch access by the optimiser
    float l0, l1, l2; float3 w0, w1, w2;
    c.eig_ok = eig_sym3(c.Sigma, l0, l1, l2, w0, w1, w2);
    const int min_id = l0 > l1 ? (l1 > l2 ? 2 : 1) : (l0 > l2 ? 2 : 0);
    const bool is0 = min_id == 0, is1 = min_id == 1;
    c.evmin = is0 ? l0 : (is1 ? l1 : l2);
    c.emin = make_float3(is0 ? w0.x : (is1 ? w1.x : w2.x), is0 ? w0.y : (is1 ? w1.y : w2.y), is0 ? w0.z : (is1 ? w1.z : w2.z));
    c.min_id = min_id;
    c.ev0 = l0; c.ev1 = l1; c.ev2 = l2; c.vc0 = w0; c.vc1 = w1; c.vc2 = w2;
    c.well = (double)c.evmin > 0.00000001;
    if (c.well) {
        const float i0 = 1 / c.ev0, i1 = 1 / c.ev1, i2 = 1 / c.ev2;
        const float v0[3] = { c.vc0.x, c.vc0.y, c.vc0.z }, v1[3] = { c.vc1.x, c.vc1.y, c.vc1.z }, v2[3] = { c.vc2.x, c.vc2.y, c.vc2.z };
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                c.Vinv.m[i][j] = (v0[i] * i0) * v0[j] + (v1[i] * i1) * v1[j] + (v2[i] * i2) * v2[j];
    } else {
        const float em[3] = { c.emin.x, c.emin.y, c.emin.z };
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) c.Vinv.m[i][j] = em[i] * em[j];
    }
}
// part 2: everything downstream of Sigma^-1 (needs c.Vinv, c.eig_ok)
__device__ __forceinline__ void cov2d_planes_finish(Cov2DCtx& c) {
    c.Cinv = m3_mul(m3_mul(c.Rwc, c.Vinv), m3_T(c.Rwc));
    c.uvh = make_float3(c.txtz, c.tytz, 1.f);
    c.uvh_m = m3_vec(c.Cinv, c.uvh);
    const float inv = 1.0f / sqrtf(dot3(c.uvh_m, c.uvh_m));
    c.uvh_mn = c.uvh_m * inv;
    c.degenerate = (c.uvh_mn.x != c.uvh_mn.x) || !c.eig_ok;
}

// What the forward's record role keeps per visible Gaussian when the caller says a plane / depth / normal gradient will come back
// (igs_refine_step with the depth-normal regulariser): Sigma^-1 with all nine entries (the products are not bit-symmetric), whether it
// is the regular inverse, and the tag of the frame.  The per-Gaussian backward then skips the eigen-solver -- 12 us of its 94 on BASELINE
// configs[4] -- unless the entry is not this frame's or the Gaussian took the rank-deficient branch (whose backward needs the eigenvectors).
#define PLANE_CACHE_F 12
__device__ __forceinline__ void plane_cache_store(float* __restrict__ dst, const Cov2DCtx& c, uint32_t tag) {
    float4* d4 = (float4*)dst;
    d4[0] = make_float4(c.Vinv.m[0][0], c.Vinv.m[0][1], c.Vinv.m[0][2], c.Vinv.m[1][0]);
    d4[1] = make_float4(c.Vinv.m[1][1], c.Vinv.m[1][2], c.Vinv.m[2][0], c.Vinv.m[2][1]);
    d4[2] = make_float4(c.Vinv.m[2][2], __uint_as_float((c.well && c.eig_ok) ? 1u : 0u), __uint_as_float(tag), 0.f);
}
// one entry, fetched early by the caller (next to its other loads: the three loads must not sit on the chain behind the moments)
struct PlaneCacheEntry { float4 a, b, d; bool have; };
__device__ __forceinline__ PlaneCacheEntry plane_cache_fetch(const float* __restrict__ src) {
    PlaneCacheEntry e;
    const float4* s4 = (const float4*)src;
    e.a = s4[0]; e.b = s4[1]; e.d = s4[2]; e.have = true;
    return e;
}
// true: c.Vinv / c.well / c.eig_ok are set from the cache (the regular branch); false: run cov2d_planes_eig
__device__ __forceinline__ bool plane_cache_load(const PlaneCacheEntry& e, Cov2DCtx& c, uint32_t tag) {
    if (!e.have || __float_as_uint(e.d.z) != tag || __float_as_uint(e.d.y) != 1u) return false;
    c.Vinv.m[0][0] = e.a.x; c.Vinv.m[0][1] = e.a.y; c.Vinv.m[0][2] = e.a.z; c.Vinv.m[1][0] = e.a.w;
    c.Vinv.m[1][1] = e.b.x; c.Vinv.m[1][2] = e.b.y; c.Vinv.m[2][0] = e.b.z; c.Vinv.m[2][1] = e.b.w;
    c.Vinv.m[2][2] = e.d.x;
    c.well = true; c.eig_ok = true;
    // (the regular branch of the backward reads none of these)
    c.min_id = 0; c.evmin = 1.f; c.emin = make_float3(0, 0, 0); c.ev0 = c.ev1 = c.ev2 = 1.f; c.vc0 = c.vc1 = c.vc2 = make_float3(0, 0, 0);
    return true;
}

__device__ __forceinline__ void cov2d_ctx(Cov2DCtx& c, float3 mean, const float* cov3D, const float* view, float fx,
                                          float fy, float tan_fovx, float tan_fovy, float kernel_size, bool with_planes = true,
                                          const PlaneCacheEntry* plane_cache = nullptr, uint32_t plane_tag = 0u) {
    float3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    c.xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    c.ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    c.txtz = t.x / t.z; c.tytz = t.y / t.z;
    c.t = t;
    c.Rwc.m[0][0] = view[0]; c.Rwc.m[0][1] = view[4]; c.Rwc.m[0][2] = view[8];
    c.Rwc.m[1][0] = view[1]; c.Rwc.m[1][1] = view[5]; c.Rwc.m[1][2] = view[9];
    c.Rwc.m[2][0] = view[2]; c.Rwc.m[2][1] = view[6]; c.Rwc.m[2][2] = view[10];
    const float j00 = fx / t.z, j02 = -(fx * t.x) / (t.z * t.z), j11 = fy / t.z, j12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        c.A[0][j] = c.Rwc.m[0][j] * j00 + c.Rwc.m[1][j] * 0.0f + c.Rwc.m[2][j] * j02;
        c.A[1][j] = c.Rwc.m[0][j] * 0.0f + c.Rwc.m[1][j] * j11 + c.Rwc.m[2][j] * j12;
    }
    c.Sigma.m[0][0] = cov3D[0]; c.Sigma.m[0][1] = cov3D[1]; c.Sigma.m[0][2] = cov3D[2];
    c.Sigma.m[1][0] = cov3D[1]; c.Sigma.m[1][1] = cov3D[3]; c.Sigma.m[1][2] = cov3D[4];
    c.Sigma.m[2][0] = cov3D[2]; c.Sigma.m[2][1] = cov3D[4]; c.Sigma.m[2][2] = cov3D[5];
    float B[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            B[i][j] = c.A[i][0] * c.Sigma.m[0][j] + c.A[i][1] * c.Sigma.m[1][j] + c.A[i][2] * c.Sigma.m[2][j];
    c.cov2[0] = B[0][0] * c.A[0][0] + B[0][1] * c.A[0][1] + B[0][2] * c.A[0][2];
    // cov[0][1] of glm = column 0, row 1 = (row 1 of A Sigma) . (row 0 of A): the transposed product rounds differently in the last bit
    c.cov2[1] = B[1][0] * c.A[0][0] + B[1][1] * c.A[0][1] + B[1][2] * c.A[0][2];
    c.cov2[2] = B[1][0] * c.A[1][0] + B[1][1] * c.A[1][1] + B[1][2] * c.A[1][2];
    // the 1e-6 literals are double in the reference: max(1e-6, float) -> double -> float
    const double d0 = (double)(c.cov2[0] * c.cov2[2] - c.cov2[1] * c.cov2[1]);
    const double d1 = (double)((c.cov2[0] + kernel_size) * (c.cov2[2] + kernel_size) - c.cov2[1] * c.cov2[1]);
    c.det0 = (float)(d0 > 1e-6 ? d0 : 1e-6);
    c.det1 = (float)(d1 > 1e-6 ? d1 : 1e-6);
    c.coef = (float)sqrt((double)c.det0 / ((double)c.det1 + 1e-6) + 1e-6);
    c.coef_zero = ((double)c.det0 <= 1e-6) || ((double)c.det1 <= 1e-6);

    if (!with_planes) {
        // the caller has no use for the plane fit (backward with no plane / normal / depth gradient): skip the eigen-solver
        c.eig_ok = false; c.min_id = 0; c.evmin = 1.f; c.emin = make_float3(0, 0, 0); c.well = false;
        c.ev0 = c.ev1 = c.ev2 = 1.f; c.vc0 = c.vc1 = c.vc2 = make_float3(0, 0, 0);
        c.uvh = make_float3(c.txtz, c.tytz, 1.f); c.uvh_m = c.uvh_mn = make_float3(0, 0, 0);
        c.degenerate = true;
        return;
    }
    if (!(plane_cache && plane_cache_load(*plane_cache, c, plane_tag))) cov2d_planes_eig(c);
    cov2d_planes_finish(c);
}
