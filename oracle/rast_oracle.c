/*
 * rast_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Scalar, single-threaded plain-C restatement of the RaDe-GS differentiable
 * Gaussian-splat rasterizer that IGS calls through GaussianRasterizer.
 * It restates the ALGORITHM of the reference CUDA extension
 *   DGR = submodules/RaDe-GS/submodules/diff-gaussian-rasterization
 *   DGR/cuda_rasterizer/auxiliary.h, forward.cu:23-851, backward.cu:21-1163,
 *   rasterizer_impl.cu:35-50,70-111,151-173,254-571
 * as loops over Gaussians / tiles / pixels.  Every function cites the
 * reference lines it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path and its CUDA sources cannot be
 * compiled here (no nvcc, glm submodule not vendored).  This oracle is
 * pinned only by (1) analytic micro-cases, (2) the reference's importable
 * pure-python helpers (sh_utils.eval_sh, graphics_utils.getProjectionMatrix;
 * fixtures in tests/golden/), (3) an independent PyTorch-autograd
 * restatement (oracle/torch_oracle.py) and float64 finite differences.
 *
 * Matrix convention: the reference uses glm (column-major; mat3 literals list
 * COLUMNS; m[i][j] = column i, row j).  The m3 type below keeps exactly that
 * indexing so each product can be checked against the reference line.
 *
 * Deliberately reproduced quirks of the reference:
 *  - backward.cu computeCov2DCUDA receives dL_dconic in its `conic_opacity`
 *    parameter (rasterizer_impl.cu:569), so its "combined_opacity" is
 *    dL_dconic[4*idx+3];
 *  - the conic backward adds kernel_size to a and c although the forward
 *    conic does not (backward.cu:377-379);
 *  - quaternions are not normalised (forward.cu:279, backward.cu:554).
 *
 * Per-Gaussian gradient sums are accumulated in double and rounded once
 * (the reference uses fp32 atomicAdd in nondeterministic order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef GSOR_DOUBLE
/* test-only float64 build (librast_oracle_f64.so): same code with every real = double, used to separate
 * formula errors from float32 rounding when checking against float64 autograd / finite differences */
typedef double real;
#define SQRT sqrt
#define EXP exp
#define FABS fabs
#define CEIL ceil
#else
typedef float real;
#define SQRT sqrtf
#define EXP expf
#define FABS fabsf
#define CEIL ceilf
#endif

#define TILE 16               /* config.h: BLOCK_X = BLOCK_Y = 16 */
#define NCH 3                 /* config.h: NUM_CHANNELS */
#define NORMALIZE_EPS 1.0E-12F

typedef struct { real v[3][3]; } m3;       /* v[col][row], glm indexing */
typedef struct { real x, y, z; } f3;

/* test-only switch: bit 0 drops the d(coef)/d(cov2D) terms of backward.cu:367-375,398-400 (whose "opacity"
 * operand is really dL_dconic.w in the reference) so the remaining chain can be checked against autograd */
static _Thread_local int g_flags = 0;      /* per thread: the tests evaluate several views concurrently */
void gsor_set_flags(int f) { g_flags = f; }
/* per-Gaussian sums of the blend backward: double by default (rounded once at the end, see the header); with flag 2 the running
 * sum is rounded to float after every add, i.e. the reference's float atomicAdd (backward.cu:878-1013) in ONE of its possible
 * orders -- tests use the difference between the two as the sensitivity of a Gaussian's gradient to that rounding */
#define ACCUM(x, v) do { (x) += (v); if (g_flags & 2) (x) = (double)(float)(x); } while (0)
/* flag 4: every finished per-Gaussian sum is scaled by 1 + 2e-6 u, u in [-1,1] a hash of its index (and of flags >> 8, the sample number): the size of the rounding
 * error a float atomicAdd accumulation over a few hundred partial sums leaves (any order).  The per-Gaussian backward that
 * follows amplifies it by up to 1e5 for splats seen edge-on or from inside; tests use the shift as that Gaussian's sensitivity. */
static double sum_jitter(size_t i, unsigned salt)
{
    if (!(g_flags & 4)) return 1.0;
    if (((g_flags >> 4) & 15) && (unsigned)((g_flags >> 4) & 15) != salt) return 1.0;    /* bits 4..7: jitter one family of sums only */
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)(salt + 16u * (unsigned)(g_flags >> 8)) * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return 1.0 + 2e-6 * ((double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

/* flag 16: every exp() of the tile blend (forward.cu:565, backward.cu:851) is scaled by 1 + 2.4e-7 u, u in [-1,1] a hash of (pixel,
 * Gaussian, flags >> 8).  2.4e-7 = 2 ulp, the documented maximum error of CUDA's expf (the reference is built without fast-math, its
 * exp(float) is that expf); x86 libm and the GPU's v_exp_f32 differ from it and from each other by as much.  The same (pixel,
 * Gaussian) pair gets the same factor in forward and backward -- as with any ONE deterministic exp -- so a perturbed forward must be
 * followed by a backward with the same flags.  Tests use the resulting shift of a Gaussian's gradient as its sensitivity to the
 * exp implementation (behind saturated pixels 1/T_final amplifies it, backward.cu:706,857). */
static real exp_jitter(real v, size_t pix, uint32_t id)
{
    if (!(g_flags & 16)) return v;
    uint64_t z = (uint64_t)pix * 0x9E3779B97F4A7C15ull + (uint64_t)id * 0xD6E8FEB86659FD93ull + (uint64_t)(g_flags >> 8) * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return (real)((double)v * (1.0 + 2.4e-7 * ((double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0)));
}

/* ------------------------------------------------------------------ */
/* small glm-like helpers                                              */
/* ------------------------------------------------------------------ */
static m3 m3_cols(real a, real b, real c, real d, real e, real f, real g, real h, real i)
{   /* glm::mat3(a..i): three columns (a,b,c),(d,e,f),(g,h,i) */
    m3 m; m.v[0][0]=a; m.v[0][1]=b; m.v[0][2]=c; m.v[1][0]=d; m.v[1][1]=e; m.v[1][2]=f;
    m.v[2][0]=g; m.v[2][1]=h; m.v[2][2]=i; return m;
}
static m3 m3_mul(m3 A, m3 B)
{   /* glm: (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
    m3 R;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++)
        R.v[c][r] = A.v[0][r]*B.v[c][0] + A.v[1][r]*B.v[c][1] + A.v[2][r]*B.v[c][2];
    return R;
}
static m3 m3_T(m3 A) { m3 R; for (int c=0;c<3;c++) for (int r=0;r<3;r++) R.v[c][r]=A.v[r][c]; return R; }
static f3 m3_vec(m3 A, f3 x)
{   /* glm: (A*x)[r] = A[0][r]*x0 + A[1][r]*x1 + A[2][r]*x2 */
    f3 y;
    y.x = A.v[0][0]*x.x + A.v[1][0]*x.y + A.v[2][0]*x.z;
    y.y = A.v[0][1]*x.x + A.v[1][1]*x.y + A.v[2][1]*x.z;
    y.z = A.v[0][2]*x.x + A.v[1][2]*x.y + A.v[2][2]*x.z;
    return y;
}
static m3 m3_scale(m3 A, real s) { for (int c=0;c<3;c++) for (int r=0;r<3;r++) A.v[c][r]*=s; return A; }
static m3 m3_div(m3 A, real s) { for (int c=0;c<3;c++) for (int r=0;r<3;r++) A.v[c][r]/=s; return A; }
static m3 m3_add(m3 A, m3 B) { for (int c=0;c<3;c++) for (int r=0;r<3;r++) A.v[c][r]+=B.v[c][r]; return A; }
static m3 m3_zero(void) { m3 m; memset(&m, 0, sizeof m); return m; }
static m3 m3_outer(f3 c, f3 r)
{   /* glm::outerProduct(c,r): column i = c * r[i] */
    real cc[3] = {c.x,c.y,c.z}, rr[3] = {r.x,r.y,r.z}; m3 m;
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) m.v[i][j] = cc[j]*rr[i];
    return m;
}
static f3 f3_mk(real x, real y, real z) { f3 r = {x,y,z}; return r; }
static real f3_dot(f3 a, f3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
static f3 f3_scale(f3 a, real s) { return f3_mk(a.x*s, a.y*s, a.z*s); }
static f3 f3_add(f3 a, f3 b) { return f3_mk(a.x+b.x, a.y+b.y, a.z+b.z); }
static f3 f3_sub(f3 a, f3 b) { return f3_mk(a.x-b.x, a.y-b.y, a.z-b.z); }
static real f3_len(f3 a) { return SQRT(f3_dot(a,a)); }
static f3 f3_normalize(f3 a) { real inv = 1.0f / SQRT(f3_dot(a,a)); return f3_scale(a, inv); } /* glm::normalize = v*inversesqrt(dot) */
static f3 m3_col(m3 A, int c) { return f3_mk(A.v[c][0], A.v[c][1], A.v[c][2]); }
static real fmaxf_(real a, real b) { return a > b ? a : b; }
static real fminf_(real a, real b) { return a < b ? a : b; }

/* real -> int conversion saturating like the GPU cvt instructions */
static int f2i_sat(real f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)f;
}

/* auxiliary.h:74-113 */
static f3 xform4x3(f3 p, const real* m)
{
    return f3_mk(m[0]*p.x + m[4]*p.y + m[8]*p.z + m[12],
                 m[1]*p.x + m[5]*p.y + m[9]*p.z + m[13],
                 m[2]*p.x + m[6]*p.y + m[10]*p.z + m[14]);
}
static void xform4x4(f3 p, const real* m, real out[4])
{
    out[0] = m[0]*p.x + m[4]*p.y + m[8]*p.z + m[12];
    out[1] = m[1]*p.x + m[5]*p.y + m[9]*p.z + m[13];
    out[2] = m[2]*p.x + m[6]*p.y + m[10]*p.z + m[14];
    out[3] = m[3]*p.x + m[7]*p.y + m[11]*p.z + m[15];
}
static f3 xformvec4x3T(f3 p, const real* m)
{
    return f3_mk(m[0]*p.x + m[1]*p.y + m[2]*p.z,
                 m[4]*p.x + m[5]*p.y + m[6]*p.z,
                 m[8]*p.x + m[9]*p.y + m[10]*p.z);
}
/* auxiliary.h:123-133 */
static f3 dnormvdv3(f3 v, f3 dv)
{
    real sum2 = v.x*v.x + v.y*v.y + v.z*v.z;
    real invsum32 = 1.0f / SQRT(sum2*sum2*sum2);
    f3 r;
    r.x = ((+sum2 - v.x*v.x)*dv.x - v.y*v.x*dv.y - v.z*v.x*dv.z) * invsum32;
    r.y = (-v.x*v.y*dv.x + (sum2 - v.y*v.y)*dv.y - v.z*v.y*dv.z) * invsum32;
    r.z = (-v.x*v.z*dv.x - v.y*v.z*dv.y + (sum2 - v.z*v.z)*dv.z) * invsum32;
    return r;
}
/* auxiliary.h:57-60 -- the 1.0 / 0.5 literals are double */
static real ndc2pix(real v, int S) { return (real)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* auxiliary.h:62-72.  grid dims are unsigned in the reference; all operands are >= 0 after max(0,.) */
static void get_rect(real px, real py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    real r = (real)max_radius;
    int a;
    a = f2i_sat((px - r) / (real)TILE); if (a < 0) a = 0; if (a > gx) a = gx; rmin[0] = a;
    a = f2i_sat((py - r) / (real)TILE); if (a < 0) a = 0; if (a > gy) a = gy; rmin[1] = a;
    a = f2i_sat((px + r + (real)TILE - 1.0f) / (real)TILE); if (a < 0) a = 0; if (a > gx) a = gx; rmax[0] = a;
    a = f2i_sat((py + r + (real)TILE - 1.0f) / (real)TILE); if (a < 0) a = 0; if (a > gy) a = gy; rmax[1] = a;
}

/* ------------------------------------------------------------------ */
/* 3x3 symmetric eigen-solver: Householder tridiagonalisation + QL     */
/* restates auxiliary.h:189-401 (D = 3, T = real, eps = 1e-7 absolute) */
/* returns 3, or 0 after more than 30 QL sweeps for one eigenvalue      */
/* ------------------------------------------------------------------ */
static int near0(real x) { return FABS(x - 0.0f) <= 0.0000001f; }
static real hyp(real a, real b)
{   /* auxiliary.h:200-214 */
    real absa = FABS(a), absb = FABS(b);
    if (absa > absb) { absb /= absa; absb *= absb; return absa * SQRT(1.0f + absb); }
    if (near0(absb)) return 0.0f;
    absa /= absb; absa *= absa;
    return absb * SQRT(1.0f + absa);
}
static real sgn_of(real v, real s) { return s >= 0 ? FABS(v) : -FABS(v); }

static int eig_sym3(const m3* S, real val[3], m3* vec)
{
    enum { D = 3 };
    real a[D][D], d[D], e[D];
    int i, j, k, l, m, iter;
    real scale, hh, h, g, f, s, r, p, c, b;

    for (i = 0; i < D; i++) for (j = 0; j < D; j++) a[i][j] = S->v[j][i];

    /* 1. Householder reduction (rows D-1 .. 1) */
    for (i = D - 1; i >= 1; i--) {
        l = i;                    /* number of leading elements in row i */
        h = scale = 0;
        if (l > 1) {
            for (k = 0; k < l; k++) scale += FABS(a[i][k]);
            if (near0(scale)) {
                e[i] = a[i][l - 1];
            } else {
                for (k = 0; k < l; k++) { a[i][k] /= scale; h += a[i][k] * a[i][k]; }
                f = a[i][l - 1];
                g = (f >= 0) ? -SQRT(h) : SQRT(h);
                e[i] = scale * g;
                h -= f * g;
                a[i][l - 1] = f - g;
                f = 0;
                for (j = 0; j < l; j++) {
                    a[j][i] = a[i][j] / h;
                    g = 0;
                    for (k = 0; k <= j; k++) g += a[j][k] * a[i][k];
                    for (k = j + 1; k < l; k++) g += a[k][j] * a[i][k];
                    e[j] = g / h;
                    f += e[j] * a[i][j];
                }
                hh = f / (h + h);
                for (j = 0; j < l; j++) {
                    f = a[i][j];
                    e[j] = g = e[j] - hh * f;
                    for (k = 0; k <= j; k++) a[j][k] -= (f * e[k] + g * a[i][k]);
                }
            }
        } else {
            e[i] = a[i][l - 1];
        }
        d[i] = h;
    }
    d[0] = 0; e[0] = 0;
    for (i = 0; i < D; i++) {
        l = i;
        if (!near0(d[i])) {
            for (j = 0; j < l; j++) {
                g = 0;
                for (k = 0; k < l; k++) g += a[i][k] * a[k][j];
                for (k = 0; k < l; k++) a[k][j] -= g * a[k][i];
            }
        }
        d[i] = a[i][i];
        a[i][i] = 1;
        for (j = 0; j < l; j++) a[j][i] = a[i][j] = 0;
    }

    /* 2. QL with implicit shifts */
    for (i = 1; i < D; i++) e[i - 1] = e[i];
    e[D - 1] = 0;
    for (l = 0; l < D; l++) {
        iter = 0;
        do {
            for (m = l; m < D - 1; m++) {
                if (near0(FABS(e[m]))) break;
            }
            if (m != l) {
                if (iter++ == 30) return 0;
                g = (d[l + 1] - d[l]) / (2 * e[l]);
                r = hyp(g, 1.0f);
                g = d[m] - d[l] + e[l] / (g + sgn_of(r, g));
                s = c = 1; p = 0;
                for (i = m - 1; i >= l; i--) {
                    f = s * e[i];
                    b = c * e[i];
                    e[i + 1] = r = hyp(f, g);
                    if (near0(r)) { d[i + 1] -= p; e[m] = 0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (k = 0; k < D; k++) {
                        f = a[k][i + 1];
                        a[k][i + 1] = s * a[k][i] + c * f;
                        a[k][i] = c * a[k][i] - s * f;
                    }
                }
                if (near0(r) && (i >= l)) continue;
                d[l] -= p; e[l] = g; e[m] = 0;
            }
        } while (m != l);
    }
    for (i = 0; i < D; i++) val[i] = d[i];
    for (i = 0; i < D; i++) for (j = 0; j < D; j++) vec->v[i][j] = a[j][i];
    return D;
}

/* ------------------------------------------------------------------ */
/* SH -> RGB  (forward.cu:23-74) and its backward (backward.cu:21-140)  */
/* ------------------------------------------------------------------ */
static const real SH_C0 = 0.28209479177387814f;
static const real SH_C1 = 0.4886025119029199f;
static const real SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const real SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

static void sh_basis(int deg, real x, real y, real z, real B[16])
{
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
            B[4] = SH_C2[0]*xy; B[5] = SH_C2[1]*yz; B[6] = SH_C2[2]*(2.0f*zz - xx - yy);
            B[7] = SH_C2[3]*xz; B[8] = SH_C2[4]*(xx - yy);
            if (deg > 2) {
                B[9]  = SH_C3[0]*y*(3.0f*xx - yy);
                B[10] = SH_C3[1]*xy*z;
                B[11] = SH_C3[2]*y*(4.0f*zz - xx - yy);
                B[12] = SH_C3[3]*z*(2.0f*zz - 3.0f*xx - 3.0f*yy);
                B[13] = SH_C3[4]*x*(4.0f*zz - xx - yy);
                B[14] = SH_C3[5]*z*(xx - yy);
                B[15] = SH_C3[6]*x*(xx - 3.0f*yy);
            }
        }
    }
}

static f3 sh_to_rgb(int idx, int deg, int M, const real* means, const real* campos, const real* shs, uint8_t* clamped)
{
    f3 pos = f3_mk(means[3*idx], means[3*idx+1], means[3*idx+2]);
    f3 dir = f3_sub(pos, f3_mk(campos[0], campos[1], campos[2]));
    real len = f3_len(dir);
    dir = f3_mk(dir.x/len, dir.y/len, dir.z/len);
    const real* sh = shs + (size_t)idx * M * 3;
    real B[16] = {0};
    int n = (deg + 1) * (deg + 1);
    sh_basis(deg, dir.x, dir.y, dir.z, B);
    real res[3];
    for (int ch = 0; ch < 3; ch++) {
        /* same association as the reference: SH_C0*sh0, then -C1*y*sh1 + C1*z*sh2 - C1*x*sh3, then each band added as one sum */
        real r = B[0] * sh[ch];
        if (deg > 0) {
            r = r + B[1]*sh[3+ch] + B[2]*sh[6+ch] + B[3]*sh[9+ch];
            if (deg > 1) {
                r = r + B[4]*sh[12+ch] + B[5]*sh[15+ch] + B[6]*sh[18+ch] + B[7]*sh[21+ch] + B[8]*sh[24+ch];
                if (deg > 2)
                    r = r + B[9]*sh[27+ch] + B[10]*sh[30+ch] + B[11]*sh[33+ch] + B[12]*sh[36+ch]
                          + B[13]*sh[39+ch] + B[14]*sh[42+ch] + B[15]*sh[45+ch];
            }
        }
        r += 0.5f;
        clamped[3*idx + ch] = (r < 0);
        res[ch] = r > 0.0f ? r : 0.0f;
    }
    (void)n;
    return f3_mk(res[0], res[1], res[2]);
}

static void sh_backward(int idx, int deg, int M, const real* means, const real* campos, const real* shs,
                        const uint8_t* clamped, const real* dL_dcolor, real* dL_dmeans, real* dL_dshs)
{
    f3 pos = f3_mk(means[3*idx], means[3*idx+1], means[3*idx+2]);
    f3 dir_orig = f3_sub(pos, f3_mk(campos[0], campos[1], campos[2]));
    real len = f3_len(dir_orig);
    f3 dir = f3_mk(dir_orig.x/len, dir_orig.y/len, dir_orig.z/len);
    const real* sh = shs + (size_t)idx * M * 3;
    real g[3];
    for (int ch = 0; ch < 3; ch++) g[ch] = dL_dcolor[3*idx+ch] * (clamped[3*idx+ch] ? 0.0f : 1.0f);
    real x = dir.x, y = dir.y, z = dir.z;
    real B[16] = {0};
    sh_basis(deg, x, y, z, B);
    real* dsh = dL_dshs + (size_t)idx * M * 3;
    int n = (deg + 1) * (deg + 1);
    for (int k = 0; k < n; k++) for (int ch = 0; ch < 3; ch++) dsh[3*k+ch] = B[k] * g[ch];

    real dx[3] = {0,0,0}, dy[3] = {0,0,0}, dz[3] = {0,0,0};   /* dRGB/d(dir) per channel */
#define SHc(k) sh[3*(k)+ch]
    for (int ch = 0; ch < 3; ch++) {
        if (deg > 0) {
            dx[ch] = -SH_C1 * SHc(3); dy[ch] = -SH_C1 * SHc(1); dz[ch] = SH_C1 * SHc(2);
            if (deg > 1) {
                real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
                dx[ch] += SH_C2[0]*y*SHc(4) + SH_C2[2]*2.f*-x*SHc(6) + SH_C2[3]*z*SHc(7) + SH_C2[4]*2.f*x*SHc(8);
                dy[ch] += SH_C2[0]*x*SHc(4) + SH_C2[1]*z*SHc(5) + SH_C2[2]*2.f*-y*SHc(6) + SH_C2[4]*2.f*-y*SHc(8);
                dz[ch] += SH_C2[1]*y*SHc(5) + SH_C2[2]*2.f*2.f*z*SHc(6) + SH_C2[3]*x*SHc(7);
                if (deg > 2) {
                    dx[ch] += (SH_C3[0]*SHc(9)*3.f*2.f*xy + SH_C3[1]*SHc(10)*yz + SH_C3[2]*SHc(11)*-2.f*xy
                               + SH_C3[3]*SHc(12)*-3.f*2.f*xz + SH_C3[4]*SHc(13)*(-3.f*xx + 4.f*zz - yy)
                               + SH_C3[5]*SHc(14)*2.f*xz + SH_C3[6]*SHc(15)*3.f*(xx - yy));
                    dy[ch] += (SH_C3[0]*SHc(9)*3.f*(xx - yy) + SH_C3[1]*SHc(10)*xz
                               + SH_C3[2]*SHc(11)*(-3.f*yy + 4.f*zz - xx) + SH_C3[3]*SHc(12)*-3.f*2.f*yz
                               + SH_C3[4]*SHc(13)*-2.f*xy + SH_C3[5]*SHc(14)*-2.f*yz + SH_C3[6]*SHc(15)*-3.f*2.f*xy);
                    dz[ch] += (SH_C3[1]*SHc(10)*xy + SH_C3[2]*SHc(11)*4.f*2.f*yz
                               + SH_C3[3]*SHc(12)*3.f*(2.f*zz - xx - yy) + SH_C3[4]*SHc(13)*4.f*2.f*xz
                               + SH_C3[5]*SHc(14)*(xx - yy));
                }
            }
        }
    }
#undef SHc
    f3 dL_ddir = f3_mk(dx[0]*g[0] + dx[1]*g[1] + dx[2]*g[2],
                       dy[0]*g[0] + dy[1]*g[1] + dy[2]*g[2],
                       dz[0]*g[0] + dz[1]*g[1] + dz[2]*g[2]);
    f3 dm = dnormvdv3(dir_orig, dL_ddir);
    dL_dmeans[3*idx+0] += dm.x; dL_dmeans[3*idx+1] += dm.y; dL_dmeans[3*idx+2] += dm.z;
}

/* ------------------------------------------------------------------ */
/* cov3D from scale / quaternion (forward.cu:270-304)                   */
/* ------------------------------------------------------------------ */
static m3 quat_R(const real* q)
{   /* forward.cu:279-290: q = (r,x,y,z), NOT normalised; literal lists columns */
    real r = q[0], x = q[1], y = q[2], z = q[3];
    return m3_cols(1.f - 2.f*(y*y + z*z), 2.f*(x*y - r*z), 2.f*(x*z + r*y),
                   2.f*(x*y + r*z), 1.f - 2.f*(x*x + z*z), 2.f*(y*z - r*x),
                   2.f*(x*z - r*y), 2.f*(y*z + r*x), 1.f - 2.f*(x*x + y*y));
}
static void cov3d_fwd(const real* scale, real mod, const real* rot, real* cov3D)
{
    m3 S = m3_zero();
    S.v[0][0] = mod*scale[0]; S.v[1][1] = mod*scale[1]; S.v[2][2] = mod*scale[2];
    m3 R = quat_R(rot);
    m3 Mx = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_T(Mx), Mx);
    cov3D[0] = Sigma.v[0][0]; cov3D[1] = Sigma.v[0][1]; cov3D[2] = Sigma.v[0][2];
    cov3D[3] = Sigma.v[1][1]; cov3D[4] = Sigma.v[1][2]; cov3D[5] = Sigma.v[2][2];
}
/* backward.cu:492-555 */
static void cov3d_bwd(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds,
                      real* dL_dscales, real* dL_drots)
{
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = quat_R(rot);
    m3 S = m3_zero();
    real s[3] = { mod*scale[0], mod*scale[1], mod*scale[2] };
    S.v[0][0] = s[0]; S.v[1][1] = s[1]; S.v[2][2] = s[2];
    m3 Mx = m3_mul(S, R);
    const real* g = dL_dcov3Ds + 6*idx;
    m3 dSigma = m3_cols(g[0], 0.5f*g[1], 0.5f*g[2], 0.5f*g[1], g[3], 0.5f*g[4], 0.5f*g[2], 0.5f*g[4], g[5]);
    m3 dM = m3_mul(m3_scale(Mx, 2.0f), dSigma);
    m3 Rt = m3_T(R);
    m3 dMt = m3_T(dM);
    dL_dscales[3*idx+0] = f3_dot(m3_col(Rt,0), m3_col(dMt,0));
    dL_dscales[3*idx+1] = f3_dot(m3_col(Rt,1), m3_col(dMt,1));
    dL_dscales[3*idx+2] = f3_dot(m3_col(Rt,2), m3_col(dMt,2));
    for (int k = 0; k < 3; k++) { dMt.v[0][k] *= s[0]; dMt.v[1][k] *= s[1]; dMt.v[2][k] *= s[2]; }
#define D_(a,b) dMt.v[a][b]
    real q0 = 2*z*(D_(0,1) - D_(1,0)) + 2*y*(D_(2,0) - D_(0,2)) + 2*x*(D_(1,2) - D_(2,1));
    real q1 = 2*y*(D_(1,0) + D_(0,1)) + 2*z*(D_(2,0) + D_(0,2)) + 2*r*(D_(1,2) - D_(2,1)) - 4*x*(D_(2,2) + D_(1,1));
    real q2 = 2*x*(D_(1,0) + D_(0,1)) + 2*r*(D_(2,0) - D_(0,2)) + 2*z*(D_(1,2) + D_(2,1)) - 4*y*(D_(2,2) + D_(0,0));
    real q3 = 2*r*(D_(0,1) - D_(1,0)) + 2*x*(D_(2,0) + D_(0,2)) + 2*y*(D_(1,2) + D_(2,1)) - 4*z*(D_(1,1) + D_(0,0));
#undef D_
    dL_drots[4*idx+0] = q0; dL_drots[4*idx+1] = q1; dL_drots[4*idx+2] = q2; dL_drots[4*idx+3] = q3;
}

/* ------------------------------------------------------------------ */
/* EWA projection + RaDe-GS plane geometry (forward.cu:77-264)          */
/* ------------------------------------------------------------------ */
static void cov2d_fwd(f3 mean, real fx, real fy, real tan_fovx, real tan_fovy, real kernel_size,
                      const real* cov3D, const real* view, real cov2D[3], real* camera_plane,
                      real* out_normal, real* ray_plane, real* coef_out)
{
    f3 t = xform4x3(mean, view);
    const real limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    real txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    txtz = t.x / t.z; tytz = t.y / t.z;

    m3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0, 0, 0);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 T = m3_mul(Wm, J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = m3_mul(m3_mul(m3_T(T), m3_T(Vrk)), T);
    cov2D[0] = cov.v[0][0]; cov2D[1] = cov.v[0][1]; cov2D[2] = cov.v[1][1];

    /* forward.cu:119-124: max(1e-6, real) is evaluated in double and rounded to real */
    double d0 = (double)(cov.v[0][0]*cov.v[1][1] - cov.v[0][1]*cov.v[0][1]);
    double d1 = (double)((cov.v[0][0] + kernel_size)*(cov.v[1][1] + kernel_size) - cov.v[0][1]*cov.v[0][1]);
    const real det_0 = (real)(d0 > 1e-6 ? d0 : 1e-6);
    const real det_1 = (real)(d1 > 1e-6 ? d1 : 1e-6);
    real coef = (real)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
    if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) coef = 0.0f;
    *coef_out = coef;

    real ev[3]; m3 evec;
    int Dn = eig_sym3(&Vrk, ev, &evec);
    unsigned min_id = ev[0] > ev[1] ? (ev[1] > ev[2] ? 2 : 1) : (ev[0] > ev[2] ? 2 : 0);
    m3 Vrk_inv;
    int well = (double)ev[min_id] > 0.00000001;
    if (well) {
        m3 diag = m3_cols(1/ev[0], 0, 0, 0, 1/ev[1], 0, 0, 0, 1/ev[2]);
        Vrk_inv = m3_mul(m3_mul(evec, diag), m3_T(evec));
    } else {
        f3 emin = m3_col(evec, min_id);
        Vrk_inv = m3_outer(emin, emin);
    }
    m3 cov_cam_inv = m3_mul(m3_mul(m3_T(Wm), Vrk_inv), Wm);
    f3 uvh = f3_mk(txtz, tytz, 1);
    f3 uvh_m = m3_vec(cov_cam_inv, uvh);
    f3 uvh_mn = f3_normalize(uvh_m);

    if (uvh_mn.x != uvh_mn.x || Dn == 0) {
        for (int ch = 0; ch < 6; ch++) camera_plane[ch] = 0;
        out_normal[0] = out_normal[1] = out_normal[2] = 0;
        ray_plane[0] = ray_plane[1] = 0;
    } else {
        real u2 = txtz*txtz, v2 = tytz*tytz, uv = txtz*tytz;
        real l = SQRT(t.x*t.x + t.y*t.y + t.z*t.z);
        m3 nJ = m3_cols(1 / t.z, 0.0f, -(t.x) / (t.z * t.z),
                        0.0f, 1 / t.z, -(t.y) / (t.z * t.z),
                        t.x / l, t.y / l, t.z / l);
        m3 nJ_inv = m3_cols(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
        real vbn = f3_dot(uvh_mn, uvh);
        real factor_normal = l / (u2 + v2 + 1);
        real den = fmaxf_(vbn, 0.0000001f);
        f3 plane = m3_vec(nJ_inv, f3_mk(uvh_mn.x/den, uvh_mn.y/den, uvh_mn.z/den));
        real nl = u2 + v2 + 1;
        camera_plane[0] = (-(v2 + 1)*t.z + plane.x*t.x) / nl / fx;
        camera_plane[1] = (uv*t.z + plane.y*t.x) / nl / fy;
        camera_plane[2] = (uv*t.z + plane.x*t.y) / nl / fx;
        camera_plane[3] = (-(u2 + 1)*t.z + plane.y*t.y) / nl / fy;
        camera_plane[4] = (t.x + plane.x*t.z) / nl / fx;
        camera_plane[5] = (t.y + plane.y*t.z) / nl / fy;
        ray_plane[0] = plane.x*l / nl / fx;
        ray_plane[1] = plane.y*l / nl / fy;
        f3 rnv = f3_mk(-plane.x*factor_normal, -plane.y*factor_normal, -1);
        f3 cnv = m3_vec(nJ, rnv);
        f3 nv = f3_normalize(cnv);
        out_normal[0] = nv.x; out_normal[1] = nv.y; out_normal[2] = nv.z;
    }
}

/* ------------------------------------------------------------------ */
/* state carried from forward to backward                              */
/* ------------------------------------------------------------------ */
typedef struct {
    int P, D, M, W, H, gx, gy, R;
    int require_coord, require_depth;
    real tan_fovx, tan_fovy, kernel_size, scale_modifier, fx, fy;
    /* geometry state (rasterizer_impl.h:29-48) */
    real *depths, *camera_planes, *ray_planes, *ts, *normals, *means2D, *view_points, *cov3D, *conic_opacity, *rgb;
    uint8_t* clamped;
    uint32_t* tiles_touched; uint32_t* point_offsets;
    int* radii;
    /* binning state */
    uint64_t* keys; uint32_t* point_list;
    /* image state */
    uint32_t* ranges;      /* [T][2] */
    uint32_t* n_contrib;   /* [2*H*W] */
    real *accum_coord, *accum_depth, *normal_length;
    /* test-only (flag 8): the per-Gaussian sums of the last blend backward on this state, kept so that further evaluations of the
     * per-Gaussian backward on the SAME upstream gradients (the jitter samples of flag 4) skip the per-pixel loop */
    double* acc_cache; int acc_cache_flags;
} gsor_state;

static void* xcalloc(size_t n, size_t sz) { void* p = calloc(n ? n : 1, sz); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }

void gsor_free(gsor_state* s)
{
    if (!s) return;
    free(s->depths); free(s->camera_planes); free(s->ray_planes); free(s->ts); free(s->normals); free(s->means2D);
    free(s->view_points); free(s->cov3D); free(s->conic_opacity); free(s->rgb); free(s->clamped);
    free(s->tiles_touched); free(s->point_offsets); free(s->radii); free(s->keys); free(s->point_list);
    free(s->ranges); free(s->n_contrib); free(s->accum_coord); free(s->accum_depth); free(s->normal_length);
    free(s->acc_cache);
    free(s);
}

/* rasterizer_impl.cu:35-50 */
static uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

typedef struct { uint64_t key; uint32_t val; uint32_t pos; } kv_t;
static uint64_t g_keymask;
static int kv_cmp(const void* a, const void* b)
{
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    uint64_t kx = x->key & g_keymask, ky = y->key & g_keymask;
    if (kx < ky) return -1; if (kx > ky) return 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);   /* stable, like an LSD radix sort */
}

/* per-pixel forward blend; forward.cu:428-693 */
static void render_tile_pixel_fwd(const gsor_state* s, int COORD, int DEPTH, int NORMAL, uint32_t r0, uint32_t r1,
                                  int px, int py, const real* bg, const real* features,
                                  real* out_color, real* out_coord, real* out_mcoord, real* out_depth,
                                  real* out_mdepth, real* out_alpha, real* out_normal)
{
    const int W = s->W, H = s->H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * py + px;
    const int GEO = DEPTH || COORD || NORMAL;
    real pixfx = (real)px, pixfy = (real)py;
    real pnx = (pixfx - W / 2.f) / s->fx, pny = (pixfy - H / 2.f) / s->fy;
    real ln = SQRT(pnx*pnx + pny*pny + 1);
    real T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0, max_contributor = (uint32_t)-1;
    real C[NCH] = {0}, weight = 0, Coord[3] = {0}, mCoord[3] = {0}, Depth = 0, mDepth = 0, Normal[3] = {0};

    for (uint32_t k = r0; k < r1; k++) {
        uint32_t id = s->point_list[k];
        contributor++;
        real dx = s->means2D[2*id] - pixfx, dy = s->means2D[2*id+1] - pixfy;
        const real* co = s->conic_opacity + 4*(size_t)id;
        real power = -0.5f * (co[0]*dx*dx + co[2]*dy*dy) - co[1]*dx*dy;
        if (power > 0.0f) continue;
        real alpha = fminf_(0.99f, co[3] * exp_jitter(EXP(power), pix_id, id));
        if (alpha < 1.0f / 255.0f) continue;
        real test_T = T * (1 - alpha);
        if (test_T < 0.0001f) break;          /* done = true: nothing after this is examined */
        const real aT = alpha * T;
        for (int ch = 0; ch < NCH; ch++) C[ch] += features[NCH*(size_t)id + ch] * aT;
        int before_median = T > 0.5;
        if (COORD) {
            const real* cp = s->camera_planes + 6*(size_t)id;
            const real* vp = s->view_points + 3*(size_t)id;
            real coord[3] = { vp[0] + cp[0]*dx + cp[1]*dy, vp[1] + cp[2]*dx + cp[3]*dy, vp[2] + cp[4]*dx + cp[5]*dy };
            for (int ch = 0; ch < 3; ch++) Coord[ch] += coord[ch] * aT;
            if (before_median) for (int ch = 0; ch < 3; ch++) mCoord[ch] = coord[ch];
        }
        if (DEPTH) {
            real t = s->ts[id] + (s->ray_planes[2*id]*dx + s->ray_planes[2*id+1]*dy);
            Depth += t * aT;
            if (before_median) mDepth = t;
        }
        if (NORMAL) for (int ch = 0; ch < 3; ch++) Normal[ch] += s->normals[3*(size_t)id+ch] * aT;
        if (GEO && before_median) max_contributor = contributor;
        weight += aT;
        T = test_T;
        last_contributor = contributor;
    }
    s->n_contrib[pix_id] = last_contributor;
    s->n_contrib[pix_id + HW] = max_contributor;
    for (int ch = 0; ch < NCH; ch++) out_color[ch*HW + pix_id] = C[ch] + T * bg[ch];
    out_alpha[pix_id] = weight;
    if (COORD) {
        for (int ch = 0; ch < 3; ch++) {
            out_coord[ch*HW + pix_id] = last_contributor ? Coord[ch] / weight : 0;
            s->accum_coord[ch*HW + pix_id] = Coord[ch];
            out_mcoord[ch*HW + pix_id] = mCoord[ch];
        }
    }
    if (DEPTH) {
        real depth_ln = Depth / ln;
        s->accum_depth[pix_id] = depth_ln;
        out_depth[pix_id] = last_contributor ? depth_ln / weight : 0;
        out_mdepth[pix_id] = mDepth / ln;
    }
    if (NORMAL) {
        if (last_contributor) {
            real len = SQRT(Normal[0]*Normal[0] + Normal[1]*Normal[1] + Normal[2]*Normal[2]);
            s->normal_length[pix_id] = len;
            len = fmaxf_(len, NORMALIZE_EPS);
            for (int ch = 0; ch < 3; ch++) out_normal[ch*HW + pix_id] = Normal[ch] / len;
        } else {
            s->normal_length[pix_id] = 1;
            for (int ch = 0; ch < 3; ch++) out_normal[ch*HW + pix_id] = 0;
        }
    }
}

/*
 * Forward: rasterizer_impl.cu:254-425 (Rasterizer::forward) with
 * forward.cu:307-423 (preprocess), rasterizer_impl.cu:70-111 (keys), 151-173 (ranges).
 * Optional pointers follow the reference's convention: NULL = "empty tensor".
 * Outputs must be zero-filled by the caller (rasterize_points.cu:71-78).
 * Returns NULL and sets *err for the `prefiltered` trap.
 */
gsor_state* gsor_forward(int P, int D, int M, int W, int H, const real* bg,
                         const real* means3D, const real* shs, const real* colors_precomp, const real* opacities,
                         const real* scales, real scale_modifier, const real* rotations, const real* cov3D_precomp,
                         const real* viewmatrix, const real* projmatrix, const real* campos,
                         real tan_fovx, real tan_fovy, real kernel_size, int prefiltered,
                         int require_coord, int require_depth,
                         real* out_color, real* out_coord, real* out_mcoord, real* out_depth, real* out_mdepth,
                         real* out_alpha, real* out_normal, int* radii_out, int* num_rendered, int* err)
{
    gsor_state* s = (gsor_state*)xcalloc(1, sizeof *s);
    *err = 0;
    s->P = P; s->D = D; s->M = M; s->W = W; s->H = H;
    s->gx = (W + TILE - 1) / TILE; s->gy = (H + TILE - 1) / TILE;
    s->require_coord = require_coord; s->require_depth = require_depth;
    s->tan_fovx = tan_fovx; s->tan_fovy = tan_fovy; s->kernel_size = kernel_size; s->scale_modifier = scale_modifier;
    s->fy = H / (2.0f * tan_fovy); s->fx = W / (2.0f * tan_fovx);
    const size_t HW = (size_t)H * W;
    const int Tn = s->gx * s->gy;
    s->depths = xcalloc(P, sizeof(real)); s->camera_planes = xcalloc((size_t)P*6, sizeof(real)); s->ray_planes = xcalloc((size_t)P*2, sizeof(real));
    s->ts = xcalloc(P, sizeof(real)); s->normals = xcalloc((size_t)P*3, sizeof(real)); s->means2D = xcalloc((size_t)P*2, sizeof(real));
    s->view_points = xcalloc((size_t)P*3, sizeof(real)); s->cov3D = xcalloc((size_t)P*6, sizeof(real)); s->conic_opacity = xcalloc((size_t)P*4, sizeof(real));
    s->rgb = xcalloc((size_t)P*3, sizeof(real)); s->clamped = xcalloc((size_t)P*3, 1);
    s->tiles_touched = xcalloc(P, 4); s->point_offsets = xcalloc(P, 4); s->radii = xcalloc(P, 4);
    s->ranges = xcalloc((size_t)Tn*2, 4); s->n_contrib = xcalloc(HW*2, 4);
    s->accum_coord = xcalloc(HW*3, sizeof(real)); s->accum_depth = xcalloc(HW, sizeof(real)); s->normal_length = xcalloc(HW, sizeof(real));

    if (P == 0) { *num_rendered = 0; return s; }   /* rasterize_points.cu:90: nothing is launched, outputs stay zero */

    /* ---- preprocess, forward.cu:307-423 ---- */
    for (int i = 0; i < P; i++) {
        s->radii[i] = 0; s->tiles_touched[i] = 0;
        f3 p_orig = f3_mk(means3D[3*i], means3D[3*i+1], means3D[3*i+2]);
        real ph[4]; xform4x4(p_orig, projmatrix, ph);
        real p_w = 1.0f / (ph[3] + 0.0000001f);
        real ppx = ph[0]*p_w, ppy = ph[1]*p_w;
        f3 p_view = xform4x3(p_orig, viewmatrix);
        if (p_view.z <= 0.2f) {               /* auxiliary.h:170 */
            if (prefiltered) { *err = 1; gsor_free(s); return NULL; }
            continue;
        }
        const real* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6*(size_t)i;
        else { cov3d_fwd(scales + 3*(size_t)i, scale_modifier, rotations + 4*(size_t)i, s->cov3D + 6*(size_t)i); cov3D = s->cov3D + 6*(size_t)i; }
        real cov2D[3], coef;
        cov2d_fwd(p_orig, s->fx, s->fy, tan_fovx, tan_fovy, kernel_size, cov3D, viewmatrix, cov2D,
                  s->camera_planes + 6*(size_t)i, s->normals + 3*(size_t)i, s->ray_planes + 2*(size_t)i, &coef);
        s->ts[i] = SQRT(p_view.x*p_view.x + p_view.y*p_view.y + p_view.z*p_view.z);
        real cx = cov2D[0], cy = cov2D[1], cz = cov2D[2];
        real det = cx*cz - cy*cy;
        if (det == 0.0f) continue;
        real det_inv = 1.f / det;
        real conic[3] = { cz*det_inv, -cy*det_inv, cx*det_inv };
        real mid = 0.5f * (cx + cz);
        real lambda1 = mid + SQRT(fmaxf_(0.1f, mid*mid - det));
        real lambda2 = mid - SQRT(fmaxf_(0.1f, mid*mid - det));
        real my_radius = CEIL(3.f * SQRT(fmaxf_(lambda1, lambda2)));
        real pix = ndc2pix(ppx, W), piy = ndc2pix(ppy, H);
        int rmin[2], rmax[2];
        get_rect(pix, piy, f2i_sat(my_radius), s->gx, s->gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            f3 c = sh_to_rgb(i, D, M, means3D, campos, shs, s->clamped);
            s->rgb[3*i] = c.x; s->rgb[3*i+1] = c.y; s->rgb[3*i+2] = c.z;
        }
        s->depths[i] = p_view.z;
        s->view_points[3*i] = p_view.x; s->view_points[3*i+1] = p_view.y; s->view_points[3*i+2] = p_view.z;
        s->radii[i] = f2i_sat(my_radius);
        s->means2D[2*i] = pix; s->means2D[2*i+1] = piy;
        s->conic_opacity[4*i] = conic[0]; s->conic_opacity[4*i+1] = conic[1]; s->conic_opacity[4*i+2] = conic[2];
        s->conic_opacity[4*i+3] = opacities[i] * coef;
        s->tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
    if (radii_out) memcpy(radii_out, s->radii, (size_t)P * 4);

    /* ---- inclusive scan, rasterizer_impl.cu:350-354 ---- */
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += s->tiles_touched[i]; s->point_offsets[i] = run; }
    const int R = P ? (int)run : 0;
    s->R = R; *num_rendered = R;

    /* ---- duplicateWithKeys, rasterizer_impl.cu:70-111 ---- */
    kv_t* kv = (kv_t*)xcalloc(R, sizeof(kv_t));
    for (int i = 0; i < P; i++) {
        if (s->radii[i] > 0) {
            uint32_t off = (i == 0) ? 0 : s->point_offsets[i-1];
            int rmin[2], rmax[2];
            get_rect(s->means2D[2*i], s->means2D[2*i+1], s->radii[i], s->gx, s->gy, rmin, rmax);
            float dflt = (float)s->depths[i]; uint32_t dbits; memcpy(&dbits, &dflt, 4);
            for (int y = rmin[1]; y < rmax[1]; y++) for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(uint32_t)(y * s->gx + x);
                key <<= 32; key |= dbits;
                kv[off].key = key; kv[off].val = (uint32_t)i; kv[off].pos = off; off++;
            }
        }
    }
    /* ---- stable sort on bits [0, 32+bit), rasterizer_impl.cu:373-381 ---- */
    uint32_t bit = higher_msb((uint32_t)Tn);
    g_keymask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
    qsort(kv, R, sizeof(kv_t), kv_cmp);
    s->keys = xcalloc(R, 8); s->point_list = xcalloc(R, 4);
    for (int k = 0; k < R; k++) { s->keys[k] = kv[k].key; s->point_list[k] = kv[k].val; }
    free(kv);
    /* ---- identifyTileRanges, rasterizer_impl.cu:151-173 (ranges zeroed first, :383) ---- */
    for (int k = 0; k < R; k++) {
        uint32_t cur = (uint32_t)(s->keys[k] >> 32);
        if (k == 0) s->ranges[2*cur] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[k-1] >> 32);
            if (cur != prev) { s->ranges[2*prev+1] = k; s->ranges[2*cur] = k; }
        }
        if (k == R - 1) s->ranges[2*cur+1] = R;
    }

    /* ---- render, forward.cu:696-742 dispatch + 428-693 ---- */
    int COORD = require_coord, DEPTH = require_depth, NORMAL = (require_coord || require_depth);
    const real* feat = colors_precomp ? colors_precomp : s->rgb;
    for (int ty = 0; ty < s->gy; ty++) for (int tx = 0; tx < s->gx; tx++) {
        uint32_t r0 = s->ranges[2*(ty*s->gx + tx)], r1 = s->ranges[2*(ty*s->gx + tx) + 1];
        for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
            int px = tx*TILE + lx, py = ty*TILE + ly;
            if (px >= W || py >= H) continue;
            render_tile_pixel_fwd(s, COORD, DEPTH, NORMAL, r0, r1, px, py, bg, feat, out_color, out_coord, out_mcoord,
                                  out_depth, out_mdepth, out_alpha, out_normal);
        }
    }
    return s;
}

/* accessors for stage-by-stage comparison against the HIP kernels */
#define GETTER(name, type, field) const type* gsor_get_##name(const gsor_state* s) { return s->field; }
GETTER(depths, real, depths) GETTER(camera_planes, real, camera_planes) GETTER(ray_planes, real, ray_planes)
GETTER(ts, real, ts) GETTER(normals, real, normals) GETTER(means2D, real, means2D) GETTER(view_points, real, view_points)
GETTER(cov3D, real, cov3D) GETTER(conic_opacity, real, conic_opacity) GETTER(rgb, real, rgb)
GETTER(clamped, uint8_t, clamped) GETTER(tiles_touched, uint32_t, tiles_touched) GETTER(point_list, uint32_t, point_list)
GETTER(ranges, uint32_t, ranges) GETTER(n_contrib, uint32_t, n_contrib) GETTER(keys, uint64_t, keys)
int gsor_num_rendered(const gsor_state* s) { return s->R; }

/* ------------------------------------------------------------------ */
/* backward                                                            */
/* ------------------------------------------------------------------ */
typedef struct {   /* per-Gaussian sums of the blend backward, accumulated in double */
    double *mean2D, *conic, *opacity, *colors, *view_points, *ts, *camera_planes, *ray_planes, *normals;
} gacc_t;

/* backward.cu:631-1016 for one pixel */
static void render_pixel_bwd(const gsor_state* s, int COORD, int DEPTH, int NORMAL, uint32_t r0, uint32_t r1,
                             int px, int py, const real* bg, const real* colors, const real* alphas,
                             const real* normalmap, const real* dL_dpixels, const real* dL_dpixel_coords,
                             const real* dL_dpixel_mcoords, const real* dL_dpixel_depths,
                             const real* dL_dpixel_mdepths, const real* dL_dalphas,
                             const real* dL_dpixel_normals, gacc_t* A)
{
    const int W = s->W, H = s->H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * py + px;
    const int GEO = COORD || DEPTH || NORMAL;
    const real fx = s->fx, fy = s->fy;
    real pixfx = (real)px, pixfy = (real)py;
    real pnx = (pixfx - W / 2.f) / fx, pny = (pixfy - H / 2.f) / fy;
    real ln = SQRT(pnx*pnx + pny*pny + 1);
    const real T_final = 1 - alphas[pix_id];
    const real w_final = alphas[pix_id];
    real T = T_final;
    const int toDo = (int)(r1 - r0);
    uint32_t contributor = (uint32_t)toDo;
    const int last_contributor = (int)s->n_contrib[pix_id];
    const int max_contributor = (int)s->n_contrib[pix_id + HW];
    real accum_rec[NCH] = {0}, dL_dpixel[NCH];
    real accum_coord_rec[3] = {0}, dL_dpixel_coord[3] = {0}, dL_dpixel_mcoord[3] = {0};
    real accum_t_rec = 0, dL_dpixel_t = 0, dL_dpixel_mt = 0;
    real accum_alpha_rec = 0, dL_dalpha;
    real accum_normal_rec[3] = {0}, dL_dpixel_normal[3] = {0};

    for (int i = 0; i < NCH; i++) dL_dpixel[i] = dL_dpixels[i*HW + pix_id];
    dL_dalpha = dL_dalphas[pix_id];
    if (GEO) {
        real ww = w_final * w_final;
        if (COORD) for (int i = 0; i < 3; i++) {
            real g = dL_dpixel_coords[i*HW + pix_id];
            dL_dalpha -= g * s->accum_coord[i*HW + pix_id] / ww;
            dL_dpixel_coord[i] = g / w_final;
            dL_dpixel_mcoord[i] = dL_dpixel_mcoords[i*HW + pix_id];
        }
        if (DEPTH) {
            real g = dL_dpixel_depths[pix_id];
            dL_dalpha -= g * s->accum_depth[pix_id] / ww;
            dL_dpixel_t = g / w_final / ln;
            dL_dpixel_mt = dL_dpixel_mdepths[pix_id] / ln;
        }
        if (NORMAL) {
            f3 gn = f3_mk(dL_dpixel_normals[pix_id], dL_dpixel_normals[HW + pix_id], dL_dpixel_normals[2*HW + pix_id]);
            f3 nn = f3_mk(normalmap[pix_id], normalmap[HW + pix_id], normalmap[2*HW + pix_id]);
            real nlen = s->normal_length[pix_id];
            f3 dL;
            if (nlen < NORMALIZE_EPS) dL = f3_mk(gn.x / NORMALIZE_EPS, gn.y / NORMALIZE_EPS, gn.z / NORMALIZE_EPS);
            else { real dt = f3_dot(gn, nn); dL = f3_mk((gn.x - dt*nn.x)/nlen, (gn.y - dt*nn.y)/nlen, (gn.z - dt*nn.z)/nlen); }
            dL_dpixel_normal[0] = dL.x; dL_dpixel_normal[1] = dL.y; dL_dpixel_normal[2] = dL.z;
        }
    }
    real last_alpha = 0, last_color[NCH] = {0}, last_coord[3] = {0}, last_t = 0, last_normal[3] = {0};
    const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);

    for (int k = (int)r1 - 1; k >= (int)r0; k--) {
        contributor--;
        if (contributor >= (uint32_t)last_contributor) continue;
        const uint32_t id = s->point_list[k];
        const real dx = s->means2D[2*id] - pixfx, dy = s->means2D[2*id+1] - pixfy;
        const real* co = s->conic_opacity + 4*(size_t)id;
        real power = -0.5f * (co[0]*dx*dx + co[2]*dy*dy) - co[1]*dx*dy;
        if (power > 0.0f) continue;
        const real G = exp_jitter(EXP(power), pix_id, id);
        const real alpha = fminf_(0.99f, co[3] * G);
        if (alpha < 1.0f / 255.0f) continue;
        T = T / (1.f - alpha);
        const real dchannel_dcolor = alpha * T;
        real dL_dopa = 0.0f;
        for (int ch = 0; ch < NCH; ch++) {
            const real c = colors[NCH*(size_t)id + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            dL_dopa += (c - accum_rec[ch]) * dL_dpixel[ch];
            ACCUM(A->colors[NCH*(size_t)id + ch], dchannel_dcolor * dL_dpixel[ch]);
        }
        real dL_dcoords[3] = {0,0,0}, dL_dt = 0;
        const real* cp = s->camera_planes + 6*(size_t)id;
        const real* rp = s->ray_planes + 2*(size_t)id;
        if (COORD) {
            const real* vp = s->view_points + 3*(size_t)id;
            real coord[3] = { vp[0] + cp[0]*dx + cp[1]*dy, vp[1] + cp[2]*dx + cp[3]*dy, vp[2] + cp[4]*dx + cp[5]*dy };
            for (int ch = 0; ch < 3; ch++) {
                const real c = coord[ch];
                accum_coord_rec[ch] = last_alpha * last_coord[ch] + (1.f - last_alpha) * accum_coord_rec[ch];
                last_coord[ch] = c;
                dL_dopa += (c - accum_coord_rec[ch]) * dL_dpixel_coord[ch];
                dL_dcoords[ch] = dchannel_dcolor * dL_dpixel_coord[ch];
                if (contributor == (uint32_t)(max_contributor - 1)) dL_dcoords[ch] += dL_dpixel_mcoord[ch];
            }
            for (int ch = 0; ch < 3; ch++) {
                ACCUM(A->view_points[3*(size_t)id + ch], dL_dcoords[ch]);
                ACCUM(A->camera_planes[6*(size_t)id + 2*ch], dL_dcoords[ch] * dx / fx);
                ACCUM(A->camera_planes[6*(size_t)id + 2*ch + 1], dL_dcoords[ch] * dy / fy);
            }
        }
        if (DEPTH) {
            real t = s->ts[id] + (rp[0]*dx + rp[1]*dy);
            accum_t_rec = last_alpha * last_t + (1.f - last_alpha) * accum_t_rec;
            last_t = t;
            dL_dopa += (t - accum_t_rec) * dL_dpixel_t;
            dL_dt = dchannel_dcolor * dL_dpixel_t;
            if (contributor == (uint32_t)(max_contributor - 1)) dL_dt += dL_dpixel_mt;
            ACCUM(A->ts[id], dL_dt);
            ACCUM(A->ray_planes[2*(size_t)id], dL_dt * dx / fx);
            ACCUM(A->ray_planes[2*(size_t)id + 1], dL_dt * dy / fy);
        }
        if (NORMAL) for (int ch = 0; ch < 3; ch++) {
            const real c = s->normals[3*(size_t)id + ch];
            accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
            last_normal[ch] = c;
            dL_dopa += (c - accum_normal_rec[ch]) * dL_dpixel_normal[ch];
            ACCUM(A->normals[3*(size_t)id + ch], dchannel_dcolor * dL_dpixel_normal[ch]);
        }
        accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
        dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
        dL_dopa *= T;
        last_alpha = alpha;
        real bg_dot_dpixel = 0;
        for (int i = 0; i < NCH; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
        dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

        const real dL_dG = co[3] * dL_dopa;
        const real gdx = G * dx, gdy = G * dy;
        const real dG_ddelx = -gdx * co[0] - gdy * co[1];
        const real dG_ddely = -gdy * co[2] - gdx * co[1];
        real dL_ddelx = dL_dG * dG_ddelx, dL_ddely = dL_dG * dG_ddely;
        if (COORD) {
            dL_ddelx += dL_dcoords[0]*cp[0] + dL_dcoords[1]*cp[2] + dL_dcoords[2]*cp[4];
            dL_ddely += dL_dcoords[0]*cp[1] + dL_dcoords[1]*cp[3] + dL_dcoords[2]*cp[5];
        }
        if (DEPTH) { dL_ddelx += dL_dt * rp[0]; dL_ddely += dL_dt * rp[1]; }
        ACCUM(A->mean2D[3*(size_t)id], dL_ddelx * ddelx_dx);
        ACCUM(A->mean2D[3*(size_t)id + 1], dL_ddely * ddely_dy);
        ACCUM(A->mean2D[3*(size_t)id + 2], FABS(dL_dG * dG_ddelx * ddelx_dx) + FABS(dL_dG * dG_ddely * ddely_dy));
        ACCUM(A->conic[4*(size_t)id], -0.5f * gdx * dx * dL_dG);
        ACCUM(A->conic[4*(size_t)id + 1], -0.5f * gdx * dy * dL_dG);
        ACCUM(A->conic[4*(size_t)id + 3], -0.5f * gdy * dy * dL_dG);
        ACCUM(A->opacity[id], G * dL_dopa);
    }
}

/* backward.cu:145-488 (computeCov2DCUDA) for one Gaussian.
 * `conic_opacity_arg` is what the reference passes in that parameter: dL_dconic (rasterizer_impl.cu:569). */
static void cov2d_bwd(int idx, const real* means, const int* radii, const real* cov3Ds, real h_x, real h_y,
                      real tan_fovx, real tan_fovy, real kernel_size, const real* view,
                      const real* dL_dconics, const real* dL_dcamera_planes, const real* dL_dray_planes,
                      const real* dL_dnormals, real* dL_dmeans, real* dL_dcov, const real* conic_opacity_arg,
                      real* dL_dopacity)
{
    if (!(radii[idx] > 0)) return;
    const real* cov3D = cov3Ds + 6*(size_t)idx;
    f3 mean = f3_mk(means[3*idx], means[3*idx+1], means[3*idx+2]);
    real dLc_x = dL_dconics[4*idx], dLc_y = dL_dconics[4*idx+1], dLc_z = dL_dconics[4*idx+3];
    f3 dL_dnormal = f3_mk(dL_dnormals[3*idx], dL_dnormals[3*idx+1], dL_dnormals[3*idx+2]);
    const real combined_opacity = conic_opacity_arg[4*idx+3];
    const real c0x = dL_dcamera_planes[6*idx],   c0y = dL_dcamera_planes[6*idx+1];
    const real c1x = dL_dcamera_planes[6*idx+2], c1y = dL_dcamera_planes[6*idx+3];
    const real c2x = dL_dcamera_planes[6*idx+4], c2y = dL_dcamera_planes[6*idx+5];
    const real drx = dL_dray_planes[2*idx], dry = dL_dray_planes[2*idx+1];

    f3 t = xform4x3(mean, view);
    const real limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    real txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
    const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
    txtz = t.x / t.z; tytz = t.y / t.z;

    m3 J = m3_cols(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 T = m3_mul(Wm, J);
    m3 cov2D = m3_mul(m3_mul(m3_T(T), m3_T(Vrk)), T);

    double d0 = (double)(cov2D.v[0][0]*cov2D.v[1][1] - cov2D.v[0][1]*cov2D.v[0][1]);
    double d1 = (double)((cov2D.v[0][0] + kernel_size)*(cov2D.v[1][1] + kernel_size) - cov2D.v[0][1]*cov2D.v[0][1]);
    const real det_0 = (real)(d0 > 1e-6 ? d0 : 1e-6);
    const real det_1 = (real)(d1 > 1e-6 ? d1 : 1e-6);
    const real coef = (real)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);

    real ev[3]; m3 evec;
    int Dn = eig_sym3(&Vrk, ev, &evec);
    unsigned min_id = ev[0] > ev[1] ? (ev[1] > ev[2] ? 2 : 1) : (ev[0] > ev[2] ? 2 : 0);
    m3 Vrk_inv; f3 eigenvector_min = f3_mk(0,0,0);
    int well = (double)ev[min_id] > 0.00000001;
    if (well) {
        m3 diag = m3_cols(1/ev[0], 0, 0, 0, 1/ev[1], 0, 0, 0, 1/ev[2]);
        Vrk_inv = m3_mul(m3_mul(evec, diag), m3_T(evec));
    } else {
        eigenvector_min = m3_col(evec, min_id);
        Vrk_inv = m3_outer(eigenvector_min, eigenvector_min);
    }
    m3 cov_cam_inv = m3_mul(m3_mul(m3_T(Wm), Vrk_inv), Wm);
    f3 uvh = f3_mk(txtz, tytz, 1);
    f3 uvh_m = m3_vec(cov_cam_inv, uvh);
    f3 uvh_mn = f3_normalize(uvh_m);
    real u2 = txtz*txtz, v2 = tytz*tytz, uv = txtz*tytz;

    m3 dL_dVrk = m3_zero(), dL_dnJ = m3_zero();
    f3 plane = f3_mk(0,0,0);
    real dL_du = 0, dL_dv = 0, dL_dl = 0, l = 1, nl = 1;
    if (!(uvh_mn.x != uvh_mn.x || Dn == 0)) {
        real vb = f3_dot(uvh_m, uvh), vbn = f3_dot(uvh_mn, uvh);
        l = SQRT(t.x*t.x + t.y*t.y + t.z*t.z);
        m3 nJ = m3_cols(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
        m3 nJ_inv = m3_cols(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
        real clamp_vb = fmaxf_(vb, 0.0000001f), clamp_vbn = fmaxf_(vbn, 0.0000001f);
        nl = u2 + v2 + 1;
        real factor_normal = l / nl;
        f3 uvh_m_vb = f3_mk(uvh_mn.x/clamp_vbn, uvh_mn.y/clamp_vbn, uvh_mn.z/clamp_vbn);
        plane = m3_vec(nJ_inv, uvh_m_vb);
        real cp0x = (-(v2 + 1)*t.z + plane.x*t.x)/nl, cp0y = (uv*t.z + plane.y*t.x)/nl;
        real cp1x = (uv*t.z + plane.x*t.y)/nl,        cp1y = (-(u2 + 1)*t.z + plane.y*t.y)/nl;
        real cp2x = (t.x + plane.x*t.z)/nl,            cp2y = (t.y + plane.y*t.z)/nl;
        real rpx = plane.x*factor_normal, rpy = plane.y*factor_normal;
        f3 rnv = f3_mk(-plane.x*factor_normal, -plane.y*factor_normal, -1);
        f3 cnv = m3_vec(nJ, rnv);
        f3 nv = f3_normalize(cnv);
        real lv = f3_len(cnv);
        f3 dL_dnormal_lv = f3_mk(dL_dnormal.x/lv, dL_dnormal.y/lv, dL_dnormal.z/lv);
        f3 dL_dcnv = f3_sub(dL_dnormal_lv, f3_scale(nv, f3_dot(nv, dL_dnormal_lv)));
        f3 dL_drnv = m3_vec(m3_T(nJ), dL_dcnv);
        dL_dnJ = m3_outer(dL_dcnv, rnv);
        dL_dl = (-plane.x*dL_drnv.x - plane.y*dL_drnv.y + plane.x*drx + plane.y*dry) / nl;
        real dpx = (t.x*c0x + t.y*c1x + t.z*c2x - l*dL_drnv.x + drx*l) / nl;
        real dpy = (t.x*c0y + t.y*c1y + t.z*c2y - l*dL_drnv.y + dry*l) / nl;
        f3 dpa = f3_mk(dpx, dpy, 0);
        real dL_dnl = (-c0x*cp0x - c0y*cp0y - c1x*cp1x - c1y*cp1y - c2x*cp2x - c2y*cp2y
                        - dL_drnv.x*rnv.x - dL_drnv.y*rnv.y - drx*rpx - dry*rpy) / nl;
        real tmp = dpx*plane.x + dpy*plane.y;
        f3 W_uvh = m3_vec(Wm, uvh);
        if (well) {
            f3 a = m3_vec(Vrk_inv, W_uvh);
            f3 inner = f3_add(f3_scale(W_uvh, -tmp), m3_vec(m3_mul(Wm, m3_T(nJ_inv)), dpa));
            f3 b = m3_vec(m3_div(Vrk_inv, clamp_vb), inner);   /* (Vrk_inv/clamp_vb) * (...) */
            dL_dVrk = m3_scale(m3_outer(a, b), -1.0f);
        } else {
            real dL_dvb = -tmp / clamp_vb;
            f3 q = m3_vec(m3_T(nJ_inv), f3_mk(dpx / clamp_vb, dpy / clamp_vb, 0));
            m3 dVi = m3_outer(W_uvh, f3_add(f3_scale(W_uvh, dL_dvb), m3_vec(Wm, q)));
            f3 dLv = m3_vec(m3_add(dVi, m3_T(dVi)), eigenvector_min);
            for (int j = 0; j < 3; j++) if ((unsigned)j != min_id) {
                real sc = f3_dot(m3_col(evec, j), dLv) / fminf_(ev[min_id] - ev[j], -0.0000001f);
                dL_dVrk = m3_add(dL_dVrk, m3_outer(f3_scale(m3_col(evec, j), sc), eigenvector_min));
            }
        }
        /* (cov_cam_inv/clamp_vb) * transpose(nJ_inv) * dpa : glm groups (A*B)*v */
        f3 dL_duvh = f3_add(f3_scale(uvh_m_vb, 2 * (-tmp)),
                            m3_vec(m3_mul(m3_div(cov_cam_inv, clamp_vb), m3_T(nJ_inv)), dpa));
        m3 dnji = m3_outer(dpa, uvh_m_vb);
        dL_du = dL_dnl*2*txtz + dL_duvh.x + (dnji.v[0][1] + dnji.v[1][0])*(-tytz) + 2*dnji.v[1][1]*txtz - dnji.v[2][0]
                + (c0y*t.y + c1x*t.y + c1y*(-2*t.x)) / nl;
        dL_dv = dL_dnl*2*tytz + dL_duvh.y + (dnji.v[0][1] + dnji.v[1][0])*(-txtz) + 2*dnji.v[0][0]*tytz - dnji.v[2][1]
                + (c0x*(-2*t.y) + c0y*t.x + c1x*t.x) / nl;
    }

    /* backward.cu:367-375; literals 1e-6, 0.5, 1., -2. are double */
    const real opacity = (real)((double)combined_opacity / ((double)coef + 1e-6));
    const real dL_dcoef = dL_dopacity[idx] * opacity;
    const real dL_dsqrtcoef = (real)((double)dL_dcoef * 0.5 * 1. / ((double)coef + 1e-6));
    const real dL_ddet0 = (real)((double)dL_dsqrtcoef / ((double)det_1 + 1e-6));
    const real dL_ddet1 = (real)((double)(dL_dsqrtcoef * det_0) * (-1.f / ((double)(det_1 * det_1) + 1e-6)));
    const real dcoef_da = dL_ddet0 * cov2D.v[1][1] + dL_ddet1 * (cov2D.v[1][1] + kernel_size);
    const real dcoef_db = (real)((double)dL_ddet0 * (-2. * (double)cov2D.v[0][1]) + (double)dL_ddet1 * (-2. * (double)cov2D.v[0][1]));
    const real dcoef_dc = dL_ddet0 * cov2D.v[0][0] + dL_ddet1 * (cov2D.v[0][0] + kernel_size);

    real a = cov2D.v[0][0] + kernel_size, b = cov2D.v[0][1], c = cov2D.v[1][1] + kernel_size;
    real denom = a*c - b*b;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    real denom2inv = 1.0f / ((denom*denom) + 0.0000001f);
    real* dcv = dL_dcov + 6*(size_t)idx;
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c*c*dLc_x + 2*b*c*dLc_y + (denom - a*c)*dLc_z);
        dL_dc = denom2inv * (-a*a*dLc_z + 2*a*b*dLc_y + (denom - a*c)*dLc_x);
        dL_db = denom2inv * 2 * (b*c*dLc_x - (denom + 2*b*b)*dLc_y + a*b*dLc_z);
        if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) {
            dL_dopacity[idx] = 0;
        } else {
            if (!(g_flags & 1)) { dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db; }
            dL_dopacity[idx] = dL_dopacity[idx] * coef;
        }
#define T_(c_,r_) T.v[c_][r_]
        dcv[0] = (T_(0,0)*T_(0,0)*dL_da + T_(0,0)*T_(1,0)*dL_db + T_(1,0)*T_(1,0)*dL_dc);
        dcv[3] = (T_(0,1)*T_(0,1)*dL_da + T_(0,1)*T_(1,1)*dL_db + T_(1,1)*T_(1,1)*dL_dc);
        dcv[5] = (T_(0,2)*T_(0,2)*dL_da + T_(0,2)*T_(1,2)*dL_db + T_(1,2)*T_(1,2)*dL_dc);
        dcv[1] = 2*T_(0,0)*T_(0,1)*dL_da + (T_(0,0)*T_(1,1) + T_(0,1)*T_(1,0))*dL_db + 2*T_(1,0)*T_(1,1)*dL_dc;
        dcv[2] = 2*T_(0,0)*T_(0,2)*dL_da + (T_(0,0)*T_(1,2) + T_(0,2)*T_(1,0))*dL_db + 2*T_(1,0)*T_(1,2)*dL_dc;
        dcv[4] = 2*T_(0,2)*T_(0,1)*dL_da + (T_(0,1)*T_(1,2) + T_(0,2)*T_(1,1))*dL_db + 2*T_(1,1)*T_(1,2)*dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dcv[i] = 0;
    }
    dcv[0] += dL_dVrk.v[0][0]; dcv[3] += dL_dVrk.v[1][1]; dcv[5] += dL_dVrk.v[2][2];
    dcv[1] += dL_dVrk.v[0][1] + dL_dVrk.v[1][0];
    dcv[2] += dL_dVrk.v[0][2] + dL_dVrk.v[2][0];
    dcv[4] += dL_dVrk.v[1][2] + dL_dVrk.v[2][1];

#define V_(c_,r_) Vrk.v[c_][r_]
    real dL_dT00 = 2*(T_(0,0)*V_(0,0) + T_(0,1)*V_(0,1) + T_(0,2)*V_(0,2))*dL_da + (T_(1,0)*V_(0,0) + T_(1,1)*V_(0,1) + T_(1,2)*V_(0,2))*dL_db;
    real dL_dT01 = 2*(T_(0,0)*V_(1,0) + T_(0,1)*V_(1,1) + T_(0,2)*V_(1,2))*dL_da + (T_(1,0)*V_(1,0) + T_(1,1)*V_(1,1) + T_(1,2)*V_(1,2))*dL_db;
    real dL_dT02 = 2*(T_(0,0)*V_(2,0) + T_(0,1)*V_(2,1) + T_(0,2)*V_(2,2))*dL_da + (T_(1,0)*V_(2,0) + T_(1,1)*V_(2,1) + T_(1,2)*V_(2,2))*dL_db;
    real dL_dT10 = 2*(T_(1,0)*V_(0,0) + T_(1,1)*V_(0,1) + T_(1,2)*V_(0,2))*dL_dc + (T_(0,0)*V_(0,0) + T_(0,1)*V_(0,1) + T_(0,2)*V_(0,2))*dL_db;
    real dL_dT11 = 2*(T_(1,0)*V_(1,0) + T_(1,1)*V_(1,1) + T_(1,2)*V_(1,2))*dL_dc + (T_(0,0)*V_(1,0) + T_(0,1)*V_(1,1) + T_(0,2)*V_(1,2))*dL_db;
    real dL_dT12 = 2*(T_(1,0)*V_(2,0) + T_(1,1)*V_(2,1) + T_(1,2)*V_(2,2))*dL_dc + (T_(0,0)*V_(2,0) + T_(0,1)*V_(2,1) + T_(0,2)*V_(2,2))*dL_db;
#undef V_
#undef T_
    real dL_dJ00 = Wm.v[0][0]*dL_dT00 + Wm.v[0][1]*dL_dT01 + Wm.v[0][2]*dL_dT02;
    real dL_dJ02 = Wm.v[2][0]*dL_dT00 + Wm.v[2][1]*dL_dT01 + Wm.v[2][2]*dL_dT02;
    real dL_dJ11 = Wm.v[1][0]*dL_dT10 + Wm.v[1][1]*dL_dT11 + Wm.v[1][2]*dL_dT12;
    real dL_dJ12 = Wm.v[2][0]*dL_dT10 + Wm.v[2][1]*dL_dT11 + Wm.v[2][2]*dL_dT12;
    real tz = 1.f / t.z, tz2 = tz*tz, tz3 = tz2*tz;
    real l3 = l*l*l;
#define N_(c_,r_) dL_dnJ.v[c_][r_]
    real dL_dtx = x_grad_mul * (-h_x*tz2*dL_dJ02 + dL_du*tz
                                 - N_(0,2)*tz2 + N_(2,0)*(1/l - t.x*t.x/l3) + N_(2,1)*(-t.x*t.y/l3) + N_(2,2)*(-t.x*t.z/l3)
                                 + (c0x*plane.x + c0y*plane.y + c2x)/nl
                                 + dL_dl*t.x/l);
    real dL_dty = y_grad_mul * (-h_y*tz2*dL_dJ12 + dL_dv*tz
                                 - N_(1,2)*tz2 + N_(2,0)*(-t.x*t.y/l3) + N_(2,1)*(1/l - t.y*t.y/l3) + N_(2,2)*(-t.y*t.z/l3)
                                 + (c1x*plane.x + c1y*plane.y + c2y)/nl
                                 + dL_dl*t.y/l);
    real dL_dtz = -h_x*tz2*dL_dJ00 - h_y*tz2*dL_dJ11 + (2*h_x*t.x)*tz3*dL_dJ02 + (2*h_y*t.y)*tz3*dL_dJ12
                   - (dL_du*t.x + dL_dv*t.y)*tz2
                   + (N_(0,0) + N_(1,1))*(-tz2) + N_(0,2)*(2*t.x*tz3) + N_(1,2)*(2*t.y*tz3)
                   + (N_(2,0)*t.x + N_(2,1)*t.y)*(-t.z/l3) + N_(2,2)*(1/l - t.z*t.z/l3)
                   + (c0x*(-(v2 + 1)) + c0y*uv + c1x*uv + c1y*(-(u2 + 1)) + c2x*plane.x + c2y*plane.y)/nl
                   + dL_dl*t.z/l;
#undef N_
    f3 dm = xformvec4x3T(f3_mk(dL_dtx, dL_dty, dL_dtz), view);
    dL_dmeans[3*idx] = dm.x; dL_dmeans[3*idx+1] = dm.y; dL_dmeans[3*idx+2] = dm.z;
}

/* backward.cu:560-628 (preprocessCUDA backward) for one Gaussian */
static void preprocess_bwd(int idx, int D, int M, const real* means, const int* radii, const real* shs,
                           const uint8_t* clamped, const real* scales, const real* rotations, real scale_modifier,
                           const real* view, const real* proj, const real* campos, const real* dL_dmean2D,
                           const real* dL_dview_points, real* dL_dmeans, const real* dL_dcolor, const real* dL_dts,
                           const real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot)
{
    if (!(radii[idx] > 0)) return;
    f3 m = f3_mk(means[3*idx], means[3*idx+1], means[3*idx+2]);
    real mh[4]; xform4x4(m, proj, mh);
    real m_w = 1.0f / (mh[3] + 0.0000001f);
    real mul1 = (proj[0]*m.x + proj[4]*m.y + proj[8]*m.z + proj[12]) * m_w * m_w;
    real mul2 = (proj[1]*m.x + proj[5]*m.y + proj[9]*m.z + proj[13]) * m_w * m_w;
    real gx = dL_dmean2D[3*idx], gy = dL_dmean2D[3*idx+1];
    f3 d1;
    d1.x = (proj[0]*m_w - proj[3]*mul1)*gx + (proj[1]*m_w - proj[3]*mul2)*gy;
    d1.y = (proj[4]*m_w - proj[7]*mul1)*gx + (proj[5]*m_w - proj[7]*mul2)*gy;
    d1.z = (proj[8]*m_w - proj[11]*mul1)*gx + (proj[9]*m_w - proj[11]*mul2)*gy;
    f3 mv = xform4x3(m, view);
    real t = SQRT(mv.x*mv.x + mv.y*mv.y + mv.z*mv.z);
    real dL_dt = dL_dts[idx];
    f3 dvp = f3_mk(dL_dview_points[3*idx], dL_dview_points[3*idx+1], dL_dview_points[3*idx+2]);
    f3 d2 = xformvec4x3T(f3_mk(dvp.x + mv.x/t*dL_dt, dvp.y + mv.y/t*dL_dt, dvp.z + mv.z/t*dL_dt), view);
    dL_dmeans[3*idx]   += d1.x + d2.x;
    dL_dmeans[3*idx+1] += d1.y + d2.y;
    dL_dmeans[3*idx+2] += d1.z + d2.z;
    if (shs) sh_backward(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
    if (scales) cov3d_bwd(idx, scales + 3*(size_t)idx, scale_modifier, rotations + 4*(size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
}

/*
 * Backward: rasterize_points.cu:135-246 + rasterizer_impl.cu:429-571.
 * All dL_d* outputs must be zero-filled by the caller (rasterize_points.cu:180-193).
 * dbg_* (optional, may be NULL) receive the intermediate per-Gaussian sums of the blend backward:
 *   dbg_view_points[P,3], dbg_ts[P], dbg_camera_planes[P,6], dbg_ray_planes[P,2], dbg_normals[P,3], dbg_conic[P,4].
 */
void gsor_backward(const gsor_state* s, const real* bg, const real* means3D, const real* shs,
                   const real* colors_precomp, const real* alphas, const real* scales, const real* rotations,
                   const real* cov3D_precomp, const real* viewmatrix, const real* projmatrix, const real* campos,
                   const real* normalmap,
                   const real* dL_dpix, const real* dL_dpix_coord, const real* dL_dpix_mcoord,
                   const real* dL_dpix_depth, const real* dL_dpix_mdepth, const real* dL_dalphas,
                   const real* dL_dpix_normal,
                   real* dL_dmean2D, real* dL_dcolor, real* dL_dopacity, real* dL_dmean3D, real* dL_dcov3D,
                   real* dL_dsh, real* dL_dscale, real* dL_drot,
                   real* dbg_view_points, real* dbg_ts, real* dbg_camera_planes, real* dbg_ray_planes,
                   real* dbg_normals, real* dbg_conic)
{
    const int P = s->P;
    gacc_t A;
    /* one block of 26 P doubles: mean2D 3 | conic 4 | opacity 1 | colors 3 | view_points 3 | ts 1 | camera_planes 6 | ray_planes 2 | normals 3 */
    gsor_state* ms = (gsor_state*)s;
    const int reuse = (g_flags & 8) && ms->acc_cache && ms->acc_cache_flags == (g_flags & 2);
    double* blk = reuse ? ms->acc_cache : xcalloc((size_t)P*26, 8);
    A.mean2D = blk; A.conic = A.mean2D + (size_t)P*3; A.opacity = A.conic + (size_t)P*4; A.colors = A.opacity + (size_t)P;
    A.view_points = A.colors + (size_t)P*3; A.ts = A.view_points + (size_t)P*3; A.camera_planes = A.ts + (size_t)P;
    A.ray_planes = A.camera_planes + (size_t)P*6; A.normals = A.ray_planes + (size_t)P*2;
    int COORD = s->require_coord, DEPTH = s->require_depth, NORMAL = (s->require_coord || s->require_depth);
    const real* color_ptr = colors_precomp ? colors_precomp : s->rgb;
    if (!reuse)
    for (int ty = 0; ty < s->gy; ty++) for (int tx = 0; tx < s->gx; tx++) {
        uint32_t r0 = s->ranges[2*(ty*s->gx + tx)], r1 = s->ranges[2*(ty*s->gx + tx) + 1];
        if (r1 <= r0) continue;
        for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
            int px = tx*TILE + lx, py = ty*TILE + ly;
            if (px >= s->W || py >= s->H) continue;
            render_pixel_bwd(s, COORD, DEPTH, NORMAL, r0, r1, px, py, bg, color_ptr, alphas, normalmap, dL_dpix,
                             dL_dpix_coord, dL_dpix_mcoord, dL_dpix_depth, dL_dpix_mdepth, dL_dalphas, dL_dpix_normal, &A);
        }
    }
    real* f_view_points = xcalloc((size_t)P*3, sizeof(real)); real* f_ts = xcalloc(P, sizeof(real)); real* f_cp = xcalloc((size_t)P*6, sizeof(real));
    real* f_rp = xcalloc((size_t)P*2, sizeof(real)); real* f_nrm = xcalloc((size_t)P*3, sizeof(real)); real* f_conic = xcalloc((size_t)P*4, sizeof(real));
    for (size_t i = 0; i < (size_t)P*3; i++) { dL_dmean2D[i] = (real)(A.mean2D[i] * sum_jitter(i, 1)); dL_dcolor[i] = (real)(A.colors[i] * sum_jitter(i, 2));
                                               f_view_points[i] = (real)(A.view_points[i] * sum_jitter(i, 3)); f_nrm[i] = (real)(A.normals[i] * sum_jitter(i, 4)); }
    for (size_t i = 0; i < (size_t)P*4; i++) f_conic[i] = (real)(A.conic[i] * sum_jitter(i, 5));
    for (size_t i = 0; i < (size_t)P; i++) { dL_dopacity[i] = (real)(A.opacity[i] * sum_jitter(i, 6)); f_ts[i] = (real)(A.ts[i] * sum_jitter(i, 7)); }
    for (size_t i = 0; i < (size_t)P*6; i++) f_cp[i] = (real)(A.camera_planes[i] * sum_jitter(i, 8));
    for (size_t i = 0; i < (size_t)P*2; i++) f_rp[i] = (real)(A.ray_planes[i] * sum_jitter(i, 9));
    if (g_flags & 8) { if (!reuse) { free(ms->acc_cache); ms->acc_cache = blk; ms->acc_cache_flags = g_flags & 2; } }
    else if (!reuse) free(blk);
    if (dbg_view_points) memcpy(dbg_view_points, f_view_points, (size_t)P*3*sizeof(real));
    if (dbg_ts) memcpy(dbg_ts, f_ts, (size_t)P*sizeof(real));
    if (dbg_camera_planes) memcpy(dbg_camera_planes, f_cp, (size_t)P*6*sizeof(real));
    if (dbg_ray_planes) memcpy(dbg_ray_planes, f_rp, (size_t)P*2*sizeof(real));
    if (dbg_normals) memcpy(dbg_normals, f_nrm, (size_t)P*3*sizeof(real));
    if (dbg_conic) memcpy(dbg_conic, f_conic, (size_t)P*4*sizeof(real));

    const real* cov3D_ptr = cov3D_precomp ? cov3D_precomp : s->cov3D;
    for (int i = 0; i < P; i++)
        cov2d_bwd(i, means3D, s->radii, cov3D_ptr, s->fx, s->fy, s->tan_fovx, s->tan_fovy, s->kernel_size, viewmatrix,
                  f_conic, f_cp, f_rp, f_nrm, dL_dmean3D, dL_dcov3D, /* conic_opacity := dL_dconic */ f_conic, dL_dopacity);
    for (int i = 0; i < P; i++)
        preprocess_bwd(i, s->D, s->M, means3D, s->radii, shs, s->clamped, scales, rotations, s->scale_modifier, viewmatrix,
                       projmatrix, campos, dL_dmean2D, f_view_points, dL_dmean3D, dL_dcolor, f_ts, dL_dcov3D, dL_dsh,
                       dL_dscale, dL_drot);
    free(f_view_points); free(f_ts); free(f_cp); free(f_rp); free(f_nrm); free(f_conic);
}

/* rasterizer_impl.cu:54-66 / auxiliary.h:155-180: markVisible */
void gsor_mark_visible(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        f3 pv = xform4x3(f3_mk(means3D[3*i], means3D[3*i+1], means3D[3*i+2]), viewmatrix);
        present[i] = pv.z <= 0.2f ? 0 : 1;
    }
}

/* exposed for unit tests of the pieces */
int gsor_eig_sym3(const real S_rowmajor[9], real val[3], real vec_cols[9])
{
    m3 S, V; for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) S.v[c][r] = S_rowmajor[3*r + c];
    int n = eig_sym3(&S, val, &V);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) vec_cols[3*c + r] = V.v[c][r];
    return n;
}
void gsor_sh_to_rgb(int deg, int M, const real mean[3], const real campos[3], const real* sh, real rgb[3], uint8_t clamped[3])
{
    f3 c = sh_to_rgb(0, deg, M, mean, campos, sh, clamped);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
