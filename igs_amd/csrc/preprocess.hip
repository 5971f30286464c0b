// preprocess.hip -- per-Gaussian forward stage (one thread per Gaussian), gfx950.
// Replaces FORWARD::preprocess / preprocessCUDA (DGR/cuda_rasterizer/forward.cu:307-423), checkFrustum
// (rasterizer_impl.cu:54-66) and the tiles_touched reduction that the reference does with cub::DeviceScan.
#include "geom_math.h"

__constant__ float SH_C0 = 0.28209479177387814f;
__constant__ float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f };
__constant__ float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

// SH -> RGB (forward.cu:23-74).  sh points at this Gaussian's [M][3] coefficients.
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float* __restrict__ sh, float3 dir, uint32_t& clamped) {
    const float3* c = (const float3*)sh;
    auto L = [&](int k) { return make_float3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
    float3 res = L(0) * SH_C0;
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        res = res - L(1) * (SH_C1 * y) + L(2) * (SH_C1 * z) - L(3) * (SH_C1 * x);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + L(4) * (SH_C2[0] * xy) + L(5) * (SH_C2[1] * yz) + L(6) * (SH_C2[2] * (2.0f * zz - xx - yy))
                  + L(7) * (SH_C2[3] * xz) + L(8) * (SH_C2[4] * (xx - yy));
            if (deg > 2) {
                res = res + L(9) * (SH_C3[0] * y * (3.0f * xx - yy)) + L(10) * (SH_C3[1] * xy * z)
                      + L(11) * (SH_C3[2] * y * (4.0f * zz - xx - yy)) + L(12) * (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy))
                      + L(13) * (SH_C3[4] * x * (4.0f * zz - xx - yy)) + L(14) * (SH_C3[5] * z * (xx - yy))
                      + L(15) * (SH_C3[6] * x * (xx - 3.0f * yy));
            }
        }
    }
    (void)c;
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamped = (res.x < 0 ? 1u : 0u) | (res.y < 0 ? 2u : 0u) | (res.z < 0 ? 4u : 0u);
    return make_float3(fmaxf(res.x, 0.f), fmaxf(res.y, 0.f), fmaxf(res.z, 0.f));
}

struct PreArgs { FwdParams p; };

// The kernel is latency-bound: a view sees ~40 % of the Gaussians, in Morton order whole waves are culled, and the ~1250 waves that do
// have work each walk one serial chain (loads -> EWA -> plane fit -> SH -> record -> binning round trips) on a SIMD they have nearly to
// themselves.  So the chain is cut in two and run by two workgroups per 256 Gaussians (round 3):
//   ROLE_BIN    : cull tests, radius / rectangle, radii[] / tiles[], instance count and the slab binning; zero-fills the accumulators
//   ROLE_RECORD : the same cull tests (same arithmetic, so the same verdict), then plane fit, SH -> RGB and the 128-byte record
// The EWA projection is computed twice; the machine has the room (preprocess 43.5 -> see DESIGN.md section 5).  ROLE_BOTH is the unsplit
// kernel (-DPRE_NO_SPLIT).
enum { ROLE_BOTH = 0, ROLE_BIN = 1, ROLE_RECORD = 2 };

#ifdef PRE_TIMELINE
// debug build only (tools/debug/preprocess_timeline.py): per wave and role, the 100 MHz clock at the marks of the chain
#define PRE_TL_MARKS 10
#define PRE_TL_WAVES 16384
__device__ unsigned long long g_pre_tl[PRE_TL_WAVES * PRE_TL_MARKS];
extern "C" int igs_debug_preprocess_timeline(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pre_tl), (size_t)n * 8);
}
#define TL(k) do { const unsigned long long m_ = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)m_) - 1u) { \
    const unsigned w_ = (blockIdx.x * 4u + (threadIdx.x >> 6)); if (w_ < PRE_TL_WAVES) g_pre_tl[w_ * PRE_TL_MARKS + (k)] = wall_clock64(); } } while (0)
#else
#define TL(k) do { } while (0)
#endif

template <int ROLE>
__device__ __forceinline__ void
preprocess_body(const FwdParams& p, const uint32_t blk, float* __restrict__ rec, uint32_t* __restrict__ tiles, uint32_t* __restrict__ depth_keys,
                uint32_t* __restrict__ ident, int* __restrict__ radii, uint32_t* __restrict__ counters,
                uint32_t* __restrict__ hist0, uint32_t per_block, uint32_t* __restrict__ tile_count,
                uint64_t* __restrict__ pairs, uint32_t slab)
{
    constexpr bool BIN = ROLE != ROLE_RECORD, RECORD = ROLE != ROLE_BIN;
    const int idx = (int)blk * 256 + threadIdx.x;
    TL(0);
    uint32_t my_tiles = 0;
    int rect_x0 = 0, rect_y0 = 0, rect_w = 1;
    uint32_t my_dkey = 0;
    if (BIN && p.zero_stats && blk == 0 && threadIdx.x < 4) p.zero_stats[threadIdx.x] = 0u;      // (nothing reads them before the tile sort)
    if (idx < p.P) {
        int radius = 0;
        uint32_t dkey = 0xFFFFFFFFu;      // culled Gaussians sort to the end (they own no instances anyway)
        float4* R4 = (float4*)(rec + (size_t)idx * REC_F);
        const float3 p_orig = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
        const float3 p_view = xform4x3(p_orig, p.view);
        TL(1);
        do {
            if (p_view.z <= 0.2f) {                       // auxiliary.h:170
                if (BIN && p.prefiltered) atomicOr(&counters[1], 1u);
                break;
            }
            const float4 p_hom = xform4x4(p_orig, p.proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float ppx = p_hom.x * p_w, ppy = p_hom.y * p_w;
            float cov3D[6];
            if (p.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov3D[k] = p.cov3D_precomp[6 * (size_t)idx + k];
            } else {
                float3 s = make_float3(p.scales[3 * idx], p.scales[3 * idx + 1], p.scales[3 * idx + 2]);
                float4 q = make_float4(p.rotations[4 * idx], p.rotations[4 * idx + 1], p.rotations[4 * idx + 2], p.rotations[4 * idx + 3]);   // caller arrays: no alignment assumption
                if (p.raw_activations) {
                    s = make_float3(act_exp(s.x), act_exp(s.y), act_exp(s.z));
                    const float inv = act_inv_norm4(q.x, q.y, q.z, q.w);
                    q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
                }
                cov3d_from_scale_rot(s, p.scale_modifier, q, cov3D);
            }
            Cov2DCtx c;
            cov2d_ctx(c, p_orig, cov3D, p.view, p.fx, p.fy, p.tan_fovx, p.tan_fovy, p.kernel_size, RECORD);      // (the binning role skips the eigen-solver)
            if (RECORD && p.plane_cache) plane_cache_store(p.plane_cache + (size_t)idx * PLANE_CACHE_F, c, p.plane_tag);
            TL(2);
            float cp[6] = { 0, 0, 0, 0, 0, 0 }, rp[2] = { 0, 0 };
            float3 nrm = make_float3(0, 0, 0);
            if (RECORD && !c.degenerate) {                 // forward.cu:169-262
                const float3 t = c.t;
                const float u = c.txtz, v = c.tytz, u2 = u * u, v2 = v * v, uv = u * v;
                const float l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
                const float vbn = dot3(c.uvh_mn, c.uvh);
                const float nl = u2 + v2 + 1;
                const float factor_normal = l / nl;
                const float den = fmaxf(vbn, 0.0000001f);
                const float3 q = make_float3(c.uvh_mn.x / den, c.uvh_mn.y / den, c.uvh_mn.z / den);
                const float plane0 = (v2 + 1) * q.x + (-uv) * q.y + (-u) * q.z;
                const float plane1 = (-uv) * q.x + (u2 + 1) * q.y + (-v) * q.z;
                cp[0] = (-(v2 + 1) * t.z + plane0 * t.x) / nl / p.fx; cp[1] = (uv * t.z + plane1 * t.x) / nl / p.fy;
                cp[2] = (uv * t.z + plane0 * t.y) / nl / p.fx;        cp[3] = (-(u2 + 1) * t.z + plane1 * t.y) / nl / p.fy;
                cp[4] = (t.x + plane0 * t.z) / nl / p.fx;             cp[5] = (t.y + plane1 * t.z) / nl / p.fy;
                rp[0] = plane0 * l / nl / p.fx; rp[1] = plane1 * l / nl / p.fy;
                const float3 rn = make_float3(-plane0 * factor_normal, -plane1 * factor_normal, -1.f);
                // nJ (std rows): (1/z, 0, x/l), (0, 1/z, y/l), (-x/z^2, -y/z^2, z/l)
                const float3 cn = make_float3((1 / t.z) * rn.x + 0.0f * rn.y + (t.x / l) * rn.z,
                                              0.0f * rn.x + (1 / t.z) * rn.y + (t.y / l) * rn.z,
                                              (-(t.x) / (t.z * t.z)) * rn.x + (-(t.y) / (t.z * t.z)) * rn.y + (t.z / l) * rn.z);
                const float inv = 1.0f / sqrtf(dot3(cn, cn));
                nrm = cn * inv;
            }
            const float ts = sqrtf(p_view.x * p_view.x + p_view.y * p_view.y + p_view.z * p_view.z);
            const float coef = c.coef_zero ? 0.0f : c.coef;
            // geometry that the reference stores even for Gaussians it drops later is irrelevant: nothing reads it
            const float cx = c.cov2[0], cy = c.cov2[1], cz = c.cov2[2];
            const float det = cx * cz - cy * cy;
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float3 conic = make_float3(cz * det_inv, -cy * det_inv, cx * det_inv);
            const float mid = 0.5f * (cx + cz);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix (auxiliary.h:57-60) is evaluated in double
            const float pix = (float)((((double)ppx + 1.0) * (double)p.W - 1.0) * 0.5);
            const float piy = (float)((((double)ppy + 1.0) * (double)p.H - 1.0) * 0.5);
            int x0, y0, x1, y1;
            get_rect(pix, piy, (int)my_radius, p.gx, p.gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) == 0) break;
            TL(3);
            float3 rgb = make_float3(0, 0, 0);
            uint32_t clamped = 0;
            if (!RECORD) {
            } else if (p.colors_precomp == nullptr) {
                float3 dir = p_orig - make_float3(p.campos[0], p.campos[1], p.campos[2]);
                const float len = sqrtf(dot3(dir, dir));
                dir = make_float3(dir.x / len, dir.y / len, dir.z / len);
                rgb = sh_to_rgb(p.D, p.shs + (size_t)idx * p.M * 3, dir, clamped);
            } else {
                rgb = make_float3(p.colors_precomp[3 * idx], p.colors_precomp[3 * idx + 1], p.colors_precomp[3 * idx + 2]);
            }
            radius = (int)my_radius;
            my_tiles = (uint32_t)((y1 - y0) * (x1 - x0));
            rect_x0 = x0; rect_y0 = y0; rect_w = x1 - x0;
            dkey = __float_as_uint(p_view.z);
            TL(4);
            my_dkey = dkey;
            if constexpr (RECORD) {
            R4[0] = make_float4(pix, piy, conic.x, conic.y);
            const float opacity = p.raw_activations ? act_sigmoid(p.opacities[idx]) : p.opacities[idx];
            R4[1] = make_float4(conic.z, opacity * coef, rgb.x, rgb.y);
            R4[2] = make_float4(rgb.z, ts, rp[0], rp[1]);
            R4[3] = make_float4(p_view.x, p_view.y, p_view.z, nrm.x);
            R4[4] = make_float4(cp[0], cp[1], cp[2], cp[3]);
            R4[5] = make_float4(cp[4], cp[5], nrm.y, nrm.z);
            R4[6] = make_float4(cov3D[0], cov3D[1], cov3D[2], cov3D[3]);
            R4[7] = make_float4(cov3D[4], cov3D[5], __uint_as_float(clamped), __uint_as_float(dkey));
            }
        } while (0);
        TL(5);
        if constexpr (BIN) {
        radii[idx] = radius;
        tiles[idx] = my_tiles;
        }
        if (BIN && hist0) {                                          // radix binning: keys + first-pass histogram of the depth sort
            depth_keys[idx] = dkey;
            ident[idx] = (uint32_t)idx;
            atomicAdd(&hist0[((uint32_t)idx / per_block) * 256u + (dkey & 255u)], 1u);
        }
    }
    if (BIN && p.zero_gacc) {
        // refine step: this kernel is latency-bound and leaves the memory pipes idle -- zero-fill the backward's accumulator
        // line of this Gaussian (and the loss shards) here instead of in a 25 MB fill of its own.  AFTER the wave's loads: issued at
        // the top, the stores sat in front of them in the memory pipeline (position loads back after 4.8 us instead of 1.0)
        if (idx < p.P) {
            float4* Z4 = (float4*)(p.zero_gacc + (size_t)idx * p.zero_gacc_stride);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < GACC_F / 4; k++)
                if (4 * k < p.zero_gacc_stride) Z4[k] = z;
        }
        if (blk == 0 && threadIdx.x < 64) {
            p.zero_loss[16 * threadIdx.x] = 0.f;
            if (p.zero_loss2) p.zero_loss2[16 * threadIdx.x] = 0.f;
        }
    }
    if constexpr (BIN) {
    __shared__ uint32_t s_incl[256], s_dkey[256];
    __shared__ int s_x0[256], s_y0[256], s_w[256];
    __shared__ uint32_t s_ws[4];
    // total instance count: wave scan, one atomic per wave that has something to add (sharded, summed later)
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t v = my_tiles;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if (lane >= (uint32_t)off) v += t; }
    // 64 counter shards, one cache line apart: same-address atomics serialise at ~12 ns each (3128 waves -> 37 us on one word)
    if (lane == 63 && v) atomicAdd(&counters[COUNTER_SHARD_STRIDE * (1 + (blk & (COUNTER_SHARDS - 1)))], v);
    if (pairs == nullptr) return;                             // radix binning: instances are emitted after the depth sort
    TL(6);

    // ---- slab binning: the workgroup drops its instances into the tile slabs cooperatively (load-balanced over the 256
    //      threads whatever the individual footprints are): slot = tile_count[t]++ ; pairs[t*slab + slot] = depth<<32 | id
    if (lane == 63) s_ws[wid] = v;
    s_dkey[threadIdx.x] = my_dkey; s_x0[threadIdx.x] = rect_x0; s_y0[threadIdx.x] = rect_y0; s_w[threadIdx.x] = rect_w;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wid; w++) woff += s_ws[w];
    const uint32_t total = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
    s_incl[threadIdx.x] = v + woff;
    __syncthreads();
    // How coherent is this workgroup?  Bounding box (in tiles) of everything its Gaussians touch.
    __shared__ int s_bb[4][4];
    {
        int bx0 = my_tiles ? rect_x0 : 0x7FFFFFFF, by0 = my_tiles ? rect_y0 : 0x7FFFFFFF;
        int bx1 = my_tiles ? rect_x0 + rect_w : -1, by1 = my_tiles ? rect_y0 + (int)(my_tiles / (uint32_t)rect_w) : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
            bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
        }
        if (lane == 0) { s_bb[wid][0] = bx0; s_bb[wid][1] = by0; s_bb[wid][2] = bx1; s_bb[wid][3] = by1; }
    }
    __syncthreads();
    const int bbx0 = min(min(s_bb[0][0], s_bb[1][0]), min(s_bb[2][0], s_bb[3][0])), bby0 = min(min(s_bb[0][1], s_bb[1][1]), min(s_bb[2][1], s_bb[3][1]));
    const int bbx1 = max(max(s_bb[0][2], s_bb[1][2]), max(s_bb[2][2], s_bb[3][2])), bby1 = max(max(s_bb[0][3], s_bb[1][3]), max(s_bb[2][3], s_bb[3][3]));
    const bool coherent = total > 0 && (long long)(bbx1 - bbx0) * (long long)(bby1 - bby0) <= 1024;       // workgroup-uniform
    TL(7);
    if (!coherent) {
    // ---- scattered workgroup: one returning global atomic per instance
    for (uint32_t k0 = 0; k0 < total; k0 += 512) {
        // two instances per thread and trip, so that two atomic round trips are in flight
        uint32_t tt[2], key_lo[2], key_hi[2]; bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t k = k0 + u * 256 + threadIdx.x;
            ok[u] = k < total;
            uint32_t lo = 0, hi = 255;
            if (ok[u]) {
#pragma unroll
                for (int it = 0; it < 8; it++) { const uint32_t mid = (lo + hi) >> 1; if (s_incl[mid] > k) hi = mid; else lo = mid + 1; }
            }
            const uint32_t j = lo;
            const uint32_t local = ok[u] ? k - (j ? s_incl[j - 1] : 0u) : 0u;
            const uint32_t w = (uint32_t)s_w[j];
            tt[u] = (uint32_t)((s_y0[j] + (int)(local / w)) * p.gx + s_x0[j] + (int)(local % w));
            key_hi[u] = s_dkey[j]; key_lo[u] = blk * 256u + j;
        }
        uint32_t slot[2];
#pragma unroll
        for (int u = 0; u < 2; u++) slot[u] = ok[u] ? atomicAdd(&tile_count[tt[u]], 1u) : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (slot[u] < slab) pairs[(size_t)tt[u] * slab + slot[u]] = ((uint64_t)key_hi[u] << 32) | (uint64_t)key_lo[u];
    }
        return;
    }
    // Slot reservation is aggregated per workgroup and tile through an LDS hash table: ONE global atomic per (workgroup, tile)
    // instead of one per instance.  Scattered returning atomics execute at the memory side at ~20 G/s for the whole chip
    // (518k instances = 26 us); when consecutive Gaussians are spatial neighbours (a Morton-sorted store: GaussianParams.
    // spatial_sort) the instances of a workgroup fall on ~5.6x fewer distinct tiles.  On an unsorted store the table costs more
    // than it saves (+22 us), hence the coherence test above.
    constexpr uint32_t HS = 2048, EMPTY = 0xFFFFFFFFu;
    __shared__ uint32_t hkey[HS], hcnt[HS];
    for (uint32_t k0 = 0; k0 < total; k0 += 1024) {
        for (uint32_t e = threadIdx.x; e < HS; e += 256) { hkey[e] = EMPTY; hcnt[e] = 0u; }
        __syncthreads();
        uint32_t tt[4], key_lo[4], key_hi[4], hh[4], rk[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = k0 + u * 256 + threadIdx.x;
            ok[u] = k < total;
            uint32_t lo = 0, hi = 255;
            if (ok[u]) {
#pragma unroll
                for (int it = 0; it < 8; it++) { const uint32_t mid = (lo + hi) >> 1; if (s_incl[mid] > k) hi = mid; else lo = mid + 1; }
            }
            const uint32_t j = lo;
            const uint32_t local = ok[u] ? k - (j ? s_incl[j - 1] : 0u) : 0u;
            const uint32_t w = (uint32_t)s_w[j];
            tt[u] = (uint32_t)((s_y0[j] + (int)(local / w)) * p.gx + s_x0[j] + (int)(local % w));
            key_hi[u] = s_dkey[j]; key_lo[u] = blk * 256u + j;
            hh[u] = 0; rk[u] = 0;
            if (ok[u]) {
                uint32_t h = (tt[u] * 2654435761u) >> 21;                      // 11 bits
                while (true) {
                    const uint32_t prev = atomicCAS(&hkey[h], EMPTY, tt[u]);
                    if (prev == EMPTY || prev == tt[u]) break;
                    h = (h + 1) & (HS - 1);                                    // (at most 1024 keys in 2048 slots: always terminates)
                }
                hh[u] = h;
                rk[u] = atomicAdd(&hcnt[h], 1u);
            }
        }
        __syncthreads();
        {   // one global atomic per occupied entry; the entry then holds the first slot of this workgroup's run in that tile
            uint32_t c[HS / 256], b[HS / 256];
#pragma unroll
            for (int i = 0; i < (int)(HS / 256); i++) {
                const uint32_t e = threadIdx.x + i * 256;
                c[i] = hcnt[e];
                b[i] = c[i] ? atomicAdd(&tile_count[hkey[e]], c[i]) : 0u;
            }
#pragma unroll
            for (int i = 0; i < (int)(HS / 256); i++)
                if (c[i]) hcnt[threadIdx.x + i * 256] = b[i];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (ok[u]) {
                const uint32_t slot = hcnt[hh[u]] + rk[u];
                if (slot < slab) pairs[(size_t)tt[u] * slab + slot] = ((uint64_t)key_hi[u] << 32) | (uint64_t)key_lo[u];
            }
        }
        __syncthreads();
    }
    TL(8);
    }      // BIN
}

// (80 VGPRs = 6 workgroups per CU: 28 of the 1564 start late.  Forcing 7 per CU -- 72 VGPRs, 12 spilled -- changes nothing: 38.0 us)
template <bool SPLIT>
__global__ void __launch_bounds__(256)
preprocess_fwd_kernel(const PreArgs a, float* __restrict__ rec, uint32_t* __restrict__ tiles, uint32_t* __restrict__ depth_keys,
                      uint32_t* __restrict__ ident, int* __restrict__ radii, uint32_t* __restrict__ counters,
                      uint32_t* __restrict__ hist0, uint32_t per_block, uint32_t* __restrict__ tile_count,
                      uint64_t* __restrict__ pairs, uint32_t slab)
{
    if constexpr (!SPLIT) {
        preprocess_body<ROLE_BOTH>(a.p, blockIdx.x, rec, tiles, depth_keys, ident, radii, counters, hist0, per_block, tile_count, pairs, slab);
    } else {
        // neighbouring workgroups take the two roles of the same 256 Gaussians: their parameter lines are fetched once
        const uint32_t blk = blockIdx.x >> 1;
        if (blockIdx.x & 1u) preprocess_body<ROLE_RECORD>(a.p, blk, rec, tiles, depth_keys, ident, radii, counters, hist0, per_block, tile_count, pairs, slab);
        else                 preprocess_body<ROLE_BIN>(a.p, blk, rec, tiles, depth_keys, ident, radii, counters, hist0, per_block, tile_count, pairs, slab);
    }
}

hipError_t launch_preprocess_fwd(hipStream_t s, const FwdParams& p, float* rec, uint32_t* tiles, uint32_t* depth_keys,
                                 uint32_t* ident, int* radii, uint32_t* counters, uint32_t* hist0, uint32_t per_block,
                                 uint32_t* tile_count, uint64_t* pairs, uint32_t slab)
{
    PreArgs a; a.p = p;
    // (staging the SH rows through LDS was tried here and lost: 50 KB/block costs more occupancy than the strided reads cost)
#ifdef PRE_NO_SPLIT
    hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((p.P + 255) / 256), dim3(256), 0, s, a, rec, tiles, depth_keys, ident, radii, counters,
                       hist0, per_block, tile_count, pairs, slab);
#else
    hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3(2 * ((p.P + 255) / 256)), dim3(256), 0, s, a, rec, tiles, depth_keys, ident, radii, counters,
                       hist0, per_block, tile_count, pairs, slab);
#endif
    return hipGetLastError();
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 pv = xform4x3(make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]), view);
    present[idx] = pv.z <= 0.2f ? 0 : 1;
}
hipError_t launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* view, uint8_t* present)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    return hipGetLastError();
}
