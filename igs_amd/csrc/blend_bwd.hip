// blend_bwd.hip -- backward of the tile blend for gfx950 (wave64).
// Replaces BACKWARD::render / renderCUDA (DGR/cuda_rasterizer/backward.cu:631-1016, dispatch 1101-1163).
//
// Same tiling as the forward (workgroup = tile, wave = 8x8 quad, lane = pixel), traversed back to front.
// What is different from the reference, by design:
//  * The reference keeps one suffix recurrence per output channel (11 of them) and issues up to 25 global float
//    atomics per (pixel, splat) pair.  Every recurrence is linear in the channel value, and the upstream gradient
//    of a channel is constant per pixel, so they collapse into ONE scalar recurrence on
//        D = sum_ch value_ch * dL/dchannel_ch        (S <- last_alpha * D_prev + (1 - last_alpha) * S),
//    which leaves 3 registers of per-pixel state instead of 22.
//  * Per-splat gradients are accumulated as 25 RAW MOMENTS of the per-pair weights (sums of w, w*dx, w*dy, q*dx^2, ...);
//    everything that is a per-splat linear combination of those (conic / mean2D / camera-plane / ray-plane gradients,
//    the 1/focal factors) is applied once per Gaussian in geom_bwd.hip instead of once per pair.
//  * The 64 lanes of a wave are summed with DPP row operations; one 25-lane, 100-byte contiguous global atomic per
//    (wave, splat) replaces 25 x 64 scalar atomics.
// T is recovered exactly like the reference does (T_final = 1 - out_alpha, T <- T / (1 - alpha), backward.cu:706,857).
#include "common.h"

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
// sum over the 64 lanes; the total is valid in lanes 48..63 (rocPRIM-style gfx9 DPP sequence)
__device__ __forceinline__ float wave_sum_hi(float v) {
    v += dpp_mov<0xB1, 0xf, 0xf>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, 0xf>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, 0xf>(v);    // row_half_mirror
    v += dpp_mov<0x140, 0xf, 0xf>(v);    // row_mirror
    v += dpp_mov<0x142, 0xa, 0xf>(v);    // row_bcast15 -> rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf>(v);    // row_bcast31 -> rows 2,3
    return v;
}

// gacc slots (raw moments), see geom_bwd.hip for how they are combined:
//  0..2  sum w*dL/dpix_ch            3..5  Sv = sum dLc_ch          6..8 Sx = sum dLc_ch*dx     9..11 Sy = sum dLc_ch*dy
//  12    St = sum dLt   13 Stx   14 Sty   15..17 sum w*dL/dnormal_ch
//  18 Q0 = sum q  19 Qx  20 Qy  21 Qxx  22 Qxy  23 Qyy  (q = dL/dG * G)   24 Z = abs-sum for dL_dmean2D.z
template <bool COORD, bool DEPTH, bool NORMAL>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(const BlendBwdArgs a)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    constexpr int NQ = GEO ? 6 : 3;
    __shared__ float4 chunk[CHUNK * NQ];
    __shared__ uint32_t chunk_id[CHUNK];
    __shared__ int wave_max[4];

    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t px = tx * TILE + (wid & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)a.W * py + px;

    const uint2 range = ((const uint2*)a.ranges)[tile];
    int last_contributor = inside ? (int)a.n_contrib[pix] : 0;
    const uint32_t max_contributor = inside ? a.n_contrib[pix + HW] : 0u;

    // ---- per-pixel upstream gradients (backward.cu:732-781); zero for pixels nothing was blended into, whose
    //      normalisations would otherwise be 0/0 (the reference never consumes those values either)
    float gp0 = 0, gp1 = 0, gp2 = 0, g_alpha = 0, T_final = 0, bg_dot = 0;
    float gc0 = 0, gc1 = 0, gc2 = 0, gm0 = 0, gm1 = 0, gm2 = 0, g_t = 0, g_mt = 0, gn0 = 0, gn1 = 0, gn2 = 0;
    if (last_contributor > 0) {
        const float w_final = a.alphas[pix];
        T_final = 1.0f - w_final;
        gp0 = a.dL_dpix[pix]; gp1 = a.dL_dpix[HW + pix]; gp2 = a.dL_dpix[2 * HW + pix];
        g_alpha = a.dL_dalpha[pix];
        bg_dot = a.bg[0] * gp0 + a.bg[1] * gp1 + a.bg[2] * gp2;
        if constexpr (GEO) {
            const float ww = w_final * w_final;
            const float pnx = (pixfx - a.W / 2.f) / a.fx, pny = (pixfy - a.H / 2.f) / a.fy;
            const float ln = sqrtf(pnx * pnx + pny * pny + 1);
            if constexpr (COORD) {
                const float w0 = a.dL_dcoord[pix], w1 = a.dL_dcoord[HW + pix], w2 = a.dL_dcoord[2 * HW + pix];
                g_alpha -= w0 * a.accum_coord[pix] / ww;
                g_alpha -= w1 * a.accum_coord[HW + pix] / ww;
                g_alpha -= w2 * a.accum_coord[2 * HW + pix] / ww;
                gc0 = w0 / w_final; gc1 = w1 / w_final; gc2 = w2 / w_final;
                gm0 = a.dL_dmcoord[pix]; gm1 = a.dL_dmcoord[HW + pix]; gm2 = a.dL_dmcoord[2 * HW + pix];
            }
            if constexpr (DEPTH) {
                const float wd = a.dL_ddepth[pix];
                g_alpha -= wd * a.accum_depth[pix] / ww;
                g_t = wd / w_final / ln;
                g_mt = a.dL_dmdepth[pix] / ln;
            }
            if constexpr (NORMAL) {
                const float d0 = a.dL_dnormal[pix], d1 = a.dL_dnormal[HW + pix], d2 = a.dL_dnormal[2 * HW + pix];
                const float n0 = a.normalmap[pix], n1 = a.normalmap[HW + pix], n2 = a.normalmap[2 * HW + pix];
                const float nlen = a.normal_length[pix];
                if (nlen < 1.0E-12F) { gn0 = d0 / 1.0E-12F; gn1 = d1 / 1.0E-12F; gn2 = d2 / 1.0E-12F; }
                else {
                    const float dt = d0 * n0 + d1 * n1 + d2 * n2;
                    gn0 = (d0 - dt * n0) / nlen; gn1 = (d1 - dt * n1) / nlen; gn2 = (d2 - dt * n2) / nlen;
                }
            }
        }
    }

    // nothing behind the deepest last_contributor of the tile is ever touched: start there
    {
        int m = last_contributor;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if (lane == 0) wave_max[wid] = m;
    }
    __syncthreads();
    const int n = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));   // elements [0, n) of the range
    const int rounds = (n + CHUNK - 1) / CHUNK;

    float T = T_final, S = 0.f, Dprev = 0.f, last_alpha = 0.f;
    const float halfW = 0.5f * a.W, halfH = 0.5f * a.H;

    for (int i = 0; i < rounds; i++) {
        __syncthreads();
        const int progress = i * CHUNK + (int)tid;          // position counted from the back of [0, n)
        if (progress < n) {
            const uint32_t id = a.point_list[range.x + (uint32_t)(n - 1 - progress)];
            const float4* src = (const float4*)(a.rec + (size_t)id * REC_F);
            float4 q0 = src[0], q1 = src[1], q2 = src[2];
            if (a.colors_precomp) {
                q1.z = a.colors_precomp[3 * (size_t)id]; q1.w = a.colors_precomp[3 * (size_t)id + 1];
                q2.x = a.colors_precomp[3 * (size_t)id + 2];
            }
            chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = q1; chunk[tid * NQ + 2] = q2;
            if constexpr (GEO) { chunk[tid * NQ + 3] = src[3]; chunk[tid * NQ + 4] = src[4]; chunk[tid * NQ + 5] = src[5]; }
            chunk_id[tid] = id;
        }
        __syncthreads();
        const int cnt = min(CHUNK, n - i * CHUNK);
        for (int j = 0; j < cnt; j++) {
            const int eidx = n - 1 - (i * CHUNK + j);       // 0-based position in the tile's list = the reference's `contributor`
            const float4 q0 = chunk[j * NQ + 0];
            const float4 q1 = chunk[j * NQ + 1];
            const float dx = q0.x - pixfx, dy = q0.y - pixfy;
            const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, q1.y * G);
            const bool valid = (eidx < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(valid) == 0ull) continue;

            const float4 q2 = chunk[j * NQ + 2];
            const float one_m = 1.f - alpha;
            T = valid ? T / one_m : T;
            const float w = valid ? alpha * T : 0.f;
            const bool is_med = valid && ((uint32_t)(eidx + 1) == max_contributor);

            float D = q1.z * gp0 + q1.w * gp1 + q2.x * gp2 + g_alpha;
            float dLc0 = 0, dLc1 = 0, dLc2 = 0, dLt = 0;
            float4 q3, q4, q5;
            if constexpr (GEO) { q3 = chunk[j * NQ + 3]; q5 = chunk[j * NQ + 5]; }
            if constexpr (COORD) {
                q4 = chunk[j * NQ + 4];
                const float c0 = q3.x + q4.x * dx + q4.y * dy;
                const float c1 = q3.y + q4.z * dx + q4.w * dy;
                const float c2 = q3.z + q5.x * dx + q5.y * dy;
                D += c0 * gc0 + c1 * gc1 + c2 * gc2;
                dLc0 = w * gc0 + (is_med ? gm0 : 0.f);
                dLc1 = w * gc1 + (is_med ? gm1 : 0.f);
                dLc2 = w * gc2 + (is_med ? gm2 : 0.f);
            }
            if constexpr (DEPTH) {
                const float t = q2.y + (q2.z * dx + q2.w * dy);
                D += t * g_t;
                dLt = w * g_t + (is_med ? g_mt : 0.f);
            }
            if constexpr (NORMAL) D += q3.w * gn0 + q5.z * gn1 + q5.w * gn2;

            const float Snew = last_alpha * Dprev + (1.f - last_alpha) * S;
            float dL_dopa = (D - Snew) * T + (-T_final / one_m) * bg_dot;
            S = valid ? Snew : S;
            Dprev = valid ? D : Dprev;
            last_alpha = valid ? alpha : last_alpha;
            const float dL_dG = valid ? q1.y * dL_dopa : 0.f;
            const float q = dL_dG * G;
            const float qdx = q * dx, qdy = q * dy;
            const float gxa = q0.z * qdx + q0.w * qdy;      // -dL/d(delx) of the Gaussian term
            const float gya = q1.x * qdy + q0.w * qdx;

            float v[GA_USED];
            v[0] = w * gp0; v[1] = w * gp1; v[2] = w * gp2;
            v[3] = dLc0; v[4] = dLc1; v[5] = dLc2;
            v[6] = dLc0 * dx; v[7] = dLc1 * dx; v[8] = dLc2 * dx;
            v[9] = dLc0 * dy; v[10] = dLc1 * dy; v[11] = dLc2 * dy;
            v[12] = dLt; v[13] = dLt * dx; v[14] = dLt * dy;
            v[15] = w * gn0; v[16] = w * gn1; v[17] = w * gn2;
            v[18] = q; v[19] = qdx; v[20] = qdy; v[21] = qdx * dx; v[22] = qdx * dy; v[23] = qdy * dy;
            v[24] = fabsf(gxa * halfW) + fabsf(gya * halfH);

            float out = 0.f;
#pragma unroll
            for (int k = 0; k < GA_USED; k++) {
                const bool live = (k < 3) || (k >= 18) || (COORD && k >= 3 && k < 12) || (DEPTH && k >= 12 && k < 15)
                                  || (NORMAL && k >= 15 && k < 18);
                if (live) {
                    const float tot = wave_sum_hi(v[k]);
                    const int s = __builtin_amdgcn_readlane(__float_as_int(tot), 63);
                    // lane k of `out` <- total (immediate lane select: no SGPR lane-select hazard)
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(out) : "s"(s), "n"(k));
                }
            }
            if (lane < GA_USED) atomicAdd(&a.gacc[(size_t)chunk_id[j] * GACC_F + lane], out);
        }
    }
}

hipError_t launch_blend_bwd(hipStream_t s, const BlendBwdArgs& a, bool coord, bool depth)
{
    const dim3 grid(a.gx * a.gy), block(256);
    if (coord && depth) hipLaunchKernelGGL((blend_bwd_kernel<true, true, true>), grid, block, 0, s, a);
    else if (coord) hipLaunchKernelGGL((blend_bwd_kernel<true, false, true>), grid, block, 0, s, a);
    else if (depth) hipLaunchKernelGGL((blend_bwd_kernel<false, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((blend_bwd_kernel<false, false, false>), grid, block, 0, s, a);
    return hipGetLastError();
}
