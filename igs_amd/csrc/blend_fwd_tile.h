// blend_fwd_tile.h -- the forward blend of ONE 16x16 tile by one 256-thread workgroup (device code shared by the stand-alone forward
// kernel, blend_fwd.hip, and the fused forward + backward tile kernel of the refine step, blend_step.hip).
// Replaces FORWARD::render / renderCUDA (DGR/cuda_rasterizer/forward.cu:428-742); see blend_fwd.hip for the design notes.
#pragma once
#include "blend_common.h"

// 192 splats staged per round by the first three waves: 18 KB of LDS per workgroup instead of 24.5 (256 splats), so that eight
// workgroups fit a CU; together with the 64-VGPR budget of the kernels (8 waves per SIMD instead of 6; two dwords spill outside the row
// loop) the fuller machine hides more of a row's dependency chain: 64.4 -> 61.0 us on the bench scene (same-box A/B, round 2)
#define FWD_CHUNK 192
#define FWD_NSW (FWD_CHUNK / 64)     // staging waves
#define FWD_NLIST 4                  // to-do lists per tile: one per quad (the half-quad experiment of round 4 -- two splats per row -- was parity-
                                     // green and slower; it lives in tools/experiments/fwd_half2.patch)

// what a pixel's lane knows when the forward of its tile is done (the fused kernel hands it straight to the backward)
struct FwdPix {
    float C0, C1, C2;            // blended colour WITHOUT the background term
    float T;                     // final transmittance
    float weight;                // sum of alpha T  (= the alpha image)
    uint32_t last_contributor;   // 1-based list position of the last splat blended into the pixel (n_contrib)
};

// LEAN = the refine step with a colour-only loss (BlendFwdArgs::skip_bwd_state): the geometry branches' backward state (accumulated
// coordinate / depth, normal length, median index: 24 of 88 bytes per pixel) is not stored -- the colour-only backward instance
// reads none of it.  (Storing the coordinate / depth / normal maps, which nothing reads again in such a step, with the
// non-temporal policy was measured too: 59.5 -> 86 us -- a quad row is 32 bytes of a line, and nt stores give up the L2's
// write combining.)  KEEP_N: false in the fused kernel -- the backward follows in the same workgroup and takes the contributor
// count from `px` instead of from memory.
// `chunk`: FWD_CHUNK * (GEO ? 6 : 3) float4 of LDS; quad_bits [4][FWD_NSW]; wave_done [4].  Every thread of the workgroup calls it.
template <bool COORD, bool DEPTH, bool NORMAL, bool LEAN, bool KEEP_N>
__device__ __forceinline__ void blend_fwd_tile(const BlendFwdArgs& a, const uint32_t tile, float4* __restrict__ chunk,
                                               uint64_t (*quad_bits)[FWD_NSW], int* wave_done, FwdPix& px_out)      // quad_bits: [FWD_NLIST][FWD_NSW]
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    constexpr int NQ = GEO ? 6 : 3;                     // float4 per staged record
    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t px = tx * TILE + (wid & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    const uint2 range = ((const uint2*)a.ranges)[tile];
    const int n = (int)(range.y - range.x);      // (a tile that overflowed its slab has an empty range; the frame is then redone)
    const int rounds = (n + FWD_CHUNK - 1) / FWD_CHUNK;

    float T = 1.0f;
    float Tl = inside ? 1.0f : 0.0f;      // live transmittance: T while the pixel still takes splats, 0 once it is finished (or outside the image)
    uint32_t last_contributor = 0, max_contributor = 0xFFFFFFFFu;
    float C0 = 0, C1 = 0, C2 = 0, weight = 0;
    float Co0 = 0, Co1 = 0, Co2 = 0, mC0 = 0, mC1 = 0, mC2 = 0, Depth = 0, mDepth = 0, N0 = 0, N1 = 0, N2 = 0;

    if (tid < 4) wave_done[tid] = 0;
    for (int i = 0; i < rounds; i++) {
        tile_barrier();                                           // previous chunk consumed, wave_done published
        if (wave_done[0] & wave_done[1] & wave_done[2] & wave_done[3]) break;
        const int progress = i * FWD_CHUNK + (int)tid;
        uint32_t qmask = 0;
        if (tid < FWD_CHUNK && progress < n) {
            const uint32_t id = a.point_list[range.x + progress];
            const float4* src = (const float4*)(a.rec + (size_t)id * REC_F);
            float4 q0 = src[0], q1 = src[1], q2 = src[2];
            if (a.colors_precomp) {                                // feature_ptr = colors_precomp (rasterizer_impl.cu:394)
                q1.z = a.colors_precomp[3 * (size_t)id]; q1.w = a.colors_precomp[3 * (size_t)id + 1];
                q2.x = a.colors_precomp[3 * (size_t)id + 2];
            }
            chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = q1; chunk[tid * NQ + 2] = q2;
            if constexpr (GEO) { chunk[tid * NQ + 3] = src[3]; chunk[tid * NQ + 4] = src[4]; chunk[tid * NQ + 5] = src[5]; }
            qmask = quad_reach_mask(q0, q1, tile_x0, tile_y0);
        }
        if (wid < FWD_NSW) {
#pragma unroll
            for (int q = 0; q < FWD_NLIST; q++) {
                const uint64_t b = __ballot((qmask >> q) & 1u);
                if (lane == 0) quad_bits[q][wid] = b;
            }
        }
        tile_barrier();
        // one (wave, splat) row; `r`: the staged record(s) the lanes read (wave-uniform address, or one per half), `slot`: its position
        // in the chunk (scalar or per-lane)
        auto blend_row = [&](const float4* r, const auto slot) {
                    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
                    const float dx = q0.x - pixfx, dy = q0.y - pixfy;
                    const float power = gauss_power(q0.z, q0.w, q1.x, dx, dy);
                    const float alpha = fminf(0.99f, q1.y * __expf(power));
                    // The reference's control flow (forward.cu:556-573: skip if power > 0 or alpha < 1/255; stop for good once
                    // T (1 - alpha) < 1e-4) without boolean state: a skipped splat blends with alpha 0, a finished pixel has the
                    // LIVE transmittance Tl = 0.  Tl stays >= 1e-4 while the pixel is live, so `test_T >= 1e-4` alone says
                    // "alive after this splat" (skipped: test_T = Tl; finished: test_T = 0), and a splat contributes iff its
                    // alpha T is positive.  Negated comparisons keep the reference's behaviour for NaN.  (Issue slots are what
                    // this loop runs out of -- tools/ubench/scalar_cost -- and the mask arithmetic of `done` / `pass` / `contrib`
                    // was nine scalar instructions per row.)
                    const float alpha_e = (!(power > 0.0f) && !(alpha < 1.0f / 255.0f)) ? alpha : 0.0f;
                    const float test_T = Tl * (1.0f - alpha_e);
                    const bool alive = !(test_T < 0.0001f);
                    const float aT = alive ? alpha_e * Tl : 0.0f;
                    const bool contrib = aT > 0.0f;
                    const uint32_t contributor = (uint32_t)(i * FWD_CHUNK + 1) + (uint32_t)slot;
                    C0 += q1.z * aT; C1 += q1.w * aT; C2 += q2.x * aT;
                    if constexpr (GEO) {
                        const float4 q3 = r[3];                            // view_point, n.x
                        const float4 q5 = r[5];                            // cp4, cp5, n.y, n.z
                        if constexpr (COORD) {
                            const float4 q4 = r[4];                        // cp0..3
                            const float c0 = q3.x + q4.x * dx + q4.y * dy;
                            const float c1 = q3.y + q4.z * dx + q4.w * dy;
                            const float c2 = q3.z + q5.x * dx + q5.y * dy;
                            Co0 += c0 * aT; Co1 += c1 * aT; Co2 += c2 * aT;
                        }
                        if constexpr (DEPTH) {
                            const float t = q2.y + (q2.z * dx + q2.w * dy);
                            Depth += t * aT;
                        }
                        if constexpr (NORMAL) { N0 += q3.w * aT; N1 += q5.z * aT; N2 += q5.w * aT; }
                        // only the index of the median splat is tracked here; its coordinate and depth are re-evaluated once
                        // per pixel after the loop (forward.cu:640-652 stores them inside the loop)
                        max_contributor = (contrib && Tl > 0.5f) ? contributor : max_contributor;
                    }
                    weight += aT;
                    T = alive ? test_T : T;                 // (skipped splat: test_T = Tl = T; finished pixel: T keeps its last value)
                    Tl = alive ? test_T : 0.0f;
                    last_contributor = contrib ? contributor : last_contributor;
        };
        if (__ballot(Tl != 0.0f) != 0ull) {
            bool wave_finished = false;
            for (int sw = 0; sw < FWD_NSW && !wave_finished; sw++) {
                uint64_t bits = uniform64(quad_bits[wid][sw]);     // wave-uniform
                while (bits != 0ull) {
                    const int jj = __builtin_ctzll(bits);
                    asm("s_bitset0_b64 %0, %1" : "+s"(bits) : "s"(jj));      // (one scalar instruction instead of the three of bits &= bits - 1)
                    const int j = sw * 64 + jj;
                    uint32_t addr;                                           // LDS byte offset of the record: ONE vector multiply (s_mul + v_mov otherwise)
                    if constexpr (NQ * 16 <= 64) asm("v_mul_u32_u24 %0, %1, %2" : "=v"(addr) : "s"(j), "n"(NQ * 16));
                    else asm("v_mul_u32_u24 %0, %1, %2" : "=v"(addr) : "s"(j), "v"(NQ * 16));
                    blend_row((const float4*)((const char*)chunk + addr), (uint32_t)j);
                }
                // "has every pixel of the quad saturated?" is asked once per 64 staged splats, not per row: the ballot
                // costs two VALU ops and a branch in the middle of the row (measured: 71.5 -> 65 us); at most the
                // rest of one 64-splat word is blended into lanes that no longer take anything
                if (__ballot(Tl != 0.0f) == 0ull) wave_finished = true;
            }
        }
        const bool all_done = __ballot(Tl != 0.0f) == 0ull;       // (the ballot must be taken by the whole wave)
        if (lane == 0) wave_done[wid] = all_done ? 1 : 0;
    }

    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)a.W * py + px;
        if constexpr (GEO) {
            if (max_contributor != 0xFFFFFFFFu) {
                const uint32_t id = a.point_list[range.x + max_contributor - 1];
                const float4* src = (const float4*)(a.rec + (size_t)id * REC_F);
                const float4 q0 = src[0];
                const float dx = q0.x - pixfx, dy = q0.y - pixfy;
                if constexpr (COORD) {
                    const float4 q3 = src[3], q4 = src[4], q5 = src[5];
                    mC0 = q3.x + q4.x * dx + q4.y * dy;
                    mC1 = q3.y + q4.z * dx + q4.w * dy;
                    mC2 = q3.z + q5.x * dx + q5.y * dy;
                }
                if constexpr (DEPTH) {
                    const float4 q2 = src[2];
                    mDepth = q2.y + (q2.z * dx + q2.w * dy);
                }
            }
        }
        if constexpr (KEEP_N) a.n_contrib[pix] = last_contributor;
        if constexpr (!LEAN) a.n_contrib[pix + HW] = max_contributor;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[HW + pix] = C1 + T * a.bg[1];
        a.out_color[2 * HW + pix] = C2 + T * a.bg[2];
        a.out_alpha[pix] = weight;
        // the per-pixel normalisations (forward.cu:696-727) use v_rcp_f32 / v_rsq_f32 (1 ulp) instead of IEEE division sequences: nine
        // divisions and two square roots were ~100 of the epilogue's 214 instructions, and an ulp is 1e-7 of a bar of 1e-4
        const float pnx = (pixfx - a.W / 2.f) / a.fx, pny = (pixfy - a.H / 2.f) / a.fy;
        const float inv_ln = __builtin_amdgcn_rsqf(pnx * pnx + pny * pny + 1);
        const float inv_w = __builtin_amdgcn_rcpf(weight);
        if constexpr (COORD) {
            a.out_coord[pix] = last_contributor ? Co0 * inv_w : 0.f;
            a.out_coord[HW + pix] = last_contributor ? Co1 * inv_w : 0.f;
            a.out_coord[2 * HW + pix] = last_contributor ? Co2 * inv_w : 0.f;
            if constexpr (!LEAN) { a.accum_coord[pix] = Co0; a.accum_coord[HW + pix] = Co1; a.accum_coord[2 * HW + pix] = Co2; }
            a.out_mcoord[pix] = mC0; a.out_mcoord[HW + pix] = mC1; a.out_mcoord[2 * HW + pix] = mC2;
        } else {
            a.out_coord[pix] = 0.f; a.out_coord[HW + pix] = 0.f; a.out_coord[2 * HW + pix] = 0.f;
            a.out_mcoord[pix] = 0.f; a.out_mcoord[HW + pix] = 0.f; a.out_mcoord[2 * HW + pix] = 0.f;
        }
        if constexpr (DEPTH) {
            const float depth_ln = Depth * inv_ln;
            if constexpr (!LEAN) a.accum_depth[pix] = depth_ln;
            a.out_depth[pix] = last_contributor ? depth_ln * inv_w : 0.f;
            a.out_mdepth[pix] = mDepth * inv_ln;
        } else {
            a.out_depth[pix] = 0.f; a.out_mdepth[pix] = 0.f;
        }
        if constexpr (NORMAL) {
            if (last_contributor) {
                float len = __builtin_amdgcn_sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
                if constexpr (!LEAN) a.normal_length[pix] = len;
                len = fmaxf(len, 1.0E-12F);
                const float inv_len = __builtin_amdgcn_rcpf(len);
                a.out_normal[pix] = N0 * inv_len; a.out_normal[HW + pix] = N1 * inv_len; a.out_normal[2 * HW + pix] = N2 * inv_len;
            } else {
                if constexpr (!LEAN) a.normal_length[pix] = 1.f;
                a.out_normal[pix] = 0.f; a.out_normal[HW + pix] = 0.f; a.out_normal[2 * HW + pix] = 0.f;
            }
        } else {
            a.out_normal[pix] = 0.f; a.out_normal[HW + pix] = 0.f; a.out_normal[2 * HW + pix] = 0.f;
        }
    }
    px_out.C0 = C0; px_out.C1 = C1; px_out.C2 = C2; px_out.T = T; px_out.weight = weight; px_out.last_contributor = last_contributor;
}
