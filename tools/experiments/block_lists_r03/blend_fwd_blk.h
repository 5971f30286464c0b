// blend_fwd_blk.h -- forward blend of one tile, BLOCK-LIST form (blend_blk.h): the wave's four 16-lane groups are the four 4x4 pixel
// blocks of its 8x8 quad and walk their own dense lists of staged splats, up to four different splats per trip.  Same arithmetic per
// (pixel, splat) pair, in the same order per pixel, as blend_fwd_tile.h (and FORWARD::renderCUDA, forward.cu:428-742).
#pragma once
#include "blend_blk.h"
#include "blend_fwd_tile.h"      // FwdPix

#ifndef FWDB_CHUNK
#define FWDB_CHUNK 160           // splats staged per round (a byte indexes a slot; slot FWDB_CHUNK is the dummy record)
#endif
#define FWDB_NSW ((FWDB_CHUNK + 63) / 64)
// LDS of the block-list forward: records (+ the dummy), reach masks, 4 waves x 4 lists
#define FWDB_CHUNK_F4(GEO) ((FWDB_CHUNK + 1) * ((GEO) ? 6 : 3))
#define FWDB_LIST_BYTES (4 * 4 * (FWDB_CHUNK + 4))

// `chunk`: FWDB_CHUNK_F4 float4; `reach`: [FWDB_CHUNK] words; `lists_all`: FWDB_LIST_BYTES bytes (4-byte aligned); wave_done [4].
template <bool COORD, bool DEPTH, bool NORMAL, bool LEAN, bool KEEP_N>
__device__ __forceinline__ void blend_fwd_tile_blk(const BlendFwdArgs& a, const uint32_t tile, float4* __restrict__ chunk,
                                                   uint32_t* __restrict__ reach, uint8_t* __restrict__ lists_all, int* wave_done, FwdPix& px_out)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    constexpr int NQ = GEO ? 6 : 3;                     // float4 per staged record
    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const BlkLane BL = blk_lane(lane, wid);
    const uint32_t px = tx * TILE + (wid & 1) * 8 + BL.lx;
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + BL.ly;
    constexpr int LSTRIDE = FWDB_CHUNK + 4;
    uint8_t* const lists = lists_all + wid * (4 * LSTRIDE);                 // this wave's four block lists
    const uint8_t* const my_list = lists + BL.blk * LSTRIDE;                // ... and this lane's block's
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    const uint2 range = ((const uint2*)a.ranges)[tile];
    const int n = (int)(range.y - range.x);      // (a tile that overflowed its slab has an empty range; the frame is then redone)
    const int rounds = (n + FWDB_CHUNK - 1) / FWDB_CHUNK;

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0, max_contributor = 0xFFFFFFFFu;
    float C0 = 0, C1 = 0, C2 = 0, weight = 0;
    float Co0 = 0, Co1 = 0, Co2 = 0, mC0 = 0, mC1 = 0, mC2 = 0, Depth = 0, mDepth = 0, N0 = 0, N1 = 0, N2 = 0;

    if (tid < 4) wave_done[tid] = 0;
    if (tid < NQ) {                                  // the dummy record a run-out list reads: opacity 0 -> alpha 0 -> not blended
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid == 0) z = make_float4(0.f, 0.f, 1.f, 0.f);
        if (tid == 1) z = make_float4(1.f, BLK_DUMMY_OPACITY, 0.f, 0.f);
        chunk[FWDB_CHUNK * NQ + tid] = z;
    }
    for (int i = 0; i < rounds; i++) {
        __syncthreads();                                           // previous chunk consumed, wave_done published
        if (wave_done[0] & wave_done[1] & wave_done[2] & wave_done[3]) break;
        const int progress = i * FWDB_CHUNK + (int)tid;
        uint32_t qmask = 0;
        if (tid < FWDB_CHUNK && progress < n) {
            const uint32_t id = a.point_list[range.x + progress];
            const float4* src = (const float4*)(a.rec + (size_t)id * REC_F);
            float4 q0 = src[0], q1 = src[1], q2 = src[2];
            if (a.colors_precomp) {                                // feature_ptr = colors_precomp (rasterizer_impl.cu:394)
                q1.z = a.colors_precomp[3 * (size_t)id]; q1.w = a.colors_precomp[3 * (size_t)id + 1];
                q2.x = a.colors_precomp[3 * (size_t)id + 2];
            }
            chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = q1; chunk[tid * NQ + 2] = q2;
            if constexpr (GEO) { chunk[tid * NQ + 3] = src[3]; chunk[tid * NQ + 4] = src[4]; chunk[tid * NQ + 5] = src[5]; }
            qmask = block_reach_mask(q0, q1, tile_x0, tile_y0);
        }
        if (tid < FWDB_CHUNK) reach[tid] = qmask;
        __syncthreads();
        if (__ballot(!done) != 0ull) {
            // ---- this wave's four block lists (dense, in list order), then one trip per list position
            {
                const uint32_t fill = (uint32_t)FWDB_CHUNK * 0x01010101u;
                for (int k = (int)lane; k < LSTRIDE; k += 64) ((uint32_t*)lists)[k] = fill;      // 4 * LSTRIDE bytes
            }
            const int no_skip[4] = {-1, -1, -1, -1};
            const int trips = build_block_lists<FWDB_CHUNK>(reach, lists, lane, wid, no_skip);
            int j = my_list[0];
            // "has every pixel of the quad saturated?" is asked every 16 trips, not per trip
            for (int t0 = 0; t0 < trips; t0 += 16) {
            const int tend = min(trips, t0 + 16);
            for (int t = t0; t < tend; t++) {
                const int jn = my_list[t + 1];                     // (next trip's slot: its LDS round trip hides under this trip)
                const float4* r = &chunk[j * NQ];
                const float4 q0 = r[0], q1 = r[1], q2 = r[2];
                const float dx = q0.x - pixfx, dy = q0.y - pixfy;
                const float power = gauss_power(q0.z, q0.w, q1.x, dx, dy);
                const float alpha = fminf(0.99f, q1.y * __expf(power));
                const float test_T = T * (1.0f - alpha);
                // negated comparisons keep the reference's behaviour for NaN (forward.cu:556-573: `if (x > 0) continue`)
                const bool pass = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                const bool contrib = pass && !(test_T < 0.0001f);
                done = done || (pass && test_T < 0.0001f);
                const uint32_t contributor = (uint32_t)(i * FWDB_CHUNK + j + 1);
                const float aT = contrib ? alpha * T : 0.0f;
                C0 += q1.z * aT; C1 += q1.w * aT; C2 += q2.x * aT;
                const bool before_median = contrib && T > 0.5f;
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
                        const float t_ = q2.y + (q2.z * dx + q2.w * dy);
                        Depth += t_ * aT;
                    }
                    if constexpr (NORMAL) { N0 += q3.w * aT; N1 += q5.z * aT; N2 += q5.w * aT; }
                    max_contributor = before_median ? contributor : max_contributor;
                }
                weight += aT;
                T = contrib ? test_T : T;
                last_contributor = contrib ? contributor : last_contributor;
                j = jn;
            }
            if (__ballot(!done) == 0ull) break;
            }
        }
        const bool all_done = __ballot(!done) == 0ull;            // (the ballot must be taken by the whole wave)
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
        const float pnx = (pixfx - a.W / 2.f) / a.fx, pny = (pixfy - a.H / 2.f) / a.fy;
        const float ln = sqrtf(pnx * pnx + pny * pny + 1);
        if constexpr (COORD) {
            a.out_coord[pix] = last_contributor ? Co0 / weight : 0.f;
            a.out_coord[HW + pix] = last_contributor ? Co1 / weight : 0.f;
            a.out_coord[2 * HW + pix] = last_contributor ? Co2 / weight : 0.f;
            if constexpr (!LEAN) { a.accum_coord[pix] = Co0; a.accum_coord[HW + pix] = Co1; a.accum_coord[2 * HW + pix] = Co2; }
            a.out_mcoord[pix] = mC0; a.out_mcoord[HW + pix] = mC1; a.out_mcoord[2 * HW + pix] = mC2;
        } else {
            a.out_coord[pix] = 0.f; a.out_coord[HW + pix] = 0.f; a.out_coord[2 * HW + pix] = 0.f;
            a.out_mcoord[pix] = 0.f; a.out_mcoord[HW + pix] = 0.f; a.out_mcoord[2 * HW + pix] = 0.f;
        }
        if constexpr (DEPTH) {
            const float depth_ln = Depth / ln;
            if constexpr (!LEAN) a.accum_depth[pix] = depth_ln;
            a.out_depth[pix] = last_contributor ? depth_ln / weight : 0.f;
            a.out_mdepth[pix] = mDepth / ln;
        } else {
            a.out_depth[pix] = 0.f; a.out_mdepth[pix] = 0.f;
        }
        if constexpr (NORMAL) {
            if (last_contributor) {
                float len = sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
                if constexpr (!LEAN) a.normal_length[pix] = len;
                len = fmaxf(len, 1.0E-12F);
                a.out_normal[pix] = N0 / len; a.out_normal[HW + pix] = N1 / len; a.out_normal[2 * HW + pix] = N2 / len;
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
