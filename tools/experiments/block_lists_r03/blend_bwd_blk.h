// blend_bwd_blk.h -- backward blend of one tile, colour-only instance, BLOCK-LIST form (blend_blk.h): the wave's four 16-lane groups
// are the four 4x4 pixel blocks of its 8x8 quad and walk their own dense lists of staged splats back to front, so one trip of the
// wave works on up to four different splats -- and ONE transpose through LDS reduces the moments of all four: every reader lane sums
// the 16 columns of its block for one moment.  The block sums go to a per-tile accumulator in LDS (one row per staged splat) and from
// there, once per round, to the global accumulator rows: one 40-byte atomic request per (tile, splat).  (Block sums sent straight to
// global memory were measured: 1.1 million requests per view instead of 0.5, four lanes of one instruction on the same address
// whenever a splat covers several blocks of a quad -- blend_bwd 61 us without its atomics, 102 us with them.)
// Same arithmetic per (pixel, splat) pair, in the same order per pixel, as blend_bwd_tile.h (BACKWARD::renderCUDA,
// backward.cu:631-1016); the per-Gaussian sums add 16-pixel partials instead of 64-pixel ones (the reference adds them one by one,
// in whatever order its atomics land).
#pragma once
#include "blend_blk.h"
#include "blend_bwd_tile.h"

#ifndef BWDB_CHUNK
#define BWDB_CHUNK 64            // splats staged per round (a byte indexes a slot; slot BWDB_CHUNK is the dummy record)
#endif
#define BWDB_LIST_BYTES (4 * 4 * (BWDB_CHUNK + 4))
#define BWDB_CHUNK_BYTES ((BWDB_CHUNK + 1) * 3 * 16)
template <bool ABS> struct BwdBlkCfg {
    static constexpr int NROWS = 9 + (ABS ? 1 : 0);
    static constexpr int RED_FLOATS = 4 * NROWS * BWD_RED_STRIDE;
    static constexpr int ACCS = NROWS;                                   // floats per staged splat in the tile's accumulator
    // LDS carve-up: records (+ dummy) | reach masks | block lists | transpose buffers | per-splat accumulators of the round
    static constexpr size_t OFF_REACH = BWDB_CHUNK_BYTES, OFF_LISTS = OFF_REACH + BWDB_CHUNK * 4, OFF_RED = (OFF_LISTS + BWDB_LIST_BYTES + 15) & ~(size_t)15;
    static constexpr size_t OFF_ACC = OFF_RED + (size_t)RED_FLOATS * 4;
    static constexpr size_t BYTES = OFF_ACC + (size_t)(BWDB_CHUNK + 1) * ACCS * 4;
};

// `smem`: BwdBlkCfg<ABS>::BYTES bytes of LDS, 16-byte aligned; wave_max [4].  Every thread of the workgroup calls it.
// PRE: the forward's per-pixel results arrive in registers -- in the BLOCK lane order (blend_fwd_tile's BLKMAP).
template <bool ABS, bool PRE>
__device__ __forceinline__ void blend_bwd_tile_blk(const BlendBwdArgs& a, const uint32_t tile, char* __restrict__ smem, int* wave_max, const FwdPix* pre)
{
    using Cfg = BwdBlkCfg<ABS>;
    constexpr int NROWS = Cfg::NROWS, NQ = 3, BCHUNK = BWDB_CHUNK, LSTRIDE = BWDB_CHUNK + 4, RED_STRIDE = BWD_RED_STRIDE;
    float4* const chunk = (float4*)smem;
    uint32_t* const reach = (uint32_t*)(smem + Cfg::OFF_REACH);
    uint8_t* const lists_all = (uint8_t*)(smem + Cfg::OFF_LISTS);
    float* const red_all = (float*)(smem + Cfg::OFF_RED);
    float* const acc_all = (float*)(smem + Cfg::OFF_ACC);
    constexpr int ACCS = Cfg::ACCS;

    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const BlkLane BL = blk_lane(lane, wid);
    const uint32_t px = tx * TILE + (wid & 1) * 8 + BL.lx;
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + BL.ly;
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)a.W * py + px;
    uint8_t* const lists = lists_all + wid * (4 * LSTRIDE);                 // this wave's four block lists
    const uint8_t* const my_list = lists + BL.blk * LSTRIDE;                // ... and this lane's block's

    const uint2 range = ((const uint2*)a.ranges)[tile];
    // clamped to the tile's list length: a corrupt image buffer must not turn into an out-of-bounds gather
    int last_contributor;
    if constexpr (PRE) last_contributor = inside ? (int)min(pre->last_contributor, range.y - range.x) : 0;
    else last_contributor = inside ? (int)min(a.n_contrib[pix], range.y - range.x) : 0;

    // ---- per-pixel upstream gradients (backward.cu:732-781); zero for pixels nothing was blended into
    float gp0 = 0, gp1 = 0, gp2 = 0, g_alpha = 0, T_final = 0, bg_dot = 0;
    if (a.l1_gt) {
        // fused L1 (loss_utils.py:17 l1_loss + its backward): every pixel of the image counts towards the loss value
        float acc = 0.f;
        if (inside) {
            float c0, c1, c2;
            if constexpr (PRE) {      // the colour image exactly as the forward stored it (blend_fwd_tile.h: C + T * bg)
                c0 = pre->C0 + pre->T * a.bg[0]; c1 = pre->C1 + pre->T * a.bg[1]; c2 = pre->C2 + pre->T * a.bg[2];
            } else { c0 = a.l1_color[pix]; c1 = a.l1_color[HW + pix]; c2 = a.l1_color[2 * HW + pix]; }
            const float d0 = c0 - a.l1_gt[pix], d1 = c1 - a.l1_gt[HW + pix];
            const float d2 = c2 - a.l1_gt[2 * HW + pix];
            acc = fabsf(d0) + fabsf(d1) + fabsf(d2);
            gp0 = d0 > 0.f ? a.l1_scale : (d0 < 0.f ? -a.l1_scale : 0.f);
            gp1 = d1 > 0.f ? a.l1_scale : (d1 < 0.f ? -a.l1_scale : 0.f);
            gp2 = d2 > 0.f ? a.l1_scale : (d2 < 0.f ? -a.l1_scale : 0.f);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0 && acc != 0.f) atomicAdd(&a.l1_loss[16 * ((tile * 4 + wid) & 63)], acc);
    }
    if (last_contributor > 0) {
        float w_final;
        if constexpr (PRE) w_final = pre->weight; else w_final = a.alphas[pix];
        T_final = 1.0f - w_final;
        if (a.dL_dpix && !a.l1_gt) { gp0 = a.dL_dpix[pix]; gp1 = a.dL_dpix[HW + pix]; gp2 = a.dL_dpix[2 * HW + pix]; }
        if (a.dL_dalpha) g_alpha = a.dL_dalpha[pix];
        bg_dot = a.bg[0] * gp0 + a.bg[1] * gp1 + a.bg[2] * gp2;
    }

    // deepest contributor of every block (a block = one DPP row of 16 lanes), of the wave, of the tile
    int bmax[4], my_wave_max;
    {
        int m = last_contributor;
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false));        // quad_perm [1,0,3,2]
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false));        // quad_perm [2,3,0,1]
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, false));       // row_half_mirror
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x140, 0xF, 0xF, false));       // row_mirror: every lane of the row holds the row's maximum
#pragma unroll
        for (int b = 0; b < 4; b++) bmax[b] = __builtin_amdgcn_readlane(m, 16 * b);
        my_wave_max = max(max(bmax[0], bmax[1]), max(bmax[2], bmax[3]));
        if (lane == 0) wave_max[wid] = my_wave_max;
    }
    for (int k = (int)tid; k < (BCHUNK + 1) * ACCS; k += 256) acc_all[k] = 0.f;
    if (tid < NQ) {                                  // the dummy record a run-out list reads: opacity 0 -> alpha 0 -> nothing happens
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid == 0) z = make_float4(0.f, 0.f, 1.f, 0.f);
        if (tid == 1) z = make_float4(1.f, BLK_DUMMY_OPACITY, 0.f, 0.f);
        chunk[BCHUNK * NQ + tid] = z;
    }
    __syncthreads();
    const int n = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));   // elements [0, n) of the range
    const int rounds = (n + BCHUNK - 1) / BCHUNK;

    float T = T_final, S = 0.f, Dprev = 0.f, last_alpha = 0.f;
    const bool has_bg = (a.bg[0] != 0.f) || (a.bg[1] != 0.f) || (a.bg[2] != 0.f);      // wave-uniform
    const float halfW = 0.5f * a.W, halfH = 0.5f * a.H;
    float* const myred = red_all + wid * (NROWS * RED_STRIDE);
    // LDS byte offset of this wave's transpose buffer (the low half of the flat address of a __shared__ object is its LDS offset)
    const unsigned myred_m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)myred);
    // ---- the reader lanes of the transpose: lane 16 b + i sums the 16 columns of block b for ONE moment.  Which moment, is chosen
    // so that the four ds_read_b128 of a row sum are conflict-free at stride 68 in the 16-lane service groups of the LDS
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md): even blocks take moment i, odd blocks moment
    // (i + SH) mod NROWS; lanes i >= NROWS re-read a row some lane of their own service group reads anyway (a broadcast).
    constexpr int SH = NROWS == 10 ? 6 : 5;
    const int ridx = (int)(lane & 15u);
    const int ridx_e = ridx < NROWS ? ridx : (ridx >= 12 ? ridx - 12 : ridx - SH);
    const int rmom = (BL.blk & 1u) ? (ridx_e + SH) % NROWS : ridx_e;
    const float4* const my_row = (const float4*)(myred + rmom * RED_STRIDE + 16 * (int)BL.blk);
    constexpr uint64_t READER_LANES = (NROWS == 10 ? 0x03FFull : 0x01FFull) * 0x0001000100010001ull;
    // LDS byte address of this reader lane's moment in accumulator row 0
    const uint32_t acc_lane_off = (uint32_t)(size_t)acc_all + 4u * (uint32_t)rmom;

    for (int i = 0; i < rounds; i++) {
        __syncthreads();
        uint32_t qmask = 0;
        if (tid < BCHUNK) {
            const int progress = i * BCHUNK + (int)tid;      // position counted from the back of [0, n)
            if (progress < n) {
                const uint32_t id = a.point_list[range.x + (uint32_t)(n - 1 - progress)];
                const float4* src = (const float4*)(a.rec + (size_t)id * REC_F);
                float4 q0 = src[0], q1 = src[1], q2 = src[2];
                if (a.colors_precomp) {
                    q1.z = a.colors_precomp[3 * (size_t)id]; q1.w = a.colors_precomp[3 * (size_t)id + 1];
                    q2.x = a.colors_precomp[3 * (size_t)id + 2];
                }
                // staged as { x, y, conic.x, conic.y | conic.z, opacity, -, - | r, g, b, id }: three aligned reads per trip
                chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = make_float4(q1.x, q1.y, 0.f, 0.f);
                chunk[tid * NQ + 2] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
                qmask = block_reach_mask(q0, q1, tile_x0, tile_y0);
            }
            reach[tid] = qmask;
        }
        __syncthreads();
        // positions -> slots for this round: list element e sits in slot base - e
        const int base = n - 1 - i * BCHUNK;
        const int j_first = base - last_contributor;                       // e < last_contributor  <=>  slot > j_first
        int skip_below[4];
#pragma unroll
        for (int b = 0; b < 4; b++) skip_below[b] = base - bmax[b];        // slots <= that lie behind every pixel of block b
        {
            const uint32_t fill = (uint32_t)BCHUNK * 0x01010101u;
            for (int k = (int)lane; k < LSTRIDE; k += 64) ((uint32_t*)lists)[k] = fill;      // 4 * LSTRIDE bytes
        }
        const int trips = build_block_lists<BCHUNK>(reach, lists, lane, wid, skip_below);
        int j = my_list[0];
        for (int t = 0; t < trips; t++) {
            const int jn = my_list[t + 1];                   // (next trip's slot: its LDS round trip hides under this trip)
            const float4* r = &chunk[j * NQ];
            const float4 q0 = r[0];
            const float2 q1a = *(const float2*)&r[1];        // conic.z, opacity
            const float dx = q0.x - pixfx, dy = q0.y - pixfy;
            const float power = gauss_power(q0.z, q0.w, q1a.x, dx, dy);
            const float G = __expf(power);
            const float alpha = fminf(0.99f, q1a.y * G);
            const bool valid = (j > j_first) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(valid) == 0ull) { j = jn; continue; }
            const float4 qc = r[2];                          // r, g, b, id
            // 1/(1-alpha) once (v_rcp_f32, 1 ulp) instead of two IEEE divisions; 1-alpha >= 0.01
            const float inv_one_m = __builtin_amdgcn_rcpf(1.f - alpha);
            T = valid ? T * inv_one_m : T;
            const float w = valid ? alpha * T : 0.f;
            const float D = qc.x * gp0 + qc.y * gp1 + qc.z * gp2 + g_alpha;
            const float Snew = last_alpha * Dprev + (1.f - last_alpha) * S;
            float dL_dopa = (D - Snew) * T;
            if (has_bg) dL_dopa += (-T_final * inv_one_m) * bg_dot;
            S = valid ? Snew : S;
            Dprev = valid ? D : Dprev;
            last_alpha = valid ? alpha : last_alpha;
            const float dL_dG = valid ? q1a.y * dL_dopa : 0.f;
            const float q = dL_dG * G;
            const float qdx = q * dx, qdy = q * dy;
            float mv[NROWS];
            mv[0] = w * gp0; mv[1] = w * gp1; mv[2] = w * gp2;
            mv[3] = q; mv[4] = qdx; mv[5] = qdy;
            mv[6] = qdx * dx; mv[7] = qdx * dy; mv[8] = qdy * dy;
            if constexpr (ABS) {
                const float gxa = q0.z * qdx + q0.w * qdy;      // -dL/d(delx) of the Gaussian term
                const float gya = q1a.x * qdy + q0.w * qdx;
                mv[9] = fabsf(gxa * halfW) + fabsf(gya * halfH);
            }
            // ---- transpose-reduce: column writes by every lane, then 16-column row sums by the reader lanes of each block
            ColWrite<RED_STRIDE * 4, 0, NROWS>::run(myred_m0, mv);
            __builtin_amdgcn_wave_barrier();          // (scheduling fence: the row reads below must stay behind the column writes)
            float4 acc4 = my_row[0];
#pragma unroll
            for (int k = 1; k < 4; k++) { const float4 t4 = my_row[k]; acc4.x += t4.x; acc4.y += t4.y; acc4.z += t4.z; acc4.w += t4.w; }
            const float tot = (acc4.x + acc4.y) + (acc4.z + acc4.w);
            __builtin_amdgcn_wave_barrier();          // ... and the next trip's column writes behind these row reads
            // block sums -> the tile's accumulator row of that splat (LDS float add; a run-out list's dummy and a splat no pixel of the
            // block took sum to exact zeros and are left out)
            const uint64_t lanes = __ballot(tot != 0.0f) & READER_LANES;
            const uint32_t aoff = (uint32_t)j * (uint32_t)(ACCS * 4) + acc_lane_off;
#if defined(BWDB_ABLATE) && (BWDB_ABLATE & 1)
            if (tot == 123.456f) a.gacc[aoff] = tot;
            j = jn; continue;
#endif
            uint64_t saved;
            asm volatile("s_and_saveexec_b64 %0, %1\n\tds_add_f32 %2, %3\n\ts_mov_b64 exec, %0"
                         : "=&s"(saved) : "s"(lanes), "v"(aoff), "v"(tot) : "memory", "scc");
            j = jn;
        }
        // ---- flush: one global atomic row per staged splat that received something; 16 lanes per splat, NROWS of them active
        __syncthreads();
#if defined(BWDB_ABLATE) && (BWDB_ABLATE & 2)
        continue;
#endif
        for (int s0 = 0; s0 < BCHUNK; s0 += 16) {
            const int sp = s0 + (int)(tid >> 4), m = (int)(tid & 15u);
            if (m < NROWS) {
                float* cell = &acc_all[sp * ACCS + m];
                const float v = *cell;
                if (v != 0.0f) {
                    *cell = 0.0f;
                    const uint32_t gid = __float_as_uint(chunk[sp * NQ + 2].w);
                    atomicAdd(&a.gacc[(size_t)gid * GACC_COMPACT_F + m], v);
                }
            }
        }
    }
}
