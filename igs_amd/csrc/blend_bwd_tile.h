// blend_bwd_tile.h -- backward of the tile blend for gfx950 (wave64): the work of ONE 16x16 tile by one 256-thread workgroup
// (device code shared by the stand-alone backward kernel, blend_bwd.hip, and the fused tile kernel of the refine step, blend_step.hip).
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
//  * The 64 lanes of a wave are summed by a transpose through LDS (25 conflict-free ds_write_b32, then every lane
//    adds up one half-row with 8 ds_read_b128): ~60 instructions per (wave, splat) instead of 25 six-step butterflies,
//    and the totals land one per lane, so ONE 25-lane, 100-byte contiguous global atomic per (wave, splat) replaces
//    the reference's 25 x 64 scalar atomics.
//  * As in the forward, each wave only walks the splats whose alpha >= 1/255 footprint reaches its 8x8 quad.
// T is recovered exactly like the reference does (T_final = 1 - out_alpha, T <- T / (1 - alpha), backward.cu:706,857).
#pragma once
#include "blend_common.h"
#include "blend_fwd_tile.h"      // FwdPix

// splats staged per round in the backward (LDS is shared with the reduction scratch): 128 for the colour-only instance, 64 for the
// instances with geometry, whose 15..25-row transpose buffers already take 16..27 KB of the workgroup's LDS
#define BCHUNK_MAX 128

// Column writes of the transpose buffer: lane l stores its value of moment r at  base + r * stride + 4 l.  That is exactly the
// address pattern of ds_write_addtid_b32 (address = M0 + offset + 4 * lane, no address VGPR), which moves 4 B per lane to the LDS
// in 2 cycles instead of the 4 of ds_write_b32 / 3 per dword of ds_write2_b32 (MI355X_MICROARCH.md, LDS: a store's cost is the
// transfer of its address and data registers) -- and this kernel is bound by the LDS pipe: 10 column writes per row
// were 28 of its 68 LDS cycles.  M0 is set inside every asm statement (the compiler knows nothing of it otherwise); up to five
// stores share one statement.  The s_nop is REQUIRED: an SALU write of M0 followed by an LDS "add-TID" instruction needs one wait
// state (ISA manual, user-inserted wait states), and the compiler's hazard recogniser does not look inside inline assembly --
// without it the first store of a statement can still see the previous M0.
template <int STRIDE_B, int R0, int N> struct ColWrite;
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 1> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" :: "v"(v[R0]), "s"(m0), "n"(R0 * STRIDE_B) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 2> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
                     :: "v"(v[R0]), "v"(v[R0 + 1]), "s"(m0), "n"(R0 * STRIDE_B), "n"((R0 + 1) * STRIDE_B) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 3> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%4\n\tds_write_addtid_b32 %1 offset:%5\n\tds_write_addtid_b32 %2 offset:%6"
                     :: "v"(v[R0]), "v"(v[R0 + 1]), "v"(v[R0 + 2]), "s"(m0), "n"(R0 * STRIDE_B), "n"((R0 + 1) * STRIDE_B), "n"((R0 + 2) * STRIDE_B) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 4> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%5\n\tds_write_addtid_b32 %1 offset:%6\n\tds_write_addtid_b32 %2 offset:%7\n\t"
                     "ds_write_addtid_b32 %3 offset:%8"
                     :: "v"(v[R0]), "v"(v[R0 + 1]), "v"(v[R0 + 2]), "v"(v[R0 + 3]), "s"(m0), "n"(R0 * STRIDE_B), "n"((R0 + 1) * STRIDE_B),
                        "n"((R0 + 2) * STRIDE_B), "n"((R0 + 3) * STRIDE_B) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 5> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%6\n\tds_write_addtid_b32 %1 offset:%7\n\tds_write_addtid_b32 %2 offset:%8\n\t"
                     "ds_write_addtid_b32 %3 offset:%9\n\tds_write_addtid_b32 %4 offset:%10"
                     :: "v"(v[R0]), "v"(v[R0 + 1]), "v"(v[R0 + 2]), "v"(v[R0 + 3]), "v"(v[R0 + 4]), "s"(m0), "n"(R0 * STRIDE_B),
                        "n"((R0 + 1) * STRIDE_B), "n"((R0 + 2) * STRIDE_B), "n"((R0 + 3) * STRIDE_B), "n"((R0 + 4) * STRIDE_B) : "memory", "m0");
    }
};
// 6..10 rows in ONE statement: one s_mov m0 per transpose instead of two (every scalar instruction costs the SIMD an issue slot of
// ~2.5 cycles, exactly like a vector one -- tools/ubench/scalar_cost)
#define CW_ST(k) "\n\tds_write_addtid_b32 %" #k " offset:%c[o" #k "]"
#define CW_IN(k) "v"(v[R0 + k])
#define CW_OFF(k) [o##k] "n"((R0 + k) * STRIDE_B)
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 9> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0" CW_ST(0) CW_ST(1) CW_ST(2) CW_ST(3) CW_ST(4) CW_ST(5) CW_ST(6) CW_ST(7) CW_ST(8)
                     :: CW_IN(0), CW_IN(1), CW_IN(2), CW_IN(3), CW_IN(4), CW_IN(5), CW_IN(6), CW_IN(7), CW_IN(8), [m] "s"(m0),
                        CW_OFF(0), CW_OFF(1), CW_OFF(2), CW_OFF(3), CW_OFF(4), CW_OFF(5), CW_OFF(6), CW_OFF(7), CW_OFF(8) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0> struct ColWrite<STRIDE_B, R0, 10> {
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0" CW_ST(0) CW_ST(1) CW_ST(2) CW_ST(3) CW_ST(4) CW_ST(5) CW_ST(6) CW_ST(7) CW_ST(8) CW_ST(9)
                     :: CW_IN(0), CW_IN(1), CW_IN(2), CW_IN(3), CW_IN(4), CW_IN(5), CW_IN(6), CW_IN(7), CW_IN(8), CW_IN(9), [m] "s"(m0),
                        CW_OFF(0), CW_OFF(1), CW_OFF(2), CW_OFF(3), CW_OFF(4), CW_OFF(5), CW_OFF(6), CW_OFF(7), CW_OFF(8), CW_OFF(9) : "memory", "m0");
    }
};
template <int STRIDE_B, int R0, int N> struct ColWrite {          // N > 5: five now, the rest recursively
    static __device__ __forceinline__ void run(unsigned m0, const float* v) {
        ColWrite<STRIDE_B, R0, 5>::run(m0, v);
        ColWrite<STRIDE_B, R0 + 5, N - 5>::run(m0, v);
    }
};

#define BWD_RED_STRIDE 68
// compile-time shape of an instance: staged record size, splats per round, live moments, LDS it needs
template <bool COORD, bool DEPTH, bool NORMAL, bool ABS> struct BwdCfg {
    static constexpr bool GEO = COORD || DEPTH || NORMAL;
    static constexpr int NQ = GEO ? 6 : 3;
    static constexpr int BCHUNK = GEO ? 64 : 128;
    static constexpr int NSW = BCHUNK / 64;
    static constexpr int NROWS = 9 + (ABS ? 1 : 0) + (COORD ? 9 : 0) + (DEPTH ? 3 : 0) + (NORMAL ? 3 : 0);
    static constexpr int RED_FLOATS = 4 * NROWS * BWD_RED_STRIDE;        // four waves' transpose buffers
};

// gacc slots (raw moments), see geom_bwd.hip for how they are combined:
//  0..2  sum w*dL/dpix_ch            3..5  Sv = sum dLc_ch          6..8 Sx = sum dLc_ch*dx     9..11 Sy = sum dLc_ch*dy
//  12    St = sum dLt   13 Stx   14 Sty   15..17 sum w*dL/dnormal_ch
//  18 Q0 = sum q  19 Qx  20 Qy  21 Qxx  22 Qxy  23 Qyy  (q = dL/dG * G)   24 Z = abs-sum for dL_dmean2D.z
// PRE: the per-pixel results of the forward arrive in registers (`pre`, fused kernel) instead of being loaded from the images.
// LDS (all provided by the kernel): chunk [BCHUNK * NQ] float4, chunk_id [BCHUNK], quad_bits [4][NSW], wave_max [4], red [RED_FLOATS]
// (16-byte aligned).  Every thread of the workgroup calls it.
template <bool COORD, bool DEPTH, bool NORMAL, bool ABS, bool PRE>
__device__ __forceinline__ void blend_bwd_tile(const BlendBwdArgs& a, const uint32_t tile, float4* __restrict__ chunk, uint32_t* __restrict__ chunk_id,
                                               uint64_t (*quad_bits)[BwdCfg<COORD, DEPTH, NORMAL, ABS>::NSW], int* wave_max,
                                               float* __restrict__ red_all, const FwdPix* pre)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    constexpr int NQ = BwdCfg<COORD, DEPTH, NORMAL, ABS>::NQ;
    constexpr int BCHUNK = BwdCfg<COORD, DEPTH, NORMAL, ABS>::BCHUNK;
    constexpr int NSW = BwdCfg<COORD, DEPTH, NORMAL, ABS>::NSW;                    // staging waves
    constexpr int NROWS = BwdCfg<COORD, DEPTH, NORMAL, ABS>::NROWS;                // live moments of this instance
    // floats per row of the per-wave transpose buffer: 16-byte aligned rows for the b128 row reads.  Columns are written with
    // ds_write_addtid_b32 (one dword per lane at consecutive addresses: conflict-free whatever the stride); the ROW reads -- 4
    // lanes per row, each summing a 16-column segment with four ds_read_b128 -- are serviced in the fixed 16-lane groups of
    // MI355X_MICROARCH.md's LDS table, and with the plain assignment "lane part p reads segment p" every group had a 2-way bank
    // conflict on every read (stride 100: 8 extra LDS cycles per row = the 27 % SQ_LDS_BANK_CONFLICT of round 1, on a kernel that is
    // bound by the LDS pipe).  Rotating the segments per row -- part p of row r reads segment (p + f[r]) mod 4, f found by search
    // over the group table (tools/lds_swizzle_search.py) -- makes the 9- and 10-row instances conflict-free at stride 68, which also
    // shrinks the buffer: 17 KB of LDS per workgroup, 8 waves per SIMD instead of 7.
    constexpr int RED_STRIDE = BWD_RED_STRIDE;
    constexpr uint32_t SEG_ROT = NROWS == 9 ? 0x1b46bu          // f = 3,2,2,1,0,1,3,2,1      (2 bits per row, row 0 lowest)
                               : NROWS == 10 ? 0x4431eu       // f = 2,3,1,0,3,0,0,1,0,1
                               : NROWS == 16 ? 0xa7dbb9c7u    // f = 3,1,0,3,1,2,3,2,3,2,1,3,3,1,2,2 (8 conflict cycles left of 32)
                               : 0u;
    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t px = tx * TILE + (wid & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)a.W * py + px;

    const uint2 range = ((const uint2*)a.ranges)[tile];
    // clamped to the tile's list length: a corrupt image buffer must not turn into an out-of-bounds gather
    int last_contributor;
    if constexpr (PRE) last_contributor = inside ? (int)min(pre->last_contributor, range.y - range.x) : 0;
    else last_contributor = inside ? (int)min(a.n_contrib[pix], range.y - range.x) : 0;
    uint32_t max_contributor = 0u;                 // (the median index: only the coordinate / depth branches use it)
    if constexpr (COORD || DEPTH) max_contributor = inside ? a.n_contrib[pix + HW] : 0u;

    // ---- per-pixel upstream gradients (backward.cu:732-781); zero for pixels nothing was blended into, whose
    //      normalisations would otherwise be 0/0 (the reference never consumes those values either)
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
    float gc0 = 0, gc1 = 0, gc2 = 0, gm0 = 0, gm1 = 0, gm2 = 0, g_t = 0, g_mt = 0, gn0 = 0, gn1 = 0, gn2 = 0;
    if (last_contributor > 0) {
        float w_final;
        if constexpr (PRE) w_final = pre->weight; else w_final = a.alphas[pix];
        T_final = 1.0f - w_final;
        if (a.dL_dpix && !a.l1_gt) { gp0 = a.dL_dpix[pix]; gp1 = a.dL_dpix[HW + pix]; gp2 = a.dL_dpix[2 * HW + pix]; }
        if (a.dL_dalpha) g_alpha = a.dL_dalpha[pix];
        bg_dot = a.bg[0] * gp0 + a.bg[1] * gp1 + a.bg[2] * gp2;
        if constexpr (GEO) {
            const float ww = w_final * w_final;
            const float pnx = (pixfx - a.W / 2.f) / a.fx, pny = (pixfy - a.H / 2.f) / a.fy;
            const float ln = sqrtf(pnx * pnx + pny * pny + 1);
            if constexpr (COORD) {
                float w0 = 0.f, w1 = 0.f, w2 = 0.f;
                if (a.dL_dcoord) { w0 = a.dL_dcoord[pix]; w1 = a.dL_dcoord[HW + pix]; w2 = a.dL_dcoord[2 * HW + pix]; }
                g_alpha -= w0 * a.accum_coord[pix] / ww;
                g_alpha -= w1 * a.accum_coord[HW + pix] / ww;
                g_alpha -= w2 * a.accum_coord[2 * HW + pix] / ww;
                gc0 = w0 / w_final; gc1 = w1 / w_final; gc2 = w2 / w_final;
                if (a.dL_dmcoord) { gm0 = a.dL_dmcoord[pix]; gm1 = a.dL_dmcoord[HW + pix]; gm2 = a.dL_dmcoord[2 * HW + pix]; }
            }
            if constexpr (DEPTH) {
                const float wd = a.dL_ddepth ? a.dL_ddepth[pix] : 0.f;
                g_alpha -= wd * a.accum_depth[pix] / ww;
                g_t = wd / w_final / ln;
                g_mt = a.dL_dmdepth ? a.dL_dmdepth[pix] / ln : 0.f;
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
    int my_wave_max;                               // ... and this wave's own deepest one (wave-uniform): rows behind it are skipped
    {
        int m = last_contributor;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if (lane == 0) wave_max[wid] = m;
        my_wave_max = __builtin_amdgcn_readfirstlane(m);
    }
    tile_barrier();
    const int n = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));   // elements [0, n) of the range
    const int rounds = (n + BCHUNK - 1) / BCHUNK;

    float T = T_final, S = 0.f, Dprev = 0.f, last_alpha = 0.f;
    const bool has_bg = (a.bg[0] != 0.f) || (a.bg[1] != 0.f) || (a.bg[2] != 0.f);      // wave-uniform
    const float halfW = 0.5f * a.W, halfH = 0.5f * a.H;
    float* myred = red_all + wid * (NROWS * RED_STRIDE);
    // LDS byte offset of this wave's transpose buffer (the low half of the flat address of a __shared__ object is its LDS offset)
    const unsigned myred_m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)myred);
    constexpr int LPR = NROWS <= 16 ? 4 : 2;             // lanes per row of the transpose buffer
#ifndef BWD_ADJ
#define BWD_ADJ 1             // (same-box A/B, round 2: blend_bwd 82.6 -> 80.2 us against the rotated 16-lanes-apart assignment + permlane swaps)
#endif
    // ADJ (rows of <= 15 moments): the LPR lanes of a row are NEIGHBOURS (lane = 4 row + part), so the active lanes fill whole 16-lane
    // service groups from the bottom (36 lanes for 9 rows: two groups and a quarter; the last group stays empty) and the partials fold
    // with two quad-permute DPP adds; at stride 68 that assignment is conflict-free as it stands (rows 4g..4g+3 x segments 0..3 of a
    // group start 16 different multiples of four banks)
    constexpr bool ADJ = BWD_ADJ && LPR == 4 && NROWS < 16;
    const int rrow = ADJ ? (int)(lane >> 2) : (int)(lane & (64 / LPR - 1)), rpart = ADJ ? (int)(lane & 3) : (int)(lane / (64 / LPR));
    const int rseg = ADJ ? rpart : LPR == 4 ? ((rpart + (int)((SEG_ROT >> (2 * (rrow & 15))) & 3u)) & 3) : rpart;      // which column segment this lane sums
    // gacc slot of compact row `lane` (rows are emitted in slot order with the dead groups left out)
    // (the colour-only instance packs its 10 moments into slots 0..9 instead: one 64-byte atomic request per row, not two --
    //  float atomics execute at the memory side in 64-byte requests, ~20 G requests/s for the whole chip)
    int slot_of_row = ADJ ? (int)(lane >> 2) : (int)lane;
    constexpr bool COMPACT = !COORD && !DEPTH && !NORMAL;
    if (!COMPACT && !COORD && slot_of_row >= 3) slot_of_row += 9;
    if (!COMPACT && !DEPTH && slot_of_row >= 12) slot_of_row += 3;
    if (!COMPACT && !NORMAL && slot_of_row >= 15) slot_of_row += 3;

    // ---- sum of the wave's 64 per-lane values of every live moment, added to Gaussian `gid`'s accumulator row.
    // Issue slots are what this kernel runs out of (tools/ubench/scalar_cost: a scalar instruction costs the SIMD ~2.5
    // cycles like a vector one, a TAKEN branch 7.7), so the reduction is written without control flow: every lane reads a row
    // (lanes beyond the last row re-read it: same addresses as the row's own lanes, a broadcast), the folds are DPP adds, and the
    // atomic is issued under an EXEC mask set by two scalar instructions instead of a divergent `if` (s_and_saveexec + s_cbranch +
    // a taken s_branch back).  Its address is a 32-bit byte offset from the accumulator base in SGPRs (the host refuses
    // accumulator arrays of 4 GB and more).
    const int rrow_c = rrow < NROWS ? rrow : NROWS - 1;
    const float4* const my_row = (const float4*)(myred + rrow_c * RED_STRIDE + rseg * (64 / LPR));
    constexpr uint64_t ATOMIC_LANES = ADJ ? (0x1111111111111111ull & ((NROWS >= 16) ? ~0ull : ((1ull << (4 * NROWS)) - 1ull)))
                                          : ((NROWS >= 64) ? ~0ull : ((1ull << NROWS) - 1ull));
    const uint32_t slot_bytes = 4u * (uint32_t)slot_of_row;
    auto reduce_row = [&](const float (&mv)[NROWS], const uint32_t gid) {
                ColWrite<RED_STRIDE * 4, 0, NROWS>::run(myred_m0, mv);
                // every lane has written its column; the row sums below read what OTHER lanes wrote.  The wave runs in lockstep and
                // LDS operations of one wave complete in issue order, so no hardware barrier is needed -- but the compiler must not
                // move the reads above the writes: wave_barrier is a scheduling fence that emits no instruction
                __builtin_amdgcn_wave_barrier();
                // LPR lanes share a row (each sums 64/LPR columns), then LPR partials are combined across lanes
                float4 acc4 = my_row[0];
#pragma unroll
                for (int k = 1; k < 16 / LPR; k++) { const float4 t4 = my_row[k]; acc4.x += t4.x; acc4.y += t4.y; acc4.z += t4.z; acc4.w += t4.w; }
                const float part = (acc4.x + acc4.y) + (acc4.z + acc4.w);
                __builtin_amdgcn_wave_barrier();      // ... and the next splat's column writes must stay behind these row reads
                float tot;
                if constexpr (ADJ) {
                    // (s_nop 1: a DPP operand needs two wait states behind the vector instruction that wrote it; s_nop is free)
                    float t1;
                    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t1) : "v"(part));
                    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(tot) : "v"(t1));
                } else {
                    // the LPR partials of a row sit 64 / LPR lanes apart: v_permlane32_swap / v_permlane16_swap (gfx950) fold them in the
                    // vector ALU -- no trip through the LDS crossbar (ds_bpermute: 6 LDS cycles each, and a round trip of latency)
                    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                    tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
                    if constexpr (LPR == 4) {
                        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tot), __float_as_uint(tot), false, false);
                        tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
                    }
                }
                const uint32_t off = gid * (uint32_t)((COMPACT ? GACC_COMPACT_F : GACC_F) * 4) + slot_bytes;
                uint64_t saved;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_atomic_add_f32 %2, %3, %4\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(ATOMIC_LANES), "v"(off), "v"(tot), "s"(a.gacc) : "memory", "scc");
    };
    // ---- one (wave, splat) row: `R` hands out the splat's record (q0 = {x, y, conic.x, conic.y}, q1 = {conic.z, opacity, r, g},
    //      q2 = {b, ...}) and its Gaussian id (LdsRec below: the staged LDS copy)
    // `j`: staged slot (wave-uniform); a pixel takes part iff the slot lies in front of its last contributor: j > j_first (per
    // lane, per round -- no per-row scalar arithmetic on list positions); j_med: the slot of the pixel's median splat
    int j_first = 0, j_med = -1;
    // background term of dL/dalpha (backward.cu:  dL_dalpha += (-T_final / (1 - alpha)) * bg_dot): zero when the background is black
    const float bg_dot_e = has_bg ? bg_dot : 0.f;
    auto process_row = [&](const int j, auto R) {
                // colour-only instances stage the record as { x, y, conic.x, conic.y | conic.z, opacity, -, - | r, g, b, id }: three
                // ALIGNED broadcast reads (b128, b64, b128) with immediate offsets; the instances with geometry keep the 96-byte record
                const float4 q0 = R.q0();
                float cz, op, col0, col1, col2;
                float4 q1, q2;
                if constexpr (GEO) { q1 = R.q1(); cz = q1.x; op = q1.y; }
                else { const float2 t2 = *(const float2*)&R.r[1]; cz = t2.x; op = t2.y; }
                const float dx = q0.x - pixfx, dy = q0.y - pixfy;
                const float power = gauss_power(q0.z, q0.w, cz, dx, dy);
                const float G = __expf(power);
                const float alpha0 = fminf(0.99f, op * G);
                const bool valid = (j > j_first) && !(power > 0.0f) && !(alpha0 < 1.0f / 255.0f);
                if (__ballot(valid) == 0ull) return;

                uint32_t gid;
                if constexpr (GEO) { q2 = R.q2(); col0 = q1.z; col1 = q1.w; col2 = q2.x; gid = chunk_id[j]; }
                else { const float4 c4 = R.r[2]; col0 = c4.x; col1 = c4.y; col2 = c4.z; gid = __float_as_uint(c4.w); }
                // A pixel the splat does not contribute to takes part with alpha 0: then 1/(1-alpha) = 1 and T stays, w = 0, and the
                // suffix recurrence below may be advanced unconditionally -- with last_alpha = 0 its next step reproduces S bit for
                // bit (0 * D + 1 * S) -- so the row needs two selects instead of six (issue slots: DESIGN.md 5).
                const float alpha = valid ? alpha0 : 0.f;
                // 1/(1-alpha) once (v_rcp_f32, 1 ulp) instead of two IEEE divisions; 1-alpha >= 0.01
                const float inv_one_m = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * inv_one_m;
                const float w = alpha * T;
                const bool is_med = valid && (j == j_med);

                float D = col0 * gp0 + col1 * gp1 + col2 * gp2 + g_alpha;
                float dLc0 = 0, dLc1 = 0, dLc2 = 0, dLt = 0;
                float4 q3, q4, q5;
                if constexpr (GEO) { q3 = R.r[3]; q5 = R.r[5]; }
                if constexpr (COORD) {
                    q4 = R.r[4];
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
                const float dL_dopa = (D - Snew) * T + (-T_final * inv_one_m) * bg_dot_e;
                S = Snew; Dprev = D; last_alpha = alpha;
                const float q = valid ? (op * dL_dopa) * G : 0.f;
                const float qdx = q * dx, qdy = q * dy;

                // ---- transpose-reduce over the 64 pixels of the wave: row r of `myred` = the 64 per-lane values of one LIVE
                // moment (rows are compacted per template instance: 25 with every branch on, 10 for colour-only gradients)
                float mv[NROWS];
                int r = 0;
                mv[r++] = w * gp0; mv[r++] = w * gp1; mv[r++] = w * gp2;
                if constexpr (COORD) {
                    mv[r++] = dLc0; mv[r++] = dLc1; mv[r++] = dLc2;
                    mv[r++] = dLc0 * dx; mv[r++] = dLc1 * dx; mv[r++] = dLc2 * dx;
                    mv[r++] = dLc0 * dy; mv[r++] = dLc1 * dy; mv[r++] = dLc2 * dy;
                }
                if constexpr (DEPTH) { mv[r++] = dLt; mv[r++] = dLt * dx; mv[r++] = dLt * dy; }
                if constexpr (NORMAL) { mv[r++] = w * gn0; mv[r++] = w * gn1; mv[r++] = w * gn2; }
                mv[r++] = q; mv[r++] = qdx; mv[r++] = qdy;
                mv[r++] = qdx * dx; mv[r++] = qdx * dy; mv[r++] = qdy * dy;
                if constexpr (ABS) {
                    const float gxa = q0.z * qdx + q0.w * qdy;      // -dL/d(delx) of the Gaussian term
                    const float gya = cz * qdy + q0.w * qdx;
                    mv[r++] = fabsf(gxa * halfW) + fabsf(gya * halfH);
                }
                reduce_row(mv, gid);
    };
    // the record of staged splat j, read back from LDS as wave-uniform (broadcast) ds_read_b128
    struct LdsRec {
        const float4* r;
        __device__ __forceinline__ float4 q0() const { return r[0]; }
        __device__ __forceinline__ float4 q1() const { return r[1]; }
        __device__ __forceinline__ float4 q2() const { return r[2]; }
    };
    // (Measured and dropped, round 2: fetching the row's record -- wave-uniform data -- with scalar loads straight from the record array
    //  into SGPRs, one row ahead (s_load_dwordx8 + x2, the vector ALU taking the values as scalar operands, no staged copy and no
    //  broadcast ds_read_b128 at all): parity-green, blend_bwd 80.0 -> 88.5 us -- the scalar cache does not keep up with 32 waves
    //  per CU pulling a fresh 40 bytes each per row.)
    for (int i = 0; i < rounds; i++) {
        tile_barrier();
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
                if constexpr (GEO) {
                    chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = q1; chunk[tid * NQ + 2] = q2;
                    chunk[tid * NQ + 3] = src[3]; chunk[tid * NQ + 4] = src[4]; chunk[tid * NQ + 5] = src[5]; chunk_id[tid] = id;
                } else {      // (see process_row: aligned reads, the id in the colour quad)
                    chunk[tid * NQ + 0] = q0; chunk[tid * NQ + 1] = make_float4(q1.x, q1.y, 0.f, 0.f);
                    chunk[tid * NQ + 2] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
                }
                qmask = quad_reach_mask(q0, q1, tile_x0, tile_y0);
            }
        }
        if (wid < NSW) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t b = __ballot((qmask >> q) & 1u);
                if (lane == 0) quad_bits[q][wid] = b;
            }
        }
        tile_barrier();
        // positions -> slots for this round: list element e sits in slot n - 1 - i BCHUNK - e
        const int base = n - 1 - i * BCHUNK;
        j_first = base - last_contributor;                       // e < last_contributor  <=>  j > j_first
        if constexpr (COORD || DEPTH) j_med = base + 1 - (int)max_contributor;
        const int j_skip = base - my_wave_max;                    // (scalar) slots <= j_skip lie behind every pixel of this quad
        for (int sw = 0; sw < NSW; sw++) {
            uint64_t bits = uniform64(quad_bits[wid][sw]);
            const int rel = j_skip - sw * 64;
            if (rel >= 63) continue;
            if (rel >= 0) bits &= ~((2ull << rel) - 1ull);
            while (bits != 0ull) {
                const int jj = __builtin_ctzll(bits);
                asm("s_bitset0_b64 %0, %1" : "+s"(bits) : "s"(jj));      // (one scalar instruction instead of the three of bits &= bits - 1)
                const int j = sw * 64 + jj;
                uint32_t addr;                                           // LDS byte offset of the record: ONE vector multiply (s_mul + v_mov otherwise)
                if constexpr (NQ * 16 <= 64) asm("v_mul_u32_u24 %0, %1, %2" : "=v"(addr) : "s"(j), "n"(NQ * 16));
                else asm("v_mul_u32_u24 %0, %1, %2" : "=v"(addr) : "s"(j), "v"(NQ * 16));      // (96 is no inline constant: from a register)
                process_row(j, LdsRec{ (const float4*)((const char*)chunk + addr) });
            }
        }
    }
}
