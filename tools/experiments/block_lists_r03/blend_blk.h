// blend_blk.h -- what the "block list" forms of the tile blend share (device only).
//
// The quad kernels (blend_fwd_tile.h / blend_bwd_tile.h) give a wave an 8x8 pixel quad and ONE splat per trip: on the bench scene
// a (quad, splat) row has 17 of its 64 lanes on pixels the splat reaches (tools/trip_stats.py), and a wave64 instruction costs the
// SIMD the same with 17 lanes as with 64.  Here the wave still owns the 8x8 quad, but its four 16-lane groups are the quad's four
// 4x4 pixel BLOCKS and every block walks its OWN list of the staged splats that reach it: one trip of the wave works on up to
// four different splats.  Trips per wave = the longest of its four lists: 0.63x the rows of the quad form on the bench scene
// (mean 95 splats per tile), 0.83x on the dense diagnostic scene (tools/trip_stats.py).
//
//  * lane = 16 * block + 4 * y + x  (block = 2 * by + bx inside the quad): a block is one DPP row and one quarter of the columns
//    of the backward's transpose buffer;
//  * reach test per (splat, block) while staging: bounding box of the alpha >= 1/255 ellipse AND the ellipse against the block's
//    circumscribed ball in the conic's own norm (both conservative; together within 3 % of the exact rectangle test);
//  * the lists are dense byte arrays in LDS, compacted per wave from the staged reach masks (ballot + mbcnt); a list that runs out
//    before its neighbours reads a dummy record (opacity 0), so a trip has no per-block control flow.
// Scalar instructions are as expensive to issue as vector ones on this part (tools/ubench/scalar_cost: ~2.5 cycles each, 4.3 for the
// 64-bit bit scans, 7.7 for a taken branch), which is why the lists are not four bit masks walked with s_ff1 / s_bitset0.
#pragma once
#include "blend_common.h"

#define BLK_DUMMY_OPACITY 0.0f

// position of the calling lane's pixel inside its tile, block-list lane order
struct BlkLane {
    uint32_t blk;        // 0..3: block of the quad this lane belongs to (= lane >> 4)
    uint32_t lx, ly;     // pixel inside the 8x8 quad
    uint32_t bit;        // bit of this lane's block in a splat's 16-bit reach mask: 4 * block_row + block_column of the TILE
};
__device__ __forceinline__ BlkLane blk_lane(uint32_t lane, uint32_t wid)
{
    BlkLane L;
    L.blk = lane >> 4;
    const uint32_t l = lane & 15u;
    L.lx = (L.blk & 1u) * 4u + (l & 3u);
    L.ly = (L.blk >> 1) * 4u + (l >> 2);
    L.bit = ((wid >> 1) * 2u + (L.blk >> 1)) * 4u + (wid & 1u) * 2u + (L.blk & 1u);
    return L;
}

// Which of the 16 4x4-pixel blocks of a tile can this splat contribute to?  bit = 4 * block_row + block_column.
// A pair contributes only if alpha = min(0.99, o exp(power)) >= 1/255, i.e. the pixel lies inside  d^T conic d <= 2 tau,
// tau = ln(255 o).  Two conservative tests, ANDed (a pair the reference blends is never removed):
//  1. the ellipse's bounding box against the block's pixel-centre rectangle [4k, 4k + 3]^2 (two 4-bit interval masks, outer product);
//  2. Q(c) <= (sqrt(2 tau) + rho)^2 at the block centre c, rho = the largest Q-norm of a half diagonal of the rectangle:
//     Q^(1/2) is a norm, so every point p of the rectangle has Q^(1/2)(p) >= Q^(1/2)(c) - rho.
// Any non-finite / non-positive-definite conic means "all blocks".
__device__ __forceinline__ uint32_t block_reach_mask(float4 q0, float4 q1, float tile_x0, float tile_y0)
{
    const float o = q1.y;
    if (o < (1.0f / 255.0f)) return 0u;                       // alpha <= o < 1/255 for every pixel
    const float cx = q0.z, cy = q0.w, cz = q1.x;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f) || !(det < 3.0e38f)) return 0xFFFFu;
    const float two_tau = 2.0f * __logf(255.0f * o) * 1.002f + 1e-3f;
    if (!(two_tau < 3.0e38f)) return 0xFFFFu;
    const float inv = 1.0f / det;
    const float ex = sqrtf(two_tau * cz * inv) * 1.001f + 0.02f;
    const float ey = sqrtf(two_tau * cx * inv) * 1.001f + 0.02f;
    if (!(ex < 3.0e38f) || !(ey < 3.0e38f)) return 0xFFFFu;
    const float ux = q0.x - tile_x0, uy = q0.y - tile_y0;     // splat centre in tile-local pixel coordinates
    // ---- 1. interval of block columns / rows the bounding box touches: k with 4k <= hi and 4k + 3 >= lo
    auto interval = [](float lo, float hi) -> uint32_t {
        const int klo = max(0, (int)ceilf((lo - 3.0f) * 0.25f));
        const int khi = min(3, (int)floorf(hi * 0.25f));
        return khi >= klo ? ((2u << khi) - (1u << klo)) : 0u;
    };
    const uint32_t xm = interval(ux - ex, ux + ex), ym = interval(uy - ey, uy + ey);
    if ((xm == 0u) || (ym == 0u)) return 0u;
    const uint32_t ys = (ym & 1u) | ((ym & 2u) << 3) | ((ym & 4u) << 6) | ((ym & 8u) << 9);
    const uint32_t box = xm * ys;                              // bit 4 r + k  <=>  row r and column k both touched
    // ---- 2. the conic's form at the 16 block centres (4k + 1.5, 4r + 1.5) against (sqrt(2 tau) + rho)^2
    const float h = 1.52f;                                     // half side of the pixel-centre rectangle, inflated by 0.02 px
    const float rho = sqrtf((cx + cz + 2.0f * fabsf(cy)) * (h * h));
    const float lim = sqrtf(two_tau) + rho;
    const float thr = lim * lim;
    float a[4], u[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { u[k] = (4.0f * k + 1.5f) - ux; a[k] = cx * u[k] * u[k]; }
    uint32_t ball = 0u;
#pragma unroll
    for (int r = 3; r >= 0; r--) {
        const float v = (4.0f * r + 1.5f) - uy;
        const float c = cz * v * v, d = 2.0f * cy * v;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const float Q = a[k] + c + d * u[k];
            ball = (ball << 1) | (!(Q > thr) ? 1u : 0u);       // (a NaN keeps the block)
        }
    }
    return box & ball;
}

// Compacts the staged reach masks into this wave's four dense block lists (bytes: staged slot of every splat that reaches the block, in
// staging order).  `reach`: [chunk] 16-bit masks (as uint32), zero beyond the staged count; `lists`: this wave's [4][chunk + 4] bytes,
// ALREADY filled with `dummy`; `skip_below[b]`: slots <= skip_below[b] are left out of block b's list (the backward drops what lies
// behind every pixel of a block; -1 = keep everything).  Returns the longest list's length (wave-uniform).
template <int CHUNKN>
__device__ __forceinline__ int build_block_lists(const uint32_t* __restrict__ reach, uint8_t* __restrict__ lists, uint32_t lane, uint32_t wid,
                                                 const int (&skip_below)[4])
{
    constexpr int NW = (CHUNKN + 63) / 64, LSTRIDE = CHUNKN + 4;
    int len[4] = {0, 0, 0, 0};
#pragma unroll
    for (int sw = 0; sw < NW; sw++) {
        const uint32_t slot = (uint32_t)sw * 64u + lane;
        const uint32_t r = (CHUNKN % 64 == 0 || slot < (uint32_t)CHUNKN) ? reach[slot] : 0u;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t bit = ((wid >> 1) * 2u + ((uint32_t)b >> 1)) * 4u + (wid & 1u) * 2u + ((uint32_t)b & 1u);      // wave-uniform
            const bool on = ((r >> bit) & 1u) && ((int)slot > skip_below[b]);
            const uint64_t m = __ballot(on);
            const int pos = len[b] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (on) lists[b * LSTRIDE + pos] = (uint8_t)slot;
            len[b] += __builtin_popcountll(m);
        }
    }
    return max(max(len[0], len[1]), max(len[2], len[3]));
}
