// blend_common.h -- helpers shared by the forward and backward tile-blend kernels (device only).
#pragma once
#include "common.h"

// Gaussian exponent exactly as the reference associates it (forward.cu:555, backward.cu:847), WITHOUT fused
// multiply-adds: the sign of a power that rounds to +-1e-8 next to a splat centre decides `if (power > 0) continue`,
// so the operation order is kept identical to the oracle's to keep such threshold flips as rare as they can be.
__device__ __forceinline__ float gauss_power(float cx, float cy, float cz, float dx, float dy)
{
#pragma clang fp contract(off)
    return -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
}

// Which of the four 8x8 quads of a tile can this splat contribute to at all?  bit q = (qy << 1) | qx.
// A pair contributes only if alpha = min(0.99, o * exp(power)) >= 1/255, i.e. power >= -tau with tau = ln(255 o):
// the pixel must lie inside the ellipse  d^T conic d <= 2 tau.  Two conservative tests (margins on tau and on the
// rectangle; any non-finite / non-positive-definite conic means "all quads"), so a pair the reference would blend is
// never removed:
//  1. bounding box of the ellipse (half extents sqrt(2 tau conic.z / det), sqrt(2 tau conic.x / det)) against the quad's
//     pixel-centre rectangle;
//  2. for the quads that survive, the exact minimum of the quadratic form over that rectangle: the form is convex with
//     its minimum (0) at the splat centre, so over a rectangle that does not contain the centre the minimum lies on a
//     face turned towards the centre -- at most two edges, each a 1-D clamped parabola.
// On the bench scene test 2 removes another 17 % of the (quad, splat) rows; 98.8 % of the rows left have a contributing pixel.
__device__ __forceinline__ uint32_t quad_reach_mask(float4 q0, float4 q1, float tile_x0, float tile_y0)
{
    const float o = q1.y;
    if (o < (1.0f / 255.0f)) return 0u;                       // alpha <= o < 1/255 for every pixel
    const float cx = q0.z, cy = q0.w, cz = q1.x;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f) || !(det < 3.0e38f)) return 0xFu;
    const float two_tau = 2.0f * __logf(255.0f * o) * 1.002f + 1e-3f;
    if (!(two_tau < 3.0e38f)) return 0xFu;
    const float inv = 1.0f / det;
    const float ex = sqrtf(two_tau * cz * inv) * 1.001f + 0.02f;
    const float ey = sqrtf(two_tau * cx * inv) * 1.001f + 0.02f;
    if (!(ex < 3.0e38f) || !(ey < 3.0e38f)) return 0xFu;
    const float lx = q0.x - ex - tile_x0, hx = q0.x + ex - tile_x0;      // reach interval in tile-local pixel coordinates
    const float ly = q0.y - ey - tile_y0, hy = q0.y + ey - tile_y0;
    const bool x0 = (hx >= 0.f) && (lx <= 7.f), x1 = (hx >= 8.f) && (lx <= 15.f);
    const bool y0 = (hy >= 0.f) && (ly <= 7.f), y1 = (hy >= 8.f) && (ly <= 15.f);
    uint32_t m = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
    if (m == 0u) return 0u;
    // ---- test 2 (coordinates relative to the splat centre; rectangle inflated by 0.02 px)
    const float nbc = -cy / cz, nba = -cy / cx;               // argmin of the form along a vertical / horizontal line
    const float bx = (tile_x0 - 0.02f) - q0.x, by = (tile_y0 - 0.02f) - q0.y;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float xl = bx + (float)((q & 1) * 8), xh = xl + 7.04f;
        const float yl = by + (float)((q >> 1) * 8), yh = yl + 7.04f;
        const bool xin = (xl <= 0.f) && (xh >= 0.f), yin = (yl <= 0.f) && (yh >= 0.f);
        const float fx = xl > 0.f ? xl : xh;                                   // x of the vertical edge facing the centre
        const float dyv = fminf(fmaxf(nbc * fx, yl), yh);
        const float qv = cx * fx * fx + 2.0f * cy * fx * dyv + cz * dyv * dyv;
        const float fy = yl > 0.f ? yl : yh;                                   // y of the horizontal edge facing the centre
        const float dxh = fminf(fmaxf(nba * fy, xl), xh);
        const float qh = cx * dxh * dxh + 2.0f * cy * dxh * fy + cz * fy * fy;
        float qmin = 3.0e38f;
        if (!xin) qmin = qv;
        if (!yin) qmin = fminf(qmin, qh);
        if (xin && yin) qmin = 0.f;
        if (qmin > two_tau) m &= ~(1u << q);                                   // (a NaN keeps the quad)
    }
    return m;
}

// The same decision for the sixteen 4x4 pixel blocks of a tile: bit b = by * 4 + bx.  The blend kernels give every 16-lane
// group of a wave its own block and its own splat list, so the finer the culling the fewer lanes idle: on the bench scene a
// splat reaches 2.3 blocks on average, and the longest of a wave's four lists is 0.62x the list of its whole 8x8 quad.
// Per-axis terms are shared between the blocks (14 VALU per block on top).
__device__ __forceinline__ uint32_t block_reach_mask(float4 q0, float4 q1, float tile_x0, float tile_y0)
{
    const float o = q1.y;
    if (o < (1.0f / 255.0f)) return 0u;
    const float cx = q0.z, cy = q0.w, cz = q1.x;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f) || !(det < 3.0e38f)) return 0xFFFFu;
    const float two_tau = 2.0f * __logf(255.0f * o) * 1.002f + 1e-3f;
    if (!(two_tau < 3.0e38f)) return 0xFFFFu;
    const float inv = 1.0f / det;
    const float ex = sqrtf(two_tau * cz * inv) * 1.001f + 0.02f;
    const float ey = sqrtf(two_tau * cx * inv) * 1.001f + 0.02f;
    if (!(ex < 3.0e38f) || !(ey < 3.0e38f)) return 0xFFFFu;
    const float lx = q0.x - ex - tile_x0, hx = q0.x + ex - tile_x0;
    const float ly = q0.y - ey - tile_y0, hy = q0.y + ey - tile_y0;
    uint32_t xm = 0u, ym = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        xm |= ((hx >= (float)(4 * k)) && (lx <= (float)(4 * k + 3))) ? (1u << k) : 0u;
        ym |= ((hy >= (float)(4 * k)) && (ly <= (float)(4 * k + 3))) ? (1u << k) : 0u;
    }
    if (xm == 0u || ym == 0u) return 0u;
    const float nbc = -cy / cz, nba = -cy / cx, cy2 = 2.0f * cy;
    const float bx0 = (tile_x0 - 0.02f) - q0.x, by0 = (tile_y0 - 0.02f) - q0.y;
    float xl[4], xh[4], t1[4], e1[4], f1[4], yl[4], yh[4], t2[4], e2[4], f2[4];
    bool xin[4], yin[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        xl[k] = bx0 + (float)(4 * k); xh[k] = xl[k] + 3.04f; xin[k] = (xl[k] <= 0.f) && (xh[k] >= 0.f);
        const float fx = xl[k] > 0.f ? xl[k] : xh[k];
        t1[k] = nbc * fx; e1[k] = cx * fx * fx; f1[k] = cy2 * fx;
        yl[k] = by0 + (float)(4 * k); yh[k] = yl[k] + 3.04f; yin[k] = (yl[k] <= 0.f) && (yh[k] >= 0.f);
        const float fy = yl[k] > 0.f ? yl[k] : yh[k];
        t2[k] = nba * fy; e2[k] = cz * fy * fy; f2[k] = cy2 * fy;
    }
    uint32_t m = 0u;
#pragma unroll
    for (int by = 0; by < 4; by++) {
#pragma unroll
        for (int bx = 0; bx < 4; bx++) {
            const float dyv = fminf(fmaxf(t1[bx], yl[by]), yh[by]);
            const float qv = e1[bx] + (f1[bx] + cz * dyv) * dyv;
            const float dxh = fminf(fmaxf(t2[by], xl[bx]), xh[bx]);
            const float qh = e2[by] + (f2[by] + cx * dxh) * dxh;
            float qmin = 3.0e38f;
            if (!xin[bx]) qmin = qv;
            if (!yin[by]) qmin = fminf(qmin, qh);
            if (xin[bx] && yin[by]) qmin = 0.f;
            const bool keep = ((xm >> bx) & (ym >> by) & 1u) && !(qmin > two_tau);        // (a NaN keeps the block)
            m |= keep ? (1u << (by * 4 + bx)) : 0u;
        }
    }
    return m;
}

// Builds the 16 per-block splat lists of one staged chunk (`bmask` = block_reach_mask of the splat this thread staged, 0 if
// none).  lists[b][0..len_b) = chunk indices in staging order; wcount is scratch.  Contains two workgroup barriers; on return
// every thread may read lists / the returned length of block `my_block`.
__device__ __forceinline__ uint32_t build_block_lists(uint32_t bmask, uint32_t tid, uint32_t my_block, uint8_t (*lists)[CHUNK],
                                                      uint16_t (*wcount)[16])
{
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint64_t bal[16];
#pragma unroll
    for (int b = 0; b < 16; b++) bal[b] = __ballot((bmask >> b) & 1u);
    if (lane < 16) {
        uint32_t c = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) c = (lane == (uint32_t)b) ? (uint32_t)__popcll(bal[b]) : c;
        wcount[wid][lane] = (uint16_t)c;
    }
    __syncthreads();
    // lane b < 16 of every wave: where this wave's entries of block b start
    uint32_t off = 0;
    if (lane < 16) { for (uint32_t w = 0; w < wid; w++) off += wcount[w][lane]; }
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)off, b);
        if ((bmask >> b) & 1u) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[b], 0u));
            lists[b][base + below] = (uint8_t)tid;
        }
    }
    const uint32_t len = (uint32_t)wcount[0][my_block] + wcount[1][my_block] + wcount[2][my_block] + wcount[3][my_block];
    __syncthreads();
    return len;
}

// wave-uniform copy of a 64-bit value (readfirstlane returns a SIGNED int: widen through uint32_t, not int)
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
