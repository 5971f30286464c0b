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
    // (v_rcp_f32 / v_sqrt_f32, 1 ulp each, instead of the IEEE sequences of `/` and sqrtf: the margins below are a thousand times wider,
    //  and this test runs once per staged splat -- 11 % of the forward's instructions went here)
    const float inv = __builtin_amdgcn_rcpf(det);
    const float ex = __builtin_amdgcn_sqrtf(two_tau * cz * inv) * 1.001f + 0.02f;
    const float ey = __builtin_amdgcn_sqrtf(two_tau * cx * inv) * 1.001f + 0.02f;
    if (!(ex < 3.0e38f) || !(ey < 3.0e38f)) return 0xFu;
    const float lx = q0.x - ex - tile_x0, hx = q0.x + ex - tile_x0;      // reach interval in tile-local pixel coordinates
    const float ly = q0.y - ey - tile_y0, hy = q0.y + ey - tile_y0;
    const bool x0 = (hx >= 0.f) && (lx <= 7.f), x1 = (hx >= 8.f) && (lx <= 15.f);
    const bool y0 = (hy >= 0.f) && (ly <= 7.f), y1 = (hy >= 8.f) && (ly <= 15.f);
    uint32_t m = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
    if (m == 0u) return 0u;
    // ---- test 2 (coordinates relative to the splat centre; rectangle inflated by 0.02 px)
    const float nbc = -cy * __builtin_amdgcn_rcpf(cz), nba = -cy * __builtin_amdgcn_rcpf(cx);      // argmin of the form along a vertical / horizontal line
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

// Workgroup barrier of the tile kernels: __syncthreads() with its release side spelled out.
// Round 3: the forward's round loop ends with `wave_done[wid] = ...` (ds_write_b32) and begins with __syncthreads() followed by the
// read of all four flags that decides `break` -- and hipcc emitted a bare s_barrier at that loop header, with no s_waitcnt lgkmcnt(0)
// behind the store (its waitcnt scoreboard took the counter for zero across the back edge).  A wave whose store is still queued
// when the barrier opens lets the waves that read first see a stale 0: they go round again while the others break.  In the
// stand-alone forward that only costs the stragglers a redundant round (their pixels are finished; waves that have ended no
// longer count at barriers).  In the fused forward + backward kernel it is fatal: the stragglers stage FORWARD records into the LDS
// the others already use for the BACKWARD -- Gaussian ids read from that are garbage, and the accumulator atomic faults
// (dense diagnostic scene, where nearly every tile ends its forward early; found with the ROCm debug agent: LDS dump of the faulting
// workgroup).  The wait costs nothing where the compiler would have put it anyway.
__device__ __forceinline__ void tile_barrier()
{
#ifndef IGS_NO_RELEASE_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
}

// wave-uniform copy of a 64-bit value (readfirstlane returns a SIGNED int: widen through uint32_t, not int)
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
