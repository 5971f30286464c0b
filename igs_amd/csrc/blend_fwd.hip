// blend_fwd.hip -- tile-wise front-to-back alpha compositing for gfx950 (wave64).
// Replaces FORWARD::render / renderCUDA (DGR/cuda_rasterizer/forward.cu:428-742).
//
// One 256-thread workgroup per 16x16 tile; each of its 4 waves owns an 8x8 pixel quad (lane = pixel).
// Splats are staged 192 at a time into LDS as packed 96-byte records gathered as whole 128-byte lines from the
// per-Gaussian record array.  While staging, the thread that holds a splat also decides which of the four quads
// it can reach at all (bounding box of the alpha >= 1/255 ellipse against the quad's pixel-centre rectangle,
// conservative); a ballot turns that into one 256-bit "to do" set per quad, so each wave walks only its own
// splats with scalar bit-scans (s_ff1) and reads them back as wave-uniform (broadcast) ds_read_b128, the next
// record being fetched while the current one is blended.
// Tiles are handed to workgroups through an XCD-aware remap so that the tiles sharing an L2 are neighbours.
// The per-tile work itself lives in blend_fwd_tile.h (shared with the fused tile kernel of the refine step, blend_step.hip).
#include "blend_fwd_tile.h"

template <bool COORD, bool DEPTH, bool NORMAL, bool LEAN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
blend_fwd_kernel(const BlendFwdArgs a)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    __shared__ float4 chunk[FWD_CHUNK * (GEO ? 6 : 3)];
    __shared__ uint64_t quad_bits[FWD_NLIST][FWD_NSW];      // [quad or half-quad][staging wave]
    __shared__ int wave_done[4];
    __shared__ uint32_t order_hist[2 * LOAD_CLASSES];

    if (blockIdx.x == 0 && threadIdx.x == 0 && a.host_dst) {
        // three words, then (once they have been acknowledged) the sequence word the host polls -- as system-scope atomic stores, which
        // go to the pinned host memory by themselves: a release FENCE here is an L2 write-back (buffer_wbl2), three of them in the
        // workgroup that opens the kernel (refine_ops.hip: l1_mean_kernel has the measurement of what those can cost; here: no measurable
        // difference, same-box A/B)
        __hip_atomic_store(&a.host_dst[0], a.stats_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_dst[1], a.stats_src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_dst[2], a.flag_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.host_dst[3], a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the host polls this word
    }
    // workgroup 0, on the side: the tile order of the backward blend (common.h: build_tile_order), before it turns to its own tile
    if (blockIdx.x == 0 && a.tile_order) build_tile_order(a.ranges, (uint32_t)(a.gx * a.gy), a.tile_order, order_hist);
    uint32_t tile;
    if (!tile_for_block(blockIdx.x, a.gx, a.gy, tile)) return;
    FwdPix px;
    blend_fwd_tile<COORD, DEPTH, NORMAL, LEAN, true>(a, tile, chunk, quad_bits, wave_done, px);
}

hipError_t launch_blend_fwd(hipStream_t s, const BlendFwdArgs& a, bool coord, bool depth)
{
    const dim3 grid(tile_grid_blocks(a.gx, a.gy)), block(256);
    // dispatch of forward.cu:732-739: NORMAL is on whenever COORD or DEPTH is
    if (a.skip_bwd_state && coord && depth) hipLaunchKernelGGL((blend_fwd_kernel<true, true, true, true>), grid, block, 0, s, a);
    else if (coord && depth) hipLaunchKernelGGL((blend_fwd_kernel<true, true, true>), grid, block, 0, s, a);
    else if (coord) hipLaunchKernelGGL((blend_fwd_kernel<true, false, true>), grid, block, 0, s, a);
    else if (depth) hipLaunchKernelGGL((blend_fwd_kernel<false, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((blend_fwd_kernel<false, false, false>), grid, block, 0, s, a);
    return hipGetLastError();
}
