// blend_bwd.hip -- backward of the tile blend for gfx950 (wave64): the stand-alone kernel (one workgroup per tile) and its dispatch.
// Replaces BACKWARD::render / renderCUDA (DGR/cuda_rasterizer/backward.cu:631-1016, dispatch 1101-1163).
// The per-tile work and its design notes live in blend_bwd_tile.h (shared with the fused tile kernel of the refine step).
#include "blend_bwd_tile.h"

template <bool COORD, bool DEPTH, bool NORMAL, bool ABS = true>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(const BlendBwdArgs a)
{
    using Cfg = BwdCfg<COORD, DEPTH, NORMAL, ABS>;
    __shared__ float4 chunk[Cfg::BCHUNK * Cfg::NQ];
    __shared__ uint32_t chunk_id[Cfg::BCHUNK];
    __shared__ uint64_t quad_bits[4][Cfg::NSW];              // [quad][staging wave]
    __shared__ int wave_max[4];
    __shared__ __attribute__((aligned(16))) float red[Cfg::RED_FLOATS];

    uint32_t tile;
    if (a.tile_order && a.tile_order[a.gx * a.gy] == 1u) {      // heavy tiles: heaviest first (common.h: build_tile_order)
        if (blockIdx.x >= (uint32_t)(a.gx * a.gy)) return;
        tile = a.tile_order[blockIdx.x];
        if (tile >= (uint32_t)(a.gx * a.gy)) return;      // (a corrupt image buffer must not turn into an out-of-bounds access)
    } else if (!tile_for_block(blockIdx.x, a.gx, a.gy, tile)) return;
    blend_bwd_tile<COORD, DEPTH, NORMAL, ABS, false>(a, tile, chunk, chunk_id, quad_bits, wave_max, red, nullptr);
}

hipError_t launch_blend_bwd(hipStream_t s, const BlendBwdArgs& a, bool coord, bool depth, bool* compact_layout, int* instance_bits)
{
    const dim3 grid(tile_grid_blocks(a.gx, a.gy)), block(256);
    // The reference instantiates (COORD, DEPTH, NORMAL) from require_coord / require_depth alone (backward.cu:1153-1160).
    // A branch whose upstream gradients are all absent (NULL = zero: the output did not take part in the loss, which is
    // the case for IGS's refine loop, whose loss only sees the colour image) contributes exact zeros everywhere, so the
    // cheaper instance without it gives the same result.
    const bool C = coord && (a.dL_dcoord || a.dL_dmcoord);
    const bool D = depth && (a.dL_ddepth || a.dL_dmdepth);
    const bool N = (coord || depth) && a.dL_dnormal;
    if (compact_layout) *compact_layout = !C && !D && !N;
    if (instance_bits) *instance_bits = (C ? 1 : 0) | (D ? 2 : 0) | (N ? 4 : 0) | ((C || (D && !N) || (!D && N) || a.want_absgrad) ? 8 : 0);
#define LAUNCH(c, d, n) hipLaunchKernelGGL((blend_bwd_kernel<c, d, n>), grid, block, 0, s, a)
    if (C) { if (D) { if (N) LAUNCH(true, true, true); else LAUNCH(true, true, false); }
             else   { if (N) LAUNCH(true, false, true); else LAUNCH(true, false, false); } }
    else   { if (D) { if (N) { if (a.want_absgrad) LAUNCH(false, true, true);
                               else hipLaunchKernelGGL((blend_bwd_kernel<false, true, true, false>), grid, block, 0, s, a); }
                      else LAUNCH(false, true, false); }
             else   { if (N) LAUNCH(false, false, true);
                      else if (a.want_absgrad) LAUNCH(false, false, false);
                      else hipLaunchKernelGGL((blend_bwd_kernel<false, false, false, false>), grid, block, 0, s, a); } }
#undef LAUNCH
    return hipGetLastError();
}
