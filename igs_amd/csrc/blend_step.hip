// blend_step.hip -- forward AND backward blend of a tile in ONE kernel, for the refine step with a per-pixel colour loss (L1).
//
// With `loss = mean |colour - gt|` (infer_batch.py:302 with lambda_dssim = 0, BASELINE configs[2]) the backward of a tile needs
// nothing another tile produces: dL/dcolour of a pixel is a function of that pixel alone.  So the workgroup that has just blended
// its tile front to back (blend_fwd_tile.h, the <coord, depth, normal> instance the reference's loop renders, images written as
// always) turns round and walks the same list back to front (blend_bwd_tile.h, colour-only instance, L1 fused in) with what it
// still holds in registers -- colour, final transmittance, alpha, last contributor -- instead of a second kernel that starts by
// loading colour (3), ground truth (3), alpha and the contributor count of every pixel again.  What it saves (same-box ablation of the
// stand-alone backward, round 3: empty grid 7 us, + per-pixel loads and the first barrier 16 us, + staging 25 us of its 78):
// one kernel boundary, the 32 B per pixel of the backward's prologue (44 MB per view), the contributor-count image (never
// written), and the records of the backward's first round come out of an L2 the forward has just pulled them through.
// Results are bit-identical to the two-kernel path: the same device functions run on the same values.
#include "blend_fwd_tile.h"
#include "blend_bwd_tile.h"

#ifdef BLEND_TIMELINE
// debug build only (tools/debug/blend_timeline.py): per workgroup, the 100 MHz clock at its start, between the passes and at its end
#define BLEND_TL_MARKS 4
#define BLEND_TL_WGS 16384
__device__ unsigned long long g_blend_tl[BLEND_TL_WGS * BLEND_TL_MARKS];
extern "C" int igs_debug_blend_timeline(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_blend_tl), (size_t)n * 8);
}
#define BTL(k, val) do { if (threadIdx.x == 0 && blockIdx.x < BLEND_TL_WGS) g_blend_tl[blockIdx.x * BLEND_TL_MARKS + (k)] = (val); } while (0)
#else
#define BTL(k, val) do { } while (0)
#endif
template <bool COORD, bool DEPTH, bool NORMAL, bool ABS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
blend_step_kernel(const BlendFwdArgs f, const BlendBwdArgs b)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    using Cfg = BwdCfg<false, false, false, ABS>;
    constexpr size_t FWD_BYTES = (size_t)FWD_CHUNK * (GEO ? 6 : 3) * 16;
    constexpr size_t BWD_CHUNK_BYTES = (size_t)Cfg::BCHUNK * Cfg::NQ * 16;
    constexpr size_t BWD_BYTES = BWD_CHUNK_BYTES + (size_t)Cfg::RED_FLOATS * 4;
    // the two passes use the same LDS one after the other
    __shared__ __attribute__((aligned(16))) char smem[FWD_BYTES > BWD_BYTES ? FWD_BYTES : BWD_BYTES];
    __shared__ uint64_t quad_bits_f[FWD_NLIST][FWD_NSW];
    __shared__ uint64_t quad_bits_b[4][Cfg::NSW];
    __shared__ int wave_done[4];
    __shared__ int wave_max[4];

    if (blockIdx.x == 0 && threadIdx.x == 0 && f.host_dst) {      // {R, overflow, prefilter flag} -> host-visible memory, as in blend_fwd_kernel
        // three words, then (once they have been acknowledged) the sequence word the host polls -- as system-scope atomic stores, which
        // go to the pinned host memory by themselves: a release FENCE here is an L2 write-back (buffer_wbl2), three of them in the
        // workgroup that opens the kernel (refine_ops.hip: l1_mean_kernel has the measurement of what those can cost; here: no measurable
        // difference, same-box A/B)
        __hip_atomic_store(&f.host_dst[0], f.stats_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&f.host_dst[1], f.stats_src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&f.host_dst[2], f.flag_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&f.host_dst[3], f.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the host polls this word
    }
    uint32_t tile;
    BTL(0, wall_clock64());
    // Dispatch order: tile_sort's extra workgroup wrote it (sort.hip: build_step_order -- the plain XCD-aware order with the light tiles
    // last); without it (dense scenes, huge images) the plain order.  Every reordering that needed a launch or empty workgroups of its
    // own cost more than the drain it removes (profiles/r03_blend_order_ab.txt).
    if (f.step_order) {
        tile = f.step_order[blockIdx.x];
        if (tile >= (uint32_t)(f.gx * f.gy)) return;
    } else if (!tile_for_block(blockIdx.x, f.gx, f.gy, tile)) return;
    if (f.reset_cursors && threadIdx.x == 0) f.reset_cursors[tile] = 0u;          // (the tile sort left this fill cursor for us to zero)
    BTL(3, (unsigned long long)(f.ranges[2 * tile + 1] - f.ranges[2 * tile]));
    FwdPix px;
    blend_fwd_tile<COORD, DEPTH, NORMAL, true, false>(f, tile, (float4*)smem, quad_bits_f, wave_done, px);
    tile_barrier();           // every wave is done with the forward's staged records: the backward takes the LDS over
    BTL(1, wall_clock64());
    blend_bwd_tile<false, false, false, ABS, true>(b, tile, (float4*)smem, nullptr, quad_bits_b, wave_max, (float*)(smem + BWD_CHUNK_BYTES), &px);
    BTL(2, wall_clock64());
}

// `f.skip_bwd_state` must be set (the lean forward); `b`: the colour-only backward with the L1 loss fused in (l1_gt set; l1_color,
// alphas, n_contrib are not read).  instance_bits as launch_blend_bwd reports them.
hipError_t launch_blend_step(hipStream_t s, const BlendFwdArgs& f, const BlendBwdArgs& b, bool coord, bool depth, int* instance_bits)
{
    if (!f.skip_bwd_state || !b.l1_gt || b.dL_dcoord || b.dL_dmcoord || b.dL_ddepth || b.dL_dmdepth || b.dL_dnormal || b.dL_dalpha || b.colors_precomp)
        return hipErrorInvalidValue;
    const dim3 grid(tile_grid_blocks(f.gx, f.gy)), block(256);
    if (instance_bits) *instance_bits = b.want_absgrad ? 8 : 0;
    if (coord && depth) {
        if (b.want_absgrad) hipLaunchKernelGGL((blend_step_kernel<true, true, true, true>), grid, block, 0, s, f, b);
        else hipLaunchKernelGGL((blend_step_kernel<true, true, true, false>), grid, block, 0, s, f, b);
    } else if (!coord && !depth) {
        if (b.want_absgrad) hipLaunchKernelGGL((blend_step_kernel<false, false, false, true>), grid, block, 0, s, f, b);
        else hipLaunchKernelGGL((blend_step_kernel<false, false, false, false>), grid, block, 0, s, f, b);
    } else return hipErrorInvalidValue;          // (the refine loop renders everything or nothing; mixed cases take the two-kernel path)
    return hipGetLastError();
}
