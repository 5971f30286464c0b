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
#include "blend_common.h"

// 192 splats staged per round by the first three waves: 18 KB of LDS per workgroup instead of 24.5 (256 splats), so that eight
// workgroups fit a CU; together with the 64-VGPR budget below (8 waves per SIMD instead of 6; two dwords spill outside the row
// loop) the fuller machine hides more of a row's dependency chain: 64.4 -> 61.0 us on the bench scene (same-box A/B, round 2)
#undef CHUNK
#define CHUNK 192
#define NSW (CHUNK / 64)     // staging waves

// LEAN = the refine step with a colour-only loss (BlendFwdArgs::skip_bwd_state): the geometry branches' backward state (accumulated
// coordinate / depth, normal length, median index: 24 of 88 bytes per pixel) is not stored -- the colour-only backward instance
// reads none of it.  (Storing the coordinate / depth / normal maps, which nothing reads again in such a step, with the
// non-temporal policy was measured too: 59.5 -> 86 us -- a quad row is 32 bytes of a line, and nt stores give up the L2's
// write combining.)
template <bool COORD, bool DEPTH, bool NORMAL, bool LEAN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
blend_fwd_kernel(const BlendFwdArgs a)
{
    constexpr bool GEO = COORD || DEPTH || NORMAL;
    constexpr int NQ = GEO ? 6 : 3;                     // float4 per staged record
    __shared__ float4 chunk[CHUNK * NQ];
    __shared__ uint64_t quad_bits[4][NSW];              // [quad][staging wave]
    __shared__ int wave_done[4];
    __shared__ uint32_t order_hist[2 * LOAD_CLASSES];

    if (blockIdx.x == 0 && threadIdx.x == 0 && a.host_dst) {
        a.host_dst[0] = a.stats_src[0]; a.host_dst[1] = a.stats_src[1]; a.host_dst[2] = a.flag_src[0];
        __threadfence_system();
        __hip_atomic_store(&a.host_dst[3], a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // the host polls this word
        __threadfence_system();
    }
    // workgroup 0, on the side: the tile order of the backward blend (common.h: build_tile_order), before it turns to its own tile
    if (blockIdx.x == 0 && a.tile_order) build_tile_order(a.ranges, (uint32_t)(a.gx * a.gy), a.tile_order, order_hist);
    uint32_t tile;
    if (!tile_for_block(blockIdx.x, a.gx, a.gy, tile)) return;
    const uint32_t tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t px = tx * TILE + (wid & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE + (wid >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixfx = (float)px, pixfy = (float)py;
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    const uint2 range = ((const uint2*)a.ranges)[tile];
    const int n = (int)(range.y - range.x);      // (a tile that overflowed its slab has an empty range; the frame is then redone)
    const int rounds = (n + CHUNK - 1) / CHUNK;

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0, max_contributor = 0xFFFFFFFFu;
    float C0 = 0, C1 = 0, C2 = 0, weight = 0;
    float Co0 = 0, Co1 = 0, Co2 = 0, mC0 = 0, mC1 = 0, mC2 = 0, Depth = 0, mDepth = 0, N0 = 0, N1 = 0, N2 = 0;

    if (tid < 4) wave_done[tid] = 0;
    for (int i = 0; i < rounds; i++) {
        __syncthreads();                                           // previous chunk consumed, wave_done published
        if (wave_done[0] & wave_done[1] & wave_done[2] & wave_done[3]) break;
        const int progress = i * CHUNK + (int)tid;
        uint32_t qmask = 0;
        if (tid < CHUNK && progress < n) {
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
        if (wid < NSW) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t b = __ballot((qmask >> q) & 1u);
                if (lane == 0) quad_bits[q][wid] = b;
            }
        }
        __syncthreads();
        if (__ballot(!done) != 0ull) {
            bool wave_finished = false;
            for (int sw = 0; sw < NSW && !wave_finished; sw++) {
                uint64_t bits = uniform64(quad_bits[wid][sw]);     // wave-uniform
                while (bits != 0ull) {
                    const int j = sw * 64 + __builtin_ctzll(bits);
                    bits &= bits - 1;
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
                    // straight-line accumulate: after the per-quad culling nearly every splat that gets here contributes
                    const uint32_t contributor = (uint32_t)(i * CHUNK + j + 1);
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
                            const float t = q2.y + (q2.z * dx + q2.w * dy);
                            Depth += t * aT;
                        }
                        if constexpr (NORMAL) { N0 += q3.w * aT; N1 += q5.z * aT; N2 += q5.w * aT; }
                        // only the index of the median splat is tracked here; its coordinate and depth are re-evaluated once
                        // per pixel after the loop (forward.cu:640-652 stores them inside the loop)
                        max_contributor = before_median ? contributor : max_contributor;
                    }
                    weight += aT;
                    T = contrib ? test_T : T;
                    last_contributor = contrib ? contributor : last_contributor;
                }
                // "has every pixel of the quad saturated?" is asked once per 64 staged splats, not per row: the ballot of the
                // `done` mask costs two VALU ops and a branch in the middle of the row (measured: 71.5 -> 65 us); at most the
                // rest of one 64-splat word is blended into lanes that no longer take anything
                if (__ballot(!done) == 0ull) wave_finished = true;
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
        a.n_contrib[pix] = last_contributor;
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
