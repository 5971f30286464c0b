// sort.hip -- binning stage for gfx950: stable LSD radix sort, instance emission, tile ranges.
//
// The reference builds 64-bit (tile | depth) keys for every (Gaussian, tile) instance and sorts all R of them with
// cub::DeviceRadixSort over 32+log2(T) bits (rasterizer_impl.cu:70-111, 373-381): 6 passes over 12-byte pairs at
// 1352x1014.  Here the same ORDER is produced with far less traffic:
//   1. sort the P Gaussians by depth bits (32-bit keys, 4 passes over P pairs);
//   2. emit the instances in that order (coalesced, load-balanced: a block of 256 Gaussians scatters its instances
//      cooperatively instead of one thread looping over its tiles);
//   3. stable-sort the R instances by tile id only (ceil(log2 T / 8) = 2 passes over 8-byte pairs).
// Stability of every pass makes the result identical to a stable sort on (tile, depth) with ties in Gaussian-index
// order, which is what the reference's stable LSD sort yields.
#include "common.h"

#define RADIX 256

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, uint32_t mask) { return (key >> shift) & mask; }

// Geometry of one sort: nb blocks, each owning `per` consecutive elements (a multiple of the 2048-element sub-tile).
void sort_geometry(uint32_t n, uint32_t* nb_out, uint32_t* per_out)
{
    uint32_t nb = (n + SORT_TILE - 1) / SORT_TILE;
    if (nb > SORT_MAX_BLOCKS) nb = SORT_MAX_BLOCKS;
    if (nb == 0) nb = 1;
    uint32_t per = (n + nb - 1) / nb;
    per = (per + SORT_TILE - 1) / SORT_TILE * SORT_TILE;
    if (per == 0) per = SORT_TILE;
    nb = (n + per - 1) / per;
    if (nb == 0) nb = 1;
    *nb_out = nb; *per_out = per;
}

// ---- single-block exclusive scan (in place) of up to a few 100k uint32 (used for the per-256 instance counts) ----
__global__ void __launch_bounds__(1024)
exclusive_scan_kernel(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ total)
{
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t beg = min(n, tid * per), end = min(n, beg + per);
    uint32_t sum = 0;
    for (uint32_t i = beg; i < end; i++) sum += data[i];
    uint32_t v = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off, 64); if (lane >= (uint32_t)off) v += t; }
    if (lane == 63) wave_sums[wid] = v;
    __syncthreads();
    if (tid == 0) { uint32_t c = 0; for (int w = 0; w < 16; w++) { uint32_t t = wave_sums[w]; wave_sums[w] = c; c += t; } carry_s = c; }
    __syncthreads();
    uint32_t run = v - sum + wave_sums[wid];
    for (uint32_t i = beg; i < end; i++) { uint32_t t = data[i]; data[i] = run; run += t; }
    if (total && tid == 0) *total = carry_s;
}

// ---- per-block digit histogram for the passes after the first (LDS atomics absorb low-entropy digits) ----
__global__ void __launch_bounds__(256)
radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t per_block, int shift, uint32_t mask, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t beg = blockIdx.x * per_block;
    const uint32_t end = min(n, beg + per_block);
    for (uint32_t e = beg + threadIdx.x; e < end; e += 256) atomicAdd(&h[digit_of(keys[e], shift, mask)], 1u);
    __syncthreads();
    hist[blockIdx.x * RADIX + threadIdx.x] = h[threadIdx.x];
}

// ---- one stable radix pass ----
// hist_cur[b * 256 + d] = number of keys with digit d in block b's slice (for the first pass it was accumulated with
// global atomics by the kernel that PRODUCED the keys -- preprocess / emit --, for later passes by radix_hist_kernel).
// There is no scan launch: every block derives its digit bases from the table itself, then ranks its keys stably
// (wave-level match-any + per-wave LDS counters) and scatters.
__global__ void __launch_bounds__(256)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                     uint32_t* __restrict__ vals_out, uint32_t n, uint32_t per_block, int shift, uint32_t mask,
                     const uint32_t* __restrict__ hist_cur)
{
    __shared__ uint32_t wave_cnt[4][RADIX];
    __shared__ uint32_t base[RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    {   // base[d] = (keys with a smaller digit, all blocks) + (keys with digit d in earlier blocks)
        // wave w sums the rows b = w, w+4, ... of the table, 4 digits (16 B) per lane, several rows in flight
        const uint32_t nb = gridDim.x, me = blockIdx.x;
        uint4 tot = make_uint4(0, 0, 0, 0), bef = make_uint4(0, 0, 0, 0);
        const uint4* tab = (const uint4*)hist_cur;
#pragma unroll 4
        for (uint32_t b = wid; b < nb; b += 4) {
            const uint4 c = tab[b * (RADIX / 4) + lane];
            tot.x += c.x; tot.y += c.y; tot.z += c.z; tot.w += c.w;
            if (b < me) { bef.x += c.x; bef.y += c.y; bef.z += c.z; bef.w += c.w; }
        }
        // wave_cnt[w][0..255] <- totals, base (via a second pass) <- befores
        ((uint4*)wave_cnt[wid])[lane] = tot;
        __syncthreads();
        const uint32_t total = wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
        __syncthreads();
        ((uint4*)wave_cnt[wid])[lane] = bef;
        __syncthreads();
        const uint32_t before = wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
        // exclusive scan of `total` over the 256 digits
        uint32_t v = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if (lane >= (uint32_t)off) v += t; }
        __syncthreads();
        if (lane == 63) base[wid] = v;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wid; w++) woff += base[w];
        __syncthreads();
        base[tid] = (v - total) + woff + before;
    }
    const uint32_t beg = blockIdx.x * per_block;
    const uint32_t end = min(n, beg + per_block);
    for (uint32_t sub = beg; sub < end; sub += SORT_TILE) {
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++) wave_cnt[w][tid] = 0;
        __syncthreads();
        uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
        const uint32_t e0 = sub + wid * (SORT_ITEMS * 64) + lane;
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; i++) {
            const uint32_t e = e0 + i * 64;
            const bool valid = e < end;
            key[i] = valid ? keys_in[e] : 0xFFFFFFFFu;
            val[i] = valid ? vals_in[e] : 0u;
        }
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; i++) {
            const uint32_t e = e0 + i * 64;
            const bool valid = e < end;
            const uint32_t d = digit_of(key[i], shift, mask);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t r = __popcll(peers & lt_mask);
            uint32_t old = 0;
            if (valid) {
                old = wave_cnt[wid][d];                       // every peer reads before the leader updates
                if (r == 0) wave_cnt[wid][d] = old + __popcll(peers);
            }
            rank[i] = old + r;
        }
        __syncthreads();
        {   // digit `tid`: turn per-wave counts into per-wave bases, advance the running base
            uint32_t run = base[tid];
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint32_t t = wave_cnt[w][tid]; wave_cnt[w][tid] = run; run += t; }
            base[tid] = run;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; i++) {
            const uint32_t e = e0 + i * 64;
            if (e < end) {
                const uint32_t pos = wave_cnt[wid][digit_of(key[i], shift, mask)] + rank[i];
                keys_out[pos] = key[i];
                vals_out[pos] = val[i];
            }
        }
    }
}

// hist: SORT_MAX_PASSES tables of RADIX*SORT_MAX_BLOCKS counters; table 0 (zeroed by the caller before the producer ran)
// already holds the first pass's histogram, filled by the kernel that wrote keys_a.
hipError_t radix_sort_pairs(hipStream_t s, uint32_t n, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                            uint32_t* hist, int bit_lo, int bit_hi, uint32_t** out_keys, uint32_t** out_vals)
{
    uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    if (n > 0) {
        uint32_t nb, per;
        sort_geometry(n, &nb, &per);
        int pass = 0;
        for (int lo = bit_lo; lo < bit_hi; lo += 8, pass++) {
            const int nbits = (bit_hi - lo) < 8 ? (bit_hi - lo) : 8;
            const uint32_t mask = (1u << nbits) - 1u;
            uint32_t* hcur = hist + (size_t)pass * RADIX * SORT_MAX_BLOCKS;
            if (pass > 0) hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(256), 0, s, kin, n, per, lo, mask, hcur);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, kin, vin, kout, vout, n, per, lo, mask, hcur);
            uint32_t* t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    *out_keys = kin; *out_vals = vin;
    return hipGetLastError();
}

// ---- instance counts in depth order: blocksum[b] = sum over sorted positions [256b, 256b+256) of tiles[order[s]] ----
__global__ void __launch_bounds__(256)
count_sorted_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ blocksum)
{
    __shared__ uint32_t ws[4];
    const int s = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = (s < P) ? tiles[order[s]] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) blocksum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
hipError_t launch_count_sorted(hipStream_t s, int P, const uint32_t* order, const uint32_t* tiles, uint32_t* blocksum)
{
    hipLaunchKernelGGL(count_sorted_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, order, tiles, blocksum);
    return hipGetLastError();
}
hipError_t launch_scan_blocksums(hipStream_t s, int nblocks, uint32_t* blocksum)
{
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, s, blocksum, (uint32_t)nblocks, (uint32_t*)nullptr);
    return hipGetLastError();
}

// ---- emission: a block owns 256 depth-consecutive Gaussians and writes all their (tile, id) pairs cooperatively ----
__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, int gx, int gy, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles,
                      const uint32_t* __restrict__ blocksum, const float* __restrict__ rec, const int* __restrict__ radii,
                      uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ hist0,
                      uint32_t per_block, uint32_t mask0)
{
    __shared__ uint32_t incl[256];     // inclusive scan of instance counts
    __shared__ uint32_t gid[256];
    __shared__ int rx0[256], ry0[256], rw[256];
    __shared__ uint32_t ws[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = blockIdx.x * 256 + tid;
    uint32_t cnt = 0, g = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (s < P) {
        g = order[s];
        cnt = tiles[g];
        if (cnt) {
            const float2 xy = *(const float2*)(rec + (size_t)g * REC_F);
            get_rect(xy.x, xy.y, radii[g], gx, gy, x0, y0, x1, y1);    // same call as the reference's duplicateWithKeys
        }
    }
    uint32_t v = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off, 64); if (lane >= (uint32_t)off) v += t; }
    if (lane == 63) ws[wid] = v;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wid; w++) woff += ws[w];
    const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3];
    incl[tid] = v + woff;
    gid[tid] = g; rx0[tid] = x0; ry0[tid] = y0; rw[tid] = x1 - x0;
    __syncthreads();
    const uint32_t base = blocksum[blockIdx.x];
    for (uint32_t k = tid; k < total; k += 256) {
        // smallest j with incl[j] > k
        uint32_t lo = 0, hi = 255;
#pragma unroll
        for (int it = 0; it < 8; it++) { const uint32_t mid = (lo + hi) >> 1; if (incl[mid] > k) hi = mid; else lo = mid + 1; }
        const uint32_t j = lo;
        const uint32_t local = k - (j ? incl[j - 1] : 0u);
        const int w = rw[j];
        const int ty = ry0[j] + (int)(local / (uint32_t)w), tx = rx0[j] + (int)(local % (uint32_t)w);
        const uint32_t tkey = (uint32_t)(ty * gx + tx);
        tile_keys[base + k] = tkey;
        vals[base + k] = gid[j];
        if (hist0) atomicAdd(&hist0[((base + k) / per_block) * RADIX + (tkey & mask0)], 1u);      // first radix pass's histogram
    }
}
hipError_t launch_emit_instances(hipStream_t s, int P, int gx, int gy, const uint32_t* order, const uint32_t* tiles,
                                 const uint32_t* blocksum, const float* rec, const int* radii, uint32_t* tile_keys,
                                 uint32_t* vals, uint32_t* hist0, uint32_t per_block, uint32_t mask0)
{
    hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, gy, order, tiles, blocksum, rec,
                       radii, tile_keys, vals, hist0, per_block, mask0);
    return hipGetLastError();
}

// ---- per-tile [start,end) in the sorted list (identifyTileRanges, rasterizer_impl.cu:151-173; ranges pre-zeroed) ----
__global__ void __launch_bounds__(256)
tile_ranges_kernel(uint32_t R, const uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ ranges)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= R) return;
    const uint32_t cur = tile_keys[idx];
    if (idx == 0) ranges[2 * cur] = 0;
    else {
        const uint32_t prev = tile_keys[idx - 1];
        if (cur != prev) { ranges[2 * prev + 1] = idx; ranges[2 * cur] = idx; }
    }
    if (idx == R - 1) ranges[2 * cur + 1] = R;
}
hipError_t launch_tile_ranges(hipStream_t s, uint32_t R, const uint32_t* tile_keys, uint32_t* ranges)
{
    if (R == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, tile_keys, ranges);
    return hipGetLastError();
}

// =====================================================================================================================
// Slab binning (default path).  The (tile, depth) order of the reference's global sort is produced per tile instead:
//   preprocess drops every instance into its tile's fixed-capacity slab with one atomic (slot = tile_count[t]++)
//   ->  tile_sort_kernel sorts each slab in LDS by the 64-bit key (depth bits << 32 | Gaussian id) and writes the tile's range.
// That total order is exactly what a stable sort on (tile, depth) yields for instances emitted in Gaussian-index order
// (rasterizer_impl.cu:70-111,376-381): ties on identical depth bits resolve by Gaussian index.  Every instance crosses HBM
// once as an 8-byte pair and once as a 4-byte id, and nothing needs the host to know R before it is launched.
// =====================================================================================================================

// bitonic network over keys[0..N) by `nthreads` cooperating threads (thread index t); SYNC() separates the stages
#define BITONIC_SORT(keys, N, t, nthreads, SYNC)                                                        \
    for (uint32_t bs_k = 2; bs_k <= (N); bs_k <<= 1) {                                                  \
        for (uint32_t bs_j = bs_k >> 1; bs_j > 0; bs_j >>= 1) {                                         \
            for (uint32_t bs_p = (t); bs_p < ((N) >> 1); bs_p += (nthreads)) {                          \
                const uint32_t bs_i = ((bs_p & ~(bs_j - 1)) << 1) | (bs_p & (bs_j - 1));                \
                const uint32_t bs_q = bs_i | bs_j;                                                      \
                const uint64_t bs_a = (keys)[bs_i], bs_b = (keys)[bs_q];                                \
                const bool bs_up = (bs_i & bs_k) == 0;                                                  \
                if ((bs_a > bs_b) == bs_up) { (keys)[bs_i] = bs_b; (keys)[bs_q] = bs_a; }               \
            }                                                                                           \
            SYNC;                                                                                       \
        }                                                                                               \
    }

// Bitonic sort of 64*E keys held in registers by one wave: element i = lane*E + r lives in v[r] of `lane`.  Strides below E
// are register-to-register, strides of E and more are lane exchanges (partner lane = lane ^ stride/E): no LDS, no barriers,
// every loop bound a compile-time constant.
// value of lane (lane ^ LM) without a trip through the LDS crossbar (ds_bpermute: ~6 LDS cycles per CU each, and a wave sorting 512-1024
// keys makes 336-672 of them): quad permutes for LM 1 / 2, a row rotate for 8, two row shifts and a select for 4, and the gfx950 row
// swaps for 16 / 32 (v_permlane16_swap / v_permlane32_swap leave each half of the exchange in one of their two results)
#ifndef SORT_DPP
#define SORT_DPP 1
#endif
template <int LM>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t x, uint32_t lane)
{
#if SORT_DPP
    if constexpr (LM == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
    else if constexpr (LM == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
    else if constexpr (LM == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);      // row_ror:8
    else if constexpr (LM == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0xF, false);                 // row_shl:4 -> lane + 4
        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);                 // row_shr:4 -> lane - 4
        return (lane & 4u) ? dn : up;
    } else if constexpr (LM == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16u) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32u) ? r[0] : r[1];
    }
#else
    return (uint32_t)__shfl_xor((int)x, LM, 64);
#endif
}
template <int LM>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, uint32_t lane)
{
    return (uint64_t)lane_xor32<LM>((uint32_t)v, lane) | ((uint64_t)lane_xor32<LM>((uint32_t)(v >> 32), lane) << 32);
}

template <int E>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&v)[E], uint32_t lane, uint32_t ibase = 0u)      // ibase: index of the wave's first key in a larger network
{
#pragma unroll
    for (int k = 2; k <= 64 * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const int lm = j / E;
                const bool lower = (lane & (uint32_t)lm) == 0;                    // this element is the lower one of its pair
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const uint32_t i = ibase + lane * E + r;
                    const uint64_t o = lm == 1 ? lane_xor64<1>(v[r], lane) : lm == 2 ? lane_xor64<2>(v[r], lane) : lm == 4 ? lane_xor64<4>(v[r], lane)
                                     : lm == 8 ? lane_xor64<8>(v[r], lane) : lm == 16 ? lane_xor64<16>(v[r], lane) : lane_xor64<32>(v[r], lane);
                    const bool up = (i & (uint32_t)k) == 0;
                    const uint64_t lo = v[r] < o ? v[r] : o, hi = v[r] < o ? o : v[r];
                    v[r] = (lower == up) ? lo : hi;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int q = r ^ j;
                    if (q > r) {
                        const uint32_t i = ibase + lane * E + r;
                        const bool up = (i & (uint32_t)k) == 0;
                        const uint64_t x = v[r], y = v[q];
                        const bool sw = (x > y) == up;
                        v[r] = sw ? y : x; v[q] = sw ? x : y;
                    }
                }
            }
        }
    }
}
// stages j = 32 E ... 1 of level k >= 128 E of that larger network: a bitonic merge inside the wave's 64 E keys (direction: one per wave)
template <int E>
__device__ __forceinline__ void wave_bitonic_merge(uint64_t (&v)[E], uint32_t lane, uint32_t ibase, uint32_t k)
{
    const bool up = (ibase & k) == 0;
#pragma unroll
    for (int j = 32 * E; j > 0; j >>= 1) {
        if (j >= E) {
            const int lm = j / E;
            const bool lower = (lane & (uint32_t)lm) == 0;
#pragma unroll
            for (int r = 0; r < E; r++) {
                const uint64_t o = lm == 1 ? lane_xor64<1>(v[r], lane) : lm == 2 ? lane_xor64<2>(v[r], lane) : lm == 4 ? lane_xor64<4>(v[r], lane)
                                 : lm == 8 ? lane_xor64<8>(v[r], lane) : lm == 16 ? lane_xor64<16>(v[r], lane) : lane_xor64<32>(v[r], lane);
                const uint64_t lo = v[r] < o ? v[r] : o, hi = v[r] < o ? o : v[r];
                v[r] = (lower == up) ? lo : hi;
            }
        } else {
#pragma unroll
            for (int r = 0; r < E; r++) {
                const int q = r ^ j;
                if (q > r) {
                    const uint64_t x = v[r], y = v[q];
                    const bool sw = (x > y) == up;
                    v[r] = sw ? y : x; v[q] = sw ? x : y;
                }
            }
        }
    }
}
// 256 E keys by the four waves of a workgroup: every wave sorts its 64 E keys in registers (direction as the network wants it), then the
// two levels that span waves exchange through LDS -- three exchanges with a barrier pair each, where the all-LDS network
// (BITONIC_SORT) has 45-66 barrier-separated stages -- and finish inside the waves again.
template <int E>
__device__ __forceinline__ void wg_sort_tile(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n, uint64_t* skeys, uint32_t tid, uint32_t id_max)
{
    const uint32_t lane = tid & 63, wid = tid >> 6, ibase = wid * 64u * E;
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; r++) { const uint32_t i = ibase + lane * E + r; v[r] = i < n ? src[i] : ~0ull; }
    wave_bitonic_sort<E>(v, lane, ibase);
#pragma unroll
    for (uint32_t k = 128u * E; k <= 256u * E; k <<= 1) {
        const bool up = (ibase & k) == 0;
#pragma unroll
        for (uint32_t j = k >> 1; j >= 64u * E; j >>= 1) {                 // the partner sits in another wave: same lane, same register
#pragma unroll
            for (int r = 0; r < E; r++) skeys[ibase + lane * E + r] = v[r];
            __syncthreads();
            const bool lower = (ibase & j) == 0;
#pragma unroll
            for (int r = 0; r < E; r++) {
                const uint64_t o = skeys[(ibase ^ j) + lane * E + r];
                const uint64_t lo = v[r] < o ? v[r] : o, hi = v[r] < o ? o : v[r];
                v[r] = (lower == up) ? lo : hi;
            }
            __syncthreads();
        }
        wave_bitonic_merge<E>(v, lane, ibase, k);
    }
#pragma unroll
    for (int r = 0; r < E; r++) { const uint32_t i = ibase + lane * E + r; if (i < n) dst[i] = min((uint32_t)v[r], id_max); }
}
template <int E>
__device__ __forceinline__ void wave_sort_tile(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n, uint32_t lane, uint32_t id_max)
{
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; r++) { const uint32_t i = lane * E + r; v[r] = i < n ? src[i] : ~0ull; }
    wave_bitonic_sort<E>(v, lane);
#pragma unroll
    for (int r = 0; r < E; r++) { const uint32_t i = lane * E + r; if (i < n) dst[i] = min((uint32_t)v[r], id_max); }
}

// A tile of 257-320 instances by one wave: the first 256 keys sorted in registers (E = 4), the up to 64 others one per lane, merged BY RANK
// instead of by a 512-slot network (E = 8: 2 400 dependent instructions, 13.5 us -- the five such tiles of the bench scene were the
// tile sort's whole span behind the 10.9 us of everything else, profiles/r03_sort_timeline.txt).  Keys are distinct (the id is part of
// the key), so the final position of a key is the number of keys below it: its index among the sorted 256 plus the extras below it, or,
// for an extra, the main keys below it (a ballot per register) plus the extras below it.  One trip per extra, ~20 instructions.
__device__ __forceinline__ void wave_sort_tile_320(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n, uint32_t lane, uint32_t id_max)
{
    uint64_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = src[lane * 4 + r];                 // (n > 256: all there)
    const uint32_t nx = n - 256u;
    const uint64_t x = lane < nx ? src[256u + lane] : ~0ull;
    wave_bitonic_sort<4>(v, lane);
    uint32_t mcnt[4] = { 0u, 0u, 0u, 0u }, xcnt = 0u, xmain = 0u;
    for (uint32_t j = 0; j < nx; j++) {                                     // (uniform trip count)
        const uint64_t xj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)j)
                          | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)j) << 32);
        uint32_t below = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool lt = v[r] < xj;
            below += (uint32_t)__popcll(__ballot(lt));
            mcnt[r] += lt ? 0u : 1u;
        }
        xcnt += xj < x ? 1u : 0u;
        xmain = lane == j ? below : xmain;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) dst[lane * 4 + r + mcnt[r]] = min((uint32_t)v[r], id_max);
    if (lane < nx) dst[xmain + xcnt] = min((uint32_t)x, id_max);
}

#ifdef SORT_TIMELINE
// debug build only (tools/debug/sort_timeline.py): per tile, its size and the 100 MHz clock at the marks of its workgroup's life
#define SORT_TL_MARKS 6
#define SORT_TL_TILES 8192
__device__ unsigned long long g_sort_tl[SORT_TL_TILES * SORT_TL_MARKS];
extern "C" int igs_debug_sort_timeline(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sort_tl), (size_t)n * 8);
}
#define STL(k, val) do { if (threadIdx.x == 0 && blockIdx.x < SORT_TL_TILES) g_sort_tl[blockIdx.x * SORT_TL_MARKS + (k)] = (val); } while (0)
#define STLT(t, k, val) do { if ((threadIdx.x & 63) == 0 && (t) < SORT_TL_TILES) g_sort_tl[(t) * SORT_TL_MARKS + (k)] = (val); } while (0)
#else
#define STL(k, val) do { } while (0)
#define STLT(t, k, val) do { } while (0)
#endif
// A workgroup of 4 waves owns 4 consecutive tiles.  Tiles of up to TILE_SORT_WAVE instances (nearly all of them) are sorted
// by one wave each in registers; the few denser ones (up to TILE_SORT_SMALL) by the whole workgroup in LDS afterwards.
// Tiles in (TILE_SORT_SMALL, slab] are left to tile_sort_big_kernel; tiles that overflowed their slab get an empty range and
// report their size in stats[1] (the host then redoes the frame with larger slabs).
// The binning counters CLEAN UP AFTER THEMSELVES: whichever kernel reads a tile's fill cursor last resets it (this one when
// `clean_counts`, otherwise tile_sort_big_kernel), and workgroup 0 folds the instance-count shards into stats[0], the prefilter
// flag into stats[2], and resets both -- an image buffer that was clean before a slab-binned forward is clean again after it, so a
// caller that keeps its buffers (igs_refine_step, scratch_clean) needs no zero-fill launch per frame.  Every id written to the
// sorted list is clamped to id_max = P - 1: whatever a caller's broken promise puts into the slabs, the blend kernels never gather
// outside the record array.
// Dispatch order of the fused blend kernel (blend_step.hip), built on the side by TWO extra workgroups of the tile sort -- the counts are
// all there when this launch starts, and nobody resets them during it in that mode.  The order is the plain XCD-aware one
// (common.h: tile_for_block) except that the LIGHT tiles come last: a tile's forward + backward takes 8 us (empty) to 85 us (250
// instances), and in plain order a heavy tile that comes up at t = 45 us keeps a few workgroups busy for 35 us after the other 2000
// slots have drained (profiles/r03_blend_timeline.txt).  Heaviest-first is not the answer (heavy tiles together run slower and the
// store-bound near-empty ones no longer hide under them); keeping the mix and moving the tiles below STEP_ORDER_T0 / T1 instances and
// the empty ones to the end makes the drain phase consist of short workgroups: blend_step 121.4 -> 117.6 us (rocprofv3, same box).
// Every XCD group (workgroups b = 8 k + x) is partitioned on its own (stable), so every tile stays on the L2 it had; one wave per group
// (two extra workgroups).
#ifndef STEP_ORDER_T0
#define STEP_ORDER_T0 64u
#endif
#ifndef STEP_ORDER_T1
#define STEP_ORDER_T1 32u
#endif
__device__ __forceinline__ void build_step_order(const uint32_t* __restrict__ tile_count, uint32_t slab, uint32_t gx, uint32_t gy,
                                                 uint32_t* __restrict__ order, uint32_t half)
{
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t per = ((gy + 7u) / 8u) * gx;                   // workgroups of one XCD group (<= 64 * STEP_ORDER_CHUNKS: step_order_usable)
    {
        const uint32_t x = 4u * half + wid;               // two extra workgroups: wave w of workgroup `half` takes XCD group 4 half + w
        // every count of the group in ONE memory round trip, then four passes over registers
        uint32_t tl[STEP_ORDER_CHUNKS], cl[STEP_ORDER_CHUNKS];
#pragma unroll
        for (int i = 0; i < STEP_ORDER_CHUNKS; i++) {
            const uint32_t k = 64u * i + lane;
            uint32_t tile = 0xFFFFFFFFu, n = 0u;
            const bool ok = k < per && tile_for_block(8u * k + x, gx, gy, tile);
            if (ok) { n = tile_count[tile]; if (n > slab) n = 0u; }                       // (an overflowed tile gets an empty range)
            tl[i] = ok ? tile : 0xFFFFFFFFu;
            cl[i] = k >= per ? 4u : !ok ? 3u : n >= STEP_ORDER_T0 ? 0u : n >= STEP_ORDER_T1 ? 1u : n > 0u ? 2u : 3u;
        }
        uint32_t out = 0;
        for (uint32_t c = 0; c < 4u; c++) {
#pragma unroll
            for (int i = 0; i < STEP_ORDER_CHUNKS; i++) {
                const bool mine = cl[i] == c;
                const unsigned long long m = __ballot(mine);
                if (mine) order[8u * (out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) + x] = tl[i];
                out += (uint32_t)__popcll(m);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
tile_sort_kernel(uint32_t T, uint32_t* __restrict__ tile_count, const uint64_t* __restrict__ pairs,
                 uint32_t* __restrict__ point_list, uint32_t* __restrict__ ranges, uint32_t slab, uint32_t* __restrict__ stats,
                 uint32_t* __restrict__ counters, uint32_t id_max, int clean_counts, uint32_t wave_max,
                 uint32_t* __restrict__ step_order, uint32_t gx, uint32_t gy)
{
    if (step_order && blockIdx.x >= gridDim.x - 2) {          // the extra workgroups (clean_counts is 0 in this mode: nobody writes the counts)
        build_step_order(tile_count, slab, gx, gy, step_order, blockIdx.x - (gridDim.x - 2));
        return;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (blockIdx.x == 0 && wid == 0 && counters) {
        // R = sum of the preprocess kernel's counter shards
        uint32_t v = counters[COUNTER_SHARD_STRIDE * (1 + lane)];
        counters[COUNTER_SHARD_STRIDE * (1 + lane)] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) { stats[0] = v; stats[2] = counters[1]; counters[1] = 0u; }
    }
    const uint32_t t0 = blockIdx.x * 4;
    STLT(t0 + wid, 0, wall_clock64());
    // the four counts of the workgroup, fetched once (lane q of every wave loads tile t0+q)
    const uint32_t my_cnt = (lane < 4 && t0 + lane < T) ? tile_count[t0 + lane] : 0u;
    const uint32_t cnt0 = (uint32_t)__shfl((int)my_cnt, 0, 64), cnt1 = (uint32_t)__shfl((int)my_cnt, 1, 64);
    const uint32_t cnt2 = (uint32_t)__shfl((int)my_cnt, 2, 64), cnt3 = (uint32_t)__shfl((int)my_cnt, 3, 64);
    if (clean_counts) {
        // all four waves hold the counts in registers before any of them resets the words
        __syncthreads();
        if (wid == 0 && lane < 4 && t0 + lane < T) tile_count[t0 + lane] = 0u;
    }
    // ---- phase 1: one wave per tile
    {
        const uint32_t t = t0 + wid;
        if (t < T) {
            const uint32_t n_true = wid == 0 ? cnt0 : wid == 1 ? cnt1 : wid == 2 ? cnt2 : cnt3;
            STLT(t, 5, (unsigned long long)n_true); STLT(t, 1, wall_clock64());
            const size_t base = (size_t)t * slab;
            if (n_true > slab) {
                if (lane == 0) { atomicMax(&stats[1], n_true); ranges[2 * t] = (uint32_t)base; ranges[2 * t + 1] = (uint32_t)base; }
            } else {
                if (lane == 0 && n_true <= TILE_SORT_SMALL) { ranges[2 * t] = (uint32_t)base; ranges[2 * t + 1] = (uint32_t)(base + n_true); }
                const uint64_t* src = pairs + base;
                uint32_t* dst = point_list + base;
                if (n_true == 0 || n_true > wave_max) {}          // (denser tiles: tile_sort_mid_kernel / tile_sort_big_kernel)
                else if (n_true <= 64) wave_sort_tile<1>(src, dst, n_true, lane, id_max);
                else if (n_true <= 128) wave_sort_tile<2>(src, dst, n_true, lane, id_max);
                else if (n_true <= 256) wave_sort_tile<4>(src, dst, n_true, lane, id_max);
                else {
                    // Every wave of the launch is resident at once (5.3 per SIMD on the bench scene), so the launch lasts as long as its
                    // longest chain -- the few tiles above 256 instances (2 400-5 300 dependent instructions): those waves go first.
                    // profiles/r03_sort_timeline.txt: a 263-instance tile took 13.5 us of a 15.1 us launch at equal priority.
#ifndef TILE_SORT_NO_PRIO
                    __builtin_amdgcn_s_setprio(3);
#endif
                    if (n_true <= 320) wave_sort_tile_320(src, dst, n_true, lane, id_max);
                    else if (n_true <= 512) wave_sort_tile<8>(src, dst, n_true, lane, id_max);
                    else if (n_true <= 1024) wave_sort_tile<16>(src, dst, n_true, lane, id_max);
                }
                STLT(t, 4, wall_clock64());
            }
        }
    }
}

// one workgroup per tile of (wave_max, TILE_SORT_SMALL] instances, in a launch of its own (only when the slabs are larger than
// TILE_SORT_WAVE_ALONE): four waves and 16 KB of LDS bring a 512-key tile home in ~45 barrier-separated stages of ONE compare-exchange
// per thread, where a single wave walks 2 400-5 300 dependent instructions for 512-1024 keys in registers.  (As a second phase of
// tile_sort_kernel the tiles above 1024 were sorted one after the other by the workgroup that owned them, behind its four register
// sorts: on the dense diagnostic scene -- mean 430 instances per tile -- that tail and the long single-wave sorts made 85 us.)
__global__ void __launch_bounds__(256)
tile_sort_mid_kernel(uint32_t* __restrict__ tile_count, const uint64_t* __restrict__ pairs, uint32_t* __restrict__ point_list,
                     uint32_t slab, uint32_t id_max, int clean_counts, uint32_t wave_max)
{
    __shared__ __attribute__((aligned(16))) uint64_t skeys[TILE_SORT_SMALL];        // 16 KB
    const uint32_t t = blockIdx.x, tid = threadIdx.x, n = tile_count[t];
    if (clean_counts) {
        __syncthreads();                               // every thread has its copy of n
        if (tid == 0) tile_count[t] = 0u;
    }
    if (n <= wave_max || n > TILE_SORT_SMALL || n > slab) return;           // (the range was written by tile_sort_kernel)
    const size_t base = (size_t)t * slab;
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t i = tid; i < N; i += 256) skeys[i] = (i < n) ? pairs[base + i] : ~0ull;
    __syncthreads();
    BITONIC_SORT(skeys, N, tid, 256, __syncthreads())
    for (uint32_t i = tid; i < n; i += 256) point_list[base + i] = min((uint32_t)skeys[i], id_max);
}

// one workgroup per tile of (TILE_SORT_SMALL, TILE_SORT_BIG] instances (only launched when the slabs are that large); the last reader
// of the fill cursors in that case, so it is the one that resets them
__global__ void __launch_bounds__(1024)
tile_sort_big_kernel(uint32_t* __restrict__ tile_count, const uint64_t* __restrict__ pairs, uint32_t* __restrict__ point_list,
                     uint32_t* __restrict__ ranges, uint32_t slab, uint32_t id_max)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t bkeys[];
    const uint32_t t = blockIdx.x, n = tile_count[t];
    __syncthreads();                                   // every thread has its copy of n
    if (threadIdx.x == 0) tile_count[t] = 0u;          // (the last reader of the fill cursors whenever it is launched)
    if (n <= TILE_SORT_SMALL || n > slab || n > TILE_SORT_BIG) return;
    const size_t base = (size_t)t * slab;
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t i = threadIdx.x; i < N; i += 1024) bkeys[i] = (i < n) ? pairs[base + i] : ~0ull;
    __syncthreads();
    BITONIC_SORT(bkeys, N, threadIdx.x, 1024, __syncthreads())
    for (uint32_t i = threadIdx.x; i < n; i += 1024) point_list[base + i] = min((uint32_t)bkeys[i], id_max);
    if (threadIdx.x == 0) { ranges[2 * t] = (uint32_t)base; ranges[2 * t + 1] = (uint32_t)(base + n); }
}

// ONE workgroup per tile, for scenes whose slabs have grown beyond TILE_SORT_WAVE_ALONE (dense views: most tiles hold several
// hundred instances).  A tile of up to TILE_SORT_ONE_WAVE instances is sorted by the workgroup's first wave in registers while the
// other three leave at once; a denser one by all four waves in LDS (a 512-key tile: 9.7 us against 13.5 for one wave in registers,
// profiles/r03_sort_timeline.txt).  Replaces tile_sort_kernel + tile_sort_mid_kernel -- two launches -- there; on the bench scene
// (densest tile 263 instances) its 5440 workgroups take 8 us to dispatch and it loses (16.5 against 16.1 us).
#ifndef TILE_SORT_ONE_WAVE
#define TILE_SORT_ONE_WAVE 256
#endif
__global__ void __launch_bounds__(256)
tile_sort_one_kernel(uint32_t T, uint32_t* __restrict__ tile_count, const uint64_t* __restrict__ pairs,
                     uint32_t* __restrict__ point_list, uint32_t* __restrict__ ranges, uint32_t slab, uint32_t* __restrict__ stats,
                     uint32_t* __restrict__ counters, uint32_t id_max, int clean_counts)
{
    __shared__ __attribute__((aligned(16))) uint64_t skeys[TILE_SORT_SMALL];        // 16 KB
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (blockIdx.x == 0 && wid == 0 && counters) {
        // R = sum of the preprocess kernel's counter shards
        uint32_t v = counters[COUNTER_SHARD_STRIDE * (1 + lane)];
        counters[COUNTER_SHARD_STRIDE * (1 + lane)] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) { stats[0] = v; stats[2] = counters[1]; counters[1] = 0u; }
    }
    const uint32_t t = blockIdx.x;
    if (t >= T) return;
    STL(0, wall_clock64());
    const uint32_t n = tile_count[t];
    STL(5, (unsigned long long)n);
    if (clean_counts) {
        __syncthreads();                               // every thread has its copy of n
        if (tid == 0) tile_count[t] = 0u;
    }
    const size_t base = (size_t)t * slab;
    if (n > slab) {
        if (tid == 0) { atomicMax(&stats[1], n); ranges[2 * t] = (uint32_t)base; ranges[2 * t + 1] = (uint32_t)base; }
        return;
    }
    if (tid == 0 && n <= TILE_SORT_SMALL) { ranges[2 * t] = (uint32_t)base; ranges[2 * t + 1] = (uint32_t)(base + n); }
    STL(1, wall_clock64());
    if (n == 0 || n > TILE_SORT_SMALL) return;         // (denser tiles: tile_sort_big_kernel)
    const uint64_t* src = pairs + base;
    uint32_t* dst = point_list + base;
    if (n <= TILE_SORT_ONE_WAVE) {
        if (wid != 0) return;
        if (n <= 64) wave_sort_tile<1>(src, dst, n, lane, id_max);
        else if (n <= 128) wave_sort_tile<2>(src, dst, n, lane, id_max);
        else if (n <= 256) wave_sort_tile<4>(src, dst, n, lane, id_max);
#if TILE_SORT_ONE_WAVE > 256
        else if (n <= 512) wave_sort_tile<8>(src, dst, n, lane, id_max);
#endif
        STL(4, wall_clock64());
        return;
    }
#ifdef TILE_SORT_ALL_LDS
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t i = tid; i < N; i += 256) skeys[i] = (i < n) ? src[i] : ~0ull;
    __syncthreads();
    BITONIC_SORT(skeys, N, tid, 256, __syncthreads())
    for (uint32_t i = tid; i < n; i += 256) dst[i] = min((uint32_t)skeys[i], id_max);
#else
    if (n <= 256) wg_sort_tile<1>(src, dst, n, skeys, tid, id_max);
    else if (n <= 512) wg_sort_tile<2>(src, dst, n, skeys, tid, id_max);
    else if (n <= 1024) wg_sort_tile<4>(src, dst, n, skeys, tid, id_max);
    else wg_sort_tile<8>(src, dst, n, skeys, tid, id_max);
#endif
    STL(4, wall_clock64());
}

// (TILE_SORT_WAVE_ALONE: common.h -- slabs up to this size: every tile by one wave in registers, no second launch)
#ifndef TILE_SORT_WAVE_WITH_MID
#define TILE_SORT_WAVE_WITH_MID 1024      // ... larger slabs: the register sort keeps the tiles up to this size, the LDS kernel takes the rest
#endif                                    // (256 / 512 measured: no better on the dense scene, and a second launch for nothing on the bench scene's 1024-slot slabs)
hipError_t launch_tile_sort(hipStream_t s, uint32_t T, uint32_t* tile_count, const uint64_t* pairs, uint32_t* point_list,
                            uint32_t* ranges, uint32_t slab, uint32_t* stats, uint32_t* counters, uint32_t P,
                            uint32_t* step_order, uint32_t gx, uint32_t gy)
{
    const bool mid = slab > TILE_SORT_WAVE_ALONE, big = slab > TILE_SORT_SMALL;
    const uint32_t id_max = P ? P - 1u : 0u;
    const uint32_t wave_max = mid ? TILE_SORT_WAVE_WITH_MID : TILE_SORT_WAVE;
#ifndef TILE_SORT_NO_ONE_PER_WG
    if (mid) {
        hipLaunchKernelGGL(tile_sort_one_kernel, dim3(T), dim3(256), 0, s, T, tile_count, pairs, point_list, ranges, slab, stats, counters, id_max, big ? 0 : 1);
    } else
#endif
    {
    // whichever launch reads the fill cursors last resets them
    // with a step order wanted (fused refine step, step_order_usable): one extra workgroup builds it from the fill cursors, and these stay
    // as they are -- the fused blend kernel zeroes them (BlendFwdArgs::reset_cursors)
    const bool ord = step_order != nullptr && !mid;
    if (step_order != nullptr && mid) return hipErrorInvalidValue;      // (api.hip asks step_order_usable() first: a dispatch order nobody writes must never be read)
    hipLaunchKernelGGL(tile_sort_kernel, dim3((T + 3) / 4 + (ord ? 2 : 0)), dim3(256), 0, s, T, tile_count, pairs, point_list, ranges, slab, stats, counters,
                       id_max, (mid || ord) ? 0 : 1, wave_max, ord ? step_order : nullptr, gx, gy);
    if (mid) hipLaunchKernelGGL(tile_sort_mid_kernel, dim3(T), dim3(256), 0, s, tile_count, pairs, point_list, slab, id_max, big ? 0 : 1, wave_max);
    }
    if (big) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)tile_sort_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_SORT_BIG * 8);
            attr_set = true;
        }
        hipLaunchKernelGGL(tile_sort_big_kernel, dim3(T), dim3(1024), TILE_SORT_BIG * 8, s, tile_count, pairs, point_list, ranges, slab, id_max);
    }
    return hipGetLastError();
}

// ---- debug / test support: gather per-tile lists (slab or compact) into the reference's compact layout --------------------
__global__ void __launch_bounds__(1024)
compact_ranges_kernel(uint32_t T, const uint32_t* __restrict__ ranges_in, uint32_t* __restrict__ ranges_out)
{
    __shared__ uint32_t wave_sums[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t per = (T + 1023) / 1024;
    const uint32_t beg = min(T, tid * per), end = min(T, beg + per);
    uint32_t sum = 0;
    for (uint32_t i = beg; i < end; i++) sum += ranges_in[2 * i + 1] - ranges_in[2 * i];
    uint32_t v = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off, 64); if (lane >= (uint32_t)off) v += t; }
    if (lane == 63) wave_sums[wid] = v;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wid; w++) woff += wave_sums[w];
    uint32_t run = v - sum + woff;
    for (uint32_t i = beg; i < end; i++) {
        const uint32_t c = ranges_in[2 * i + 1] - ranges_in[2 * i];
        // empty tiles read (0,0) as in the reference (rasterizer_impl.cu:383 zero-fills ranges)
        ranges_out[2 * i] = c ? run : 0u; run += c; ranges_out[2 * i + 1] = c ? run : 0u;
    }
}
__global__ void __launch_bounds__(256)
compact_lists_kernel(const uint32_t* __restrict__ ranges_in, const uint32_t* __restrict__ ranges_out, const uint32_t* __restrict__ list_in,
                     uint32_t* __restrict__ list_out, uint32_t out_capacity)
{
    const uint32_t t = blockIdx.x;
    const uint32_t src = ranges_in[2 * t], n = ranges_in[2 * t + 1] - src, dst = ranges_out[2 * t];
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        if (dst + i < out_capacity) list_out[dst + i] = list_in[src + i];
}
hipError_t launch_compact_lists(hipStream_t s, uint32_t T, const uint32_t* ranges_in, const uint32_t* list_in, uint32_t* ranges_out,
                                uint32_t* list_out, uint32_t out_capacity)
{
    hipLaunchKernelGGL(compact_ranges_kernel, dim3(1), dim3(1024), 0, s, T, ranges_in, ranges_out);
    if (list_in && list_out && out_capacity)
        hipLaunchKernelGGL(compact_lists_kernel, dim3(T), dim3(256), 0, s, ranges_in, ranges_out, list_in, list_out, out_capacity);
    return hipGetLastError();
}

// ---- Morton (Z-order) ordering of the Gaussians' positions (GaussianParams.spatial_sort; no reference counterpart: the binning
// stage reserves its slots per (workgroup, tile), which pays when consecutive Gaussians project to neighbouring tiles) ----------
// key = 3 x `bits` interleaved bits of the position quantised inside the bounding box lohi = {lo.xyz, hi.xyz} (device memory: no
// host read-back); written with the identity permutation and the first radix pass's per-block histogram, then sorted by the
// library's own stable LSD radix sort -- the order torch.argsort(code, stable=True) gives.
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v = (v | (v << 16)) & 0x30000FFu;
    v = (v | (v << 8)) & 0x300F00Fu;
    v = (v | (v << 4)) & 0x30C30C3u;
    return (v | (v << 2)) & 0x9249249u;
}
__global__ void __launch_bounds__(256)
morton_keys_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ lohi, int bits, uint32_t* __restrict__ keys,
                   uint32_t* __restrict__ vals, uint32_t* __restrict__ hist0, uint32_t per_block)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float top = (float)((1u << bits) - 1u);
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = lohi[k], ext = fmaxf(lohi[3 + k] - lo, 1e-12f);
        const float t = (xyz[3 * (size_t)i + k] - lo) / ext * top;          // (same expression, same order as the PyTorch form it replaces)
        const long long v = (long long)t;                                     // truncation towards zero, as .long()
        q[k] = (uint32_t)(v < 0 ? 0 : (v > (long long)top ? (long long)top : v));
    }
    const uint32_t code = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    keys[i] = code; vals[i] = (uint32_t)i;
    atomicAdd(&hist0[((uint32_t)i / per_block) * RADIX + (code & 255u)], 1u);
}
size_t morton_scratch_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 0);
    return 4 * ((n * 4 + 255) & ~(size_t)255) + (size_t)SORT_MAX_PASSES * RADIX * SORT_MAX_BLOCKS * 4 + 256;
}
hipError_t launch_morton_order(hipStream_t s, int P, const float* xyz, const float* lohi, int bits, void* scratch, int* perm)
{
    if (P <= 0) return hipSuccess;
    const size_t seg = ((size_t)P * 4 + 255) & ~(size_t)255;
    char* b = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    uint32_t *ka = (uint32_t*)b, *kb = (uint32_t*)(b + seg), *va = (uint32_t*)(b + 2 * seg), *vb = (uint32_t*)(b + 3 * seg);
    uint32_t* hist = (uint32_t*)(b + 4 * seg);
    hipError_t e = zero_fill_async(s, hist, (size_t)SORT_MAX_PASSES * RADIX * SORT_MAX_BLOCKS * 4);
    if (e != hipSuccess) return e;
    uint32_t nb, per;
    sort_geometry((uint32_t)P, &nb, &per);
    hipLaunchKernelGGL(morton_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, xyz, lohi, bits, ka, va, hist, per);
    uint32_t *sk = nullptr, *sv = nullptr;
    e = radix_sort_pairs(s, (uint32_t)P, ka, kb, va, vb, hist, 0, 3 * bits, &sk, &sv);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(perm, sv, (size_t)P * 4, hipMemcpyDeviceToDevice, s);
}
