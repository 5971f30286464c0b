// Micro-benchmark (VERDICT r1 next #4a): how many cycles does one wave64 VALU instruction cost a SIMD on gfx950 as a function of the
// waves resident on that SIMD?  MI355X_MICROARCH.md says v_fma_f32 = 2 cycles with co-resident waves, 4 for a wave alone, and 8 for the
// transcendentals; DESIGN.md (round 1) priced the blend kernels at 4.  This program measures it:
//   * workgroups of 256 threads = 4 waves = ONE wave per SIMD; W workgroups per CU resident at once = W waves per SIMD
//     (grid = 256 CUs x W, tiny register / LDS footprint so that all of them are resident; a census of s_getreg(HW_ID) checks it);
//   * every wave issues ITERS x 64 independent instructions (8 chains) between two s_memtime stamps;
//   * reported: median cycles per instruction seen by ONE wave (delta / count) and the SIMD-side cost = that / W.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum { OP_FMA = 0, OP_EXP = 1, OP_RCP = 2, OP_MIX = 3, OP_LDSB128 = 4, OP_DPP = 5, OP_SWAP32 = 6, OP_SWAP16 = 7, OP_CNDMASK = 8, OP_BPERM = 9, OP_PKFMA = 10, OP_PKADD = 11, OP_MUL = 12, OP_FMA_LO32 = 13, OP_FMA_LO16 = 14, OP_FMA_HI32 = 15, OP_FMA_ROWS02 = 16 };

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc, float seed)
{
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 0.999f + seed * 1e-6f, c = 1e-3f;
    float4 q = make_float4(0, 0, 0, 0);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    // partial EXEC masks: does the SIMD skip a 32-lane half (or a 16-lane row) of a wave64 instruction in which no lane is active?
    constexpr bool MASKED = OP >= OP_FMA_LO32;
    constexpr unsigned long long EXECMASK = OP == OP_FMA_LO32 ? 0x00000000FFFFFFFFull : OP == OP_FMA_LO16 ? 0x000000000000FFFFull
                                            : OP == OP_FMA_HI32 ? 0xFFFFFFFF00000000ull : 0x0000FFFF0000FFFFull;
    if (MASKED) asm volatile("s_mov_b64 exec, %0" :: "s"(EXECMASK));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == OP_FMA || MASKED) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == OP_EXP) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == OP_RCP) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == OP_DPP) {      // cross-lane add inside a 16-lane row (what a DPP wave reduction is made of)
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == OP_SWAP32) {   // gfx950: exchange the upper half of one register with the lower half of another
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == OP_SWAP16) {
                asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == OP_CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (OP == OP_BPERM) {    // ds_bpermute_b32: cross-lane gather through the LDS crossbar (no memory access)
                const int addr = ((threadIdx.x * 5 + u) & 63) << 2;
                a0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a0)));
                a1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a1)));
                a2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a2)));
                a3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, a3)));
            } else if (OP == OP_PKFMA) {    // packed fp32: two FMAs per lane and instruction on a 64-bit register pair
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (OP == OP_PKADD) {
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
            } else if (OP == OP_MUL) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == OP_MIX) {      // the blend row's mix: 1 transcendental per ~30 plain ops -> here 1 exp + 7 fma
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else {                          // broadcast ds_read_b128 (what the blend kernels fetch their staged records with) + 7 fma
                const float4 r = lds[(i + u) & 63];
                q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MASKED) asm volatile("s_mov_b64 exec, -1");
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q.x + q.y + q.z + q.w + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> static void run(const char* name, int per_iter)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 8 * 4 * 8);
    const int iters = 2000;
    printf("%-34s", name);
    for (int W : {1, 2, 4, 7, 8}) {
        const int blocks = 256 * W;
        std::vector<unsigned long long> h(blocks * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc, 0.5f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc, 0.5f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_wave = (double)h[h.size() / 2] / ((double)iters * per_iter);      // s_memtime ticks at 100 MHz on this part? report raw + wall
        const double wall_cyc = ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * W);         // at the 2.4 GHz peak clock: upper bound on cycles per instruction and SIMD
        printf("  W=%d: %.2f tick/inst/wave, wall %.1f us -> <= %.2f cyc/inst/SIMD", W, per_wave, ms * 1e3, wall_cyc);
    }
    printf("\n");
}

int main()
{
    printf("wave64 VALU issue cost on gfx950 (W = waves per SIMD; 'cyc/inst/SIMD' from wall time at 2.4 GHz, i.e. an upper bound if the clock is lower)\n");
    run<OP_FMA>("v_fma_f32 (8 independent chains)", 64);
    run<OP_FMA_LO32>("v_fma_f32, EXEC = lanes 0-31", 64);
    run<OP_FMA_HI32>("v_fma_f32, EXEC = lanes 32-63", 64);
    run<OP_FMA_LO16>("v_fma_f32, EXEC = lanes 0-15", 64);
    run<OP_FMA_ROWS02>("v_fma_f32, EXEC = lanes 0-15, 32-47", 64);
    run<OP_EXP>("v_exp_f32", 64);
    run<OP_RCP>("v_rcp_f32", 64);
    run<OP_MIX>("1 v_exp_f32 + 7 v_fma_f32", 64);
    run<OP_LDSB128>("1 ds_read_b128 (bcast) + 4 v_add + 4 v_fma", 72);
    run<OP_DPP>("v_add_f32_dpp (row ops)", 64);
    run<OP_SWAP32>("v_permlane32_swap_b32", 64);
    run<OP_SWAP16>("v_permlane16_swap_b32", 64);
    run<OP_CNDMASK>("v_cndmask_b32", 64);
    run<OP_BPERM>("ds_bpermute_b32 (4 independent)", 32);
    run<OP_MUL>("4 v_mul_f32 + 4 v_add_f32", 64);
    run<OP_PKFMA>("v_pk_fma_f32 (2 FMAs per lane)", 64);
    run<OP_PKADD>("v_pk_add_f32", 64);
    return 0;
}
