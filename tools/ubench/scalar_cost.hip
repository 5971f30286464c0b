// Micro-benchmark (round 3): what does ONE scalar-side instruction cost a SIMD that is busy with wave64 VALU work (8 waves resident)?
// tools/ubench/salu_rate showed that scalar ALU work is NOT free beside a VALU-bound stream on gfx950 (8 v_fma = 26.5 cycles per
// SIMD and iteration; + 32 scalar instructions = 166.7).  This one prices the kinds a blend row is made of, each as
//   ITERS x { 8 independent v_fma_f32 + 8 x <instruction under test> }     cost = (cycles - cycles of the bare fma loop) / 8
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/scalar_cost tools/ubench/scalar_cost.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum { BARE = 0, NOP, WAITCNT, SADD_INDEP, SADD_DEP, SAVEEXEC, CBRANCH_NT, READFIRSTLANE, MOV_M0, VCMP_SAND, FF1_BITSET, VCNDMASK, SMOV_EXEC, SBRANCH_TAKEN, BALLOT_BRANCH, NKINDS };

template <int K>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned seed_u, float seed)
{
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 0.999f + seed * 1e-6f, c = 1e-3f;
    unsigned s0 = seed_u, s1 = seed_u + 1, s2 = seed_u + 2, s3 = seed_u + 3, s4 = seed_u + 4, s5 = seed_u + 5, s6 = seed_u + 6, s7 = seed_u + 7;
    unsigned long long m = 0x9E3779B97F4A7C15ull ^ seed_u;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (K == NOP) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        else if (K == WAITCNT) asm volatile("s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)");
        else if (K == SADD_INDEP) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                                           : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
        else if (K == SADD_DEP) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3"
                                         : "+s"(s0) :: "scc");
        else if (K == SAVEEXEC) asm volatile("s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, -1\n s_or_b64 exec, exec, %0"
                                         : "+s"(m) :: "scc");
        else if (K == CBRANCH_NT) asm volatile("s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %0, 0x7ffffff1\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %0, 0x7ffffff2\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %0, 0x7ffffff3\n s_cbranch_scc1 1f\n1:"
                                           :: "s"(s0) : "scc");
        else if (K == READFIRSTLANE) asm volatile("v_readfirstlane_b32 %0, %8\n v_readfirstlane_b32 %1, %9\n v_readfirstlane_b32 %2, %8\n v_readfirstlane_b32 %3, %9\n v_readfirstlane_b32 %4, %8\n v_readfirstlane_b32 %5, %9\n v_readfirstlane_b32 %6, %8\n v_readfirstlane_b32 %7, %9"
                                              : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(a0), "v"(a1));
        else if (K == MOV_M0) asm volatile("s_mov_b32 m0, %0\n s_mov_b32 m0, %1\n s_mov_b32 m0, %0\n s_mov_b32 m0, %1\n s_mov_b32 m0, %0\n s_mov_b32 m0, %1\n s_mov_b32 m0, %0\n s_mov_b32 m0, %1" :: "s"(s0), "s"(s1) : "m0");
        else if (K == VCMP_SAND) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %1, %3\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %3, %2\n s_and_b64 %0, %0, vcc"
                                          : "+s"(m) : "v"(a0), "v"(a1), "v"(a2) : "vcc", "scc");
        else if (K == FF1_BITSET) { unsigned j; asm volatile("s_ff1_i32_b64 %0, %1\n s_bitset0_b64 %1, %0\n s_ff1_i32_b64 %0, %1\n s_bitset0_b64 %1, %0\n s_ff1_i32_b64 %0, %1\n s_bitset0_b64 %1, %0\n s_ff1_i32_b64 %0, %1\n s_bitset0_b64 %1, %0"
                                                             : "=&s"(j), "+s"(m)); s1 += j; if (m == 0) m = 0x9E3779B97F4A7C15ull; }
        else if (K == VCNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
        else if (K == SMOV_EXEC) asm volatile("s_mov_b64 exec, %0\n s_mov_b64 exec, -1\n s_mov_b64 exec, %0\n s_mov_b64 exec, -1\n s_mov_b64 exec, %0\n s_mov_b64 exec, -1\n s_mov_b64 exec, %0\n s_mov_b64 exec, -1" :: "s"(m | 1ull));
        else if (K == SBRANCH_TAKEN) asm volatile("s_branch 1f\n1: s_branch 2f\n2: s_branch 3f\n3: s_branch 4f\n4: s_branch 5f\n5: s_branch 6f\n6: s_branch 7f\n7: s_branch 8f\n8:");
        else if (K == BALLOT_BRANCH) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_cmp_lt_f32 vcc, %1, %2\n s_cbranch_vccz 1f\n v_cmp_lt_f32 vcc, %0, %2\n s_cbranch_vccz 1f\n v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n1:"
                                              :: "v"(a0), "v"(a1), "v"(a2) : "vcc");
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f || (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + (unsigned)m) == 0x12345u) out[0] = s;
}

static double g_bare[2];
template <int K> static void run(const char* name, int n_under_test)
{
    float* out;
    hipMalloc(&out, 64);
    const int iters = 4000;
    printf("%-52s", name);
    int wi = 0;
    for (int W : {4, 8}) {
        const int blocks = 256 * W;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 0.5f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 0.5f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * W);
        if (K == BARE) g_bare[wi] = cyc;
        printf("  W=%d: %6.1f cyc/iter/SIMD (%.2f per instruction under test)", W, cyc, n_under_test ? (cyc - g_bare[wi]) / n_under_test : 0.0);
        wi++;
    }
    printf("\n");
}

int main()
{
    printf("cost of scalar-side instructions beside 8 v_fma_f32 per iteration, SIMD cycles at 2.4 GHz (upper bounds)\n");
    run<BARE>("8 v_fma_f32 (+ loop: s_add, s_cmp, s_cbranch)", 0);
    run<NOP>("+ 8 s_nop 0", 8);
    run<WAITCNT>("+ 8 s_waitcnt (nothing outstanding)", 8);
    run<SADD_INDEP>("+ 8 independent s_add_u32", 8);
    run<SADD_DEP>("+ 8 dependent s_add_u32", 8);
    run<FF1_BITSET>("+ 4 x (s_ff1_i32_b64, s_bitset0_b64)", 8);
    run<SAVEEXEC>("+ 4 x (s_and_saveexec_b64, s_or_b64 exec)", 8);
    run<SMOV_EXEC>("+ 8 s_mov_b64 exec", 8);
    run<CBRANCH_NT>("+ 4 x (s_cmp, s_cbranch not taken)", 8);
    run<SBRANCH_TAKEN>("+ 8 s_branch taken (to the next instruction)", 8);
    run<BALLOT_BRANCH>("+ 4 x (v_cmp, s_cbranch_vccz not taken)", 8);
    run<READFIRSTLANE>("+ 8 v_readfirstlane_b32", 8);
    run<MOV_M0>("+ 8 s_mov_b32 m0", 8);
    run<VCMP_SAND>("+ 4 x (v_cmp_lt_f32 vcc, s_and_b64)", 8);
    run<VCNDMASK>("+ 4 x (v_cmp_lt_f32 vcc, v_cndmask_b32)", 8);
    return 0;
}
