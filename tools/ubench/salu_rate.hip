// Micro-benchmark (round 3): does scalar work ride for free beside a VALU-bound wave64 stream on gfx950, and up to how many scalar
// instructions per vector instruction?  (The block-list blend kernels walk four bit masks per trip in the scalar ALU: ~30 SALU
// instructions beside ~85 VALU.)  Also: a ds_read_b128 whose four 16-lane blocks read four DIFFERENT records against the broadcast
// form the quad kernels use.
//   every wave: ITERS x { 8 independent v_fma_f32 + NS scalar ops (s_ff1_i32_b64 / s_bitset0_b64 / s_min_u32 chain) }
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/salu_rate tools/ubench/salu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int NS, int LDSMODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long seedmask, float seed)
{
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 0.999f + seed * 1e-6f, c = 1e-3f;
    unsigned long long m0 = seedmask, m1 = ~seedmask, m2 = seedmask * 3, m3 = seedmask * 5;
    unsigned acc = 0;
    float4 q = make_float4(0, 0, 0, 0);
    const unsigned lane = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#pragma unroll
        for (int s = 0; s < NS / 4; s++) {        // 4 scalar instructions per group: ff1, bitset0, min, add
            unsigned j;
            unsigned long long& m = (s & 3) == 0 ? m0 : (s & 3) == 1 ? m1 : (s & 3) == 2 ? m2 : m3;
            asm volatile("s_ff1_i32_b64 %0, %1\n s_bitset0_b64 %1, %0\n s_min_u32 %0, %0, 64\n s_add_u32 %2, %2, %0"
                         : "=&s"(j), "+s"(m), "+s"(acc));
        }
        if (LDSMODE == 1) {             // broadcast record read
            const float4 r = lds[(i * 3) & 255];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
        } else if (LDSMODE == 2) {      // four records, one per 16-lane block
            const float4 r = lds[((i * 3) + (lane >> 4) * 37) & 255];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
        } else if (LDSMODE == 3) {      // 64 records
            const float4 r = lds[((i * 3) + lane * 37) & 255];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
        }
        if ((m0 | m1 | m2 | m3) == 0) { m0 = seedmask; m1 = ~seedmask; m2 = seedmask * 3; m3 = seedmask * 5; }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q.x + q.y + q.z + q.w;
    if (s == 12345.678f || acc == 0x12345u) out[0] = s;
}

template <int NS, int LDSMODE> static void run(const char* name)
{
    float* out;
    hipMalloc(&out, 64);
    const int iters = 4000;
    printf("%-58s", name);
    for (int W : {4, 8}) {
        const int blocks = 256 * W;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<NS, LDSMODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x9E3779B97F4A7C15ull, 0.5f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NS, LDSMODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x9E3779B97F4A7C15ull, 0.5f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc_per_iter = ms * 1e-3 * 2.4e9 / ((double)iters * W);      // SIMD cycles per loop iteration (8 VALU + NS SALU [+ 1 LDS read + 4 add])
        printf("  W=%d: %.1f cyc/iter/SIMD", W, cyc_per_iter);
    }
    printf("\n");
}

int main()
{
    printf("SIMD cycles per loop iteration at 2.4 GHz (upper bound); 8 v_fma_f32 alone cost ~21\n");
    run<0, 0>("8 fma");
    run<4, 0>("8 fma + 4 salu");
    run<8, 0>("8 fma + 8 salu");
    run<16, 0>("8 fma + 16 salu");
    run<32, 0>("8 fma + 32 salu");
    run<0, 1>("8 fma + 4 add + ds_read_b128 broadcast");
    run<0, 2>("8 fma + 4 add + ds_read_b128, 4 records (per 16 lanes)");
    run<0, 3>("8 fma + 4 add + ds_read_b128, 64 records");
    return 0;
}
