// Micro-benchmark: what does a (nearly) empty kernel cost as a function of grid size, workgroup size, dynamic LDS and VGPR
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/launch_cost tools/ubench/launch_cost.hip ; run it on the GPU box
// allocation?  (Question behind it: the fused per-Gaussian backward + Adam kernel takes ~80 us whatever its body does.)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float dyn[];
__global__ void k_small(float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n && dyn[0] == 123.f) out[i] = 1.f; }
__global__ void __attribute__((amdgpu_num_vgpr(184))) k_big(float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n && dyn[0] == 123.f) out[i] = 1.f; }
struct Big { float a[100]; };
__global__ void __attribute__((amdgpu_num_vgpr(184))) k_bigarg(Big b, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n && dyn[0] == 123.f) out[i] = b.a[3]; }
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / 20.f;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const int n = 200000;
    for (int wg : {64, 256}) for (size_t lds : {(size_t)0, (size_t)12544, (size_t)50176}) {
        const int blocks = (n + wg - 1) / wg;
        float t1 = timeit([&] { hipLaunchKernelGGL(k_small, dim3(blocks), dim3(wg), lds, 0, out, n); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_big, dim3(blocks), dim3(wg), lds, 0, out, n); });
        Big b{}; float t3 = timeit([&] { hipLaunchKernelGGL(k_bigarg, dim3(blocks), dim3(wg), lds, 0, b, out, n); });
        printf("wg %3d blocks %5d lds %6zu : small-vgpr %.1f us   184-vgpr %.1f us   184-vgpr+400B-args %.1f us\n", wg, blocks, lds, t1, t2, t3);
    }
    return 0;
}
