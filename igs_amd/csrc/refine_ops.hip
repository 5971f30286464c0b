// refine_ops.hip -- bandwidth-bound helpers of the per-frame refine loop (gfx950): fused Adam step and fused L1
// loss forward+backward.  Counterparts in the reference are plain PyTorch: torch.optim.Adam(lr=0, eps=1e-15) built in
// GaussianModel.load_fromstream (igs/models/gaussian_model.py:295-348) and l1_loss (igs/utils/loss_utils.py:17-18).
#include "common.h"
#include "../../include/igs_rast.h"

// torch.optim.Adam semantics (no weight decay, no amsgrad):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256)
adam_kernel(size_t n4, size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            float lr_over_bc1, float b1, float b2, float eps, float inv_sqrt_bc2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P4 = ((float4*)p)[i], M4 = ld_moment((const float4*)m + i), V4 = ld_moment((const float4*)v + i);
        const float4 G4 = ((const float4*)g)[i];
        float* pp = (float*)&P4; float* mm = (float*)&M4; float* vv = (float*)&V4; const float* gg = (const float*)&G4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
            pp[k] -= lr_over_bc1 * mm[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
        }
        ((float4*)p)[i] = P4; st_moment((float4*)m + i, M4); st_moment((float4*)v + i, V4);
    }
    // tail
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

extern "C" int igs_adam_step(void* stream, size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt)
{
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return IGS_RAST_E_INVALID;
    const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, n, param, grad, exp_avg, exp_avg_sq,
                       lr / bias_correction1, beta1, beta2, eps, 1.0f / bias_correction2_sqrt);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

struct AdamGroups { int n; size_t off[8]; size_t cnt[8]; float lr_over_bc1[8]; };
__global__ void __launch_bounds__(256)
adam_groups_kernel(const AdamGroups G, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                   float b1, float b2, float eps, float inv_sqrt_bc2)
{
    // blockIdx.y = group; grid-stride over the group's span (float4 body when the span is 16-byte aligned)
    const int k = blockIdx.y;
    if (k >= G.n) return;
    const size_t o = G.off[k], n = G.cnt[k];
    const float lr = G.lr_over_bc1[k];
    float* pp = p + o; const float* gg = g + o; float* mm = m + o; float* vv = v + o;
    const bool aligned = (((uintptr_t)pp | (uintptr_t)gg | (uintptr_t)mm | (uintptr_t)vv) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P4 = ((float4*)pp)[i], M4 = ld_moment((const float4*)mm + i), V4 = ld_moment((const float4*)vv + i);
        const float4 G4 = ((const float4*)gg)[i];
        float* a = (float*)&P4; float* b = (float*)&M4; float* c = (float*)&V4; const float* d = (const float*)&G4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            b[q] = b1 * b[q] + (1.f - b1) * d[q];
            c[q] = b2 * c[q] + (1.f - b2) * d[q] * d[q];
            a[q] -= lr * b[q] / (sqrtf(c[q]) * inv_sqrt_bc2 + eps);
        }
        ((float4*)pp)[i] = P4; st_moment((float4*)mm + i, M4); st_moment((float4*)vv + i, V4);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = gg[i];
        const float mi = b1 * mm[i] + (1.f - b1) * gi;
        const float vi = b2 * vv[i] + (1.f - b2) * gi * gi;
        mm[i] = mi; vv[i] = vi;
        pp[i] -= lr * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}
extern "C" int igs_adam_step_groups(void* stream, int ngroups, const size_t* offset, const size_t* count, const float* lr, float* param,
                                    const float* grad, float* exp_avg, float* exp_avg_sq, float beta1, float beta2, float eps,
                                    float bias_correction1, float bias_correction2_sqrt)
{
    if (ngroups <= 0) return 0;
    if (ngroups > 8 || !offset || !count || !lr || !param || !grad || !exp_avg || !exp_avg_sq) return IGS_RAST_E_INVALID;
    AdamGroups G; G.n = ngroups;
    size_t nmax = 0;
    for (int k = 0; k < ngroups; k++) { G.off[k] = offset[k]; G.cnt[k] = count[k]; G.lr_over_bc1[k] = lr[k] / bias_correction1; if (count[k] > nmax) nmax = count[k]; }
    size_t blocks = (nmax / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(adam_groups_kernel, dim3((unsigned)blocks, (unsigned)ngroups), dim3(256), 0, (hipStream_t)stream, G, param, grad, exp_avg,
                       exp_avg_sq, beta1, beta2, eps, 1.0f / bias_correction2_sqrt);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

// The same update over up to 8 SEPARATE tensors in one launch (igs_amd/optim.py: a torch.optim.Optimizer whose parameters are ordinary
// nn.Parameters, each with its own gradient / moment allocation and its own step count).  blockIdx.y = tensor.
struct AdamMulti { int n; float* p[8]; const float* g[8]; float* m[8]; float* v[8]; size_t cnt[8]; float lr_over_bc1[8]; float inv_sqrt_bc2[8]; float* step[8]; unsigned* done; };
#define ADAM_DONE_GROUP 64u          // workgroups per first-level done-counter
#define ADAM_DONE_STRIDE 32u         // words between counters (one cache line each)
#define ADAM_DONE_WORDS ((1u + (8u * 1024u + ADAM_DONE_GROUP - 1u) / ADAM_DONE_GROUP) * ADAM_DONE_STRIDE)
// DEV_STEP: the step counts live in device memory (one float per tensor, torch.optim.Adam's `capturable` layout), so that the launch can
// be replayed from a hipGraph: the bias corrections are computed here (in double, as the host does) instead of arriving as arguments,
// and lr_over_bc1 holds the plain learning rate.  step[k] = number of COMPLETED steps: every workgroup reads it when it starts and uses
// step + 1; the workgroup that finishes LAST -- after every other one has started, hence read -- advances the counts for the next launch
// (two-level done-counter in `done`, relaxed agent-scope atomics, no fences: refine_ops.hip l1_mean_kernel has the reason), so the
// update stays ONE launch.
template <bool DEV_STEP>
__global__ void __launch_bounds__(256)
adam_multi_kernel(const AdamMulti G, float b1, float b2, float eps)
{
    const int k = blockIdx.y;
    if (k >= G.n) return;
    const size_t n = G.cnt[k];
    float lr = G.lr_over_bc1[k], isb = G.inv_sqrt_bc2[k];
    if (DEV_STEP) {
        __shared__ float bc[2];
        if (threadIdx.x == 0) {
            const double t = (double)__hip_atomic_load(G.step[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.0;
            bc[0] = (float)((double)lr / (1.0 - pow((double)b1, t)));
            bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
        }
        __syncthreads();
        lr = bc[0]; isb = bc[1];
    }
    float* pp = G.p[k]; const float* gg = G.g[k]; float* mm = G.m[k]; float* vv = G.v[k];
    const bool aligned = (((uintptr_t)pp | (uintptr_t)gg | (uintptr_t)mm | (uintptr_t)vv) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P4 = ((float4*)pp)[i], M4 = ld_moment((const float4*)mm + i), V4 = ld_moment((const float4*)vv + i);
        const float4 G4 = ((const float4*)gg)[i];
        float* a = (float*)&P4; float* b = (float*)&M4; float* c = (float*)&V4; const float* d = (const float*)&G4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            b[q] = b1 * b[q] + (1.f - b1) * d[q];
            c[q] = b2 * c[q] + (1.f - b2) * d[q] * d[q];
            a[q] -= lr * b[q] / (sqrtf(c[q]) * isb + eps);
        }
        ((float4*)pp)[i] = P4; st_moment((float4*)mm + i, M4); st_moment((float4*)vv + i, V4);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = gg[i];
        const float mi = b1 * mm[i] + (1.f - b1) * gi;
        const float vi = b2 * vv[i] + (1.f - b2) * gi * gi;
        mm[i] = mi; vv[i] = vi;
        pp[i] -= lr * mi / (sqrtf(vi) * isb + eps);
    }
    if (DEV_STEP) {
        __syncthreads();                                   // (thread 0 read the counts long ago; every wave of the workgroup is done streaming)
        if (threadIdx.x == 0) {
            const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
            const unsigned g = id / ADAM_DONE_GROUP, ng = (total + ADAM_DONE_GROUP - 1u) / ADAM_DONE_GROUP;
            const unsigned gsize = (g == ng - 1u) ? total - g * ADAM_DONE_GROUP : ADAM_DONE_GROUP;
            unsigned* gc = G.done + ADAM_DONE_STRIDE * (1u + g);
            if (__hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
                __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(G.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u) {
                    __hip_atomic_store(G.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int q = 0; q < G.n; q++) {
                        const float c = __hip_atomic_load(G.step[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(G.step[q], c + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
}
static int adam_multi(void* stream, int ntensors, float* const* param, const float* const* grad, float* const* exp_avg,
                      float* const* exp_avg_sq, const size_t* count, const float* lr, const float* bias_correction1,
                      const float* bias_correction2_sqrt, float* const* step, unsigned* done, float beta1, float beta2, float eps)
{
    if (ntensors <= 0) return 0;
    if (ntensors > 8 || !param || !grad || !exp_avg || !exp_avg_sq || !count || !lr) return IGS_RAST_E_INVALID;
    if (!step && (!bias_correction1 || !bias_correction2_sqrt)) return IGS_RAST_E_INVALID;
    if (step && !done) return IGS_RAST_E_INVALID;
    AdamMulti G; G.n = ntensors; G.done = done;
    size_t nmax = 0;
    for (int k = 0; k < ntensors; k++) {
        if (count[k] && (!param[k] || !grad[k] || !exp_avg[k] || !exp_avg_sq[k])) return IGS_RAST_E_INVALID;
        if (step && !step[k]) return IGS_RAST_E_INVALID;
        G.p[k] = param[k]; G.g[k] = grad[k]; G.m[k] = exp_avg[k]; G.v[k] = exp_avg_sq[k]; G.cnt[k] = count[k];
        G.lr_over_bc1[k] = step ? lr[k] : lr[k] / bias_correction1[k]; G.inv_sqrt_bc2[k] = step ? 1.0f : 1.0f / bias_correction2_sqrt[k];
        G.step[k] = step ? step[k] : nullptr;
        if (count[k] > nmax) nmax = count[k];
    }
    size_t blocks = (nmax / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    const dim3 grid((unsigned)blocks, (unsigned)ntensors);
    if (step) {
        for (int a = 0; a < ntensors; a++)
            for (int b = a + 1; b < ntensors; b++)
                if (step[a] == step[b]) return IGS_RAST_E_INVALID;              // (one counter per tensor: each is advanced once)
        hipLaunchKernelGGL(adam_multi_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, G, beta1, beta2, eps);
    } else {
        hipLaunchKernelGGL(adam_multi_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, G, beta1, beta2, eps);
    }
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}
extern "C" int igs_adam_step_multi(void* stream, int ntensors, float* const* param, const float* const* grad, float* const* exp_avg,
                                   float* const* exp_avg_sq, const size_t* count, const float* lr, const float* bias_correction1,
                                   const float* bias_correction2_sqrt, float beta1, float beta2, float eps)
{
    if (!bias_correction1 || !bias_correction2_sqrt) return ntensors <= 0 ? 0 : IGS_RAST_E_INVALID;
    return adam_multi(stream, ntensors, param, grad, exp_avg, exp_avg_sq, count, lr, bias_correction1, bias_correction2_sqrt, nullptr, nullptr,
                      beta1, beta2, eps);
}
extern "C" size_t igs_adam_step_multi_dev_scratch_words(void) { return ADAM_DONE_WORDS; }
extern "C" int igs_adam_step_multi_dev(void* stream, int ntensors, float* const* param, const float* const* grad, float* const* exp_avg,
                                       float* const* exp_avg_sq, const size_t* count, const float* lr, float* const* step,
                                       unsigned* done_scratch, float beta1, float beta2, float eps)
{
    if (!step || !done_scratch) return ntensors <= 0 ? 0 : IGS_RAST_E_INVALID;
    return adam_multi(stream, ntensors, param, grad, exp_avg, exp_avg_sq, count, lr, nullptr, nullptr, step, done_scratch, beta1, beta2, eps);
}

// mean |pred - gt| and its gradient in ONE launch, the value finished on the device: every workgroup leaves its partial sum in
// partials[blockIdx.x]; the workgroup that finishes last (self-resetting counter) adds the partials IN INDEX ORDER -- the value does not
// depend on which workgroup that was -- and stores the mean.  For `igs_amd.losses.l1_loss` (loss_utils.py:17-18) as an autograd Function:
// PyTorch's sub / abs / mean and their three backward kernels become this launch plus one scale.
__global__ void __launch_bounds__(256)
l1_mean_kernel(size_t n4, size_t n, const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ grad,
               float* __restrict__ mean_out, float* __restrict__ partials, unsigned* __restrict__ counter, float inv_n)
{
    __shared__ float ws[4];
    __shared__ bool last;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    // four float4 pairs per thread in flight before the first is used: with one pair per loop trip (and `s_waitcnt vmcnt(0)` behind it) a
    // wave had 2 KB outstanding and the kernel streamed its 49 MB at 1.2 TB/s (40 us at 1352 x 1014)
    for (size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; base < n4; base += stride * 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = base + (size_t)u * 256;
            if (i < n4) { a[u] = ((const float4*)pred)[i]; b[u] = ((const float4*)gt)[i]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = base + (size_t)u * 256;
            if (i < n4) {
                float4 g;
                float d;
                d = a[u].x - b[u].x; acc += fabsf(d); g.x = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
                d = a[u].y - b[u].y; acc += fabsf(d); g.y = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
                d = a[u].z - b[u].z; acc += fabsf(d); g.z = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
                d = a[u].w - b[u].w; acc += fabsf(d); g.w = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
                ((float4*)grad)[i] = g;
            }
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float d = pred[i] - gt[i];
        acc += fabsf(d);
        grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // Publish the partial sum WITHOUT a device-scope release fence: on this GPU `__threadfence()` is `buffer_wbl2 sc1` -- write back
        // every dirty line of the XCD's L2 -- and this kernel has just stored 16 MB of gradient through that L2: a thousand workgroups
        // each waiting for such a write-back made the kernel take 40 us for 49 MB of traffic (round 4; the two-level counter and the
        // four loads in flight below changed nothing until the fences went).  An agent-scope atomic store is written through by itself;
        // `s_waitcnt vmcnt(0)` holds the counter increment back until it has been acknowledged.
        __hip_atomic_store(partials + blockIdx.x, ws[0] + ws[1] + ws[2] + ws[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // "who is last" over two levels (groups of 32 workgroups, then the groups; the group words 256 bytes apart): returning atomics on
        // one address retire one by one, and the workgroups of a streaming kernel all finish together
        const unsigned g = blockIdx.x >> 5, ng = (gridDim.x + 31u) >> 5;
        const unsigned gsize = (g == ng - 1u) ? gridDim.x - (g << 5) : 32u;
        bool l = false;
        if (__hip_atomic_fetch_add(counter + 64u * (1u + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
            __hip_atomic_store(counter + 64u * (1u + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u;
        }
        last = l;
    }
    __syncthreads();
    if (!last) return;
    float t = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)                       // (fixed order per lane; agent-scope loads: not from a stale L2 line)
        t += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { mean_out[0] = (ws[0] + ws[1] + ws[2] + ws[3]) * inv_n; __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
extern "C" int igs_l1_mean_fwd_bwd(void* stream, size_t n, const float* pred, const float* gt, float* grad, float* mean_out, float* partials,
                                   unsigned* counter)
{
    if (n == 0) return IGS_RAST_E_INVALID;
    if (!pred || !gt || !grad || !mean_out || !partials || !counter) return IGS_RAST_E_INVALID;
    const bool aligned = (((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)grad) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    size_t blocks = (n / 4 + 1023) / 1024;          // (1024 float4 per workgroup and trip)
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(l1_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, n, pred, gt, grad, mean_out, partials, counter,
                       (float)(1.0 / (double)n));
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

// L1: loss_sum += sum |pred - gt| ; grad = sign(pred - gt) * scale        (mean => scale = upstream / n)
__global__ void __launch_bounds__(256)
l1_kernel(size_t n4, size_t n, const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ grad, float* __restrict__ loss_sum, float scale)
{
    __shared__ float ws[4];
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 a = ((const float4*)pred)[i], b = ((const float4*)gt)[i];
        float4 g;
        float d;
        d = a.x - b.x; acc += fabsf(d); g.x = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        d = a.y - b.y; acc += fabsf(d); g.y = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        d = a.z - b.z; acc += fabsf(d); g.z = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        d = a.w - b.w; acc += fabsf(d); g.w = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        ((float4*)grad)[i] = g;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float d = pred[i] - gt[i];
        acc += fabsf(d);
        grad[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    // 64 shards one cache line apart (same-address atomics serialise); the caller sums loss_sum[16*s], s < 64
    if (threadIdx.x == 0) atomicAdd(&loss_sum[16 * (blockIdx.x & 63)], ws[0] + ws[1] + ws[2] + ws[3]);
}

extern "C" int igs_l1_loss_fwd_bwd(void* stream, size_t n, const float* pred, const float* gt, float* grad, float* loss_sum, float scale)
{
    if (n == 0) return 0;
    if (!pred || !gt || !grad || !loss_sum) return IGS_RAST_E_INVALID;
    const bool aligned = (((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)grad) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, n, pred, gt, grad, loss_sum, scale);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

// ---- fused activations of the refine loop (igs/models/gaussian_model.py:90-127: sigmoid / exp / F.normalize) ----
__global__ void __launch_bounds__(256)
activate_fwd_kernel(int P, const float* __restrict__ logit, const float* __restrict__ log_scale, const float* __restrict__ rot,
                    float* __restrict__ opacity, float* __restrict__ scale, float* __restrict__ rot_n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    opacity[i] = act_sigmoid(logit[i]);
#pragma unroll
    for (int k = 0; k < 3; k++) scale[3 * i + k] = act_exp(log_scale[3 * i + k]);
    const float q0 = rot[4 * i], q1 = rot[4 * i + 1], q2 = rot[4 * i + 2], q3 = rot[4 * i + 3];
    const float inv = act_inv_norm4(q0, q1, q2, q3);
    rot_n[4 * i] = q0 * inv; rot_n[4 * i + 1] = q1 * inv; rot_n[4 * i + 2] = q2 * inv; rot_n[4 * i + 3] = q3 * inv;
}
__global__ void __launch_bounds__(256)
activate_bwd_kernel(int P, const float* __restrict__ opacity, const float* __restrict__ scale, const float* __restrict__ rot,
                    const float* __restrict__ d_opacity, const float* __restrict__ d_scale, const float* __restrict__ d_rot,
                    float* __restrict__ g_logit, float* __restrict__ g_log_scale, float* __restrict__ g_rot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float s = opacity[i];
    g_logit[i] = d_opacity[i] * s * (1.0f - s);
#pragma unroll
    for (int k = 0; k < 3; k++) g_log_scale[3 * i + k] = d_scale[3 * i + k] * scale[3 * i + k];
    const float q0 = rot[4 * i], q1 = rot[4 * i + 1], q2 = rot[4 * i + 2], q3 = rot[4 * i + 3];
    const float g0 = d_rot[4 * i], g1 = d_rot[4 * i + 1], g2 = d_rot[4 * i + 2], g3 = d_rot[4 * i + 3];
    const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    const float inv = 1.0f / nrm;
    const float n0 = q0 * inv, n1 = q1 * inv, n2 = q2 * inv, n3 = q3 * inv;
    const float dt = n0 * g0 + n1 * g1 + n2 * g2 + n3 * g3;
    g_rot[4 * i] = (g0 - n0 * dt) * inv; g_rot[4 * i + 1] = (g1 - n1 * dt) * inv;
    g_rot[4 * i + 2] = (g2 - n2 * dt) * inv; g_rot[4 * i + 3] = (g3 - n3 * dt) * inv;
}
extern "C" int igs_activate_fwd(void* stream, int P, const float* logit, const float* log_scale, const float* rot, float* opacity,
                                float* scale, float* rot_n)
{
    if (P <= 0) return 0;
    if (!logit || !log_scale || !rot || !opacity || !scale || !rot_n) return IGS_RAST_E_INVALID;
    hipLaunchKernelGGL(activate_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, logit, log_scale, rot, opacity, scale, rot_n);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}
extern "C" int igs_activate_bwd(void* stream, int P, const float* opacity, const float* scale, const float* rot, const float* d_opacity,
                                const float* d_scale, const float* d_rot, float* g_logit, float* g_log_scale, float* g_rot)
{
    if (P <= 0) return 0;
    if (!opacity || !scale || !rot || !d_opacity || !d_scale || !d_rot || !g_logit || !g_log_scale || !g_rot) return IGS_RAST_E_INVALID;
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, opacity, scale, rot, d_opacity, d_scale, d_rot, g_logit, g_log_scale, g_rot);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

// ---- densification support (igs/models/gaussian_model.py:586-663,865-868; infer_batch.py:308-321) -------------------------
// per-step statistics: for every Gaussian visible in the view (radii > 0):
//   xyz_gradient_accum += || dL/dmean2D[:2] ||,  denom += 1,  max_radii2D = max(max_radii2D, radii)
__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const float* __restrict__ dL_dmean2D, const int* __restrict__ radii, float* __restrict__ grad_accum,
                     float* __restrict__ denom, float* __restrict__ max_radii)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r > 0) {
        const float gx = dL_dmean2D[3 * i], gy = dL_dmean2D[3 * i + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);
        denom[i] += 1.0f;
        max_radii[i] = fmaxf(max_radii[i], (float)r);
    }
}
extern "C" int igs_densify_stats(void* stream, int P, const float* dL_dmean2D, const int* radii, float* grad_accum, float* denom,
                                 float* max_radii)
{
    if (P <= 0) return 0;
    if (!dL_dmean2D || !radii || !grad_accum || !denom || !max_radii) return IGS_RAST_E_INVALID;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, dL_dmean2D, radii, grad_accum, denom, max_radii);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

// Rebuilds the flat optimiser state after clone / split / prune in ONE pass (the reference re-creates five nn.Parameters and
// ten Adam-state tensors with masks and torch.cat: _prune_optimizer / cat_tensors_to_optimizer, gaussian_model.py:466-557).
// New Gaussian i is a copy of old Gaussian src[i]; fresh[i] != 0: a Gaussian created by this densification (Adam moments
// start at zero); ovr[i] >= 0: a split child whose position and log-scale come from row ovr[i] of ovr_xyz / ovr_scale.
struct RemapArgs {
    int P_new, M;
    const int *src, *fresh, *ovr;
    const float *ovr_xyz, *ovr_scale;
    const float *p_old, *m_old, *v_old; float *p_new, *m_new, *v_new;
    size_t off_old[5], off_new[5];          // xyz, rotation, shs, opacity, scaling
};
__global__ void __launch_bounds__(256)
densify_remap_kernel(const RemapArgs a)
{
    const int per = 11 + 3 * a.M;                                   // floats per Gaussian
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)a.P_new * per) return;
    // group-major work order so that consecutive threads write consecutive addresses
    const size_t P = (size_t)a.P_new;
    int g; size_t rem = t;
    const size_t n0 = 3 * P, n1 = 4 * P, n2 = (size_t)3 * a.M * P, n3 = P;
    if (rem < n0) g = 0; else if ((rem -= n0) < n1) g = 1; else if ((rem -= n1) < n2) g = 2; else if ((rem -= n2) < n3) g = 3; else { rem -= n3; g = 4; }
    const int k = g == 0 ? 3 : g == 1 ? 4 : g == 2 ? 3 * a.M : g == 3 ? 1 : 3;
    const size_t i = rem / k; const int c = (int)(rem - i * k);
    const int s = a.src[i];
    const size_t so = a.off_old[g] + (size_t)s * k + c, dn = a.off_new[g] + rem;
    float pv = a.p_old[so];
    const int o = a.ovr ? a.ovr[i] : -1;
    if (o >= 0) { if (g == 0) pv = a.ovr_xyz[3 * (size_t)o + c]; else if (g == 4) pv = a.ovr_scale[3 * (size_t)o + c]; }
    const bool fresh = a.fresh && a.fresh[i] != 0;
    a.p_new[dn] = pv;
    a.m_new[dn] = fresh ? 0.f : a.m_old[so];
    a.v_new[dn] = fresh ? 0.f : a.v_old[so];
}
extern "C" int igs_densify_remap(void* stream, int P_new, int M, const int* src, const int* fresh, const int* ovr, const float* ovr_xyz,
                                 const float* ovr_scale, const float* param_old, const float* exp_avg_old, const float* exp_avg_sq_old,
                                 const size_t* off_old, float* param_new, float* exp_avg_new, float* exp_avg_sq_new, const size_t* off_new)
{
    if (P_new <= 0) return 0;
    if (!src || !param_old || !exp_avg_old || !exp_avg_sq_old || !off_old || !param_new || !exp_avg_new || !exp_avg_sq_new || !off_new || M < 0)
        return IGS_RAST_E_INVALID;
    if (ovr && (!ovr_xyz || !ovr_scale)) return IGS_RAST_E_INVALID;
    RemapArgs a;
    a.P_new = P_new; a.M = M; a.src = src; a.fresh = fresh; a.ovr = ovr; a.ovr_xyz = ovr_xyz; a.ovr_scale = ovr_scale;
    a.p_old = param_old; a.m_old = exp_avg_old; a.v_old = exp_avg_sq_old; a.p_new = param_new; a.m_new = exp_avg_new; a.v_new = exp_avg_sq_new;
    for (int g = 0; g < 5; g++) { a.off_old[g] = off_old[g]; a.off_new[g] = off_new[g]; }
    const size_t total = (size_t)P_new * (11 + 3 * (size_t)M);
    hipLaunchKernelGGL(densify_remap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

__global__ void __launch_bounds__(256) zero_fill_kernel(uint32_t* __restrict__ p, size_t words, uint32_t* __restrict__ extra, int extra_words)
{
    if (blockIdx.x == 0 && (int)threadIdx.x < extra_words) extra[threadIdx.x] = 0u;      // (a second, tiny range: at most 256 words)
    // 16-byte stores over the aligned middle, single words at the ragged ends
    const size_t head = min(words, (size_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4);
    const size_t n4 = (words - head) / 4;
    uint4* q = (uint4*)(p + head);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) q[i] = z;
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) p[threadIdx.x] = 0u;
        const size_t tail0 = head + 4 * n4;
        if (tail0 + threadIdx.x < words) p[tail0 + threadIdx.x] = 0u;
    }
}
hipError_t zero_fill_async(hipStream_t s, void* p, size_t bytes, void* extra, int extra_words)
{
    if (bytes == 0 && extra_words == 0) return hipSuccess;
    if (extra_words < 0 || extra_words > 256) return hipErrorInvalidValue;
    if (extra_words == 0 && ((((uintptr_t)p) | bytes) & 3u)) return hipMemsetAsync(p, 0, bytes, s);      // (not word-granular: never the case in this library)
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, words, (uint32_t*)extra, extra_words);
    return hipGetLastError();
}

