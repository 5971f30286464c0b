// io_ops.hip -- the start-Gaussian PLY table <-> the flat parameter store, on the device.
// Reference: igs/models/gs.py:400-462 (load_ply: column gathering, channel-major SH -> [P, K, 3], the Mip-Splatting 3-D filter folded
// into scale and opacity, :480-490) and :297-343 (save_ply: property order, normals = 0, f_dc / f_rest channel-major).
// The file itself is read / written by the host (igs_amd/io.py); what runs here is the per-Gaussian re-layout and arithmetic that the
// reference does with a dozen numpy / torch passes over [P, 62] tables.
#include "common.h"
#include "../../include/igs_rast.h"

struct PlyCols { int xyz[3], f_dc[3], f_rest[45], opacity, scale[3], rot[4], filter; };

// one thread per Gaussian: its `stride` floats of the vertex table -> its rows of the five parameter arrays
__global__ void __launch_bounds__(256)
ply_to_params_kernel(int P, const float* __restrict__ table, int stride, const PlyCols c, int K,
                     float* __restrict__ xyz, float* __restrict__ rot, float* __restrict__ shs, float* __restrict__ opacity, float* __restrict__ scaling)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* v = table + (size_t)i * stride;
#pragma unroll
    for (int k = 0; k < 3; k++) xyz[3 * (size_t)i + k] = v[c.xyz[k]];
#pragma unroll
    for (int k = 0; k < 4; k++) rot[4 * (size_t)i + k] = v[c.rot[k]];
    // SH: the file holds f_dc_[channel] and f_rest_[channel * (K - 1) + coefficient - 1] (channel-major, gs.py:327-332); the store is
    // [P][K][3] (coefficient-major): shs[i][0][ch] = f_dc[ch], shs[i][k][ch] = f_rest[ch * (K - 1) + k - 1]
    float* s = shs + (size_t)i * K * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) s[ch] = v[c.f_dc[ch]];
    for (int k = 1; k < K; k++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) s[3 * k + ch] = v[c.f_rest[ch * (K - 1) + k - 1]];
    float ls[3] = { v[c.scale[0]], v[c.scale[1]], v[c.scale[2]] };
    float logit = v[c.opacity];
    if (c.filter >= 0) {
        // get_scaling_n_opacity_with_3D_filter (gs.py:480-490) followed by inverse_sigmoid / log (:451-455)
        const float f = v[c.filter], f2 = f * f;
        const float op = 1.0f / (1.0f + expf(-logit));
        float det1 = 1.f, det2 = 1.f, after[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { const float e = expf(ls[k]); const float s2 = e * e; after[k] = s2 + f2; det1 *= s2; det2 *= after[k]; }
        const float o2 = op * sqrtf(det1 / det2);
        logit = logf(o2 / (1.0f - o2));
#pragma unroll
        for (int k = 0; k < 3; k++) ls[k] = logf(sqrtf(after[k]));
    }
    opacity[i] = logit;
#pragma unroll
    for (int k = 0; k < 3; k++) scaling[3 * (size_t)i + k] = ls[k];
}

// the reverse: the table save_ply writes (gs.py:297-343): x y z | nx ny nz = 0 | f_dc 3 | f_rest 45 | opacity | scale 3 | rot 4  (62 floats)
__global__ void __launch_bounds__(256)
params_to_ply_kernel(int P, int K, const float* __restrict__ xyz, const float* __restrict__ rot, const float* __restrict__ shs,
                     const float* __restrict__ opacity, const float* __restrict__ scaling, float* __restrict__ table)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int stride = 6 + 3 * K + 1 + 3 + 4;
    float* v = table + (size_t)i * stride;
    const float* s = shs + (size_t)i * K * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k] = xyz[3 * (size_t)i + k]; v[3 + k] = 0.f; v[6 + k] = s[k]; }
    for (int ch = 0; ch < 3; ch++)
        for (int k = 1; k < K; k++) v[9 + ch * (K - 1) + k - 1] = s[3 * k + ch];
    float* t = v + 6 + 3 * K;
    t[0] = opacity[i];
#pragma unroll
    for (int k = 0; k < 3; k++) t[1 + k] = scaling[3 * (size_t)i + k];
#pragma unroll
    for (int k = 0; k < 4; k++) t[4 + k] = rot[4 * (size_t)i + k];
}

extern "C" int igs_ply_to_params(void* stream, int P, const float* table, int stride, const int* cols, int n_cols, int K,
                                 float* xyz, float* rotation, float* shs, float* opacity, float* scaling)
{
    if (P < 0 || K < 1 || K > 16 || stride <= 0) return IGS_RAST_E_INVALID;
    if (P == 0) return 0;
    const int need = 3 + 3 + 3 * (K - 1) + 1 + 3 + 4 + 1;
    if (!table || !cols || n_cols != need || !xyz || !rotation || !shs || !opacity || !scaling) return IGS_RAST_E_INVALID;
    PlyCols c;
    int j = 0;
    for (int k = 0; k < 3; k++) c.xyz[k] = cols[j++];
    for (int k = 0; k < 3; k++) c.f_dc[k] = cols[j++];
    for (int k = 0; k < 45; k++) c.f_rest[k] = k < 3 * (K - 1) ? cols[j++] : 0;
    c.opacity = cols[j++];
    for (int k = 0; k < 3; k++) c.scale[k] = cols[j++];
    for (int k = 0; k < 4; k++) c.rot[k] = cols[j++];
    c.filter = cols[j++];
    for (int k = 0; k < need - 1; k++) if (cols[k] < 0 || cols[k] >= stride) return IGS_RAST_E_INVALID;      // (only the filter column may be absent: -1)
    if (c.filter >= stride) return IGS_RAST_E_INVALID;
    hipLaunchKernelGGL(ply_to_params_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, table, stride, c, K, xyz, rotation, shs,
                       opacity, scaling);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

extern "C" int igs_params_to_ply(void* stream, int P, int K, const float* xyz, const float* rotation, const float* shs, const float* opacity,
                                 const float* scaling, float* table)
{
    if (P < 0 || K < 1 || K > 16) return IGS_RAST_E_INVALID;
    if (P == 0) return 0;
    if (!xyz || !rotation || !shs || !opacity || !scaling || !table) return IGS_RAST_E_INVALID;
    hipLaunchKernelGGL(params_to_ply_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, K, xyz, rotation, shs, opacity, scaling, table);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}
