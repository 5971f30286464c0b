// igs_torch_ext.cpp -- the compiled `_C` module of the drop-in packages: the four functions the reference's pybind module exports
// (DGR/ext.cpp:15-20; DGR = submodules/RaDe-GS/submodules/diff-gaussian-rasterization) as torch glue over the C ABI of
// include/igs_rast.h.  Counterpart of DGR/rasterize_points.cu:35-267 (RasterizeGaussiansCUDA, RasterizeGaussiansBackwardCUDA,
// markVisible): tensor checks, output allocation, pointer extraction, PyTorch's CURRENT stream -- no arithmetic.
//
// Built by igs_amd/build_ext.py with the host compiler (no device code in this file); links libigs_rast.so.
//
// Extensions over the reference's signatures are keyword-only extras with defaults (the positional lists are the reference's):
//   rasterize_gaussians(..., scratch=None, out_images=None, out_radii=None, mode=0, scratch_clean=False)
//   rasterize_gaussians_backward(..., workspace=None, out_*=None)         any upstream gradient may be None (= zeros)
//   rasterize_gaussians_backward_ex(...same..., nan_report=0|1|2, clamp=0.0) -> (8 gradients, nan flag, verdict word, sequence number)
#include <torch/extension.h>
// PyTorch-ROCm presents its HIP devices under the device type "cuda" (so that `device="cuda"` callers run unchanged): guards and
// the current stream come from the classes that know about that
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <limits.h>
#include <map>
#include <mutex>

#include "../../include/igs_rast.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

struct RasterizerError : public std::runtime_error { using std::runtime_error::runtime_error; };

// uint8 scratch tensors grown on demand by the library (rasterize_points.cu:27-33, resizeFunctional).  `persistent` sets are born
// zero-filled and only grow (by 25 %): the library leaves its binning counters zeroed after every frame, which lets a caller that
// keeps its set skip the per-frame zero-fill launch (igs_rast_hint_scratch_clean).
struct Grow { at::Tensor* t; bool persistent; };
char* grow_cb(void* user, size_t n)
{
    Grow* g = (Grow*)user;
    try {
        if ((size_t)g->t->numel() < n) {
            auto o = g->t->options();
            *g->t = g->persistent ? at::zeros({(int64_t)(n + n / 4)}, o) : at::empty({(int64_t)n}, o);
        }
        return (char*)g->t->data_ptr();
    } catch (...) {                    // an exception must not cross the C frame
        return nullptr;
    }
}
struct ScratchSet {
    Tensor geom, binning, img, workspace;
    bool persistent;
    c10::Device device;
    Grow g_geom, g_binning, g_img;      // the `user` words of the three growth callbacks (also handed to igs_refine_step by address)
    ScratchSet(c10::Device dev, bool persistent_) : persistent(persistent_), device(dev)
    {
        auto o = at::TensorOptions().dtype(at::kByte).device(dev);
        geom = at::empty({0}, o); binning = at::empty({0}, o); img = at::empty({0}, o); workspace = at::empty({0}, o);
        g_geom = Grow{ &geom, persistent }; g_binning = Grow{ &binning, persistent }; g_img = Grow{ &img, persistent };
    }
    ScratchSet(const ScratchSet&) = delete;
    ScratchSet& operator=(const ScratchSet&) = delete;
    Tensor& ensure_workspace(int64_t P)
    {
        const int64_t need = (int64_t)igs_rast_backward_workspace_bytes((int)P);
        if (workspace.numel() < need) workspace = at::empty({need}, at::TensorOptions().dtype(at::kByte).device(device));
        return workspace;
    }
};

// float32, contiguous, on `dev`; the reference's "empty tensor = absent" convention gives NULL
struct In {
    Tensor keep; const float* p = nullptr;
    In() {}
    In(const OptTensor& t, const c10::Device& dev, const char* what)
    {
        if (!t.has_value() || !t->defined() || t->numel() == 0) return;
        if (t->device() != dev) throw RasterizerError(std::string(what) + " must live on " + dev.str() + " (got " + t->device().str() + ")");
        keep = (t->scalar_type() == at::kFloat && t->is_contiguous()) ? *t : t->to(at::kFloat).contiguous();
        p = keep.data_ptr<float>();
    }
};

// a caller-provided destination tensor: right device, dtype, size, and contiguous (the kernels write raw pointers)
void check_out(const Tensor& t, int64_t numel, at::ScalarType dt, const c10::Device& dev, const char* what)
{
    if (!t.defined() || t.device() != dev || t.scalar_type() != dt || t.numel() != numel || !t.is_contiguous())
        throw RasterizerError(std::string(what) + ": must be a contiguous " + c10::toString(dt) + " tensor of " + std::to_string(numel) +
                              " elements on " + dev.str());
}

void check(int rc, const char* what)
{
    if (rc < 0) throw RasterizerError(std::string(what) + " failed (" + std::to_string(rc) + "): " + igs_rast_last_error());
}

using FwdTuple = std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;

// _C.rasterize_gaussians (RasterizeGaussiansCUDA, DGR/rasterize_points.cu:35-133).
// Returns (num_rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geomBuffer, binningBuffer, imgBuffer).
// mode 0: one host wait for the instance count (as the reference); 1: igs_rast_forward_async (caller must call forward_finish);
// 2: igs_rast_forward_nowait (stream capture).  In modes 1 / 2 num_rendered is INT_MAX, which the backward accepts.
FwdTuple rasterize_gaussians(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix, const Tensor& projmatrix,
    double tan_fovx, double tan_fovy, double kernel_size, int64_t image_height, int64_t image_width, const Tensor& sh, int64_t degree,
    const Tensor& campos, bool prefiltered, bool require_coord, bool require_depth, bool debug,
    const std::shared_ptr<ScratchSet>& scratch, const OptTensor& out_images, const OptTensor& out_radii, int64_t mode, bool scratch_clean)
{
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw RasterizerError("means3D must have dimensions (num_points, 3)");
    if (!means3D.is_cuda()) throw RasterizerError("igs_amd rasterizer: tensors must be on a GPU (no CPU fallback)");
    const c10::Device dev = means3D.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0), H = image_height, W = image_width;
    In m3(means3D, dev, "means3D"), col(colors, dev, "colors_precomp"), op(opacity, dev, "opacities"), sc(scales, dev, "scales"),
       rot(rotations, dev, "rotations"), cov(cov3D_precomp, dev, "cov3D_precomp"), shs(sh, dev, "shs"), bg(background, dev, "bg"),
       view(viewmatrix, dev, "viewmatrix"), proj(projmatrix, dev, "projmatrix"), cam(campos, dev, "campos");
    const int64_t M = shs.p ? shs.keep.size(1) : 0;
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    // one allocation for the seven images; every pixel is written by the kernels when P > 0
    if (H <= 0 || W <= 0) throw RasterizerError("image_height and image_width must be positive");
    if (out_images.has_value()) check_out(*out_images, 15 * H * W, at::kFloat, dev, "out_images");
    if (out_radii.has_value()) check_out(*out_radii, P, at::kInt, dev, "out_radii");
    Tensor imgs = out_images.has_value() ? *out_images : (P > 0 ? at::empty({15, H, W}, fopt) : at::zeros({15, H, W}, fopt));
    Tensor radii = out_radii.has_value() ? *out_radii : (P > 0 ? at::empty({P}, fopt.dtype(at::kInt)) : at::zeros({0}, fopt.dtype(at::kInt)));
    if (scratch && scratch->device != dev) throw RasterizerError("scratch set lives on " + scratch->device.str() + ", the tensors on " + dev.str());
    std::shared_ptr<ScratchSet> ss = scratch ? scratch : std::make_shared<ScratchSet>(dev, false);
    Tensor color = imgs.narrow(0, 0, 3), coord = imgs.narrow(0, 3, 3), mcoord = imgs.narrow(0, 6, 3), depth = imgs.narrow(0, 9, 1),
           mdepth = imgs.narrow(0, 10, 1), alpha = imgs.narrow(0, 11, 1), normal = imgs.narrow(0, 12, 3);
    int64_t rendered = 0;
    if (P != 0) {
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        auto fwd = mode == 1 ? igs_rast_forward_async : (mode == 2 ? igs_rast_forward_nowait : igs_rast_forward);
        if (scratch_clean) igs_rast_hint_scratch_clean(1);
        float* ib = imgs.data_ptr<float>();
        const size_t HW = (size_t)H * W;
        rendered = fwd(stream, grow_cb, &ss->g_geom, grow_cb, &ss->g_binning, grow_cb, &ss->g_img, (int)P, (int)degree, (int)M, bg.p, (int)W, (int)H, m3.p, shs.p, col.p,
                       op.p, sc.p, (float)scale_modifier, rot.p, cov.p, view.p, proj.p, cam.p, (float)tan_fovx, (float)tan_fovy,
                       (float)kernel_size, prefiltered ? 1 : 0, ib, ib + 3 * HW, ib + 6 * HW, ib + 9 * HW, ib + 10 * HW, ib + 11 * HW,
                       ib + 12 * HW, radii.data_ptr<int>(), require_coord ? 1 : 0, require_depth ? 1 : 0, debug ? 1 : 0);
        check((int)rendered, "igs_rast_forward");
    }
    return FwdTuple(rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, ss->geom, ss->binning, ss->img);
}

using BwdTuple = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;

// Body of _C.rasterize_gaussians_backward (RasterizeGaussiansBackwardCUDA, DGR/rasterize_points.cu:135-246).  The seven small
// gradients are carved from ONE [23 P] block (m2d 3 | colors 3 | opacity 1 | means3D 3 | scales 3 | rot 4 | cov3D 6); every element is
// written by the kernels (no zero fills).  Returns the reference's 8-tuple and, with nan_report, whether a NaN was written.
struct BwdResult { BwdTuple grads; int64_t nan = 0; int64_t nan_word = 0; int64_t nan_seq = 0; };
BwdResult backward_body(
    const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& scales, const Tensor& rotations,
    double scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy,
    double kernel_size, const OptTensor& dL_dout_color, const OptTensor& dL_dout_coord, const OptTensor& dL_dout_mcoord,
    const OptTensor& dL_dout_depth, const OptTensor& dL_dout_mdepth, const OptTensor& dL_dout_alpha, const OptTensor& dL_dout_normal,
    const Tensor& normalmap, const Tensor& sh, int64_t degree, const Tensor& campos, const Tensor& geomBuffer, int64_t R,
    const Tensor& binningBuffer, const Tensor& imageBuffer, const Tensor& alphas, bool require_coord, bool require_depth, bool debug,
    const OptTensor& workspace, const OptTensor& out_means2D, const OptTensor& out_colors, const OptTensor& out_opacity,
    const OptTensor& out_means3D, const OptTensor& out_cov3D, const OptTensor& out_sh, const OptTensor& out_scales,
    const OptTensor& out_rotations, int64_t nan_report, double clamp)
{
    if (!means3D.is_cuda()) throw RasterizerError("igs_amd rasterizer: tensors must be on a GPU (no CPU fallback)");
    const c10::Device dev = means3D.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0);
    const int64_t H = alphas.size(-2), W = alphas.size(-1);
    In shs(sh, dev, "shs");
    const int64_t M = shs.p ? shs.keep.size(1) : 0;
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor dL_dsh = out_sh.has_value() ? *out_sh : (P > 0 ? at::empty({P, M, 3}, fopt) : at::zeros({P, M, 3}, fopt));
    Tensor block = P > 0 ? at::empty({23 * P}, fopt) : at::zeros({0}, fopt);
    int64_t o = 0;
    if (out_sh.has_value()) check_out(*out_sh, P * M * 3, at::kFloat, dev, "out_sh");
    auto carve = [&](int64_t k, const OptTensor& given) {
        if (given.has_value()) check_out(*given, k * P, at::kFloat, dev, "gradient destination");
        Tensor t = given.has_value() ? *given : block.narrow(0, o, k * P).view({P, k});
        o += k * P;
        return t;
    };
    Tensor dL_dmeans2D = carve(3, out_means2D), dL_dcolors = carve(3, out_colors), dL_dopacity = carve(1, out_opacity),
           dL_dmeans3D = carve(3, out_means3D), dL_dscales = carve(3, out_scales), dL_drotations = carve(4, out_rotations),
           dL_dcov3D = carve(6, out_cov3D);
    BwdResult res;
    if (P != 0) {
        In m3(means3D, dev, "means3D"), col(colors, dev, "colors_precomp"), sc(scales, dev, "scales"), rot(rotations, dev, "rotations"),
           cov(cov3D_precomp, dev, "cov3D_precomp"), bg(background, dev, "bg"), view(viewmatrix, dev, "viewmatrix"),
           proj(projmatrix, dev, "projmatrix"), cam(campos, dev, "campos"), al(alphas, dev, "alphas"), nm(normalmap, dev, "normalmap");
        In g0(dL_dout_color, dev, "grad"), g1(dL_dout_coord, dev, "grad"), g2(dL_dout_mcoord, dev, "grad"), g3(dL_dout_depth, dev, "grad"),
           g4(dL_dout_mdepth, dev, "grad"), g5(dL_dout_alpha, dev, "grad"), g6(dL_dout_normal, dev, "grad");
        Tensor radii_c = radii.contiguous();
        const int64_t need = (int64_t)igs_rast_backward_workspace_bytes((int)P);
        Tensor ws = (workspace.has_value() && workspace->numel() >= need) ? *workspace
                                                                           : at::empty({need}, at::TensorOptions().dtype(at::kByte).device(dev));
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        if (nan_report || clamp > 0.0) igs_rast_next_backward_options(nan_report ? 1 : 0, (float)clamp);
        auto bp = [](const Tensor& t) { return t.numel() ? (const char*)t.data_ptr() : nullptr; };
        const int rc = igs_rast_backward(
            stream, (int)P, (int)degree, (int)M, (int)std::min<int64_t>(R, INT_MAX), bg.p, (int)W, (int)H, m3.p, shs.p, col.p, al.p, sc.p,
            (float)scale_modifier, rot.p, cov.p, view.p, proj.p, cam.p, (float)tan_fovx, (float)tan_fovy, (float)kernel_size,
            radii_c.data_ptr<int>(), nm.p, bp(geomBuffer), bp(binningBuffer), bp(imageBuffer), g0.p, g1.p, g2.p, g3.p, g4.p, g5.p, g6.p,
            ws.data_ptr(), dL_dmeans2D.data_ptr<float>(), dL_dcolors.data_ptr<float>(), dL_dopacity.data_ptr<float>(),
            dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(), M > 0 ? dL_dsh.data_ptr<float>() : nullptr,
            dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(), require_coord ? 1 : 0, require_depth ? 1 : 0, debug ? 1 : 0);
        check(rc, "igs_rast_backward");
        if (nan_report == 1) {                 // wait here (the reference's place for its asserts)
            const int v = igs_rast_nan_report_wait();
            check(v, "igs_rast_nan_report_wait");
            res.nan = v;
        } else if (nan_report == 2) {          // hand the verdict's address out: the caller waits once the whole backward pass is enqueued
            const void* word = nullptr; unsigned seq = 0;
            check(igs_rast_nan_report_handle(&word, &seq), "igs_rast_nan_report_handle");
            res.nan_word = (int64_t)(uintptr_t)word; res.nan_seq = (int64_t)seq;
        }
    }
    res.grads = BwdTuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
    return res;
}

// _C.mark_visible (DGR/rasterize_points.cu:248-267)
Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix)
{
    if (!means3D.is_cuda()) throw RasterizerError("igs_amd rasterizer: tensors must be on a GPU (no CPU fallback)");
    const c10::Device dev = means3D.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0);
    Tensor present = at::zeros({P}, at::TensorOptions().dtype(at::kBool).device(dev));
    if (P != 0) {
        In m(means3D, dev, "means3D"), v(viewmatrix, dev, "viewmatrix"), p(projmatrix, dev, "projmatrix");
        check(igs_rast_mark_visible(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(), (int)P, m.p, v.p, p.p, (uint8_t*)present.data_ptr()),
              "igs_rast_mark_visible");
    }
    return present;
}

// igs_adam_step_multi over lists of tensors (igs_amd/optim.py): one launch for up to 8 parameters
void adam_step_multi(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                     const std::vector<Tensor>& exp_avg_sqs, const std::vector<double>& lrs, const std::vector<double>& bc1,
                     const std::vector<double>& bc2_sqrt, double beta1, double beta2, double eps, const std::vector<Tensor>& steps,
                     const OptTensor& done_scratch)
{
    // `steps` (optional): one float32 GPU scalar per tensor = the step count, advanced on the device (igs_adam_step_multi_dev; bc1 /
    // bc2_sqrt are then ignored) -- the form a hipGraph can replay; `done_scratch`: int32 GPU tensor of
    // igs_adam_step_multi_dev_scratch_words() zeros the caller keeps between calls
    const size_t n = params.size();
    if (n == 0) return;
    const bool dev_step = !steps.empty();
    if (n > 8 || grads.size() != n || exp_avgs.size() != n || exp_avg_sqs.size() != n || lrs.size() != n
        || (dev_step ? steps.size() != n : (bc1.size() != n || bc2_sqrt.size() != n)))
        throw RasterizerError("adam_step_multi: between 1 and 8 tensors, all lists of the same length");
    float* p[8]; const float* g[8]; float* m[8]; float* v[8]; size_t cnt[8]; float lr[8], b1c[8], b2c[8]; float* st[8];
    std::vector<Tensor> keep;
    const c10::Device dev = params[0].device();
    for (size_t k = 0; k < n; k++) {
        const Tensor& P_ = params[k];
        if (!P_.is_cuda() || P_.device() != dev || P_.scalar_type() != at::kFloat || !P_.is_contiguous() || !exp_avgs[k].is_contiguous()
            || !exp_avg_sqs[k].is_contiguous() || grads[k].numel() != P_.numel() || exp_avgs[k].numel() != P_.numel() || exp_avg_sqs[k].numel() != P_.numel())
            throw RasterizerError("adam_step_multi: parameters and state must be contiguous float32 tensors on one GPU (no CPU fallback)");
        Tensor G = (grads[k].is_contiguous() && grads[k].scalar_type() == at::kFloat) ? grads[k] : grads[k].to(at::kFloat).contiguous();
        keep.push_back(G);
        p[k] = P_.data_ptr<float>(); g[k] = G.data_ptr<float>(); m[k] = exp_avgs[k].data_ptr<float>(); v[k] = exp_avg_sqs[k].data_ptr<float>();
        cnt[k] = (size_t)P_.numel(); lr[k] = (float)lrs[k];
        if (dev_step) {
            const Tensor& S_ = steps[k];
            if (!S_.is_cuda() || S_.device() != dev || S_.scalar_type() != at::kFloat || S_.numel() != 1)
                throw RasterizerError("adam_step_multi: every step count must be a one-element float32 tensor on the parameters' GPU");
            st[k] = S_.data_ptr<float>();
        } else {
            b1c[k] = (float)bc1[k]; b2c[k] = (float)bc2_sqrt[k];
        }
    }
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    unsigned* done = nullptr;
    if (dev_step) {
        if (!done_scratch.has_value() || !done_scratch->is_cuda() || done_scratch->device() != dev || done_scratch->scalar_type() != at::kInt
            || (size_t)done_scratch->numel() < igs_adam_step_multi_dev_scratch_words() || !done_scratch->is_contiguous())
            throw RasterizerError("adam_step_multi: device-side step counts need `done_scratch`, an int32 tensor of adam_dev_scratch_words() zeros on the parameters' GPU");
        done = (unsigned*)done_scratch->data_ptr<int>();
    }
    const int rc = dev_step ? igs_adam_step_multi_dev(stream, (int)n, p, g, m, v, cnt, lr, st, done, (float)beta1, (float)beta2, (float)eps)
                            : igs_adam_step_multi(stream, (int)n, p, g, m, v, cnt, lr, b1c, b2c, (float)beta1, (float)beta2, (float)eps);
    if (rc != 0) throw RasterizerError("igs_adam_step_multi failed: " + std::to_string(rc));
}

// ---- the two image losses of the refine loop as single calls (igs_amd/losses.py wraps them in autograd Functions) ----
// small per-(device, stream) scratch kept for the life of the process
struct LossScratch { Tensor l1; Tensor ssim; int64_t ssim_w = 0, ssim_h = 0; };
bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
LossScratch& loss_scratch(const c10::Device& dev, hipStream_t stream)
{
    static std::mutex mu;
    static std::map<std::pair<int, void*>, LossScratch> table;
    std::lock_guard<std::mutex> lock(mu);
    return table[{ (int)dev.index(), (void*)stream }];
}

// mean |a - b| and sign(a - b) / n in one launch (igs_l1_mean_fwd_bwd; loss_utils.py:17-18)
std::tuple<Tensor, Tensor> l1_mean(const Tensor& a, const Tensor& b)
{
    if (!a.is_cuda() || !b.is_cuda() || a.numel() == 0 || a.numel() != b.numel()) throw RasterizerError("l1_mean: two GPU tensors of the same, non-zero size (no CPU fallback)");
    const c10::Device dev = a.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    In x(a, dev, "a"), y(b, dev, "b");
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    // 1024 partial sums | the self-resetting counter words: kept per (device, stream) -- except on a capturing stream, where the scratch
    // must be memory the graph owns (allocated here, from the capture's private pool, zeroed by a node of the graph)
    Tensor l1s;
    if (stream_is_capturing(stream)) l1s = at::zeros({1024 + 33 * 64}, fopt);
    else {
        LossScratch& sc = loss_scratch(dev, stream);
        if (!sc.l1.defined()) sc.l1 = at::zeros({1024 + 33 * 64}, fopt);
        l1s = sc.l1;
    }
    Tensor grad = at::empty_like(x.keep), out = at::empty({}, fopt);
    const int rc = igs_l1_mean_fwd_bwd(stream, (size_t)x.keep.numel(), x.p, y.p, grad.data_ptr<float>(), out.data_ptr<float>(),
                                       l1s.data_ptr<float>(), (unsigned*)(l1s.data_ptr<float>() + 1024));
    if (rc != 0) throw RasterizerError("igs_l1_mean_fwd_bwd failed: " + std::to_string(rc));
    return { out, grad };
}

// mean SSIM(a, b) over all elements (finished on the device) and d(mean SSIM)/da in two launches (igs_ssim_mean_fwd_bwd; loss_utils.py:34-63 with the 11x11 window).  a, b: [3, H, W] (or anything that reshapes to it)
std::tuple<Tensor, Tensor> ssim_mean(const Tensor& a, const Tensor& b)
{
    if (!a.is_cuda() || !b.is_cuda() || a.dim() < 3 || a.numel() != b.numel()) throw RasterizerError("ssim_mean: two GPU images of the same size (no CPU fallback)");
    const c10::Device dev = a.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    In x(a, dev, "a"), y(b, dev, "b");
    const int64_t H = a.size(-2), W = a.size(-1);
    if (a.numel() != 3 * H * W) throw RasterizerError("ssim_mean: one 3-channel image per side");
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    Tensor scratch;
    const auto bopt = at::TensorOptions().dtype(at::kByte).device(dev);
    if (stream_is_capturing(stream)) scratch = at::empty({(int64_t)igs_ssim_l1_scratch_bytes((int)W, (int)H)}, bopt);      // (graph-owned, as in l1_mean)
    else {
        LossScratch& sc = loss_scratch(dev, stream);
        if (!sc.ssim.defined() || sc.ssim_w != W || sc.ssim_h != H) {
            sc.ssim = at::empty({(int64_t)igs_ssim_l1_scratch_bytes((int)W, (int)H)}, bopt);
            sc.ssim_w = W; sc.ssim_h = H;
        }
        scratch = sc.ssim;
    }
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor grad = at::empty_like(x.keep), mean = at::empty({}, fopt);
    const int rc = igs_ssim_mean_fwd_bwd(stream, (int)W, (int)H, x.p, y.p, scratch.data_ptr(), grad.data_ptr<float>(), mean.data_ptr<float>());
    if (rc != 0) throw RasterizerError("igs_ssim_mean_fwd_bwd failed: " + std::to_string(rc));
    return { mean, grad };               // grad = d(mean SSIM)/da
}

}      // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    namespace py = pybind11;
    py::register_exception<RasterizerError>(m, "RasterizerError", PyExc_RuntimeError);
    py::class_<ScratchSet, std::shared_ptr<ScratchSet>>(m, "ScratchSet")
        .def(py::init([](const py::object& device, bool persistent) {
                 return std::make_shared<ScratchSet>(torch::python::detail::py_object_to_device(device), persistent);
             }), py::arg("device"), py::arg("persistent") = true)
        .def_readonly("geom", &ScratchSet::geom).def_readonly("binning", &ScratchSet::binning).def_readonly("img", &ScratchSet::img)
        .def_readonly("persistent", &ScratchSet::persistent)
        .def("workspace", [](ScratchSet& s, int64_t P) { return s.ensure_workspace(P); }, py::arg("P"))
        // (callback address, user word) x 3 for callers that fill a C struct themselves (igs_refine_step_args through ctypes);
        // valid for as long as this object lives
        .def("callbacks", [](ScratchSet& s) {
            return std::make_tuple((uintptr_t)&grow_cb, (uintptr_t)&s.g_geom, (uintptr_t)&s.g_binning, (uintptr_t)&s.g_img);
        });

    const auto none = py::none();
    m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"), py::arg("opacity"),
          py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"),
          py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("kernel_size"), py::arg("image_height"),
          py::arg("image_width"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("prefiltered"), py::arg("require_coord"),
          py::arg("require_depth"), py::arg("debug"), py::kw_only(), py::arg("scratch") = std::shared_ptr<ScratchSet>(),
          py::arg("out_images") = none, py::arg("out_radii") = none, py::arg("mode") = 0, py::arg("scratch_clean") = false,
          py::call_guard<py::gil_scoped_release>());

#define BWD_ARGS \
    py::arg("background"), py::arg("means3D"), py::arg("radii"), py::arg("colors"), py::arg("scales"), py::arg("rotations"), \
    py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), \
    py::arg("kernel_size"), py::arg("dL_dout_color"), py::arg("dL_dout_coord"), py::arg("dL_dout_mcoord"), py::arg("dL_dout_depth"), \
    py::arg("dL_dout_mdepth"), py::arg("dL_dout_alpha"), py::arg("dL_dout_normal"), py::arg("normalmap"), py::arg("sh"), py::arg("degree"), \
    py::arg("campos"), py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("alphas"), \
    py::arg("require_coord"), py::arg("require_depth"), py::arg("debug"), py::kw_only(), py::arg("workspace") = none, \
    py::arg("out_means2D") = none, py::arg("out_colors") = none, py::arg("out_opacity") = none, py::arg("out_means3D") = none, \
    py::arg("out_cov3D") = none, py::arg("out_sh") = none, py::arg("out_scales") = none, py::arg("out_rotations") = none

    m.def("rasterize_gaussians_backward",
          [](const Tensor& a0, const Tensor& a1, const Tensor& a2, const Tensor& a3, const Tensor& a4, const Tensor& a5, double a6, const Tensor& a7,
             const Tensor& a8, const Tensor& a9, double a10, double a11, double a12, const OptTensor& a13, const OptTensor& a14, const OptTensor& a15,
             const OptTensor& a16, const OptTensor& a17, const OptTensor& a18, const OptTensor& a19, const Tensor& a20, const Tensor& a21, int64_t a22,
             const Tensor& a23, const Tensor& a24, int64_t a25, const Tensor& a26, const Tensor& a27, const Tensor& a28, bool a29, bool a30, bool a31,
             const OptTensor& ws, const OptTensor& o0, const OptTensor& o1, const OptTensor& o2, const OptTensor& o3, const OptTensor& o4,
             const OptTensor& o5, const OptTensor& o6, const OptTensor& o7) {
              return backward_body(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23,
                                   a24, a25, a26, a27, a28, a29, a30, a31, ws, o0, o1, o2, o3, o4, o5, o6, o7, 0, 0.0).grads;
          }, BWD_ARGS, py::call_guard<py::gil_scoped_release>());
    m.def("rasterize_gaussians_backward_ex",
          [](const Tensor& a0, const Tensor& a1, const Tensor& a2, const Tensor& a3, const Tensor& a4, const Tensor& a5, double a6, const Tensor& a7,
             const Tensor& a8, const Tensor& a9, double a10, double a11, double a12, const OptTensor& a13, const OptTensor& a14, const OptTensor& a15,
             const OptTensor& a16, const OptTensor& a17, const OptTensor& a18, const OptTensor& a19, const Tensor& a20, const Tensor& a21, int64_t a22,
             const Tensor& a23, const Tensor& a24, int64_t a25, const Tensor& a26, const Tensor& a27, const Tensor& a28, bool a29, bool a30, bool a31,
             const OptTensor& ws, const OptTensor& o0, const OptTensor& o1, const OptTensor& o2, const OptTensor& o3, const OptTensor& o4,
             const OptTensor& o5, const OptTensor& o6, const OptTensor& o7, int64_t nan_report, double clamp) {
              // nan_report: 0 none; 1 wait for the kernel's verdict here -> (grads, 0 / 1, 0, 0); 2 deferred -> (grads, 0, word, seq) for nan_report_wait
              BwdResult r = backward_body(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23,
                                          a24, a25, a26, a27, a28, a29, a30, a31, ws, o0, o1, o2, o3, o4, o5, o6, o7, nan_report, clamp);
              return std::make_tuple(r.grads, r.nan, r.nan_word, r.nan_seq);
          }, BWD_ARGS, py::arg("nan_report") = 0, py::arg("clamp") = 0.0, py::call_guard<py::gil_scoped_release>());
    m.def("nan_report_wait", [](int64_t word, int64_t seq) {
        const int v = igs_rast_nan_report_wait_at((const void*)(uintptr_t)word, (unsigned)seq);
        check(v, "igs_rast_nan_report_wait_at");
        return v != 0;
    }, py::arg("word"), py::arg("seq"), py::call_guard<py::gil_scoped_release>());
    m.def("mark_visible", &mark_visible, py::arg("means3D"), py::arg("viewmatrix"), py::arg("projmatrix"), py::call_guard<py::gil_scoped_release>());
    m.def("integrate_gaussians_to_points", [](const py::args&, const py::kwargs&) -> py::object {
        // GOF tetrahedra integration (DGR/rasterize_points.cu:269-387): mesh extraction only, never reached from IGS
        PyErr_SetString(PyExc_NotImplementedError, "integrate_gaussians_to_points is outside the IGS hot path and is not implemented");
        throw py::error_already_set();
    });
    m.def("forward_finish", []() -> py::object {
        const int rc = igs_rast_forward_finish();
        if (rc == IGS_RAST_E_RETRY) return py::none();
        check(rc, "igs_rast_forward_finish");
        return py::int_(rc);
    });
    m.def("adam_step_multi", &adam_step_multi, py::arg("params"), py::arg("grads"), py::arg("exp_avgs"), py::arg("exp_avg_sqs"), py::arg("lrs"),
          py::arg("bias_correction1"), py::arg("bias_correction2_sqrt"), py::arg("beta1"), py::arg("beta2"), py::arg("eps"),
          py::arg("steps") = std::vector<Tensor>(), py::arg("done_scratch") = py::none(), py::call_guard<py::gil_scoped_release>());
    m.def("adam_dev_scratch_words", []() { return (int64_t)igs_adam_step_multi_dev_scratch_words(); });
    m.def("l1_mean", &l1_mean, py::arg("a"), py::arg("b"), py::call_guard<py::gil_scoped_release>());
    m.def("ssim_mean", &ssim_mean, py::arg("a"), py::arg("b"), py::call_guard<py::gil_scoped_release>());
    m.def("abi_version", []() { return igs_rast_version(); });
}
