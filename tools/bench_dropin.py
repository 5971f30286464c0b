"""What an unchanged IGS caller sees: the step driven through the reference's Python API (GaussianRasterizer autograd Function,
torch activations, torch loss, loss.backward()), against the library's fused single-call step.  Same scene as bench.py."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

def run(ref, steps=100, warm=20):
    for _ in range(warm):
        ref.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / steps

def graph_row(raw, cams, gts, bg, dev, loss, steps=100, warm=20):
    """The same autograd step as a user of the reference API would capture it: torch activations, GaussianRasterizer, torch loss,
    loss.backward(), torch.optim.Adam(capturable=True) -- recorded once with torch.cuda.graph and replayed per step; the camera and
    the ground-truth image are static tensors overwritten before each replay.  (The reference's rasterizer cannot be captured: it
    reads its instance count back in the middle of the forward.)"""
    import copy
    from igs_amd.refine import _gaussian_window
    from igs_amd.rasterizer import capture_status
    import torch.nn.functional as F
    window = _gaussian_window(11, 1.5, 3, dev)          # (the reference's ssim() builds it on the host per call: not capturable)

    def ssim(a, b):                                      # igs/utils/loss_utils.py:34-63
        x, y = a.unsqueeze(0), b.unsqueeze(0)
        mu1, mu2 = F.conv2d(x, window, padding=5, groups=3), F.conv2d(y, window, padding=5, groups=3)
        s1 = F.conv2d(x * x, window, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(y * y, window, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(x * y, window, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1 * mu1 + mu2 * mu2 + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))).mean()
    p = GaussianParams(raw, dev); p.spatial_sort()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.leaves.items()}
    opt = torch.optim.Adam([dict(params=[leaves[n]], lr=p.lrs[n]) for n in leaves], lr=0.0, eps=1e-15, capturable=True)
    cam_s = copy.copy(cams[0])
    cam_s.world_view_transform, cam_s.full_proj_transform = cams[0].world_view_transform.clone(), cams[0].full_proj_transform.clone()
    cam_s.camera_center = cams[0].camera_center.clone()
    gt_s = gts[0].clone()

    def one():
        opt.zero_grad(set_to_none=True)
        act = dict(means3D=leaves["xyz"], shs=leaves["shs"], opacities=torch.sigmoid(leaves["opacity"]),
                   scales=torch.exp(leaves["scaling"]), rotations=torch.nn.functional.normalize(leaves["rotation"]))
        img = render(act, cam_s, bg)["images_pred"]
        l = torch.abs(img - gt_s).mean()
        if loss == "l1_ssim":
            l = 0.8 * l + 0.2 * (1.0 - ssim(img, gt_s))
        l.backward()
        opt.step()
        return l.detach()

    def load(i):
        c = cams[i % len(cams)]
        cam_s.world_view_transform.copy_(c.world_view_transform); cam_s.full_proj_transform.copy_(c.full_proj_transform)
        cam_s.camera_center.copy_(c.camera_center); gt_s.copy_(gts[i % len(cams)])

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for i in range(warm):
            load(i); one()
        side.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):                          # the same loop, eager: what the graph is compared with
            load(i); one()
        side.synchronize()
        eager_ms = 1000 * (time.perf_counter() - t0) / steps
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        l_s = one()
    first = None
    for i in range(warm):
        load(i); g.replay()
        if first is None:
            first = float(l_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        load(i); g.replay()
    torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / steps
    n, overflow = capture_status()
    assert overflow == 0 and float(l_s) < first, (overflow, float(l_s), first)      # (the replays really optimise)
    return eager_ms, ms


def main():
    dev = torch.device("cuda:0")
    raw, cams, bg = sear_steak_like_scene()
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    out = {}
    for name, kw, nan in (("python_api_autograd_nan_checks_on", dict(native=False), True),
                          ("python_api_autograd", dict(native=False), False),
                          ("python_api_autograd_fused_activations", dict(native=False), False),
                          ("c_abi_unfused", dict(native=True, fused=False), False),
                          ("igs_refine_step", dict(native=True, fused=True), False)):
        rasterizer.NAN_CHECKS = nan
        for loss in ("l1", "l1_ssim"):
            p = GaussianParams(raw, dev); p.spatial_sort()
            r = Refiner(p, cams, gts, bg, loss=loss, **kw)
            r.fused_activations = name.endswith("fused_activations")
            r.direct_adam = True         # autograd path: gradients straight from autograd into the fused Adam (set_to_none semantics, infer_batch.py:324)
            out["%s/%s" % (name, loss)] = round(run(r), 4)
    # (L1 only: PyTorch's conv2d SSIM alone is 8 ms per step, eager or captured)
    e, g = graph_row(raw, cams, gts, bg, dev, "l1")
    out["plain_torch_loop_torch_adam_eager/l1"], out["plain_torch_loop_torch_adam_in_a_cuda_graph/l1"] = round(e, 4), round(g, 4)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
