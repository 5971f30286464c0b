"""TEST INFRASTRUCTURE -- PyTorch (autograd-differentiable) restatements of the caller-side losses the HIP loss kernels are checked
against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; no product module does.

Pinned by reference-produced data: `tests/golden/ref_torch_only.npz` holds values and autograd gradients of the reference's own
`submodules/RaDe-GS/utils/loss_utils.py` (= `igs/utils/loss_utils.py:17-63`) for two image pairs
(`tests/test_oracle_golden.py::test_loss_restatements_match_reference_loss_utils`).  The depth-normal regulariser
(`submodules/RaDe-GS/utils/graphics_utils.py:97-126`, `train.py:143-160`) is NOT pinned that way (the file needs cv2, absent here):
it is pinned by the analytic-plane case of tests/test_host_logic.py only -- parity unpinned.
"""
import math

import torch
import torch.nn.functional as F


def l1_mean(a, b):
    """igs/utils/loss_utils.py:17-18."""
    return (a - b).abs().mean()


def _window(size, sigma, channels, like):
    """loss_utils.py:21-31: normalised 1-D Gaussian, outer product, one copy per channel (grouped convolution weights)."""
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    g = g / g.sum()
    w = torch.outer(g, g)
    return w.expand(channels, 1, size, size).contiguous().to(device=like.device, dtype=like.dtype)


def ssim_map(img1, img2, window_size=11):
    """loss_utils.py:41-58: the SSIM index per element, zero-padded 11x11 window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2.
    Inputs [C,H,W] or [B,C,H,W] (broadcast against each other like the reference's conv2d calls)."""
    a = img1 if img1.dim() == 4 else img1.unsqueeze(0)
    b = img2 if img2.dim() == 4 else img2.unsqueeze(0)
    C = a.size(1)
    w = _window(window_size, 1.5, C, a)
    blur = lambda t: F.conv2d(t, w, padding=window_size // 2, groups=C)
    ma, mb = blur(a), blur(b)
    va = blur(a * a) - ma * ma
    vb = blur(b * b) - mb * mb
    cab = blur(a * b) - ma * mb
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))


def ssim_reference_call(img1, img2, window_size=11, size_average=True):
    """The reference's return convention (loss_utils.py:60-63): `(mean, map)` with size_average, per-batch means without."""
    m = ssim_map(img1, img2, window_size)
    if size_average:
        return m.mean(), (m if img1.dim() == 4 else m.squeeze(0))
    return m.mean(1).mean(1).mean(1)


def ssim_mean(img, gt):
    """Scalar mean SSIM of one image pair."""
    return ssim_map(img, gt).mean()


def psnr(img, gt):
    """infer_batch.py:350-353."""
    return -10.0 * torch.log10(torch.mean((torch.clamp(img, 0, 1) - gt) ** 2))


# ---- RaDe-GS depth-normal consistency (graphics_utils.py:97-126, train.py:143-160) ----------------------------------------
def backproject(cam, depth):
    """graphics_utils.py:97-112 for one depth map [1,H,W]: points = depth * K^-1 (x + 0.5, y + 0.5, 1)."""
    W, H = cam.width, cam.height
    fx = W / (2 * math.tan(cam.FoVx / 2.0))
    fy = H / (2 * math.tan(cam.FoVy / 2.0))
    dev = depth.device
    ys = (torch.arange(H, device=dev, dtype=torch.float32) + 0.5 - H / 2.0) / fy
    xs = (torch.arange(W, device=dev, dtype=torch.float32) + 0.5 - W / 2.0) / fx
    rays = torch.stack([xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W), torch.ones(H, W, device=dev)], dim=0)
    return rays * depth.reshape(1, H, W)


def normal_from_points(points):
    """graphics_utils.py:116-123 for one point map [3,H,W]: normalised cross product of the central differences along y and x
    (in the reference's index order), zero on the one-pixel border."""
    out = torch.zeros_like(points)
    d_rows = points[:, 2:, 1:-1] - points[:, :-2, 1:-1]
    d_cols = points[:, 1:-1, 2:] - points[:, 1:-1, :-2]
    out[:, 1:-1, 1:-1] = F.normalize(torch.linalg.cross(d_rows, d_cols, dim=0), dim=0)
    return out


def depth_pair_to_normals(cam, depth1, depth2):
    """[2,3,H,W]: normals implied by the expected and the median depth map."""
    return torch.stack([normal_from_points(backproject(cam, depth1)), normal_from_points(backproject(cam, depth2))], dim=0)


def depth_normal_loss(pkg, cam, require_depth=True, depth_ratio=0.6):
    """train.py:143-160.  `pkg`: rasterizer outputs by name (depth_pred, mdepth | coord, mcoord; normal)."""
    n = pkg["normal"]
    if require_depth:
        implied = depth_pair_to_normals(cam, pkg["depth_pred"], pkg["mdepth"])
    else:
        implied = torch.stack([normal_from_points(pkg["coord"]), normal_from_points(pkg["mcoord"])], dim=0)
    err = 1 - (n.unsqueeze(0) * implied).sum(dim=1)
    return (1 - depth_ratio) * err[0].mean() + depth_ratio * err[1].mean()
