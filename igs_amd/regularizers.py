"""Depth-normal consistency regulariser of RaDe-GS, PyTorch ops on the rasterizer's outputs.

`submodules/RaDe-GS/utils/graphics_utils.py:97-126` (depths_double_to_points, point_double_to_normal, depth_double_to_normal)
and `submodules/RaDe-GS/train.py:143-164` (the loss): the normal implied by the rendered expected / median depth (or coord)
maps is compared with the rendered normal map; its gradient reaches the Gaussians through dL/d depth, mdepth (or coord, mcoord)
and normal, i.e. the full backward instance of the blend kernel.
"""
import math

import torch
import torch.nn.functional as F


def depths_double_to_points(cam, depthmap1, depthmap2):
    """graphics_utils.py:97-112: back-projects two depth maps along the rays of the pixel CENTRES (+0.5)."""
    W, H = cam.width, cam.height
    fx = W / (2 * math.tan(cam.FoVx / 2.))
    fy = H / (2 * math.tan(cam.FoVy / 2.))
    dev = depthmap1.device
    intrins_inv = torch.tensor([[1 / fx, 0., -W / (2 * fx)], [0., 1 / fy, -H / (2 * fy)], [0., 0., 1.0]]).float().to(dev)
    grid_x, grid_y = torch.meshgrid(torch.arange(W) + 0.5, torch.arange(H) + 0.5, indexing='xy')
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=0).reshape(3, -1).float().to(dev)
    rays_d = intrins_inv @ points
    points1 = depthmap1.reshape(1, -1) * rays_d
    points2 = depthmap2.reshape(1, -1) * rays_d
    return points1.reshape(3, H, W), points2.reshape(3, H, W)


def point_double_to_normal(points1, points2):
    """graphics_utils.py:116-123: central differences of the point maps, cross product, normalised; zero on the border."""
    points = torch.stack([points1, points2], dim=0)
    output = torch.zeros_like(points)
    dx = points[..., 2:, 1:-1] - points[..., :-2, 1:-1]
    dy = points[..., 1:-1, 2:] - points[..., 1:-1, :-2]
    normal_map = F.normalize(torch.cross(dx, dy, dim=1), dim=1)
    output[..., 1:-1, 1:-1] = normal_map
    return output


def depth_double_to_normal(cam, depth1, depth2):
    return point_double_to_normal(*depths_double_to_points(cam, depth1, depth2))


def depth_normal_loss(pkg, cam, require_depth=True, depth_ratio=0.6):
    """train.py:143-160.  `pkg`: dict of the rasterizer outputs (depth_pred, mdepth, coord, mcoord, normal)."""
    rendered_normal = pkg["normal"]
    if require_depth:
        dn = depth_double_to_normal(cam, pkg["depth_pred"], pkg["mdepth"])
    else:
        dn = point_double_to_normal(pkg["coord"], pkg["mcoord"])
    err = 1 - (rendered_normal.unsqueeze(0) * dn).sum(dim=1)
    return (1 - depth_ratio) * err[0].mean() + depth_ratio * err[1].mean()
