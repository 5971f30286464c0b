"""Camera matrices in the conventions the rasterizer's callers use.

Mirrors the reference's `Camera` / `getProjectionMatrix`
(igs/models/gs.py:61-113; same formula in igs/utils/graphics_utils.py:51-71):
matrices are handed to the rasterizer TRANSPOSED (row-vector convention),
`world_view_transform = w2c.T`, `full_proj_transform = view_T @ proj_T`,
`camera_center = inverse(view_T)[3, :3]`, znear 0.01, zfar 100.
"""
import math

import torch


def get_projection_matrix(znear, zfar, fovX, fovY):
    """igs/models/gs.py:61-81."""
    tan_half_y = math.tan(fovY / 2)
    tan_half_x = math.tan(fovX / 2)
    top = tan_half_y * znear
    bottom = -top
    right = tan_half_x * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


class Camera:
    """igs/models/gs.py:90-113.  `resolution` = (height, width)."""

    def __init__(self, w2c, FoVx, FoVy, resolution):
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.height = int(resolution[0])
        self.width = int(resolution[1])
        self.world_view_transform = w2c.transpose(0, 1).contiguous()
        self.zfar = 100.0
        self.znear = 0.01
        self.projection_matrix = get_projection_matrix(self.znear, self.zfar, self.FoVx, self.FoVy) \
            .transpose(0, 1).to(device=w2c.device, dtype=w2c.dtype)
        self.full_proj_transform = self.world_view_transform.unsqueeze(0).bmm(
            self.projection_matrix.unsqueeze(0)).squeeze(0).contiguous()
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()

    @staticmethod
    def from_c2w(c2w, FOV, resolution):
        w2c = torch.inverse(c2w)
        return Camera(w2c=w2c, FoVx=FOV[0], FoVy=FOV[1], resolution=resolution)

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        self.world_view_transform = self.world_view_transform.to(device)
        self.projection_matrix = self.projection_matrix.to(device)
        self.full_proj_transform = self.full_proj_transform.to(device)
        self.camera_center = self.camera_center.to(device)
        return self


def look_at_c2w(eye, target, up=(0.0, -1.0, 0.0)):
    """OpenCV/COLMAP-style camera-to-world (x right, y down, z forward)."""
    eye = torch.as_tensor(eye, dtype=torch.float32)
    target = torch.as_tensor(target, dtype=torch.float32)
    upv = torch.as_tensor(up, dtype=torch.float32)
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(-upv, z)   # y points down: x = (-up) x z  ->  right-handed (x, y, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
    return c2w
