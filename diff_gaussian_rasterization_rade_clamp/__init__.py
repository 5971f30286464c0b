"""Drop-in for `diff_gaussian_rasterization_rade_clamp` (the clamp variant used by igs/models/gs.py:39): identical native
library, gradients of means3D / sh / opacities / scales / rotations clamped to [-15, 15]
(submodules/RaDe-GS/submodules/diff-gaussian-rasterization-clamp/diff_gaussian_rasterization_rade_clamp/__init__.py:156-162)."""
from igs_amd.rasterizer import GaussianRasterizationSettings, cpu_deep_copy_tuple  # noqa: F401
from igs_amd.rasterizer import GaussianRasterizerClamp as GaussianRasterizer  # noqa: F401
from igs_amd.rasterizer import _RasterizeGaussiansClamp as _RasterizeGaussians  # noqa: F401
from igs_amd.rasterizer import rasterize_gaussians_autograd_clamp as rasterize_gaussians  # noqa: F401
from diff_gaussian_rasterization_rade import _C  # noqa: F401
