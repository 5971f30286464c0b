"""Drop-in for the reference package `diff_gaussian_rasterization_rade`
(submodules/RaDe-GS/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization_rade/__init__.py),
backed by the MI355X-native HIP library.  `infer_batch.py:23` imports exactly
`GaussianRasterizationSettings` and `GaussianRasterizer` from here."""
from igs_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    cpu_deep_copy_tuple,
)
from igs_amd.rasterizer import rasterize_gaussians_autograd as rasterize_gaussians  # noqa: F401
from . import _C  # noqa: F401
