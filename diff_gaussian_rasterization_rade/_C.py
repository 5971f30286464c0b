"""The four functions the reference's pybind module exports (DGR/ext.cpp:15-20), served by libigs_rast.so."""
from igs_amd.rasterizer import (  # noqa: F401
    rasterize_gaussians,
    rasterize_gaussians_backward,
    mark_visible,
    integrate_gaussians_to_points,
)
