"""The four functions the reference's pybind module exports (DGR/ext.cpp:15-20), from the compiled module igs_amd/_C.*.so
(igs_amd/csrc_torch/igs_torch_ext.cpp: torch glue over the C ABI of libigs_rast.so)."""
from igs_amd._cabi import ext as _ext

_m = _ext()
rasterize_gaussians = _m.rasterize_gaussians
rasterize_gaussians_backward = _m.rasterize_gaussians_backward
mark_visible = _m.mark_visible
integrate_gaussians_to_points = _m.integrate_gaussians_to_points
