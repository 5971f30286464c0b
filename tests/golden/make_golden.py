"""Generates tests/golden/ref_helpers.npz from the reference's own importable pure-python helpers.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference has no tests/fixtures for the rasterizer path; the only reference code that
can be executed here are igs/utils/sh_utils.py (eval_sh: the SH basis and sign conventions that
computeColorFromSH, forward.cu:23-74, must reproduce) and igs/utils/graphics_utils.py
(getProjectionMatrix, getWorld2View2, fov2focal, focal2fov: the matrix conventions of the callers).
They are loaded by file path (the `igs` package itself cannot be imported: jaxtyping/omegaconf absent).
Only inputs and outputs (data) are stored, never reference source.
"""
import importlib.util
import math
import os

import numpy as np
import torch

REF = "/root/reference/igs/utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    sh_utils = load("sh_utils")
    gu = load("graphics_utils")
    g = torch.Generator().manual_seed(1234)
    N = 64
    means = torch.randn(N, 3, generator=g) * 2.0
    campos = torch.tensor([0.3, -0.2, -4.0])
    dirs = means - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.randn(N, 16, 3, generator=g) * 0.5
    out = {"sh_means": means.numpy(), "sh_campos": campos.numpy(), "sh_coeffs": sh.numpy()}
    for deg in range(4):
        # eval_sh wants [..., C, (deg+1)^2]
        rgb = sh_utils.eval_sh(deg, sh[:, :(deg + 1) ** 2, :].transpose(1, 2), dirs) + 0.5
        out["sh_rgb_deg%d" % deg] = rgb.numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(torch.tensor([0.0, 0.25, 1.0])).numpy()
    fovs = [(math.radians(50.0), math.radians(50.0)), (1.4944, 1.2138), (0.6, 0.9)]
    out["proj_fovs"] = np.array(fovs, dtype=np.float64)
    out["proj_mats"] = np.stack([gu.getProjectionMatrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
    R = np.array([[0.9, -0.1, 0.42], [0.2, 0.95, -0.2], [-0.38, 0.27, 0.88]], dtype=np.float64)
    q, _ = np.linalg.qr(R)
    t = np.array([0.5, -1.0, 3.0])
    out["w2v_R"], out["w2v_t"] = q, t
    out["w2v_mat"] = gu.getWorld2View2(q, t)
    out["w2v_mat_ts"] = gu.getWorld2View2(q, t, np.array([0.1, 0.2, -0.3]), 2.0)
    out["fov2focal"] = np.array([gu.fov2focal(1.2, 1352), gu.fov2focal(0.9, 1014)])
    out["focal2fov"] = np.array([gu.focal2fov(730.0, 1352), gu.focal2fov(730.0, 1014)])
    np.savez(os.path.join(HERE, "ref_helpers.npz"), **out)
    print("wrote ref_helpers.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
