"""Pins the CPU oracle and the host-side camera maths against the only reference code that is runnable:
igs/utils/sh_utils.py and igs/utils/graphics_utils.py (fixtures made by tests/golden/make_golden.py)."""
import math

import numpy as np
import torch

from igs_amd import camera
from oracle import c_oracle as co


def test_sh_basis_matches_reference_eval_sh(golden):
    co.set_precision("float32")
    means, campos, sh = golden["sh_means"], golden["sh_campos"], golden["sh_coeffs"]
    for deg in range(4):
        want = golden["sh_rgb_deg%d" % deg]
        for i in range(means.shape[0]):
            rgb, clamped = co.sh_to_rgb(deg, sh[i], means[i], campos)
            np.testing.assert_allclose(rgb, np.maximum(want[i], 0.0), atol=2e-6)
            assert (clamped == (want[i] < 0)).all() or np.abs(want[i]).min() < 1e-6


def test_projection_matrix_matches_reference(golden):
    for (fx, fy), want in zip(golden["proj_fovs"], golden["proj_mats"]):
        got = camera.get_projection_matrix(0.01, 100.0, float(fx), float(fy)).numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=0)


def test_fov_focal_helpers(golden):
    np.testing.assert_allclose([camera.fov2focal(1.2, 1352), camera.fov2focal(0.9, 1014)], golden["fov2focal"], rtol=1e-12)
    np.testing.assert_allclose([camera.focal2fov(730.0, 1352), camera.focal2fov(730.0, 1014)], golden["focal2fov"], rtol=1e-12)


def test_camera_view_matrix_convention(golden):
    """Camera.world_view_transform = w2c.T, same matrix getWorld2View2(R, t) builds (R = c2w rotation, t = w2c translation)."""
    R, t = golden["w2v_R"], golden["w2v_t"]
    w2c = torch.tensor(golden["w2v_mat"], dtype=torch.float32)
    c2w = torch.inverse(w2c)
    cam = camera.Camera.from_c2w(c2w, (1.0, 0.8), (64, 96))
    np.testing.assert_allclose(cam.world_view_transform.numpy(), golden["w2v_mat"].T, atol=1e-5)
    np.testing.assert_allclose(cam.world_view_transform.numpy()[:3, :3], R, atol=1e-5)   # rows of w2c.T[:3,:3] = R
    np.testing.assert_allclose(cam.camera_center.numpy(), c2w[:3, 3].numpy(), atol=1e-5)
    assert cam.height == 64 and cam.width == 96
    # full projection = view_T @ proj_T
    P = camera.get_projection_matrix(0.01, 100.0, 1.0, 0.8).t()
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), (cam.world_view_transform @ P).numpy(), atol=1e-6)
