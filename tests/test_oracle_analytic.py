"""Analytic micro-cases for the CPU oracle (the reference holds no golden vectors for this path,
so the oracle is pinned by cases whose answer is known in closed form)."""
import math

import numpy as np
import pytest
import torch

from igs_amd.camera import Camera
from oracle import c_oracle as co

SH_C0 = 0.28209479177387814


def _cam(size=64, fov_deg=60.0, z=-4.0):
    c2w = torch.eye(4)
    c2w[2, 3] = z
    f = math.radians(fov_deg)
    return Camera.from_c2w(c2w, (f, f), (size, size))


def _render(means, scales, quats, opac, sh, cam, bg=(0, 0, 0), deg=0, **kw):
    co.set_precision("float32")
    return co.rasterize_forward(np.array(bg, np.float32), means, None, opac, scales, quats, 1.0, None,
                                cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                cam.height, cam.width, sh, deg, cam.camera_center, **kw)


def test_single_isotropic_gaussian_centre_pixel():
    cam = _cam()
    W = cam.width
    # put the Gaussian so that it projects exactly on the pixel centre (32, 32): ndc2Pix(v) = ((v+1)*W-1)/2
    fx = W / (2 * cam.tanfovx)
    depth = 4.0
    x = (32 - (W - 1) / 2.0) / fx * depth       # pixel 32 <-> ndc = (2*32+1)/W - 1
    means = np.array([[x, x, 0.0]], np.float32)
    s = 0.05
    col = np.array([0.3, -0.1, 0.8], np.float32)
    sh = np.zeros((1, 1, 3), np.float32)
    sh[0, 0] = col
    o = 0.7
    nr, out, st = _render(means, np.full((1, 3), s, np.float32), np.array([[1, 0, 0, 0]], np.float32),
                          np.array([[o]], np.float32), sh, cam, bg=(0.1, 0.2, 0.3))
    it = st.intermediates()
    np.testing.assert_allclose(it["means2D"][0], [32.0, 32.0], atol=1e-4)
    det = float(it["conic_opacity"][0, 0] * it["conic_opacity"][0, 2] - it["conic_opacity"][0, 1] ** 2)
    cov_det = 1.0 / det
    coef = math.sqrt(cov_det / (cov_det + 1e-6) + 1e-6)
    alpha = min(0.99, o * coef)
    rgb = np.maximum(SH_C0 * col + 0.5, 0)
    np.testing.assert_allclose(out["alpha"][0, 32, 32], alpha, rtol=2e-5)
    np.testing.assert_allclose(out["color"][:, 32, 32], rgb * alpha + (1 - alpha) * np.array([0.1, 0.2, 0.3]), rtol=2e-5)
    # expected depth map = ray distance / ray length = view-space z for a fronto-parallel isotropic Gaussian
    ln = math.sqrt(((32 - W / 2) / fx) ** 2 * 2 + 1)
    tcen = math.sqrt(x * x * 2 + depth * depth)
    np.testing.assert_allclose(out["depth"][0, 32, 32], tcen / ln, rtol=1e-4)
    np.testing.assert_allclose(out["mdepth"][0, 32, 32], tcen / ln, rtol=1e-4)
    np.testing.assert_allclose(out["coord"][:, 32, 32], [x, x, depth], rtol=1e-3, atol=1e-4)
    # far away from the splat only the background is seen and geometry outputs stay zero
    np.testing.assert_allclose(out["color"][:, 2, 2], [0.1, 0.2, 0.3], atol=1e-7)
    assert out["alpha"][0, 2, 2] == 0 and out["depth"][0, 2, 2] == 0 and (out["normal"][:, 2, 2] == 0).all()
    assert nr == it["tiles_touched"][0] and out["radii"][0] > 0


def test_sphere_normal_faces_camera():
    """For an isotropic Gaussian the RaDe-GS plane is perpendicular to the ray: normal = -ray direction."""
    cam = _cam()
    means = np.array([[0.4, -0.3, 0.5]], np.float32)
    sh = np.zeros((1, 1, 3), np.float32)
    nr, out, st = _render(means, np.full((1, 3), 0.1, np.float32), np.array([[1, 0, 0, 0]], np.float32),
                          np.array([[0.9]], np.float32), sh, cam)
    it = st.intermediates()
    pv = it["view_points"][0]
    np.testing.assert_allclose(it["normals"][0], -pv / np.linalg.norm(pv), atol=2e-4)
    np.testing.assert_allclose(it["ts"][0], np.linalg.norm(pv), rtol=1e-6)


def test_two_gaussian_occlusion_order_and_median():
    cam = _cam()
    means = np.array([[0, 0, 1.0], [0, 0, 0.0]], np.float32)    # index 0 is FARTHER (z view = 5) than index 1 (z view = 4)
    sh = np.zeros((2, 1, 3), np.float32)
    sh[0, 0] = (1.0 - 0.5) / SH_C0 * np.array([1, 0, 0])   # red, far
    sh[1, 0] = (1.0 - 0.5) / SH_C0 * np.array([0, 1, 0])   # green, near
    sh[:, 0] += -0.5 / SH_C0 * (1 - np.array([[1, 0, 0], [0, 1, 0]]))
    o = np.array([[0.6], [0.6]], np.float32)
    nr, out, st = _render(means, np.full((2, 3), 0.2, np.float32), np.tile(np.array([[1, 0, 0, 0]], np.float32), (2, 1)), o, sh, cam)
    it = st.intermediates()
    # the near Gaussian (index 1) must come first in every tile it shares with the far one
    c = cam.width // 2
    tile = (c // 16) * ((cam.width + 15) // 16) + c // 16
    r0, r1 = it["ranges"][tile]
    assert list(it["point_list"][r0:r1]) == [1, 0]
    px = out["color"][:, c, c]
    a_near = it["conic_opacity"][1, 3] * math.exp(-0.5 * (it["conic_opacity"][1, 0] * 0.25 * 2 + 2 * it["conic_opacity"][1, 1] * 0.25))
    assert px[1] > px[0] > 0                  # green over red
    # median depth: T after the first splat is 1-0.6*G < 0.5 only if alpha > 0.5 -> median is the near one
    np.testing.assert_allclose(out["alpha"][0, c, c], 1 - (1 - px[1]) * (1 - px[0] / (1 - px[1])), rtol=1e-5)


def test_near_plane_cull_and_mark_visible():
    cam = _cam(z=-4.0)
    means = np.array([[0, 0, -3.9], [0, 0, -3.79], [0, 0, 0.0], [0, 0, -10.0]], np.float32)   # view z: 0.1, 0.21, 4, -6
    sh = np.zeros((4, 1, 3), np.float32)
    nr, out, st = _render(means, np.full((4, 3), 0.01, np.float32), np.tile(np.array([[1, 0, 0, 0]], np.float32), (4, 1)),
                          np.full((4, 1), 0.5, np.float32), sh, cam)
    assert (out["radii"] > 0).tolist() == [False, True, True, False]
    vis = co.mark_visible(means, cam.world_view_transform, cam.full_proj_transform)
    assert vis.tolist() == [False, True, True, False]
    with pytest.raises(RuntimeError):
        _render(means, np.full((4, 3), 0.01, np.float32), np.tile(np.array([[1, 0, 0, 0]], np.float32), (4, 1)),
                np.full((4, 1), 0.5, np.float32), sh, cam, prefiltered=True)


def test_empty_inputs():
    cam = _cam()
    nr, out, st = _render(np.zeros((0, 3), np.float32), None, None, None, None, cam, bg=(0.5, 0.5, 0.5))
    assert nr == 0 and (out["color"] == 0).all()      # P == 0 short-circuits: outputs stay zero (rasterize_points.cu:90)
    # all Gaussians culled: R == 0, the image is the background
    means = np.array([[0, 0, -30.0]], np.float32)
    nr, out, st = _render(means, np.full((1, 3), 0.01, np.float32), np.array([[1, 0, 0, 0]], np.float32),
                          np.array([[0.5]], np.float32), np.zeros((1, 1, 3), np.float32), cam, bg=(0.5, 0.25, 0.125))
    assert nr == 0
    np.testing.assert_array_equal(out["color"][:, 5, 7], [0.5, 0.25, 0.125])


def test_eigen_solver_against_numpy():
    co.set_precision("float32")
    rng = np.random.default_rng(3)
    for _ in range(200):
        A = rng.standard_normal((3, 3)).astype(np.float32) * rng.uniform(0.05, 2.0)
        S = (A @ A.T).astype(np.float32)
        n, val, vec = co.eig_sym3(S)
        assert n == 3
        w = np.linalg.eigvalsh(S.astype(np.float64))
        np.testing.assert_allclose(np.sort(val), w, atol=5e-6 * max(1.0, w.max()) + 3e-7)
        # reconstruction V diag(val) V^T = S, up to the solver's ABSOLUTE 1e-7 convergence threshold
        np.testing.assert_allclose(vec @ np.diag(val) @ vec.T, S, atol=2e-5 * max(1.0, np.abs(S).max()))
    # diagonal input needs no iteration and is returned as is
    n, val, vec = co.eig_sym3(np.diag([3.0, 1.0, 2.0]).astype(np.float32))
    assert n == 3 and sorted(val.tolist()) == [1.0, 2.0, 3.0]
