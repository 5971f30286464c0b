"""On-disk formats: start-Gaussian PLY with filter_3D, cameras.json, frame images.  CPU only."""
import json
import math
import os

import numpy as np
import torch

from igs_amd import io as gio
from igs_amd.scenes import cfg1_scene


def _raw(P=257):
    raw, _, _ = cfg1_scene(P=P, size=32)
    return raw


def test_ply_property_order_is_the_reference_writers():
    # igs/models/gs.py:297-312: x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3
    names = gio.gaussian_ply_attributes()
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54] == "opacity"
    assert names[55:58] == ["scale_0", "scale_1", "scale_2"] and names[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"] and len(names) == 62


def test_ply_round_trip_and_channel_major_sh_layout(tmp_path):
    raw = _raw()
    p = str(tmp_path / "g" / "start.ply")
    gio.write_gaussian_ply(p, raw)
    v = gio.read_ply_vertices(p)
    assert v.shape[0] == raw["xyz"].shape[0] and v.dtype.names == tuple(gio.gaussian_ply_attributes())
    # f_dc_c = shs[:, 0, c]; f_rest is channel-major: f_rest_{c*15 + k-1} = shs[:, k, c]   (gs.py:325-326, transpose + flatten)
    np.testing.assert_array_equal(v["f_dc_1"], raw["shs"][:, 0, 1].numpy())
    np.testing.assert_array_equal(v["f_rest_%d" % (2 * 15 + 6)], raw["shs"][:, 7, 2].numpy())
    assert np.all(v["nx"] == 0)
    back = gio.load_start_gaussians(p)                      # no filter_3D column: values come back unchanged
    for k in raw:
        torch.testing.assert_close(back[k].reshape(raw[k].shape), raw[k].float(), rtol=0, atol=0)


def test_reads_a_ply_assembled_byte_for_byte_like_the_reference_writer(tmp_path):
    """The reference's `save_ply` (igs/models/gs.py:317-342) fills a numpy record array of '<f4' fields in
    `construct_list_of_attributes()` order and hands it to plyfile, which writes the header
    `ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... / end_header` followed by the packed
    records.  The file is assembled here with struct / numpy only -- not with this repository's writer -- and must come back through
    `load_start_gaussians` with the channel-major f_rest layout undone (`load_ply`, gs.py:400-462).  (No reference-written PLY
    exists in the reference tree to test against; this pins the reader against the FORMAT, independently of our own writer.)"""
    import struct
    N = 5
    rng = np.random.default_rng(3)
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] \
        + ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]
    assert len(names) == 62
    vals = rng.standard_normal((N, 62)).astype("<f4")
    p = str(tmp_path / "ref_like.ply")
    with open(p, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N).encode("ascii"))
        for n in names:
            f.write(("property float %s\n" % n).encode("ascii"))
        f.write(b"end_header\n")
        for row in vals:
            f.write(struct.pack("<62f", *[float(v) for v in row]))
    back = gio.load_start_gaussians(p)
    col = {n: vals[:, i] for i, n in enumerate(names)}
    np.testing.assert_array_equal(back["xyz"].numpy(), vals[:, 0:3])
    np.testing.assert_array_equal(back["opacity"].numpy().reshape(-1), col["opacity"])
    np.testing.assert_array_equal(back["scaling"].numpy(), vals[:, 55:58])
    np.testing.assert_array_equal(back["rotation"].numpy(), vals[:, 58:62])
    shs = back["shs"].numpy().reshape(N, 16, 3)
    for c in range(3):
        np.testing.assert_array_equal(shs[:, 0, c], col["f_dc_%d" % c])
        for k in range(1, 16):                               # features_extra.reshape(P, 3, 15) then transposed (gs.py:421-424)
            np.testing.assert_array_equal(shs[:, k, c], col["f_rest_%d" % (c * 15 + k - 1)])


def test_filter_3d_is_folded_into_scale_and_opacity(tmp_path):
    raw = _raw(64)
    f3 = torch.rand(64, 1) * 0.05 + 0.01
    p = str(tmp_path / "f.ply")
    gio.write_gaussian_ply(p, raw, filter_3D=f3)
    back = gio.load_start_gaussians(p)
    s2 = torch.exp(raw["scaling"]) ** 2
    exp_scale = torch.sqrt(s2 + f3 ** 2)
    coef = torch.sqrt(s2.prod(dim=1) / (s2 + f3 ** 2).prod(dim=1))
    exp_op = torch.sigmoid(raw["opacity"]).view(-1) * coef
    torch.testing.assert_close(torch.exp(back["scaling"]), exp_scale, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(torch.sigmoid(back["opacity"]).view(-1), exp_op, rtol=1e-5, atol=1e-7)
    assert torch.all(torch.exp(back["scaling"]) >= torch.exp(raw["scaling"]))


def test_ascii_and_big_endian_ply(tmp_path):
    p = str(tmp_path / "a.ply")
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment hello\nelement vertex 2\nproperty float x\nproperty double y\nproperty uchar z\nend_header\n")
        f.write("1.5 2.5 3\n-4 5 6\n")
    v = gio.read_ply_vertices(p)
    assert v["x"].tolist() == [1.5, -4.0] and v["y"].tolist() == [2.5, 5.0] and v["z"].tolist() == [3, 6]
    q = str(tmp_path / "b.ply")
    with open(q, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty float x\nproperty int y\nend_header\n")
        f.write(np.array([(1.25, 7), (-2.0, -9)], dtype=[("x", ">f4"), ("y", ">i4")]).tobytes())
    w = gio.read_ply_vertices(q)
    assert w["x"].tolist() == [1.25, -2.0] and w["y"].tolist() == [7, -9]


def test_cameras_json_and_frame_images(tmp_path, golden):
    from PIL import Image
    cams = []
    for i in range(4):
        a = 0.1 * i
        R = [[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]
        cams.append(dict(id=i, img_name="cam%02d" % i, width=1352, height=1014, position=[0.1 * i, 0.2, -3.0], rotation=R, fx=730.0, fy=731.0))
    jp = str(tmp_path / "cameras.json")
    json.dump(cams, open(jp, "w"))
    ent = gio.load_cameras_json(jp)
    assert [e["name"] for e in ent] == ["cam01", "cam02", "cam03"]            # the first entry is the held-out test view
    e = ent[1]
    assert e["c2w"][3].tolist() == [0, 0, 0, 1] and abs(float(e["c2w"][0, 3]) - 0.2) < 1e-7
    assert abs(e["FOV"][0] - 2 * math.atan(1352 / (2 * 730.0))) < 1e-12
    # focal2fov against the reference's own helper (tests/golden/ref_helpers.npz, from igs/utils/graphics_utils.py)
    np.testing.assert_allclose([ent[0]["FOV"][0], gio.focal2fov(730.0, 1014)], golden["focal2fov"], rtol=1e-12)
    cs = gio.cameras_from_json(jp, resolution=(507, 676))
    assert len(cs) == 3 and cs[0].height == 507 and cs[0].width == 676
    torch.testing.assert_close(cs[1].world_view_transform, torch.inverse(ent[1]["c2w"]).t().contiguous())
    fd = tmp_path / "frame000001" / "images_r2"
    os.makedirs(fd)
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (3, 5, 7, 3), dtype=np.uint8)
    for k, e in enumerate(ent):
        Image.fromarray(px[k]).save(str(fd / (e["name"] + ".png")))
    imgs = gio.load_frame_images(str(tmp_path / "frame000001"), ent)
    assert imgs[2].shape == (3, 5, 7) and imgs[2].dtype == torch.float32
    np.testing.assert_allclose(imgs[2].numpy(), np.transpose(px[2], (2, 0, 1)) / 255.0, rtol=0, atol=1e-7)
