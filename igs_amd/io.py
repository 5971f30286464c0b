"""On-disk formats either side of the hot path: the start-Gaussian PLY (with `filter_3D`) and the
`cameras.json` + per-frame image layout, so the real N3DV assets can replace the synthetic stand-ins when they are mounted.

`plyfile` is not installed here (and must not be): the reader below parses the PLY header itself and maps the single `vertex`
element onto a numpy structured array (binary_little_endian, binary_big_endian or ascii; scalar properties only, which is all
a Gaussian PLY holds).

Reference: `igs/models/gs.py:297-312` (property order), `:318-343` (save_ply), `:400-462` (load_ply), `:480-490`
(get_scaling_n_opacity_with_3D_filter); `igs/data/infer_data.py:93-95,405-455` (cameras.json, images_r2/<img_name>.png).
"""
import json
import os

import numpy as np
import torch

from .camera import Camera, focal2fov

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_vertices(path):
    """Returns the `vertex` element of a PLY file as a numpy structured array (field names = property names)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = dict(name=tok[1], count=int(tok[2]), props=[])
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError("%s: list properties are not supported (element %s)" % (path, cur["name"]))
                cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError("%s: the first element must be `vertex`" % path)
        el = elements[0]
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=el["count"], ndmin=2)
            out = np.empty(el["count"], dtype=[(n, t) for n, t in el["props"]])
            for i, (n, _) in enumerate(el["props"]):
                out[n] = rows[:, i]
            return out
        order = "<" if fmt == "binary_little_endian" else ">" if fmt == "binary_big_endian" else None
        if order is None:
            raise ValueError("%s: unknown PLY format %r" % (path, fmt))
        dt = np.dtype([(n, order + t) for n, t in el["props"]])
        data = np.frombuffer(f.read(el["count"] * dt.itemsize), dtype=dt, count=el["count"])
        return data.astype(dt.newbyteorder("="))


def gaussian_ply_attributes(n_scale=3, n_rot=4):
    """Property order of the reference's writer (gs.py:297-312): always 45 `f_rest_*` columns."""
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)]
    return names + ["opacity"] + ["scale_%d" % i for i in range(n_scale)] + ["rot_%d" % i for i in range(n_rot)]


def write_gaussian_ply(path, raw, filter_3D=None):
    """save_ply (gs.py:318-343): raw leaves {xyz [P,3], shs [P,16,3], opacity [P,1] (logit), scaling [P,3] (log), rotation [P,4]};
    normals are zeros; f_dc / f_rest are channel-major (shs transposed).  `filter_3D` [P,1] adds the column the loader expects."""
    g = lambda k: raw[k].detach().cpu().float()
    xyz, shs = g("xyz"), g("shs")
    P = xyz.shape[0]
    f_dc = shs[:, 0:1].transpose(1, 2).flatten(start_dim=1)
    f_rest = shs[:, 1:].transpose(1, 2).flatten(start_dim=1)
    cols = [xyz, torch.zeros_like(xyz), f_dc, f_rest, g("opacity").view(P, 1), g("scaling"), g("rotation")]
    names = gaussian_ply_attributes()
    if filter_3D is not None:
        cols.append(filter_3D.detach().cpu().float().view(P, 1))
        names = names + ["filter_3D"]
    table = torch.cat(cols, dim=1).numpy().astype("<f4")
    assert table.shape[1] == len(names)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(("element vertex %d\n" % P).encode())
        for n in names:
            f.write(("property float %s\n" % n).encode())
        f.write(b"end_header\n")
        f.write(np.ascontiguousarray(table).tobytes())


def scaling_n_opacity_with_3D_filter(log_scale, opacity_logit, filter_3D):
    """gs.py:480-490: the Mip-Splatting style 3-D low-pass filter folded into scale and opacity."""
    opacity = torch.sigmoid(opacity_logit)
    scales_square = torch.square(torch.exp(log_scale))
    det1 = scales_square.prod(dim=1)
    after = scales_square + torch.square(filter_3D)
    det2 = after.prod(dim=1)
    coef = torch.sqrt(det1 / det2)
    return torch.sqrt(after), opacity * coef[..., None]


def load_start_gaussians(path, max_sh_degree=3):
    """load_ply (gs.py:400-462): the raw leaves of a start-Gaussian PLY, with `filter_3D` folded in:
    {xyz [P,3], rotation [P,4], shs [P,(d+1)^2,3], opacity [P,1] (logit), scaling [P,3] (log)}."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], dtype=np.float64)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opac = col("opacity")[..., None]
    f_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[..., None]                 # [P,3,1]
    rest_names = sorted((n for n in names if n.startswith("f_rest_")), key=lambda x: int(x.split("_")[-1]))
    K = (max_sh_degree + 1) ** 2
    if len(rest_names) != 3 * K - 3:
        raise ValueError("%s: %d f_rest columns, expected %d for SH degree %d" % (path, len(rest_names), 3 * K - 3, max_sh_degree))
    f_rest = np.stack([col(n) for n in rest_names], axis=1).reshape(xyz.shape[0], 3, K - 1)
    scales = np.stack([col(n) for n in sorted((n for n in names if n.startswith("scale_")), key=lambda x: int(x.split("_")[-1]))], axis=1)
    rots = np.stack([col(n) for n in sorted((n for n in names if n.startswith("rot")), key=lambda x: int(x.split("_")[-1]))], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float)
    shs = torch.cat((t(f_dc).transpose(1, 2).contiguous(), t(f_rest).transpose(1, 2).contiguous()), dim=1)
    log_scale, logit = t(scales), t(opac)
    if "filter_3D" in names:
        s_act, o_act = scaling_n_opacity_with_3D_filter(log_scale, logit, t(col("filter_3D")[..., None]))
        logit = torch.log(o_act / (1 - o_act))                    # inverse_sigmoid
        log_scale = torch.log(s_act)
    return dict(xyz=t(xyz), rotation=t(rots), shs=shs, opacity=logit, scaling=log_scale)


def _ply_columns(names, K):
    """Column indices, in the order igs_ply_to_params wants them (include/igs_rast.h)."""
    idx = {n: i for i, n in enumerate(names)}
    by_num = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda x: int(x.split("_")[-1]))
    rest, scales, rots = by_num("f_rest_"), by_num("scale_"), by_num("rot")
    if len(rest) != 3 * K - 3 or len(scales) != 3 or len(rots) != 4:
        raise ValueError("PLY columns do not describe SH degree %d Gaussians (%d f_rest, %d scale, %d rot columns)"
                         % (int(round(K ** 0.5)) - 1, len(rest), len(scales), len(rots)))
    cols = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + rest + ["opacity"] + scales + rots
    return [idx[c] for c in cols] + [idx.get("filter_3D", -1)]


def load_start_gaussians_gpu(path, device, max_sh_degree=3):
    """`load_start_gaussians` with the per-Gaussian work on the GPU: the host only parses the header and hands the vertex table over as it
    lies in the file; column gathering, the channel-major -> [P, K, 3] SH re-layout and the `filter_3D` fold (gs.py:400-462, 480-490) are
    ONE launch (igs_ply_to_params).  Tables whose properties are all float32 (what the reference writes); anything else raises.
    Returns the raw leaves on `device`."""
    import ctypes as C
    from . import _cabi
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("load_start_gaussians_gpu: needs a GPU device (the host-only reader is load_start_gaussians)")
    v = read_ply_vertices(path)
    names = v.dtype.names
    if any(v.dtype[n] != np.dtype("f4") for n in names):
        raise ValueError("%s: non-float32 vertex properties; use load_start_gaussians" % path)
    P, stride = v.shape[0], len(names)
    K = (max_sh_degree + 1) ** 2
    cols = _ply_columns(names, K)
    table = torch.from_numpy(np.ascontiguousarray(v).view(np.float32).reshape(P, stride)).to(device)
    f = dict(dtype=torch.float32, device=device)
    out = dict(xyz=torch.empty((P, 3), **f), rotation=torch.empty((P, 4), **f), shs=torch.empty((P, K, 3), **f),
               opacity=torch.empty((P, 1), **f), scaling=torch.empty((P, 3), **f))
    with torch.cuda.device(device):
        rc = _cabi.lib().igs_ply_to_params(torch.cuda.current_stream(device).cuda_stream, P, table.data_ptr(), stride, (C.c_int * len(cols))(*cols),
                                           len(cols), K, out["xyz"].data_ptr(), out["rotation"].data_ptr(), out["shs"].data_ptr(),
                                           out["opacity"].data_ptr(), out["scaling"].data_ptr())
    if rc != 0:
        raise RuntimeError("igs_ply_to_params failed: %d" % rc)
    return out


def write_gaussian_ply_gpu(path, raw):
    """`write_gaussian_ply` for leaves that live on the GPU: the [P, 62] table (gs.py:297-343) is laid out by ONE launch
    (igs_params_to_ply) and leaves the device as a single copy; the host writes header + bytes."""
    from . import _cabi
    xyz = raw["xyz"].detach().contiguous().float()
    dev, P = xyz.device, xyz.shape[0]
    if dev.type != "cuda":
        raise RuntimeError("write_gaussian_ply_gpu: leaves must be on a GPU (the host-only writer is write_gaussian_ply)")
    g = lambda k: raw[k].detach().contiguous().float()
    shs, rot, op, sc = g("shs"), g("rotation"), g("opacity"), g("scaling")
    K = shs.shape[1]
    table = torch.empty((P, 14 + 3 * K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().igs_params_to_ply(torch.cuda.current_stream(dev).cuda_stream, P, K, xyz.data_ptr(), rot.data_ptr(), shs.data_ptr(),
                                           op.data_ptr(), sc.data_ptr(), table.data_ptr())
    if rc != 0:
        raise RuntimeError("igs_params_to_ply failed: %d" % rc)
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(3 * K - 3)] \
        + ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(("element vertex %d\n" % P).encode())
        for n in names:
            f.write(("property float %s\n" % n).encode())
        f.write(b"end_header\n")
        f.write(table.cpu().numpy().astype("<f4").tobytes())


def load_cameras_json(path, training_only=True):
    """infer_data.py:93-95,412-440: entries {img_name, rotation 3x3 (c2w), position, fx, fy, width, height}; the first entry is
    the held-out test view (`cameras_data[1:]` are the training views).  Returns (list of dicts with c2w 4x4, FOV, size, name)."""
    with open(path) as f:
        data = json.load(f)
    if training_only:
        data = data[1:]
    out = []
    for cam in data:
        c2w = np.zeros((4, 4))
        c2w[:3, :3] = np.array(cam["rotation"])
        c2w[:3, 3] = np.array(cam["position"])
        c2w[3, 3] = 1
        out.append(dict(name=cam["img_name"], c2w=torch.from_numpy(c2w).to(torch.float),
                        FOV=(focal2fov(cam["fx"], cam["width"]), focal2fov(cam["fy"], cam["height"])),
                        width=int(cam["width"]), height=int(cam["height"])))
    return out


def cameras_from_json(path, training_only=True, resolution=None):
    """`Camera.from_c2w(c2w, FOV, image.shape[-2:])` for every entry (infer_batch.py:295); `resolution` = (H, W) overrides the
    size in the file (the reference takes it from the loaded image, e.g. the half-resolution `images_r2`)."""
    cams = []
    for e in load_cameras_json(path, training_only):
        res = resolution if resolution is not None else (e["height"], e["width"])
        cams.append(Camera.from_c2w(e["c2w"], e["FOV"], res))
    return cams


def load_frame_images(frame_dir, entries, subdir="images_r2", ext=".png"):
    """infer_data.py:417-431: `<frame_dir>/<subdir>/<img_name><ext>` -> float [3,H,W] in [0,1]."""
    from PIL import Image
    imgs = []
    for e in entries:
        a = np.array(Image.open(os.path.join(frame_dir, subdir, e["name"] + ext)))
        imgs.append(torch.from_numpy(a / 255.0).permute(2, 0, 1)[:3].to(torch.float))
    return imgs
