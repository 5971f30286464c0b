"""ctypes/numpy front end of the CPU oracle (oracle/rast_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under igs_amd/ or the drop-in packages
may import this module.

The call surface mirrors the reference's `_C.rasterize_gaussians` /
`_C.rasterize_gaussians_backward` (DGR/rasterize_points.cu:35-246) but on
numpy float32 arrays; `None` plays the role of the reference's empty tensors.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}                      # numpy scalar type -> loaded library (float32 and float64 builds of the same source)
_DTYPE = np.float32             # process-wide default precision
_TLS = threading.local()        # per-thread override (tests evaluate several views on worker threads)
_LOCK = threading.Lock()


def _dtype():
    return getattr(_TLS, "dtype", None) or _DTYPE


def set_precision(name):
    """'float32' (default; the parity oracle) or 'float64' (formula-check build, -DGSOR_DOUBLE).  Process-wide default;
    `thread_precision` overrides it for the calling thread only."""
    global _DTYPE
    _DTYPE = np.dtype(name).type


class thread_precision:
    """`with thread_precision("float64"): ...` -- precision of the oracle calls THIS thread makes inside the block."""

    def __init__(self, name):
        self.dtype = np.dtype(name).type

    def __enter__(self):
        self.prev = getattr(_TLS, "dtype", None)
        _TLS.dtype = self.dtype
        return self

    def __exit__(self, *exc):
        _TLS.dtype = self.prev
        return False


def build(force=False):
    """Compiles both precisions (make -C oracle) when the source is newer; returns the path for the current precision."""
    src = os.path.join(_HERE, "rast_oracle.c")
    sos = [os.path.join(_HERE, n) for n in ("librast_oracle.so", "librast_oracle_f64.so")]
    if force or any(not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)) for so in sos):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return sos[0] if _dtype() is np.float32 else sos[1]


def lib():
    dt = _dtype()
    with _LOCK:
        L = _LIBS.get(dt)
        if L is None:
            L = _LIBS[dt] = _load(build(), dt)
    return L


def _load(path, dt):
    L = C.CDLL(path)
    fp = C.c_void_p
    real = C.c_float if dt is np.float32 else C.c_double
    L.gsor_forward.restype = C.c_void_p
    L.gsor_forward.argtypes = ([C.c_int] * 5 + [fp] * 5 + [fp, real, fp, fp, fp, fp, fp]
                               + [real] * 3 + [C.c_int] * 3 + [fp] * 8 + [fp, fp])
    L.gsor_backward.restype = None
    L.gsor_backward.argtypes = [fp] * 34
    L.gsor_free.argtypes = [fp]
    L.gsor_free.restype = None
    L.gsor_num_rendered.argtypes = [fp]
    L.gsor_num_rendered.restype = C.c_int
    for name in ("depths", "camera_planes", "ray_planes", "ts", "normals", "means2D", "view_points", "cov3D",
                 "conic_opacity", "rgb", "clamped", "tiles_touched", "point_list", "ranges", "n_contrib", "keys"):
        f = getattr(L, "gsor_get_" + name)
        f.argtypes = [fp]
        f.restype = C.c_void_p
    L.gsor_mark_visible.argtypes = [C.c_int, fp, fp, fp, fp]
    L.gsor_mark_visible.restype = None
    L.gsor_eig_sym3.argtypes = [fp, fp, fp]
    L.gsor_eig_sym3.restype = C.c_int
    L.gsor_sh_to_rgb.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp]
    L.gsor_sh_to_rgb.restype = None
    L.gsor_set_flags.argtypes = [C.c_int]
    L.gsor_set_flags.restype = None
    return L


def _f32(a):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(a, dtype=_dtype()))
    return a if a.size else None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleState:
    """Owns the C state (geometry/binning/image buffers) between forward and backward."""

    def __init__(self, handle, P, W, H, T):
        self.h, self.P, self.W, self.H, self.T = handle, P, W, H, T
        self._lib = lib()          # the float32 and float64 builds have different struct layouts
        self.dtype = _dtype()

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "_lib", None) is not None:
            try:
                self._lib.gsor_free(self.h)
            except Exception:
                pass
            self.h = None

    @property
    def num_rendered(self):
        return self._lib.gsor_num_rendered(self.h)

    def _arr(self, name, dtype, shape):
        ptr = getattr(self._lib, "gsor_get_" + name)(self.h)
        n = int(np.prod(shape))
        if n == 0 or not ptr:
            return np.zeros(shape, dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def intermediates(self):
        P, R = self.P, self.num_rendered
        return dict(
            depths=self._arr("depths", self.dtype, (P,)),
            camera_planes=self._arr("camera_planes", self.dtype, (P, 6)),
            ray_planes=self._arr("ray_planes", self.dtype, (P, 2)),
            ts=self._arr("ts", self.dtype, (P,)),
            normals=self._arr("normals", self.dtype, (P, 3)),
            means2D=self._arr("means2D", self.dtype, (P, 2)),
            view_points=self._arr("view_points", self.dtype, (P, 3)),
            cov3D=self._arr("cov3D", self.dtype, (P, 6)),
            conic_opacity=self._arr("conic_opacity", self.dtype, (P, 4)),
            rgb=self._arr("rgb", self.dtype, (P, 3)),
            clamped=self._arr("clamped", np.uint8, (P, 3)),
            tiles_touched=self._arr("tiles_touched", np.uint32, (P,)),
            point_list=self._arr("point_list", np.uint32, (R,)),
            keys=self._arr("keys", np.uint64, (R,)),
            ranges=self._arr("ranges", np.uint32, (self.T, 2)),
            n_contrib=self._arr("n_contrib", np.uint32, (2, self.H, self.W)),
        )


def rasterize_forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, image_height, image_width, sh, degree,
                      campos, prefiltered=False, require_coord=True, require_depth=True):
    """Same argument order as `_C.rasterize_gaussians` (DGR/rasterize_points.cu:35-58), minus `debug`.

    Returns (num_rendered, dict(color, coord, mcoord, alpha, normal, depth, mdepth, radii), OracleState)."""
    DT = _dtype()
    means3D = _f32(means3D)
    if means3D is None:
        means3D = np.zeros((0, 3), DT)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    sh, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (sh, colors_precomp, scales, rotations, cov3D_precomp))
    M = sh.shape[1] if sh is not None else 0
    bg, opacities, viewmatrix, projmatrix, campos = map(_f32, (bg, opacities, viewmatrix, projmatrix, campos))
    out = dict(color=np.zeros((3, H, W), DT), coord=np.zeros((3, H, W), DT),
               mcoord=np.zeros((3, H, W), DT), depth=np.zeros((1, H, W), DT),
               mdepth=np.zeros((1, H, W), DT), alpha=np.zeros((1, H, W), DT),
               normal=np.zeros((3, H, W), DT), radii=np.zeros((P,), np.int32))
    nr, err = C.c_int(0), C.c_int(0)
    h = lib().gsor_forward(P, int(degree), M, W, H, _p(bg), _p(means3D), _p(sh), _p(colors_precomp), _p(opacities),
                           _p(scales), float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(viewmatrix),
                           _p(projmatrix), _p(campos), float(tan_fovx), float(tan_fovy), float(kernel_size),
                           int(bool(prefiltered)), int(bool(require_coord)), int(bool(require_depth)),
                           _p(out["color"]), _p(out["coord"]), _p(out["mcoord"]), _p(out["depth"]), _p(out["mdepth"]),
                           _p(out["alpha"]), _p(out["normal"]), _p(out["radii"]), C.byref(nr), C.byref(err))
    if not h:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return nr.value, out, OracleState(h, P, W, H, T)


def rasterize_backward(state, bg, means3D, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                       campos, sh, alpha, normalmap, grad_color, grad_coord, grad_mcoord, grad_depth, grad_mdepth,
                       grad_alpha, grad_normal, debug_intermediates=False):
    """Returns dict(means2D, colors, opacity, means3D, cov3D, sh, scales, rotations) like the 8-tuple of
    `_C.rasterize_gaussians_backward` (DGR/rasterize_points.cu:245)."""
    DT = _dtype()
    P, H, W = state.P, state.H, state.W
    means3D, sh, colors_precomp, scales, rotations, cov3D_precomp = map(
        _f32, (means3D, sh, colors_precomp, scales, rotations, cov3D_precomp))
    M = sh.shape[1] if sh is not None else 0
    z = lambda a, shape: (np.zeros(shape, DT) if a is None else _f32(a))
    g = dict(color=z(grad_color, (3, H, W)), coord=z(grad_coord, (3, H, W)), mcoord=z(grad_mcoord, (3, H, W)),
             depth=z(grad_depth, (1, H, W)), mdepth=z(grad_mdepth, (1, H, W)), alpha=z(grad_alpha, (1, H, W)),
             normal=z(grad_normal, (3, H, W)))
    out = dict(means2D=np.zeros((P, 3), DT), colors=np.zeros((P, 3), DT),
               opacity=np.zeros((P, 1), DT), means3D=np.zeros((P, 3), DT),
               cov3D=np.zeros((P, 6), DT), sh=np.zeros((P, M, 3), DT),
               scales=np.zeros((P, 3), DT), rotations=np.zeros((P, 4), DT))
    dbg = dict(view_points=np.zeros((P, 3), DT), ts=np.zeros((P,), DT),
               camera_planes=np.zeros((P, 6), DT), ray_planes=np.zeros((P, 2), DT),
               normals=np.zeros((P, 3), DT), conic=np.zeros((P, 4), DT))
    if P:
        bg, viewmatrix, projmatrix, campos, alpha, normalmap = map(_f32, (bg, viewmatrix, projmatrix, campos, alpha, normalmap))
        lib().gsor_backward(state.h, _p(bg), _p(means3D), _p(sh), _p(colors_precomp), _p(alpha), _p(scales), _p(rotations),
                            _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), _p(normalmap),
                            _p(g["color"]), _p(g["coord"]), _p(g["mcoord"]), _p(g["depth"]), _p(g["mdepth"]),
                            _p(g["alpha"]), _p(g["normal"]),
                            _p(out["means2D"]), _p(out["colors"]), _p(out["opacity"]), _p(out["means3D"]),
                            _p(out["cov3D"]), _p(out["sh"]), _p(out["scales"]), _p(out["rotations"]),
                            _p(dbg["view_points"]), _p(dbg["ts"]), _p(dbg["camera_planes"]), _p(dbg["ray_planes"]),
                            _p(dbg["normals"]), _p(dbg["conic"]))
    if debug_intermediates:
        out["_dbg"] = dbg
    return out


def set_flags(flags):
    """bit 0: drop the d(coef)/d(cov2D) terms in the backward; bit 1 (2): keep the per-Gaussian sums of the blend backward in
    float like the reference's atomicAdd (default: double, rounded once); bit 2 (4): scale every finished sum by 1 + 2e-6 u,
    u a hash of (index, flags >> 8); bits 4..7: jitter only that family of sums.  Test-only, see rast_oracle.c."""
    lib().gsor_set_flags(int(flags))


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D, viewmatrix, projmatrix = map(_f32, (means3D, viewmatrix, projmatrix))
    P = 0 if means3D is None else means3D.shape[0]
    present = np.zeros((P,), np.uint8)
    if P:
        lib().gsor_mark_visible(P, _p(means3D), _p(viewmatrix), _p(projmatrix), _p(present))
    return present.astype(bool)


def eig_sym3(S):
    DT = _dtype()
    S = _f32(S).reshape(3, 3)
    val = np.zeros(3, DT)
    vec = np.zeros(9, DT)
    n = lib().gsor_eig_sym3(_p(S), _p(val), _p(vec))
    return n, val, vec.reshape(3, 3).T.copy()   # columns = eigenvectors


def sh_to_rgb(deg, sh, mean, campos):
    DT = _dtype()
    sh = _f32(sh)
    rgb = np.zeros(3, DT)
    cl = np.zeros(3, np.uint8)
    lib().gsor_sh_to_rgb(int(deg), sh.shape[0], _p(_f32(mean)), _p(_f32(campos)), _p(sh), _p(rgb), _p(cl))
    return rgb, cl.astype(bool)
