"""Every-element gradient certificate (test infrastructure; VERDICT r1 "next #1").

north_star asks for "gradients within 1e-3 rel".  The reference's own arithmetic does not let a float32 implementation promise that
for EVERY element against another float32 implementation: it recovers the transmittance as T_final = 1 - sum(alpha T) and divides it
back out (backward.cu:706,857), which amplifies the last bits of expf by 1/T_final behind saturated pixels; its per-Gaussian sums are
float atomicAdds in arbitrary order (backward.cu:878-1013); and with kernel_size = 0 its coef backward adds the rounding residue of two
equal terms (backward.cu:367-375).  So the bar is stated against a float64 evaluation of the same formulas, per element:

    |hip - f64|  <=  REL * |f64| + FLOOR * max|f64 of the tensor|                                  ("plain": 1e-3 relative)
                     + K * |oracle32 - f64|_g + K * jitter_g                                        (the allowance)

where, for the Gaussian g the element belongs to, |oracle32 - f64|_g is the distance of the float32 ORACLE itself from float64 (max over
the Gaussian's components of that tensor) and jitter_g is how far the oracle's result moves (a) when its per-Gaussian sums are kept in
float (flag 2) and each finished sum is scaled by 1 + 2e-6 u (flag 4; `samples` draws of u), and (b) when every exp() of the blend, forward
and backward alike, is off by up to 2 ulp (flag 16; `exp_samples` draws) -- 2 ulp is the documented bound of CUDA's expf, which the
reference calls (no fast-math in its setup.py); x86 libm (the oracle) and v_exp_f32 (the HIP kernels) differ from it by as much.
An element that needs the allowance is an element on which the reference itself -- CUDA atomics in another order, another exp --
would differ by as much.  The function returns
how many elements needed it, so a test can print the number and bound it; any element outside even the allowance fails.
"""
import numpy as np

from oracle import c_oracle as co

KEYS = ["color", "coord", "mcoord", "depth", "mdepth", "alpha", "normal"]
GNAMES = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations"]

REL = 1e-3          # north_star's relative bar
FLOOR = 1e-5        # absolute floor, as a fraction of the tensor's largest |f64| entry (elements that cancel to ~0)
K = 5.0             # multiple of the oracle's own float32 error / jitter shift that is still "the same arithmetic"


def _np(x, dt):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=dt)


def oracle_all(a, cam, bg, grads, req=(True, True), deg=3, kernel_size=0.0, colors=None, cov=None, scale_modifier=1.0,
               samples=16, exp_samples=3):
    """Everything the certificate needs from the CPU oracle for one view, callable from a worker thread (ctypes releases the GIL,
    the oracle's flags are thread-local): float32 forward + backward (the parity oracle proper), float64 forward + backward, and
    the per-Gaussian jitter shifts.  `grads`: dict of the seven upstream image gradients (numpy, None = zeros).
    Returns dict(nr, out, state, g32, g64, shift)."""
    P = a["means3D"].shape[0]

    def fwd_bwd(dt, fwd_flags, flags_list, each):
        with co.thread_precision(dt):
            T = np.float32 if dt == "float32" else np.float64
            arg = {k: _np(v, T) for k, v in a.items()}
            c, v = _np(colors, T), _np(cov, T)
            sc, ro = (None, None) if v is not None else (arg["scales"], arg["rotations"])
            sh = None if c is not None else arg["shs"]
            V, Pm, cc = _np(cam.world_view_transform, T), _np(cam.full_proj_transform, T), _np(cam.camera_center, T)
            try:
                co.set_flags(fwd_flags)
                nr, oo, st = co.rasterize_forward(_np(bg, T), arg["means3D"], c, arg["opacities"], sc, ro, scale_modifier, v, V, Pm,
                                                  cam.tanfovx, cam.tanfovy, kernel_size, cam.height, cam.width, sh, deg, cc,
                                                  require_coord=req[0], require_depth=req[1])
                for fl in flags_list:
                    co.set_flags(fl)
                    each(co.rasterize_backward(st, _np(bg, T), arg["means3D"], c, sc, ro, v, V, Pm, cc, sh, oo["alpha"], oo["normal"],
                                               *[None if grads.get(k) is None else _np(grads[k], T) for k in KEYS]))
            finally:
                co.set_flags(0)
            return nr, oo, st

    # float32: plain (double sums rounded once = the parity oracle), then float sums + jitter samples on the cached accumulators
    box = {"g32": None, "shift": {}}

    def each32(g):
        if box["g32"] is None:
            box["g32"] = g
            return
        for n in GNAMES:
            if g[n].size == 0:
                continue
            d = np.abs(g[n].astype(np.float64).reshape(P, -1) - box["g32"][n].astype(np.float64).reshape(P, -1)).max(1)
            box["shift"][n] = d if n not in box["shift"] else np.maximum(box["shift"][n], d)
    nr, oo, st = fwd_bwd("float32", 0, [0] + [2 + 4 + 8 + 256 * s for s in range(samples)], each32)
    g32, shift = box["g32"], box["shift"]
    for n in GNAMES:
        if g32[n].size and n not in shift:
            shift[n] = np.zeros(P)
    # exp sensitivity: forward AND backward with every blend exp() off by up to 2 ulp; the shifts ADD to those of the sum jitter
    sum_shift = {n: v.copy() for n, v in shift.items()}
    box["shift"] = {}
    for s_ in range(exp_samples):
        fwd_bwd("float32", 16 + 256 * s_, [16 + 256 * s_], each32)
    for n in sum_shift:
        shift[n] = sum_shift[n] + box["shift"].get(n, 0.0)
    res64 = []
    fwd_bwd("float64", 0, [0], res64.append)
    return dict(nr=nr, out=oo, state=st, g32=g32, g64=res64[0], shift=shift)


def certify(gout, ob, label="", max_allowance_frac=None, verbose=True, allowance_floor=0, oracle_factor=None):
    """`gout`: the HIP gradients (8-tuple of tensors / arrays in GNAMES order); `ob`: result of `oracle_all`.
    Raises AssertionError naming the worst element if any element is outside plain + allowance.  Returns
    {tensor: (n_elements, n_needing_allowance, n_where_the_f32_oracle_itself_is_outside_plain)}."""
    g32, g64, shift = ob["g32"], ob["g64"], ob["shift"]
    stats = {}
    for n, t in zip(GNAMES, gout):
        if g32[n].size == 0:
            continue
        P = g32[n].shape[0]
        A = _np(t, np.float64).reshape(P, -1)
        assert not np.isnan(A).any(), (label, n, "NaN in the HIP gradient")
        G, G32 = g64[n].reshape(P, -1), g32[n].astype(np.float64).reshape(P, -1)
        scale = max(np.abs(G).max(), 1e-30)
        plain = REL * np.abs(G) + FLOOR * scale
        e_hip = np.abs(A - G)
        e_or_g = np.abs(G32 - G).max(1, keepdims=True)
        allow = K * e_or_g + K * shift[n].reshape(P, 1)
        need = e_hip > plain
        bad = e_hip > plain + allow
        if bad.any():
            i = np.unravel_index(int(np.argmax(np.where(bad, e_hip - plain - allow, -1))), bad.shape)
            raise AssertionError("%s %s: element %s of Gaussian %d: hip %.9g, f64 %.9g, oracle32 %.9g; |hip-f64| %.3g > plain %.3g + 5*|o32-f64|_g %.3g "
                                 "+ 5*jitter_g %.3g  (%d of %d elements outside)" % (label, n, i[1], i[0], A[i], G[i], G32[i], e_hip[i], plain[i],
                                                                                    K * e_or_g[i[0], 0], K * shift[n][i[0]], int(bad.sum()), bad.size))
        stats[n] = (int(need.size), int(need.sum()), int((np.abs(G32 - G) > plain).sum()))
    tot = sum(v[0] for v in stats.values()); used = sum(v[1] for v in stats.values()); orc = sum(v[2] for v in stats.values())
    if verbose:
        print("certificate %s: %d gradient elements, every one within 1e-3 rel of float64 + allowance; %d (%.4f %%) needed the allowance "
              "(the float32 oracle itself is outside plain 1e-3 on %d = %.4f %%); per tensor: %s"
              % (label, tot, used, 100.0 * used / max(tot, 1), orc, 100.0 * orc / max(tot, 1),
                 ", ".join("%s %d/%d" % (k, v[1], v[0]) for k, v in stats.items())))
    if max_allowance_frac is not None:
        # (`allowance_floor`: scenes of a handful of Gaussians, where ONE ill-conditioned Gaussian is a large fraction of everything;
        #  `oracle_factor`: a scene that is ill-conditioned as a whole -- a few hundred Gaussians stacked 60 deep, every pixel
        #  saturated -- on which the float32 ORACLE itself leaves plain 1e-3 on more elements than that: the kernels may then need
        #  the allowance on oracle_factor times as many elements as the oracle is outside on)
        limit = max(max_allowance_frac * tot, allowance_floor, (oracle_factor * orc + allowance_floor) if oracle_factor else 0)
        assert used <= limit, (label, used, tot, orc)
    return stats
