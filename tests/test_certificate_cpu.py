"""CPU checks of the gradient-certificate machinery (tests/certificate.py): the float32 oracle certifies against itself on worker
threads (thread-local precision / flags, cached accumulators for the jitter samples), and a corrupted gradient is caught."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from igs_amd.scenes import cfg1_scene, activate
import certificate as cert


def _case(seed):
    raw, cams, bg = cfg1_scene(P=600, seed=seed, size=64)
    a = activate(raw)
    rng = np.random.default_rng(seed)
    grads = {k: None for k in cert.KEYS}
    grads["color"] = rng.standard_normal((3, 64, 64)).astype(np.float32)
    grads["depth"] = rng.standard_normal((1, 64, 64)).astype(np.float32)
    return a, cams[0], bg, grads


def test_oracle_certifies_itself_from_worker_threads_and_catches_a_corrupted_gradient():
    cases = [_case(s) for s in range(3)]
    with ThreadPoolExecutor(3) as ex:
        obs = list(ex.map(lambda c: cert.oracle_all(*c, samples=4), cases))
    # the same on the main thread, one after the other: identical results (flags / precision did not leak between threads)
    ob0 = cert.oracle_all(*cases[0], samples=4)
    for n in cert.GNAMES:
        np.testing.assert_array_equal(ob0["g32"][n], obs[0]["g32"][n])
        np.testing.assert_array_equal(ob0["g64"][n], obs[0]["g64"][n])
        np.testing.assert_array_equal(ob0["shift"][n], obs[0]["shift"][n])
    assert obs[0]["g64"]["means3D"].dtype == np.float64 and obs[0]["g32"]["means3D"].dtype == np.float32
    assert max(float(v.max()) for v in obs[0]["shift"].values()) > 0.0          # the jitter samples did something
    for ob in obs:
        g = [ob["g32"][n] for n in cert.GNAMES]
        st = cert.certify(g, ob, "self", verbose=False)
        assert all(v[1] == v[2] for v in st.values())        # "hip" = oracle32 here: needs the allowance exactly where the oracle is outside plain
    # a gradient that is wrong by 1 % on one well-conditioned element must fail
    ob = obs[1]
    g = [ob["g32"][n].copy() for n in cert.GNAMES]
    G = ob["g64"]["means3D"]
    quiet = ob["shift"]["means3D"] + np.abs(ob["g32"]["means3D"] - G).max(1)
    i = int(np.argmin(np.where(np.abs(G).max(1) > 0.1 * np.abs(G).max(), quiet, np.inf)))
    j = int(np.argmax(np.abs(G[i])))
    g[3][i, j] *= 1.01
    with pytest.raises(AssertionError, match="means3D"):
        cert.certify(g, ob, "corrupted", verbose=False)
