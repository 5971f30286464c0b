"""Builds the compiled `_C` module (igs_amd/csrc_torch/igs_torch_ext.cpp: torch glue over the C ABI, the counterpart of the
reference's pybind module DGR/ext.cpp:15-20) in-tree: igs_amd/_C.<abi>.so, linked against igs_amd/lib/libigs_rast.so.

Host code only (no kernels in that file), so the host compiler is used directly with PyTorch's include / library paths -- the same
result `torch.utils.cpp_extension` would give, without its ninja / JIT cache (a cache under ~/.cache does not travel to the GPU box).
"""
import hashlib
import importlib.machinery
import os
import subprocess
import sys
import sysconfig

from . import build as _hip

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc_torch", "igs_torch_ext.cpp")
HEADER = os.path.join(HERE, "..", "include", "igs_rast.h")
TARGET = os.path.join(HERE, "_C" + importlib.machinery.EXTENSION_SUFFIXES[0])
STAMP = os.path.join(_hip.LIBDIR, "build_ext.stamp")
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-sign-compare",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H"]


def _fingerprint():
    import torch
    h = hashlib.sha256()
    for f in (SRC, HEADER):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update((" ".join(CXXFLAGS) + torch.__version__ + sys.version).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(TARGET) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _fingerprint()


def build(force=False, verbose=False):
    lib = _hip.build(verbose=verbose)                    # the C-ABI library first: _C links against it
    if not force and not needs_build():
        return TARGET
    import fcntl
    os.makedirs(_hip.LIBDIR, exist_ok=True)
    with open(os.path.join(_hip.LIBDIR, "build_ext.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return TARGET
            return _build_locked(lib, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(lib, verbose):
    import torch
    from torch.utils import cpp_extension as ce
    cxx = os.environ.get("CXX", "g++")
    inc = ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)
    tmp = TARGET + ".tmp.%d" % os.getpid()
    cmd = ([cxx] + CXXFLAGS + [abi] + inc + [SRC, "-o", tmp, "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
                                                "-ltorch_python", "-L" + os.path.dirname(lib), "-ligs_rast",
                                                "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath," + tlib, "-Wl,--no-as-needed"])
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("building the _C module failed:\n" + r.stdout[-6000:])
    if verbose and r.stdout.strip():
        print(r.stdout)
    os.replace(tmp, TARGET)
    with open(STAMP, "w") as f:
        f.write(_fingerprint())
    return TARGET


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
