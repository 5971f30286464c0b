"""Builds the HIP library in-tree: igs_amd/lib/libigs_rast.so (gfx950 code object, C ABI of include/igs_rast.h).

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libigs_rast.so")
# files whose inner loops are independent FMA streams: let the SLP vectorizer form v_pk_fma_f32 (2 FMAs per lane and instruction);
# everywhere else packing only added register moves
SLP_OK = {"loss_ops.hip"}
SOURCES = ["api.hip", "preprocess.hip", "sort.hip", "blend_fwd.hip", "blend_bwd.hip", "geom_bwd.hip", "refine_ops.hip", "loss_ops.hip"]
# -fno-slp-vectorize: on gfx950 v_pk_*_f32 runs at the scalar-f32 rate per element, so SLP packing only adds v_mov shuffles
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-fno-slp-vectorize"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "igs_rast.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        flags = [f for f in FLAGS if not (f == "-fno-slp-vectorize" and src in SLP_OK)]
        cmd = [hipcc] + flags + os.environ.get("IGS_EXTRA_FLAGS", "").split() + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
