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
SLP_OK = {"loss_ops.hip"} | set(os.environ.get("IGS_SLP_FILES", "").split())
SOURCES = ["api.hip", "preprocess.hip", "sort.hip", "blend_fwd.hip", "blend_bwd.hip", "blend_step.hip", "geom_bwd.hip", "refine_ops.hip", "loss_ops.hip", "io_ops.hip"]
# -fno-slp-vectorize: on gfx950 v_pk_*_f32 runs at the scalar-f32 rate per element, so SLP packing only adds v_mov shuffles
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-fno-slp-vectorize", "-Wno-inline-asm"]      # (-Wno-inline-asm: the column-write statements clobber m0 on purpose)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


STAMP = os.path.join(LIBDIR, "build.stamp")


def _fingerprint():
    """Content hash of everything the library is made from (sources, headers, flags): file times do not survive a copy to
    another machine, and a spurious rebuild by eight ranks at once is exactly what must not happen."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "igs_rast.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + sorted(SLP_OK) + [os.environ.get("IGS_EXTRA_FLAGS", "")]).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _fingerprint()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, "build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)                    # one builder at a time (ranks of a multi-GPU run share the tree)
        try:
            if not force and not needs_build():             # somebody else built it while we waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
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
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    os.replace(tmp, LIB)                                    # a process that has the old file mapped keeps it; nobody sees half a file
    with open(STAMP, "w") as f:
        f.write(_fingerprint())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
