"""ISA audit of every kernel in igs_amd/lib/libigs_rast.so (gfx950 code objects, `llvm-objdump -d`; runs without a GPU).

Two questions, both raised by GPU faults of earlier rounds:

1. BARRIERS.  Round 3's fault: hipcc emitted a bare `s_barrier` at a loop header with an LDS store (`wave_done[wid] = ...`) still
   outstanding -- no `s_waitcnt lgkmcnt(0)` behind the store -- so another wave could pass the barrier and read a stale word.  For
   every `s_barrier` of every kernel this tool runs a forward may-analysis over the kernel's control-flow graph:
     LDS-store-pending  = on SOME path into the barrier a DS instruction that writes LDS (ds_write* / ds_add* / ds_*xchg* / ...,
                          not ds_read* / ds_bpermute / ds_swizzle) has been issued with no `s_waitcnt ... lgkmcnt(0)` after it;
     VMEM-store-pending = the same for global / buffer / flat / scratch stores and atomics and `vmcnt(0)` (gfx9 counts stores in vmcnt).
   Only an explicit count of ZERO clears a bit (lgkmcnt also counts scalar loads, which return out of order).
   A barrier with LDS-store-pending is a DEFECT (exit code 1): waves synchronise on data that may not have landed.
   VMEM-store-pending is reported but not a defect by itself: no kernel of this library hands data to another wave of the same workgroup
   through global memory across a barrier (the ones that publish through global memory use atomics + __threadfence()).

2. M0 -> ds_write_addtid_b32.  Round 2's abort: the ISA requires one wait state between an SALU write of M0 and an LDS "add TID" store
   that reads it; inside inline assembly the compiler's hazard recogniser does not insert it.  For every ds_write_addtid_b32 the tool
   reports the number of instructions since the last write to m0 on the straight-line path before it (0 = back to back = DEFECT).

usage: python tools/audit_barriers.py [--json out.json] [--verbose]
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "igs_amd", "lib", "libigs_rast.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

DS_NO_WRITE = re.compile(r"^ds_(read|load|bpermute|permute|swizzle|nop|gws|consume|append|ordered_count|bvh|read_addtid)")
VMEM_STORE = re.compile(r"^(global|buffer|flat|scratch)_(store|atomic)")
INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
TARGET = re.compile(r"<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>\s*$")


def code_objects(lib):
    tmp = tempfile.mkdtemp(prefix="igs_audit_")
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    return tmp, sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def parse(path):
    """{kernel: [(addr, mnemonic, operands, branch_target_addr or None)]}"""
    txt = subprocess.run([OBJDUMP, "-d", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    funcs, cur, base = collections.OrderedDict(), None, {}
    for line in txt.splitlines():
        m = FUNC.match(line)
        if m:
            cur = m.group(2)
            base[cur] = int(m.group(1), 16)
            funcs[cur] = []
            continue
        m = INSN.match(line)
        if m and cur is not None:
            mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
            tgt = None
            if mn.startswith("s_cbranch") or mn == "s_branch":
                t = TARGET.search(line)
                if t and t.group(1) in base:
                    tgt = base[t.group(1)] + (int(t.group(2), 16) if t.group(2) else 0)
                else:
                    tgt = -1          # unknown target: handled conservatively
            funcs[cur].append((addr, mn, ops, tgt))
    return funcs


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def analyse(insns):
    """Forward may-analysis.  Returns ([(barrier addr, lds_pending, vmem_pending)], [(addtid addr, distance since m0 write)])."""
    addr_index = {a: i for i, (a, _, _, _) in enumerate(insns)}
    n = len(insns)
    succ = [[] for _ in range(n)]
    for i, (a, mn, ops, tgt) in enumerate(insns):
        if mn == "s_endpgm":
            continue
        if mn == "s_branch":
            if tgt in addr_index:
                succ[i].append(addr_index[tgt])
            continue
        if mn in ("s_setpc_b64", "s_swappc_b64"):
            continue                                   # (no indirect control flow inside these kernels; a call would end the analysis of the path)
        if i + 1 < n:
            succ[i].append(i + 1)
        if mn.startswith("s_cbranch") and tgt in addr_index:
            succ[i].append(addr_index[tgt])
    IN = [0] * n          # bit 0: LDS store pending, bit 1: VMEM store pending
    seen = [False] * n
    seen[0] = True
    work = collections.deque([0])
    while work:
        i = work.popleft()
        a, mn, ops, _ = insns[i]
        st = IN[i]
        if mn.startswith("ds_") and not DS_NO_WRITE.match(mn):
            st |= 1
        elif VMEM_STORE.match(mn):
            st |= 2
        elif mn == "s_waitcnt":
            if re.search(r"lgkmcnt\(0\)", ops):
                st &= ~1
            if re.search(r"vmcnt\(0\)", ops):
                st &= ~2
        for j in succ[i]:
            new = IN[j] | st
            if not seen[j] or new != IN[j]:
                seen[j] = True
                IN[j] = new
                work.append(j)
    barriers = [(insns[i][0], bool(IN[i] & 1), bool(IN[i] & 2)) for i in range(n) if insns[i][1] == "s_barrier" and seen[i]]
    addtid = []
    for i, (a, mn, ops, _) in enumerate(insns):
        if mn == "ds_write_addtid_b32":
            d, j = None, i - 1
            while j >= 0 and i - j <= 64:
                pm, po = insns[j][1], insns[j][2]
                if po.split(",")[0].strip() == "m0" and pm.startswith("s_"):
                    d = i - j - 1
                    break
                if pm.startswith("s_cbranch") or pm == "s_branch" or pm == "s_barrier":
                    break                               # (another basic block: the write is at least a branch away)
                j -= 1
            addtid.append((a, d))
    return barriers, addtid


def main():
    verbose = "--verbose" in sys.argv
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        from igs_amd import build
        build.build()
    tmp, cos = code_objects(LIB)
    report, defects = [], 0
    try:
        for co in cos:
            funcs = parse(co)
            names = demangle(list(funcs))
            for f, insns in funcs.items():
                if not insns:
                    continue
                bars, addtid = analyse(insns)
                nice = re.sub(r"\(.*$", "", names[f]).replace("void ", "")
                lds_bad = [hex(a) for a, l, v in bars if l]
                vm = [hex(a) for a, l, v in bars if v]
                tight = [hex(a) for a, d in addtid if d == 0]
                dist = [d for a, d in addtid if d is not None]
                report.append({"kernel": nice, "instructions": len(insns), "barriers": len(bars), "lds_store_pending_at": lds_bad,
                               "vmem_store_pending_at": vm, "addtid_stores": len(addtid), "addtid_min_distance_from_m0_write": min(dist) if dist else None,
                               "addtid_back_to_back_at": tight})
                defects += len(lds_bad) + len(tight)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    report.sort(key=lambda r: r["kernel"])
    tot_b = sum(r["barriers"] for r in report)
    print("%-78s %6s %5s %9s %9s %7s %7s" % ("kernel", "insns", "bar", "LDS-pend", "VMEM-pend", "addtid", "min-m0"))
    for r in report:
        if r["barriers"] or r["addtid_stores"] or verbose:
            print("%-78s %6d %5d %9d %9d %7d %7s" % (r["kernel"][:78], r["instructions"], r["barriers"], len(r["lds_store_pending_at"]),
                                                     len(r["vmem_store_pending_at"]), r["addtid_stores"],
                                                     "-" if r["addtid_min_distance_from_m0_write"] is None else r["addtid_min_distance_from_m0_write"]))
    print("%d kernels, %d barriers; barriers with an LDS store possibly outstanding: %d; add-TID stores back to back with an m0 write: %d"
          % (len(report), tot_b, sum(len(r["lds_store_pending_at"]) for r in report), sum(len(r["addtid_back_to_back_at"]) for r in report)))
    if "--json" in sys.argv:
        json.dump(report, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    return 1 if defects else 0


if __name__ == "__main__":
    sys.exit(main())
