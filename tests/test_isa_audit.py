"""ISA-level guards on the built gfx950 code objects (no GPU needed): tools/audit_barriers.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_barrier_with_an_outstanding_lds_store_and_no_m0_hazard():
    """Every `s_barrier` of every kernel in libigs_rast.so: on no path into it may an LDS store be outstanding (round 3's GPU fault was a
    bare s_barrier behind `wave_done[wid] = ...`; built with -DIGS_NO_RELEASE_WAIT the audit flags exactly that barrier in every
    blend_step instance); every ds_write_addtid_b32 has at least one instruction between it and the last write of M0 (round 2's abort)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from igs_amd import build
    build.build()
    import audit_barriers as A
    tmp, cos = A.code_objects(build.LIB)
    try:
        kernels = barriers = addtid_n = 0
        for co in cos:
            for f, insns in A.parse(co).items():
                if not insns:
                    continue
                kernels += 1
                bars, addtid = A.analyse(insns)
                barriers += len(bars)
                addtid_n += len(addtid)
                assert not [hex(a) for a, lds, _ in bars if lds], (f, "s_barrier with an LDS store possibly outstanding")
                assert not [hex(a) for a, d in addtid if d == 0], (f, "ds_write_addtid_b32 right behind a write of m0")
        assert kernels >= 40 and barriers >= 100 and addtid_n >= 100, (kernels, barriers, addtid_n)      # (the audit really saw the library)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def test_audit_finds_a_planted_hazard():
    """The analysis on a hand-written instruction stream: a DS store, no wait, a barrier in a loop -> flagged; with lgkmcnt(0) -> clean."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_barriers as A
    bad = [(0, "v_mov_b32_e32", "v1, 0", None), (4, "s_barrier", "", None), (8, "ds_read_b32", "v2, v1", None),
           (12, "ds_write_b32", "v1, v2", None), (16, "s_cbranch_scc1", "65530", 4), (20, "s_endpgm", "", None)]
    bars, _ = A.analyse(bad)
    assert bars == [(4, True, False)]
    good = bad[:4] + [(14, "s_waitcnt", "lgkmcnt(0)", None)] + bad[4:]
    bars, _ = A.analyse(good)
    assert bars == [(4, False, False)]
    m0 = [(0, "s_mov_b32", "m0, s5", None), (4, "ds_write_addtid_b32", "v3 offset:16", None), (8, "s_mov_b32", "m0, s6", None),
          (12, "s_nop", "0", None), (16, "ds_write_addtid_b32", "v4", None), (20, "s_endpgm", "", None)]
    _, addtid = A.analyse(m0)
    assert addtid == [(4, 0), (16, 1)]


def test_no_kernel_writes_back_its_l2():
    """No `buffer_wbl2` in any kernel of libigs_rast.so: on gfx950 a device- or system-scope RELEASE fence (`__threadfence()`,
    `__threadfence_system()`, a release atomic) is a write-back of every dirty line of the XCD's L2, and a kernel that has just streamed its
    output through that L2 waits for all of it -- round 4: a thousand workgroups fencing in front of a done-counter made a 10 us kernel
    take 40 (profiles/r04_l1_mean_fence_ab.txt).  Cross-workgroup hand-offs inside a kernel use agent-scope atomic stores / loads and
    `s_waitcnt vmcnt(0)` instead (refine_ops.hip: l1_mean_kernel; blend_fwd.hip: the status words for the host)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from igs_amd import build
    build.build()
    import audit_barriers as A
    tmp, cos = A.code_objects(build.LIB)
    try:
        seen = 0
        for co in cos:
            for f, insns in A.parse(co).items():
                seen += len(insns)
                hits = [hex(a) for a, mn, _, _ in insns if mn.startswith("buffer_wbl2")]
                assert not hits, (f, "L2 write-back (a release fence) inside a kernel", hits)
        assert seen > 50000, seen          # (the scan really saw the library: ~99 000 instructions)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
