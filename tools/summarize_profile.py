"""Turns rocprofv3 CSV output under gpurun_out/ into the tracked summaries under profiles/.

usage: python tools/summarize_profile.py <round-tag> <kernel_stats_dir> [--fetch DIR] [--write DIR] [--sq DIR] [--misc DIR]
HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, both are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) reads, so the read
side is doubled before it is compared with a byte count.
"""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"blend_step": "blend_step_kernel", "blend_fwd": "blend_fwd_kernel", "blend_bwd": "blend_bwd_kernel", "preprocess": "preprocess_fwd_kernel",
        "geom_bwd": "geom_bwd_kernel", "radix_scatter": "radix_scatter_kernel", "radix_hist": "radix_hist_kernel",
        "emit": "emit_instances_kernel", "adam": "adam_groups_kernel", "l1": "l1_kernel", "tile_sort": "tile_sort_kernel",
        "activate_fwd": "activate_fwd_kernel", "activate_bwd": "activate_bwd_kernel"}


def counters(d):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def per_kernel(agg, counter):
    out = {}
    for short, pat in KEYS.items():
        vals = [v for k, c in agg.items() if pat in k for v in c.get(counter, [])]
        if vals:
            out[short] = sum(vals) / len(vals)
    return out


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    opts = dict(zip(sys.argv[3::2], sys.argv[4::2]))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    f = glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    dst = os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag)
    with open(dst, "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print("wrote", dst)
    summary = {}
    for short, pat in KEYS.items():
        for r in rows:
            if pat in r["Name"]:
                summary.setdefault(short, {})["avg_us"] = float(r["AverageNs"]) / 1e3
                summary[short]["calls"] = int(r["Calls"])
                break
    fetch = per_kernel(counters(opts["--fetch"]), "FETCH_SIZE") if "--fetch" in opts else {}
    write = per_kernel(counters(opts["--write"]), "WRITE_SIZE") if "--write" in opts else {}
    for k in summary:
        if k in fetch or k in write:
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            summary[k]["FETCH_SIZE_KiB"] = fk
            summary[k]["WRITE_SIZE_KiB"] = wk
            summary[k]["hbm_bytes_per_launch"] = (2.0 * fk + wk) * 1024.0      # gfx950 correction: read side doubled
            summary[k]["hbm_bytes_per_launch_uncorrected"] = (fk + wk) * 1024.0
    for name in ("--sq", "--misc"):
        if name in opts:
            agg = counters(opts[name])
            cs = set(c for v in agg.values() for c in v)
            for c in sorted(cs):
                for k, val in per_kernel(agg, c).items():
                    summary.setdefault(k, {})[c] = val
    # which build of the kernels these counters belong to (bench.py reports them only next to launch times of the same build)
    sys.path.insert(0, ROOT)
    try:
        from igs_amd import build as _b
        summary["_meta"] = {"build_fingerprint": _b._fingerprint()}
    except Exception as e:  # noqa: BLE001
        summary["_meta"] = {"build_fingerprint": None, "error": str(e)}
    dst = os.path.join(ROOT, "profiles", "%s_pmc.json" % tag)
    json.dump(summary, open(dst, "w"), indent=1, sort_keys=True)
    json.dump(summary, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", dst)
    for k, v in summary.items():
        if k == "_meta":
            continue
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
