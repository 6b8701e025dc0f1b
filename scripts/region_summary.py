#!/usr/bin/env python
"""Per-region kernel summary from a `rocprofv3 --kernel-trace --marker-trace` output directory.

For every roctx region of the driver (timed_region_model, timed_region_pcg, ecmech_kernel, krylov_solver, ...) and every kernel
dispatched inside it: number of dispatches, mean / min / max duration, and the resource columns of the dispatch records (VGPRs,
AGPRs, SGPRs, scratch bytes per lane, LDS bytes per block, workgroup size).  This is what makes the roofline figure of bench.py
reproducible from profiles/ alone: `timed_region_model` holds exactly the constitutive launches of the timed loop.
Without a marker trace the last N dispatches of each kernel are summarised instead (N = --last).
Usage: python scripts/region_summary.py <trace_dir> <out.csv> [--last 20]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    for pre in ("void ", "ecmdev::"):
        n = n.replace(pre, "")
    return n[:110]


def main():
    d, out = sys.argv[1], sys.argv[2]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 20
    kfiles = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    mfiles = glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)
    disp = []
    for f in kfiles:
        for r in csv.DictReader(open(f)):
            disp.append(r)
    regions = []     # (name, start, end)
    for f in mfiles:
        for r in csv.DictReader(open(f)):
            name = r.get("Function", "")
            if name.startswith("roctx"):
                name = r.get("Message", name)
            try:
                regions.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            except (KeyError, ValueError):
                pass
    rows = defaultdict(list)
    if regions:
        names = sorted({n for n, _, _ in regions})
        for n in names:
            spans = sorted((s, e) for nn, s, e in regions if nn == n)
            for r in disp:
                t = int(r["Start_Timestamp"])
                # kernels are attributed by start time: launches are enqueued and completed inside the host-side range (the ranges end after a
                # stream synchronisation)
                for s, e in spans:
                    if s <= t <= e:
                        rows[(n, short(r["Kernel_Name"]))].append(r)
                        break
    else:
        by = defaultdict(list)
        for r in disp:
            by[short(r["Kernel_Name"])].append(r)
        for k, v in by.items():
            v.sort(key=lambda r: int(r["Start_Timestamp"]))
            rows[("last_%d_dispatches" % last, k)] = v[-last:]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["region", "kernel", "dispatches", "mean_us", "min_us", "max_us", "total_ms", "VGPR", "AGPR", "SGPR", "scratch_B_per_lane", "LDS_B_per_block",
                    "workgroup", "grid", "dynamic_LDS_B_per_block"])
        for (reg, k), v in sorted(rows.items(), key=lambda kv: (kv[0][0], -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))):
            du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in v]
            r0 = v[-1]
            w.writerow([reg, k, len(v), "%.2f" % (sum(du) / len(du)), "%.2f" % min(du), "%.2f" % max(du), "%.3f" % (sum(du) * 1e-3), r0.get("VGPR_Count", ""),
                        r0.get("Accum_VGPR_Count", ""), r0.get("SGPR_Count", ""), r0.get("Scratch_Size", ""), r0.get("LDS_Block_Size", ""), r0.get("Workgroup_Size_X", ""),
                        r0.get("Grid_Size_X", ""), dynamic_lds(k, r0)])
    print("wrote", out, "(%d regions x kernels, marker trace: %s)" % (len(rows), "yes" if regions else "no"))


def dynamic_lds(kernel, row):
    """rocprofv3's LDS_Block_Size is the STATIC group segment; the constitutive launch asks for its LDS dynamically (model_kernels.hip, model_lds_bytes):
    per-lane stash of ST_SLOTS = 38 doubles x block size (+ 8 x 12 doubles of slip table for the Kocks-Mecking kinds 2, 3, 6, 7; + the shape table in the
    dense tail launch / reference layout, not counted here).  Other kernels: unknown to this script (empty)."""
    import re
    m = re.match(r"k_model_setup<(\d+),", kernel)
    if not m:
        return ""
    try:
        bs = int(row.get("Workgroup_Size_X", 128))
    except ValueError:
        bs = 128
    return 38 * bs * 8 + (8 * 12 * 8 if int(m.group(1)) in (2, 3, 6, 7) else 0)


if __name__ == "__main__":
    main()
