"""Summarise the FETCH_SIZE / WRITE_SIZE passes written by scripts/profile_round.sh into
gpurun_out/<tag>_pmc_summary.csv and gpurun_out/<tag>_pmc_traffic.json (bytes per quadrature point of the two hot kernels).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts wide coalesced reads at 1/2, so
traffic = 2 * FETCH_SIZE KB + WRITE_SIZE KB; only the steady-state half of the dispatches of a kernel is used (median)."""
import collections
import csv
import glob
import json
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows_out = []
med = {}
grid = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}_pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                grid[r["Kernel_Name"]] = float(r["Grid_Size"])
    for k, v in vals.items():
        m = statistics.median(v[len(v) // 2:])
        med[(c, k)] = m
        rows_out.append((c, k, len(v), m, m * 1024 / 1e9))
with open(f"gpurun_out/{tag}_pmc_summary.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "calls", "median_counter_value_KB(last half of calls)", "GB_uncorrected"])
    for r in sorted(rows_out, key=lambda r: -r[4]):
        w.writerow(r)
out = {}
for name, key in (("k_model_setup", "k_model_setup<0"), ("k_grad_apply_p1", "k_grad_apply_p1")):
    ks = [k for (c, k) in med if key in k and c == "FETCH_SIZE"]
    if not ks:
        continue
    k = max(ks, key=lambda k: med[("FETCH_SIZE", k)])
    qpts = grid[k] if name == "k_model_setup" else grid[k] * 8   # model: one thread per point; apply: one thread per element
    out[name] = (2 * med[("FETCH_SIZE", k)] + med.get(("WRITE_SIZE", k), 0.0)) * 1024 / qpts
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (scripts/profile_round.sh {tag}), bench.py at 128^3",
           "correction": "gfx950: traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section); the x2 over-corrects 8-byte strided loads by ~15 %",
           "bytes_per_qpt": out}, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
print(out)
