#!/bin/bash
# run on the GPU box: SQ counter pass + FETCH/WRITE passes + kernel trace over one named kernel of an arbitrary command
#   usage: scripts/pmc_kernel.sh <tag> <kernel-substring> <command...>
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- "$@" > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_sq -- "$@" > gpurun_out/${tag}_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- "$@" > gpurun_out/${tag}_$c.log 2>&1
done
python - <<PY
import csv, collections, glob, statistics
kern = "$kern"
for f in glob.glob("gpurun_out/${tag}_trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]: print("trace:", r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
for f in glob.glob("gpurun_out/${tag}_sq/*/*counter_collection.csv"):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"] = float(r["Grid_Size"])
    if d:
        k = sorted(d)[len(d) // 2]; c = d[k]; waves = c["grid"] / 64
        print("per-wave: VALU %.0f SALU %.0f LDS %.0f | wave_cycles %.0f  valu_active %.1f%%  wait_any %.1f%%  wait_inst %.1f%%" % (c["SQ_INSTS_VALU"]/waves, c["SQ_INSTS_SALU"]/waves, c.get("SQ_INSTS_LDS", 0)/waves, 4*c["SQ_WAVE_CYCLES"]/waves, 100*c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]*2, 100*c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], 100*c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]))
tot = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []; g = 1
    for f in glob.glob("gpurun_out/${tag}_%s/*/*counter_collection.csv" % cn):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == cn: v.append(float(r["Counter_Value"])); g = float(r["Grid_Size"])
    if v: tot[cn] = statistics.median(v); tot["grid"] = g
if "FETCH_SIZE" in tot:
    b = (2 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0)) * 1024
    print("traffic: %.3f GB per launch (2*FETCH+WRITE), %.0f B per thread" % (b / 1e9, b / tot["grid"]))
PY
rm -rf gpurun_out/${tag}_sq gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE; rm -f gpurun_out/${tag}_trace/*/*kernel_trace.csv
