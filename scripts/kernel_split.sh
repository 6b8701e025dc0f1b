#!/bin/bash
# run on the GPU box: per-dispatch durations of the constitutive kernel (main launch and tail launch of the split) for one model
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-ksplit}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python bench.py --model ${MODEL:-bcc_kmdd} --steps 20 --warmup 2 --pcg-iters 10 --no-cpu-baseline > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/$tag/*/*kernel_trace.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "k_model_setup" in r["Kernel_Name"]]
    rows=rows[-40:]     # the plastic-regime timed passes
    by=collections.defaultdict(list)
    for r in rows: by[(int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), r["Kernel_Name"][:60])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for k,v in by.items(): print("${MODEL:-bcc_kmdd}", k, "n=%d avg %.3f ms min %.3f max %.3f" % (len(v), sum(v)/len(v), min(v), max(v)))
PY
