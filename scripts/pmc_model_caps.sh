#!/bin/bash
# run on the GPU box: VALU instructions per wave of the constitutive launch as a function of the evaluation cap
# (cap K: the launch stops a point after K evaluations) -> cost per evaluation and fixed cost of a point update
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in ${CAPS:-1 2 3 4 off}; do
  EXA_NEWTON_CAP=$cap rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d gpurun_out/cap_$cap -- python bench.py --model ${MODEL:-fcc_voce} --steps 2 --warmup 1 --pcg-iters 2 --no-cpu-baseline > gpurun_out/cap_$cap.log 2>&1
  python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/cap_$cap/*/*counter_collection.csv"):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"] = float(r["Grid_Size"])
    ks = sorted(d)[-4:]
    print("cap $cap:", [(int(d[k]["grid"]), round(d[k]["SQ_INSTS_VALU"] / (d[k]["grid"] / 64))) for k in ks])
PY
  rm -rf gpurun_out/cap_$cap
done
