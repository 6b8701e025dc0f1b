#!/bin/bash
# run on the GPU box: instruction-cache counters of the constitutive kernel (kinematic state)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export EXA_BENCH_SOLVE_STEPS=0
CMD="python bench.py --model ${MODEL:-fcc_voce} --steps 3 --warmup 1 --pcg-iters 10 --no-cpu-baseline"
tag=${1:-pmc_icache}
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_IFETCH --output-format csv -d gpurun_out/$tag -- $CMD > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/$tag/*/*counter_collection.csv"):
    d=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"]=float(r["Grid_Size"])
    k=max(d); c=d[k]; waves=c["grid"]/64
    print({n: (v/waves if n!="grid" else v) for n,v in c.items()})
PY
tail -3 gpurun_out/$tag.log
