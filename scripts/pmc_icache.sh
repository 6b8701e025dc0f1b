#!/bin/bash
# run on the GPU box: instruction-cache counters of the constitutive kernel (code size 100-170 KB against a 64 KB instruction cache per CU pair)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-pmc_icache}
CMD="python bench.py --model ${MODEL:-fcc_voce} --steps 3 --warmup 1 --pcg-iters 10 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_IFETCH --output-format csv -d gpurun_out/$tag -- $CMD > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/$tag/*/*counter_collection.csv"):
    d=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    for k in sorted(d)[-2:]:
        c=d[k]; print("${MODEL:-fcc_voce}", k, {n:"%.4g"%v for n,v in c.items()}, "miss rate %.3f" % (c.get("SQC_ICACHE_MISSES",0)/max(c.get("SQC_ICACHE_REQ",1),1)))
PY
