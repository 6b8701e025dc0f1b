#!/bin/bash
# run on the GPU box: non-FP64 part of the VALU instruction mix of the constitutive kernel (plastic pass = last dispatch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02_valu_mix}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/$tag -- python bench.py --model ${MODEL:-fcc_voce} --steps 3 --warmup 1 --pcg-iters 10 --no-cpu-baseline > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/$tag/*/*counter_collection.csv"):
    d=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"]=float(r["Grid_Size"])
    k=max(d); c=d[k]; w=c.pop("grid")/64
    print("per wave:", {n: round(v/w,1) for n,v in c.items()})
PY
