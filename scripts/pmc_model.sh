#!/bin/bash
# run on the GPU box: SQ counter pass over the model kernel (last dispatch = plastic-regime pass of bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --model ${MODEL:-fcc_voce} --steps 3 --warmup 1 --pcg-iters 10 --no-cpu-baseline"
tag=${1:-pmc_model}
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_FLAT --output-format csv -d gpurun_out/$tag -- $CMD > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/$tag/*/*counter_collection.csv"):
    d=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"]=float(r["Grid_Size"])
    gmax=max(v["grid"] for v in d.values()); full=sorted(i for i,v in d.items() if v["grid"]==gmax)
    # a capped launch and its tail launch have the same kernel name and grid (Kocks-Mecking): of the last two full-size dispatches the one with
    # more VALU work is the main launch, the other one the tail launch; without a tail split both are plastic-regime passes of the same launch
    last2=full[-2:]; main=max(last2, key=lambda i: d[i]["SQ_INSTS_VALU"]); rest=[i for i in last2 if i!=main]
    for label,k in [("main launch",main)]+[("tail launch (or previous pass)",i) for i in rest]:
        c=d[k]; waves=c["grid"]/64
        print(label+": per-wave VALU %.0f SALU %.0f FLAT %.1f | wave_cycles %.0f  valu_active %.1f%%  wait_any %.1f%%  wait_inst %.1f%%" % (c["SQ_INSTS_VALU"]/waves, c["SQ_INSTS_SALU"]/waves, c["SQ_INSTS_FLAT"]/waves, 4*c["SQ_WAVE_CYCLES"]/waves, 100*c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]*2, 100*c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], 100*c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]))
PY
