#!/bin/bash
# run on the GPU box: FP64 operation counts of the constitutive kernel (plastic-regime pass = last dispatch of bench.py)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/${tag}_pmc_flops -- python bench.py --steps 3 --warmup 1 --pcg-iters 10 --no-cpu-baseline > gpurun_out/${tag}_pmc_flops.log 2>&1
python - <<PY
import csv, collections, glob, json
d=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/${tag}_pmc_flops/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["qpts"]=float(r["Grid_Size"])
def summ(c):
    q=c["qpts"]; fl=64*(2*c["SQ_INSTS_VALU_FMA_F64"]+c["SQ_INSTS_VALU_ADD_F64"]+c["SQ_INSTS_VALU_MUL_F64"]+c["SQ_INSTS_VALU_TRANS_F64"])/q
    return {"flop_per_qpt": fl, "valu_insts_per_wave": c["SQ_INSTS_VALU"]/(q/64), "salu_insts_per_wave": c["SQ_INSTS_SALU"]/(q/64),
            "fp64_insts_per_wave": (c["SQ_INSTS_VALU_FMA_F64"]+c["SQ_INSTS_VALU_ADD_F64"]+c["SQ_INSTS_VALU_MUL_F64"]+c["SQ_INSTS_VALU_TRANS_F64"])/(q/64)}
ks=sorted(d)
out={"source":"rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 SQ_INSTS_VALU SQ_INSTS_SALU (scripts/pmc_flops.sh), bench.py at 128^3; flop = 64 lanes x (2 FMA + ADD + MUL + TRANS), exec mask ignored",
     "k_model_setup": {"plastic": summ(d[ks[-1]]), "elastic": summ(d[ks[0]])}}
json.dump(out, open("gpurun_out/${tag}_pmc_flops.json","w"), indent=1); print(out)
PY
