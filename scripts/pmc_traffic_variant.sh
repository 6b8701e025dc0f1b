#!/bin/bash
# run on the GPU box: rebuild the model kernel with TUNE flags and print its HBM traffic per point (raw FETCH_SIZE / WRITE_SIZE)
cd $GRAFT_REPO_ROOT/exaconstit_amd/csrc
for cfg in "$@"; do
  rm -f model_kernels.o; make -s -j8 TUNE="$cfg" 2>&1 | grep -E "error"
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/var_pmc_$c; rocprofv3 --pmc $c --output-format csv -d gpurun_out/var_pmc_$c -- python bench.py --steps 3 --warmup 1 --pcg-iters 4 --no-cpu-baseline > /dev/null 2>&1; done
  python - <<PY
import csv, glob
out=[]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/var_pmc_{c}/*/*counter_collection.csv"):
        v=[(int(r["Dispatch_Id"]), float(r["Counter_Value"])*1024/16777216) for r in csv.DictReader(open(f)) if "k_model_setup" in r["Kernel_Name"] and r["Counter_Name"]==c]
        v=[round(x[1]) for x in sorted(v)]; out.append((c, v[0], v[-1]))
print("TUNE=[$cfg]", out)
PY
  cd $GRAFT_REPO_ROOT/exaconstit_amd/csrc
done
rm -f model_kernels.o
