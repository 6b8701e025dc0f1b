#!/bin/bash
# run on the GPU box: stochastic PC sampling of the constitutive launch (rocprofv3 beta feature); output under gpurun_out/$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-pcs}
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
CMD="python bench.py --model ${MODEL:-fcc_voce} --steps ${STEPS:-20} --warmup 1 --pcg-iters 10 --no-cpu-baseline"
for method in stochastic host_trap; do
  unit=cycles; intv=65536
  if [ $method = host_trap ]; then unit=time; intv=1; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $intv --kernel-trace --output-format csv -d gpurun_out/${tag}_$method -- $CMD > gpurun_out/${tag}_$method.log 2>&1
  echo "$method rc=$?"
  ls -la gpurun_out/${tag}_$method/* 2>/dev/null | head
  tail -3 gpurun_out/${tag}_$method.log
  if ls gpurun_out/${tag}_$method/*/*pc_sampling*.csv >/dev/null 2>&1; then break; fi
done
