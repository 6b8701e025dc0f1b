#!/bin/bash
# run on the GPU box: rebuild the integrator kernels with different tuning macros and time the 128^3 PCG loop / gradient action
cd $GRAFT_REPO_ROOT/exaconstit_amd/csrc
for cfg in "$@"; do
  rm -f pa_kernels.o
  make -s -j8 TUNE="$cfg" 2>&1 | grep -E "error"
  (cd $GRAFT_REPO_ROOT && python bench.py ${BENCH_ARGS} --steps 5 --warmup 1 --pcg-iters 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TUNE=[$cfg]', 'pcg ms/iter %.4f' % d['pcg_ms_per_iter'], 'apply ms %.4f' % d['roofline_pcg_apply']['avg_kernel_ms'], 'it/s %.1f' % d['pcg_iters_per_s'])")
done
rm -f pa_kernels.o
