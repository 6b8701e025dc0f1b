#!/bin/bash
# run on the GPU box: time the 128^3 constitutive pass of the built library under different environment switches (one bench run each)
# usage: MODEL=fcc_voce scripts/ab_env.sh "VAR=val" "" "VAR2=val" ...   (an empty string = the defaults)
MODEL=${MODEL:-fcc_voce}
STEPS=${STEPS:-20}
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python bench.py --model $MODEL --steps $STEPS --warmup 5 --pcg-iters 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODEL ENV=[$cfg]', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'elastic_ms %.3f' % d['elastic_regime']['avg_kernel_ms'], 'fail', d['nonconverged_points'], 'nfev_mean %.4f' % d['local_solver_evals']['mean'])"
done
