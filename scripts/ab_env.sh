#!/bin/bash
# run on the GPU box: the default bench (real-solve state) under different environment switches, one run each
# usage: MODEL=fcc_voce scripts/ab_env.sh "VAR=val" "" "VAR2=val" ...   (an empty string = the defaults)
MODEL=${MODEL:-fcc_voce}
STEPS=${STEPS:-20}
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python bench.py --model $MODEL --steps $STEPS --warmup 5 --pcg-iters 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('newton_pcg_solve') or {}; sp=s.get('steady_plastic') or {}
print('$MODEL ENV=[$cfg]', 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'kin_ms %.3f' % d['kinematic_state']['avg_kernel_ms'], 'elastic_ms %.3f' % d['elastic_regime']['avg_kernel_ms'], 'fail', d['nonconverged_points'], 'nfev_mean %.4f max %d' % (d['local_solver_evals']['mean'], d['local_solver_evals']['max']), 'steady_ms %.3f' % sp.get('kernel_ms_per_call', 0), 'solve_wall %.1f' % s.get('wall_s', 0))"
done
