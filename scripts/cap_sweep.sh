#!/bin/bash
# run on the GPU box: constitutive launch time of one model at 128^3 for a list of EXA_NEWTON_CAP settings ("auto", "K", "K,K2")
#   usage: MODEL=fcc_kmdd SOLVE_STEPS=0 scripts/cap_sweep.sh auto 5 6 7 6,10
MODEL=${MODEL:-fcc_kmdd}
cd $GRAFT_REPO_ROOT
for cap in "$@"; do
  if [ "$cap" = default ]; then unset EXA_NEWTON_CAP; else export EXA_NEWTON_CAP=$cap; fi
  python bench.py --model $MODEL --steps ${STEPS:-20} --warmup 5 --pcg-iters 10 --no-cpu-baseline --solve-steps ${SOLVE_STEPS:-0} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODEL cap=$cap', 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'fail', d['nonconverged_points'], 'nfev_mean %.4f' % d['local_solver_evals']['mean'])"
done
