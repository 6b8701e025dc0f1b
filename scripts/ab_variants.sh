#!/bin/bash
# run on the GPU box: time the 128^3 constitutive pass (kinematic state: --solve-steps 0) with timing variants of the library (built locally
# by `make variant TAG=...`); usage: MODEL=fcc_voce scripts/ab_variants.sh base bs64 ...     ("main" = the product library)
MODEL=${MODEL:-fcc_voce}
STEPS=${STEPS:-20}
cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  lib=$GRAFT_REPO_ROOT/exaconstit_amd/variants/libexaconstit_hip_$tag.so
  [ $tag = main ] && lib=$GRAFT_REPO_ROOT/exaconstit_amd/libexaconstit_hip.so
  EXA_LIB=$lib python bench.py --model $MODEL --steps $STEPS --warmup 5 --pcg-iters 10 --no-cpu-baseline --solve-steps ${SOLVE_STEPS:-0} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODEL variant=$tag', 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'elastic_ms %.3f' % d['elastic_regime']['avg_kernel_ms'], 'fail', d['nonconverged_points'], 'nfev_mean %.4f' % d['local_solver_evals']['mean'])"
done
