#!/bin/bash
# run on the GPU box: Kocks-Mecking constitutive pass at 128^3 for fixed evaluation caps of the main launch and for the controller
# usage: MODEL=fcc_kmdd scripts/km_caps.sh 4 5 6 ""      ("" = controller)
MODEL=${MODEL:-fcc_kmdd}
cd $GRAFT_REPO_ROOT
for cap in "$@"; do
  env ${cap:+EXA_NEWTON_CAP=$cap} python bench.py --model $MODEL --steps 5 --warmup 1 --pcg-iters 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODEL cap=[${cap:-controller}]', 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'elastic_ms %.3f' % d['elastic_regime']['avg_kernel_ms'], 'fail', d['nonconverged_points'])"
done
