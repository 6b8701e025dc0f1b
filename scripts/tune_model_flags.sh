#!/bin/bash
# run on the GPU box: rebuild the constitutive translation unit with different MODEL_FLAGS (Makefile) and time the 128^3 pass
# usage: MODEL=fcc_voce scripts/tune_model_flags.sh "<MODEL_FLAGS>" ["<MODEL_FLAGS>" ...]
MODEL=${MODEL:-fcc_voce}
cd $GRAFT_REPO_ROOT/exaconstit_amd/csrc
for cfg in "$@"; do
  rm -f model_kernels.o
  make -s -j8 MODEL_FLAGS="$cfg" 2>&1 | grep -E "error"
  (cd $GRAFT_REPO_ROOT && python bench.py --model $MODEL --steps 5 --warmup 1 --pcg-iters 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODEL MODEL_FLAGS=[$cfg]', 'kernel_ms %.3f' % d['roofline']['avg_kernel_ms'], 'elastic_ms %.3f' % d['elastic_regime']['avg_kernel_ms'], 'fail', d['nonconverged_points'])")
done
rm -f model_kernels.o; make -s -j8 2>&1 | grep -E "error"
