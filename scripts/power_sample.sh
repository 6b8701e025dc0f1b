#!/bin/bash
# run on the GPU box: sample power / clocks (rocm-smi) while the constitutive launch runs back to back
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^=\|^$" | head -30
echo "---- under load (${MODEL:-fcc_voce}, EXA_LIB=${EXA_LIB:-product})"
python bench.py --model ${MODEL:-fcc_voce} --steps 3000 --warmup 5 --pcg-iters 10 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|busy" | tr '\n' ';'; echo
  sleep 1.5
done
wait $pid
python -c "import json; d=json.load(open('/tmp/b.json')); print('kernel_ms %.3f' % d['roofline']['avg_kernel_ms'])"
