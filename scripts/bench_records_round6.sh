#!/bin/bash
# run on the GPU box: the bench records of the round-6 evidence set -> gpurun_out/r06_*.json.  bench.py fills roofline.traffic from profiles/r06_pmc_traffic.json when that file
# carries the loaded library's kernel_build_id (scripts/profile_round6.sh puts it there before it calls this script), so these records are taken AFTER the counter passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r06
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${T}_bench_driver_form.err | grep '^{"metric"' > gpurun_out/${T}_bench_driver_form.json
python bench.py --model bcc_kmdd --steps 50 --warmup 5 --no-cpu-baseline --no-adapter-route 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_n128_bcc_kmdd.json
python bench.py --model fcc_kmdd --steps 30 --warmup 5 --no-cpu-baseline --no-adapter-route --solve-steps-total 14 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_n128_fcc_kmdd.json
python bench.py --n 64 --steps 100 --warmup 5 --no-cpu-baseline --pcg-iters 400 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_n64.json
python bench.py --n 64 --jacobi --steps 100 --warmup 5 --no-cpu-baseline --no-adapter-route --pcg-iters 400 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_n64_jacobi.json
python bench.py --jacobi --steps 50 --warmup 5 --no-cpu-baseline --no-adapter-route --solve-steps-total 14 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_n128_jacobi.json
python bench.py --order 2 --bbar --steps 50 --warmup 5 --pcg-iters 200 2>/dev/null | grep '^{"metric"' > gpurun_out/${T}_bench_order2_bbar_n64.json
python scripts/bench_config5.py 64 2>/dev/null | grep '^{' > gpurun_out/${T}_config5_rates.json
python scripts/adapter_route.py --model bcc_kmdd --steps 10 --iters 10 2>/dev/null | grep '^{' > gpurun_out/${T}_adapter_route_bcc_kmdd.json
