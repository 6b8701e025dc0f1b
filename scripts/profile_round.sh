#!/bin/bash
# run on the GPU box: the judged profile set of a round -> gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (MI355X_MICROARCH.md: they do not fit in one pass)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps ${STEPS:-100} --warmup 5 --pcg-iters 100 --no-cpu-baseline ${BENCH_ARGS}"
# kernel + marker trace (roctx regions of the driver: timed_region_model, timed_region_pcg, ecmech_kernel, krylov_solver); no counters in this pass
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- $CMD > gpurun_out/${tag}_trace.log 2>&1
python scripts/region_summary.py gpurun_out/${tag}_trace gpurun_out/${tag}_region_summary.csv
cp $(ls gpurun_out/${tag}_trace/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
grep '^{"metric"' gpurun_out/${tag}_trace.log > gpurun_out/${tag}_bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python bench.py --steps 3 --warmup 1 --pcg-iters 20 --no-cpu-baseline > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python scripts/pmc_summary.py $tag
