#!/bin/bash
# run on the GPU box: the judged profile set of a round -> gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (MI355X_MICROARCH.md: they do not fit in one pass)
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 10 --warmup 2 --pcg-iters 50 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- $CMD > gpurun_out/${tag}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python bench.py --steps 3 --warmup 1 --pcg-iters 20 --no-cpu-baseline > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python scripts/pmc_summary.py $tag
