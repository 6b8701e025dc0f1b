#!/bin/bash
# run on the GPU box: the driver and multi-rank tests under every A/B switch of DESIGN 7b (the non-default paths must keep working)
cd $GRAFT_REPO_ROOT
for sw in EXA_QLAYOUT=aos EXA_UNFUSED_SETUP=1 EXA_APPLY_GEO=off EXA_TANGENT_FORM=full EXA_EA_ASSEMBLED=1 EXA_PCG_UNFUSED=1 EXA_NEWTON_CAP=auto EXA_NEWTON_CAP=off EXA_PCG_TWO_REDUCTIONS=1 EXA_PCG_REDUCE_LAUNCH=1 EXA_DETERMINISTIC=1 EXA_PCG_GRAPH=0 EXA_PCG_GRAPH=all EXA_TANGENT_RECORDS=off EXA_HALO_OVERLAP=off EXA_KM_PQ1=off EXA_TAIL_RESUME=off EXA_NEWTON_CAP=4,7 EXA_JAC_FIELD=on EXA_VOCE_XN_CT=off EXA_LOOPBACK_SYNC=1 EXA_NT_MIN_MB=0 EXA_NT_MIN_MB=1000000 EXA_P2_PREPASS=off EXA_GRAD_SETUP_LAZY=off; do
  echo "== $sw"
  env $sw python -m pytest tests/test_gpu_driver.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -1
done
