"""Inputs of the 1/2/4/8-GPU prediction in DESIGN.md section 7, measured on ONE MI355X: per-rank kernel times of the strong-scaling
decomposition of the 128^3 problem (128^3, 2 x 64 x 128 x 128 ... approximated by cubes of the same element count per rank) and the
latency floor of the two RCCL calls of a PCG iteration.  Run on the GPU box: python scripts/scaling_inputs.py gpurun_out/r06_scaling_inputs.json"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exaconstit_amd.lib as L

out = {}
o = (C.c_double * 2)(); err = C.create_string_buffer(256)
for n in (12800, 3 * 65 * 65, 3 * 129 * 129):      # ~100 kB, one 64^3 face (3 dofs x 65^2 nodes), one 128^2 face
    assert L.exa_rccl_microbench(2000, n, o, err, 256) == 0, err.value
    out["rccl_n%d" % n] = {"us_allreduce_16B": o[0], "us_sendrecv_self": o[1], "bytes": 8 * n}
def bench(N, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(N), "--steps", "100", "--pcg-iters", "400", "--no-cpu-baseline", "--no-adapter-route", "--solve-steps", "0"],
                       capture_output=True, text=True, env=dict(os.environ, EXA_PCG_GRAPH="0", **env))
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert line, r.stderr[-2000:]
    return json.loads(line[-1])


for N in (128, 101, 80, 64):                           # elements per edge with 1, 2, 4, 8 ranks' share of 128^3 (cube of equal volume); kinematic state
    j = bench(N)
    out["n%d" % N] = {"model_ms": j["roofline"]["avg_kernel_ms"], "apply_ms": j["roofline_pcg_apply"]["avg_kernel_ms"], "pcg_ms_per_iter": j["pcg_ms_per_iter"]}
    # the loop several ranks run (single fused all-reduce per iteration, five launches) on a one-rank RCCL communicator: its kernels without an exchange ...
    j = bench(N, EXA_FORCE_RCCL="1")
    out["n%d" % N]["pcg_ms_per_iter_multirank_loop"] = j["pcg_ms_per_iter"]
    # ... and with the grouped send / recv of one face's halo to the own rank in every action (in line, the default over RCCL)
    j = bench(N, EXA_FORCE_RCCL="1", EXA_HALO_SELFTEST="1")
    out["n%d" % N]["pcg_ms_per_iter_multirank_loop_self_exchange"] = j["pcg_ms_per_iter"]
out["library"] = {"kernel_build_id": L.exa_kernel_build_id().decode()}
if len(sys.argv) > 1:      # (RCCL prints its version banner to stdout when the process ends: a file keeps the JSON clean)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
else:
    print(json.dumps(out, indent=1))
