"""The RCCL transport of the stand-alone driver (ncclCommInitRank, ncclAllReduce, grouped ncclSend/ncclRecv; reference coupling sites
src/mechanics_driver.cpp:312, src/system_driver.cpp:167, src/mechanics_kernels.hpp:119,124) on hardware.
  * one GPU: EXA_FORCE_RCCL=1 runs every RCCL call on a one-rank communicator (the loopback tests of test_gpu_multirank.py cover the
    partition logic, this covers the library calls);
  * a torchrun-launched 2-rank run against the one-rank run: over RCCL with two or more GPUs, through the shared-device inter-process
    transport on a one-GPU box (the driver's 8-GPU scaling run uses the same launch path through bench.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(tmp_path, tag, case, nsteps, env_extra=None, nproc=1, jacobi=False):
    import orc
    out = os.path.join(str(tmp_path), tag + ".json")
    env = dict(os.environ); env.update(env_extra or {}); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = [os.path.join(HERE, "rccl_worker.py"), os.path.join(orc.REFDATA, case + ".toml"), str(nsteps), out] + (["jacobi"] if jacobi else [])
    if nproc == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", "29571"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("case,jacobi", [("voce_pa", False), ("voce_ea", True)])
def test_forced_rccl_on_one_rank(oracle, tmp_path, case, jacobi):
    n = 4
    plain = _worker(tmp_path, "plain", case, n, jacobi=jacobi)
    forced = _worker(tmp_path, "forced", case, n, {"EXA_FORCE_RCCL": "1"}, jacobi=jacobi)
    assert plain["ok"] and forced["ok"] and forced["forced"] and not plain["forced"]
    a, b = np.array(plain["avg_stress"]), np.array(forced["avg_stress"])
    # the forced run also takes the multi-rank PCG (single fused 16-byte all-reduce per iteration) instead of the one-rank loop: same answers
    # to the Krylov tolerance, same Newton counts, Krylov counts within a few iterations
    assert np.max(np.abs(a - b)) < 1e-7 * np.abs(a).max()
    assert plain["newton"] == forced["newton"]
    assert all(abs(x - y) <= max(3, 0.03 * x) for x, y in zip(plain["krylov"], forced["krylov"]))


@pytest.mark.parametrize("overlap", ["on", "off"])
def test_forced_rccl_self_exchange(oracle, tmp_path, overlap):
    """EXA_HALO_SELFTEST=1 with the forced one-rank communicator: the rank is its own neighbour across its x-max face and every halo exchange of the solve is
    a grouped ncclSend / ncclRecv of zeros of the real face size to itself - with EXA_HALO_OVERLAP=on issued on the communication stream while the interior
    element blocks run on the main stream, between the fused all-reduces of the PCG on the main stream: the two-stream use of ONE communicator that the
    overlapped halo makes of RCCL on several GPUs, executed on the hardware a one-GPU box has (reference coupling sites: src/mechanics_operator_ext.cpp:149-157
    for the exchange, src/system_driver.cpp:167 for the dot products).  Zeros added to the shared dofs: the answers are the plain run's."""
    n = 4
    plain = _worker(tmp_path, "plain", "voce_pa", n)
    st = _worker(tmp_path, "selftest", "voce_pa", n, {"EXA_FORCE_RCCL": "1", "EXA_HALO_SELFTEST": "1", "EXA_HALO_OVERLAP": overlap})
    assert plain["ok"] and st["ok"] and st["forced"] and st["transport"] == "rccl"
    assert st["comm"]["neighbours"] == 1 and st["comm"]["halo_bytes_per_exchange"] > 0 and st["comm"]["halo_overlap"] == (overlap == "on")
    a, b = np.array(plain["avg_stress"]), np.array(st["avg_stress"])
    assert np.max(np.abs(a - b)) < 1e-7 * np.abs(a).max()
    assert plain["newton"] == st["newton"]
    assert all(abs(x - y) <= max(3, 0.03 * x) for x, y in zip(plain["krylov"], st["krylov"]))


def test_two_ranks_two_processes(oracle, tmp_path):
    """torchrun --nproc-per-node 2: two processes, TCP rendez-vous of torch beside this library's communicator, per-rank device mapping.  With
    two GPUs the ranks talk over RCCL; on a one-GPU box RCCL refuses the second rank on the device, and the same launch goes through the
    shared-device inter-process transport (POSIX shared memory + hipIpcMemHandle, host/driver.hip) - every line but the RCCL calls, which
    test_forced_rccl_on_one_rank covers."""
    import torch
    n = 5
    one = _worker(tmp_path, "one", "voce_pa", n)
    two = _worker(tmp_path, "two", "voce_pa", n, nproc=2)
    assert one["ok"] and two["ok"] and two["world"] == 2 and two["comm_ranks"] == 2
    assert two["transport"] == ("rccl" if torch.cuda.device_count() >= 2 else "ipc")
    a, b = np.array(one["avg_stress"]), np.array(two["avg_stress"])
    assert np.max(np.abs(a - b)) < 1e-9 * np.abs(a).max()
    assert one["newton"] == two["newton"]


@pytest.mark.parametrize("nproc", [2, 8])
def test_bench_ranks_under_torchrun(tmp_path, nproc):
    """The driver's scaling launch line, `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, with N = 2 and N = 8 (the first
    SCALE run of the driver is N = 8: a 2 x 2 x 2 decomposition, 7 neighbours per rank): torch's process group and the library's own communicator
    side by side, max-over-ranks timing, one JSON line from rank 0 that names the transport, the rank count the transport itself reports
    (ncclCommCount over RCCL; the shared-device transport on a one-GPU box) and every rank's share of the partition."""
    import torch
    root = os.path.dirname(HERE)
    ss = 3 if nproc == 2 else 2      # (eight processes on one GPU through the host-synchronous transport: one step fewer)
    env = dict(os.environ, EXA_BENCH_N="32", EXA_BENCH_SOLVE_STEPS=str(ss), EXA_BENCH_SOLVE_STEPS_TOTAL=str(ss + 1)); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (the driver's line carries no size flags)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(29573 + nproc),
           os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == nproc and d["comm"]["ranks_reported_by_transport"] == nproc
    assert d["comm"]["transport"] == ("rccl" if torch.cuda.device_count() >= nproc else "ipc")
    assert d["nonconverged_points"] == 0 and d["value"] > 0 and d["newton_pcg_solve"]["steps"] == ss
    pr = d["comm"]["per_rank"]
    assert [p["rank"] for p in pr] == list(range(nproc)) and sum(p["elements"] for p in pr) == 32 ** 3
    assert all(p["neighbours"] == (1 if nproc == 2 else 7) and p["halo_bytes_per_exchange"] > 0 for p in pr)
    assert d["roofline"]["in_solve"]["solved_to_step"] == ss + 1 and os.path.basename(d["library"]["path"]).startswith("libexaconstit_hip")


def test_bench_order2_line():
    """`python bench.py --order 2 --bbar`: BASELINE config 5's shape in a line of the contract's form (its own workload, named as such; small RVE here)"""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--order", "2", "--bbar", "--n", "8", "--steps", "3", "--warmup", "1", "--pcg-iters", "10"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["config"]["qpts"] == 27 * 8 ** 3 and "NOT the headline" in d["config"]["workload"]
    assert d["value"] > 0 and d["nonconverged_points"] == 0 and d["pcg_iters"] == 10 and d["local_solver_evals"]["mean"] > 4
    assert 0 < d["roofline"]["frac"] < 1 and 0 < d["roofline_pcg_apply"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"
