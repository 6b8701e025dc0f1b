#!/usr/bin/env python
"""bench.py — headline benchmark of the ExaConstit hot path on MI355X.

Metric (BASELINE.json): "quadrature-point constitutive updates/s + Newton-PCG iter/s, 128^3 hex RVE".
  * `value`            = quadrature-point constitutive updates/s.  One STEP is one full constitutive pass of the residual evaluation
                         (L->E restriction, geometric factors, fused velocity-gradient + ExaCMech update + tangent kernel) over ALL
                         quadrature points of the 128^3 FCC-Voce RVE, restarting from the same begin-of-step state exactly as every
                         residual evaluation of a Newton solve does.  The state is the one a REAL Newton/PCG solve of the
                         reference schedule reaches in step --solve-steps (default 14: plastic regime); the passes repeat
                         that step's converged (last) residual evaluation.  --solve-steps 0 times the kinematically driven state of the
                         earlier rounds instead (reported beside it as `kinematic_state` in the default run).
  * `pcg_iters_per_s`  = partial-assembly PCG iterations/s on the same RVE (second timed region of the same run).
Inputs are synthetic (seeded orientations, reference test properties) and resident in HBM before the timed regions.
Multi-GPU: one process per GPU (torchrun), block domain decomposition of the SAME 128^3 problem (strong scaling, BASELINE config 4),
RCCL halo-sum + dot all-reduce on the PCG path; the constitutive pass needs no communication.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

# the host driver of this pool only supports dmabuf IPC: without this RCCL fails across processes (hipIpcGetMemHandle); must be set
# before the HIP runtime comes up
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)
FP64_VEC_PEAK_TFLOPS = 78.6       # vendor FP64 vector peak, for the informational compute fraction only
MODEL_BYTES_PER_QPT = 928.0       # read v 3 + J 9 + state 28 + sigma 6, write state 28 + sigma 6 + tangent 36 doubles
APPLY_BYTES_PER_QPT = 408.0       # tangent 36 + Jacobian 9 + x 3 + y 3 doubles
APPLY_MOVED_BYTES_PER_QPT = 328.0 # what the geometry-recomputing kernel has to move: tangent 36 doubles + L-vector x/coords/y (~5 doubles)
APPLY_MOVED_BYTES_COMPACT = 248.0 # ... with the tangent in its deviatoric-block + bulk form (26 doubles)
PCG_VEC_BYTES_PER_DOF = 128.0
MODEL_NAMES = {"fcc_voce": "FCC Voce power-law", "bcc_voce": "BCC Voce power-law", "fcc_voce_nl": "FCC non-linear Voce",
               "fcc_kmdd": "FCC Kocks-Mecking dislocation density", "bcc_kmdd": "BCC Kocks-Mecking dislocation density"}
SETTLE_PASSES = 60                # untimed constitutive passes in front of a timed region that follows a COLD prepare phase (kinematic state, --solve-steps 0), see main()
SOLVE_STEPS_DEFAULT = 14          # real Newton/PCG steps of the reference schedule before the timed passes (plastic regime: steps >= 10)
SOLVE_STEPS_TOTAL_DEFAULT = 25    # ... and the solve goes on to this step after the timed regions: per-step in-solve kernel rates up to the plateau of the
                                  # dt = 0.1 segment (steps 3..21 of test/data/custom_dt.txt) and into the dt = 0.2 segment (22..27)
PREP_DTS = [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]   # first 10 steps of the reference schedule: 0.1 % strain, plastic


def host_cores():
    """Threads this process may actually use: affinity mask, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_config1(props, gpu_step=None):
    """BASELINE config 1 on the host: 16^3 auto-generated hex RVE, FCC Voce power law, partial-assembly PCG, ONE load step (dt = 0.005 of
    test/data/custom_dt.txt) through the oracle's driver - the restatement of mechanics_driver's step loop with the reference's timers: step
    wall around the solve (src/mechanics_driver.cpp:865-892), the constitutive region (src/mechanics_ecmech.cpp:237-257) and krylov_solver
    (src/mechanics_solver.cpp:99-103).  One thread (rtmodel=CPU) and all host cores (element loops under OpenMP: stand-in for mpirun -np cores)."""
    import orc
    N = 16
    case = orc.load_case("voce_pa.toml")
    rng = np.random.default_rng(20240928)
    quats = rng.standard_normal((N ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    case.update(nx=N, ny=N, nz=N, length=[1.0, 1.0, 1.0], elem_grain=np.arange(N ** 3, dtype=np.int32), quats=quats,
                dts=np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:1])
    cores = host_cores()
    legs = {}
    for label, nt in (("one_thread", 1), ("all_cores", cores)):
        orc.lib().orc_set_threads(nt)
        t0 = time.perf_counter()
        r = orc.run_case(case)
        wall = time.perf_counter() - t0
        kit = int(r["krylov_iters"].sum())
        legs[label] = {"threads": nt, "step_wall_s": r["t_total"], "t_model_s": r["t_model"], "t_krylov_s": r["t_krylov"], "newton_iters": int(r["newton_iters"].sum()),
                       "pcg_iters": kit, "pcg_iters_per_s": kit / r["t_krylov"] if r["t_krylov"] > 0 else None, "model_calls": int(r["model_calls"].sum()),
                       "qpt_updates_per_s_in_model": r["qpt_updates"] / r["t_model"] if r["t_model"] > 0 else None, "wall_incl_setup_s": wall,
                       "avg_stress_zz": float(r["avg_stress"][-1, 2])}
    orc.lib().orc_set_threads(1)
    out = {"workload": "BASELINE config 1: 16^3 hex RVE p=1 (4 096 elements, 32 768 qpts, 14 739 dofs), FCC Voce, partial-assembly PCG (rel 1e-7, 1000 it), NR, "
                       "1 load step dt = 0.005 (elastic), uniaxial z-velocity BCs of test/data/voce_pa.toml",
           "timers": "step wall = around the step's solve; t_model = constitutive region summed over ModelSetup calls; t_krylov = inside the PCG solves",
           **legs}
    if gpu_step is not None:
        out["gpu_same_step"] = gpu_step
    return out


def gpu_config1_step(L, props):
    """the same 16^3 load step through the HIP driver (one GPU): wall of the step, constitutive and krylov region times, PCG rate"""
    N = 16
    rng = np.random.default_rng(20240928)
    quats = rng.standard_normal((N ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:1]
    best = None
    for rep in range(3):      # the first repetition pays module load / allocation; report the fastest of the three
        d = L.Driver.synthetic(N, props, quats.ravel(), sched)
        d.reset_timers()
        import torch
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok = d.step(1)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        tm = d.timers(); nw, kr, mc = d.stats()
        row = {"step_wall_s": wall, "t_model_s": tm["model_ms"] * 1e-3, "t_krylov_s": tm["krylov_ms"] * 1e-3, "newton_iters": int(nw[-1]), "pcg_iters": int(kr[-1]),
               "pcg_iters_per_s": int(kr[-1]) / (tm["krylov_ms"] * 1e-3) if tm["krylov_ms"] > 0 else None, "model_calls": int(mc[-1]),
               "qpt_updates_per_s_in_model": 8 * N ** 3 * int(mc[-1]) / (tm["model_ms"] * 1e-3) if tm["model_ms"] > 0 else None,
               "avg_stress_zz": float(d.avgs(0, 6)[-1, 2]), "converged": bool(ok)}
        d.close()
        if best is None or row["step_wall_s"] < best["step_wall_s"]:
            best = row
    best["note"] = "a 16^3 problem is 16 waves of elements: launch latency bound on one MI355X (the headline configuration is 128^3)"
    return best


def cpu_baseline(props, seconds_target=12.0, gpu_step=None):
    """Oracle (CPU restatement of the reference's serial loops) timed on rank 0's host: one thread, bounded sample."""
    import hipref
    import orc
    import tempfile
    orc.build()
    native_dir = tempfile.mkdtemp(prefix="exa_oracle_native_")
    orc.use_lib(orc.build_native(native_dir))      # this leg times the oracle built for this host: g++ -O3 -march=native -fopenmp
    N = 20
    rve = hipref.make_rve(orc, N)
    P = rve["E"] * rve["Q"]
    quats = hipref.random_quats(rve["E"])
    hist = np.zeros(26)
    orc.lib().orc_hist_init(0, 0, orc._p(props), len(props), orc._p(hist))
    sv0 = np.tile(np.concatenate([hist, [1.0, 0.0]]), P).reshape(P, 28)
    sv0[:, 9:13] = np.repeat(quats, rve["Q"], axis=0)
    sv0 = sv0.ravel().copy()
    s0 = np.zeros(6 * P)
    v = hipref.velocity_field(rve)
    ve = hipref.l_to_e(rve, v)
    x = rve["X"].copy()
    s1 = np.zeros(6 * P); sv1 = np.zeros(28 * P); cm = np.zeros(36 * P)

    def one_pass(dt, J):
        return orc.lib().orc_model_setup(0, 0, orc._p(props), len(props), rve["Q"], rve["E"], rve["n"], 28, C.c_double(dt), C.c_double(298.0),
                                         orc._p(J), orc._p(rve["G"]), orc._p(ve), orc._p(s0), orc._p(sv0), orc._p(s1), orc._p(sv1), orc._p(cm), None, 1, 0, 0)
    J = np.zeros(9 * P)
    orc.lib().orc_set_threads(host_cores())     # preparation passes are not timed
    for dt in PREP_DTS:
        x = x + v * dt
        orc.lib().orc_jacobians(1, rve["E"], orc._p(hipref.l_to_e(rve, x)), orc._p(J))
        one_pass(dt, J)
        s0[:] = s1; sv0[:] = sv1
    def timed(budget):
        t0 = time.perf_counter(); n = 0
        while True:
            one_pass(PREP_DTS[-1], J); n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= 200:
                return n, el
    orc.lib().orc_set_threads(1)
    n1, el1 = timed(0.4 * seconds_target)
    cores = host_cores()
    orc.lib().orc_set_threads(cores)
    nc, elc = timed(0.4 * seconds_target)
    orc.lib().orc_set_threads(1)
    c1 = cpu_baseline_config1(props, gpu_step)
    return {"value": P * nc / elc, "unit": "qpt-updates/s", "cores": cores, "cpu": cpu_model(), "kind": "port", "single_thread_value": P * n1 / el1,
            "pcg_iters_per_s": c1["all_cores"]["pcg_iters_per_s"], "step_wall_s": c1["all_cores"]["step_wall_s"],
            "pcg_iters_per_s_one_thread": c1["one_thread"]["pcg_iters_per_s"], "step_wall_s_one_thread": c1["one_thread"]["step_wall_s"],
            "config1_load_step": c1,
            "what": "oracle/ = this repo's own C++ restatement of the ExaCMech update and of the reference's driver loop, built for this host with "
                    "g++ -O3 -march=native -fopenmp for this leg; NOT the ExaCMech / MFEM libraries (absent from the image)",
            "sample": f"{N}^3-element FCC-Voce RVE ({P} qpts), same kinematic drive to the plastic regime; constitutive passes of the oracle "
                      f"(element/qpt loops of the reference's CPU path): {n1} serial passes in {el1:.1f} s (rtmodel=CPU analogue) and {nc} passes "
                      f"in {elc:.1f} s with an OpenMP loop over all {cores} host threads (rtmodel=OPENMP analogue; `value`)"}


def bench_order2(args, L, N, props, quats, mk, rank, world, fresh_uid, barrier, max_over_ranks):
    """--order 2 [--bbar]: BASELINE config 5's shape (64^3 triquadratic hexahedra, element assembly, B-bar) in a line of the contract's form.  A step is one
    constitutive pass over the RVE (geometry pre-pass + the fused p = 2 launch, which writes the records of the matrix-free action) on the kinematically driven
    plastic state; the PCG block times fixed-length CG on the same state (action computed from the point records, no 81 x 81 matrices)."""
    drv = L.Driver.synthetic(N, props, quats.ravel(), np.array(PREP_DTS), assembly=1, order=2, bbar=args.bbar, nrls=True, krylov=(1000, 1e-7, 1e-27),
                             rank=rank, nranks=world, uid=fresh_uid(), **mk)
    P_global = 27 * N ** 3
    drv.bench_prepare(PREP_DTS[:1], advance=False); drv.bench_model(2); me = drv.bench_model(max(1, args.steps // 2))
    el_ms = max_over_ranks(me["kernel_ms"]) / max(1, args.steps // 2)
    drv.bench_prepare(PREP_DTS)
    P_local = L.exa_driver_local_qpts(drv.h)
    drv.bench_model(max(args.warmup, 1))
    barrier(); t0 = time.perf_counter()
    m = drv.bench_model(args.steps)
    barrier(); t_model = max_over_ranks(time.perf_counter() - t0)
    kern_ms = max_over_ranks(m["kernel_ms"]) / args.steps
    nfev = drv.nfev_hist()
    drv.bench_pcg(10)
    barrier(); t0 = time.perf_counter()
    pc = drv.bench_pcg(args.pcg_iters)
    barrier(); t_pcg = max_over_ranks(time.perf_counter() - t0)
    it_ms = max_over_ranks(pc["pcg_ms"]) / max(pc["iters"], 1); ap_ms = max_over_ranks(pc["apply_ms"]) / max(pc["iters"], 1)
    if rank == 0:
        kid = L.exa_kernel_build_id().decode(); traffic = {}; tsrc = None
        for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", f)))
            except Exception:
                continue
            if d.get("kernel_build_id") == kid and "config5_bytes_per_qpt" in d:
                traffic = d["config5_bytes_per_qpt"]; tsrc = "profiles/" + f
                break
        tr_model = (traffic["k_model_setup_p2_vg_records"] + traffic["k_geom_p2"]) * P_local if args.bbar and "k_geom_p2" in traffic else None
        gbs = MODEL_BYTES_PER_QPT * P_local / (kern_ms * 1e-3) / 1e9
        out = {"metric": "quadrature-point constitutive updates/s + Newton-PCG iter/s, 128^3 hex RVE",
               "value": P_global * args.steps / t_model, "unit": "qpt-updates/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "warmup_requested": args.warmup,
               "ms_per_step": t_model / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"NOT the headline: BASELINE config 5's shape - {N}^3 hex RVE p=2{', B-bar' if args.bbar else ''}, element assembly served matrix-free, "
                                      f"{MODEL_NAMES[args.model]}, kinematically driven plastic state", "elements": N ** 3, "qpts": P_global, "decomposition": f"{world} block(s)"},
               "library": {"path": L.LIB_PATH, "build_id": L.exa_build_id().decode(), "kernel_build_id": kid},
               "pcg_iters_per_s": 1e3 / it_ms, "pcg_iters": pc["iters"], "pcg_ms_per_iter": it_ms, "pcg_wall_s": t_pcg, "nonconverged_points": m["failed"],
               "local_solver_evals": {"mean": float((nfev * np.arange(64)).sum() / max(nfev.sum(), 1)), "max": int(np.nonzero(nfev)[0].max()) if nfev.any() else 0},
               "elastic_regime": {"avg_kernel_ms": el_ms, "value": P_local * world / (el_ms * 1e-3), "unit": "qpt-updates/s"},
               "roofline": {"kernel": "k_geom_p2 + k_model_setup<.., 27, element-blocked, records> (geometry pre-pass over the elements, then the fused ExaCMech update that writes the "
                                      "18-pair records of the matrix-free action)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": tr_model, "traffic_source": tsrc, "bytes_per_qpt": MODEL_BYTES_PER_QPT, "avg_kernel_ms": kern_ms,
                            "note": "HIP events around the two launches of a pass; frac on SURVEY 8(d)'s 928 B/qpt; traffic = L2-boundary bytes of both kernels from the PMC passes of "
                                    "this kernel build (profiles/*_pmc_traffic.json, config5_bytes_per_qpt), null otherwise"},
               "roofline_pcg_apply": {"kernel": "k_mf_apply_p2<BBAR, TRANS, compact records> (action of the element matrices computed from the point records)", "bound": "hbm",
                                      "bytes_per_qpt": 288.0, "avg_kernel_ms": ap_ms, "achieved": 288.0 * P_local / (ap_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": 288.0 * P_local / (ap_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "traffic": traffic["k_mf_apply_p2"] * P_local if "k_mf_apply_p2" in traffic else None,
                                      "note": "288 B/qpt = the 18 16-byte pairs of a point record; the kernel also reads the element-average gradients (B-bar) and gathers / scatters "
                                              "the 81 element dofs (PMC: 385 B/qpt)"},
               "cpu_baseline": None, "cpu_baseline_note": "reported with the headline workload only (python bench.py)"}
        print(json.dumps(out))
    drv.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed constitutive passes (default 300: a timed region of about 2 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=int(os.environ["EXA_BENCH_N"]) if "EXA_BENCH_N" in os.environ else None, help="elements per edge of the RVE (default 128; 64 with --order 2)")
    ap.add_argument("--order", type=int, default=1, choices=[1, 2], help="2: the shape of BASELINE config 5 (triquadratic elements, element assembly served matrix-free; with --bbar "
                    "the B-bar integrator) in a line of the same form - its own workload, named in config.workload, never the headline")
    ap.add_argument("--bbar", action="store_true", help="B-bar integrator (with --order 2)")
    ap.add_argument("--pcg-iters", type=int, default=100)
    ap.add_argument("--assembly", default="PA")
    ap.add_argument("--model", default="fcc_voce", choices=["fcc_voce", "bcc_voce", "fcc_voce_nl", "fcc_kmdd", "bcc_kmdd"],
                    help="crystal model of the RVE; the headline metric is quoted on fcc_voce (BASELINE config 4 also names bcc_kmdd)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adapter-route", action="store_true", help="skip the `adapter_route` block (the calls the MFEM adapters make, timed at the same state)")
    ap.add_argument("--jacobi", action="store_true", help="true Jacobi preconditioner (refreshed every Newton iteration) instead of the reference's effective identity (SURVEY fact 9)")
    ap.add_argument("--solve-steps", type=int, default=int(os.environ.get("EXA_BENCH_SOLVE_STEPS", str(SOLVE_STEPS_DEFAULT))),
                    help="real Newton/PCG time steps of the reference schedule that bring the RVE to the benchmark state (0: kinematically driven state only)")
    ap.add_argument("--solve-steps-total", type=int, default=int(os.environ.get("EXA_BENCH_SOLVE_STEPS_TOTAL", str(SOLVE_STEPS_TOTAL_DEFAULT))),
                    help="after the timed regions the solve continues (committing --solve-steps) up to this step of the schedule, for the in-solve rates")
    args = ap.parse_args()
    if args.n is None:
        args.n = 128 if args.order == 1 else 64

    import torch
    import exaconstit_amd.lib as L
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (a launcher that shows every rank one device numbers it 0)
    torch.cuda.set_device(local)
    shared = False
    if world > 1:
        # torch's group only carries the barrier, the max-over-ranks of the wall time and the 128-byte id: host-side data, so it runs on gloo
        # and cannot interfere with the library's own RCCL communicator (which carries every collective of the solve).  The ranks compare
        # the identity of their devices: one rank per physical GPU -> RCCL over xGMI; several ranks on one GPU (a one-GPU box running
        # `torchrun --nproc-per-node 2 bench.py --gpus 2`) -> the library's shared-device inter-process transport, since RCCL refuses two
        # ranks on one device (plumbing check of the launch path, not a performance configuration; the JSON line names the transport)
        import socket
        import torch.distributed as dist
        dist.init_process_group("gloo")
        pci = C.create_string_buffer(64)
        ident = (socket.gethostname(), pci.value.decode() if L.exa_device_identity(pci, 64) == 0 else f"device-index-{local}-of-{ndev}")
        ids = [None] * world
        dist.all_gather_object(ids, ident)
        shared = len(set(ids)) < world or bool(os.environ.get("EXA_BENCH_SAME_DEVICE"))
        os.environ["EXA_TRANSPORT"] = "ipc" if shared else "rccl"

    def fresh_uid():
        """A NEW 128-byte id from rank 0 for every communicator: a RCCL unique id serves exactly one ncclCommInitRank per rank (its bootstrap root
        answers one clique and goes away), and this run creates two drivers one after the other."""
        if world == 1:
            return None
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            assert L.exa_comm_unique_id(buf, world) == 0
        t = torch.tensor(list(buf), dtype=torch.uint8)
        dist.broadcast(t, 0)
        return (C.c_ubyte * 128)(*t.tolist())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    N = args.n
    xt, sl = args.model.split("_", 1)
    pfile = {"voce": "props_cp_voce.txt", "voce_nl": "props_cp_vocenl.txt", "kmdd": "props_cp_mts.txt"}[sl]
    mk = dict(bcc=(xt == "bcc"), slip={"voce": 0, "voce_nl": 1, "kmdd": 2}[sl], temp_k=298.0)
    props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", pfile)).ravel()
    rng = np.random.default_rng(20240928)
    quats = rng.standard_normal((N ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    asm = 0 if args.assembly.upper() == "PA" else 1
    if args.order == 2:
        bench_order2(args, L, N, props, quats, mk, rank, world, fresh_uid, barrier, max_over_ranks)
        if world > 1:
            dist.destroy_process_group()
        return
    drv = L.Driver.synthetic(N, props, quats.ravel(), np.array(PREP_DTS), assembly=asm,
                             krylov=(1000, 1e-7, 1e-27), rank=rank, nranks=world, uid=fresh_uid(), jacobi=args.jacobi, **mk)
    del quats
    P_global = 8 * N ** 3
    # Untimed passes in front of a timed region.  The kinematic state is timed right after a cold prepare phase, where the first dozens of launches read
    # ~4 % slow: SETTLE_PASSES there.  The headline region follows a 14-step real solve (24 s of GPU work): measured in round 5, --warmup 5 and 60 settle
    # passes give the same value (3.771e9 | 3.759e9 qpt/s in one call), so it runs exactly the --warmup passes the contract names.
    settle_cold = max(args.warmup, int(os.environ.get("EXA_BENCH_SETTLE", str(SETTLE_PASSES))))
    settle = settle_cold if args.solve_steps <= 0 else max(args.warmup, int(os.environ.get("EXA_BENCH_SETTLE", "0")))

    def hist_dict(h):
        return {"mean": float((h * np.arange(64)).sum() / max(h.sum(), 1)), "max": int(np.nonzero(h)[0].max()) if h.any() else 0,
                "hist": {str(i): int(c) for i, c in enumerate(h) if c}}

    def timed_passes(d, steps, nsettle):
        """untimed settle passes, then `steps` timed passes bracketed by barrier + synchronize; wall s (max over ranks), kernel ms per pass, failed points, histogram"""
        if nsettle > 0:
            d.bench_model(nsettle)
        barrier(); t0 = time.perf_counter()
        mm = d.bench_model(steps)
        barrier(); tw = max_over_ranks(time.perf_counter() - t0)
        return tw, max_over_ranks(mm["kernel_ms"]) / steps, mm["failed"], d.nfev_hist()

    # elastic regime (first step of the schedule from the virgin state), reported beside the headline plastic-regime value
    drv.bench_prepare(PREP_DTS[:1], advance=False)
    drv.bench_model(1)
    barrier(); t0 = time.perf_counter()
    me = drv.bench_model(max(1, args.steps // 2))
    barrier(); t_el = max_over_ranks(time.perf_counter() - t0)
    elastic = {"value": P_global * max(1, args.steps // 2) / t_el, "unit": "qpt-updates/s", "avg_kernel_ms": max_over_ranks(me["kernel_ms"]) / max(1, args.steps // 2),
               "regime": "elastic (step 1 of the schedule, dt = 0.005, virgin state)"}
    # kinematically driven plastic state (the benchmark state of rounds 1-3): 10 prescribed-velocity passes, no equilibrium solve
    t0 = time.perf_counter(); drv.bench_prepare(PREP_DTS); barrier(); prep_s = time.perf_counter() - t0
    P_local = L.exa_driver_local_qpts(drv.h)
    kin_steps = args.steps if args.solve_steps <= 0 else max(5, args.steps // 2)
    t_kin, kin_kern_ms, kin_failed, kin_hist = timed_passes(drv, kin_steps, settle_cold)
    kinematic = {"value": P_global * kin_steps / t_kin, "unit": "qpt-updates/s", "avg_kernel_ms": kin_kern_ms, "passes": kin_steps, "nonconverged_points": kin_failed,
                 "local_solver_evals": hist_dict(kin_hist), "prepare_wall_s": prep_s,
                 "state": "10 prescribed-velocity passes (v = L0 x + seeded perturbation) through the elastic-plastic transition, no equilibrium solve"}
    solve = None
    if args.solve_steps > 0:
        # ---- real time stepping (Newton + PCG with the reference's settings) on a fresh driver: the benchmark state ------------------------------
        drv.close()
        rng = np.random.default_rng(20240928)
        quats = rng.standard_normal((N ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
        sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:max(args.solve_steps, args.solve_steps_total)]
        drv = L.Driver.synthetic(N, props, quats.ravel(), sched, assembly=asm, rank=rank, nranks=world, uid=fresh_uid(), jacobi=args.jacobi, **mk)
        del quats
        barrier(); t0 = time.perf_counter()
        rows = []; hist_pl = np.zeros(64, dtype=np.int64); ms_pl = 0.0; calls_pl = 0; kms_tot = 0.0; kit_tot = 0; model_ms_tot = 0.0; calls_tot = 0

        def solve_step(ti, commit):
            """one step of the schedule; returns the row of in-solve figures (kernel time = HIP events around every constitutive launch of the step,
            the reference's ecmech_kernel region, src/mechanics_ecmech.cpp:237-257; max over ranks)"""
            drv.reset_timers()
            assert drv.step(ti, commit=commit), f"Newton failed at step {ti}"
            tm = drv.timers(); nw, kr, mc = drv.stats()
            mms = max_over_ranks(tm["model_ms"]); calls = int(mc[-1])
            return {"step": ti, "dt": float(sched[ti - 1]), "newton_iters": int(nw[-1]), "krylov_iters": int(kr[-1]), "model_calls": calls,
                    "kernel_ms_per_call": mms / max(calls, 1), "qpt_updates_per_s_in_kernel": P_global * calls / (mms * 1e-3),
                    "_model_ms": mms, "_krylov_ms": max_over_ranks(tm["krylov_ms"]), "_krylov_iters": tm["krylov_iters"]}
        for ti in range(1, args.solve_steps + 1):
            last = ti == args.solve_steps     # the last step is solved but not committed: the timed passes repeat its converged launch
            row = solve_step(ti, commit=not last); rows.append(row)
            kms_tot += row["_krylov_ms"]; kit_tot += row["_krylov_iters"]; model_ms_tot += row["_model_ms"]; calls_tot += row["model_calls"]
            if ti >= 10:     # steady plastic regime ("steps >= 5 of the schedule"; the transition has passed by step 10)
                ms_pl += row["_model_ms"]; calls_pl += row["model_calls"]; hist_pl += drv.nfev_hist(which=1 if last else 0)
        barrier(); wall = max_over_ranks(time.perf_counter() - t0)
        dg = drv.diagnostics()
        solve = {"steps": args.solve_steps, "wall_s": wall, "per_step": [{k: v for k, v in r.items() if not k.startswith("_")} for r in rows],
                 "qpt_updates_per_s_in_kernel": P_global * calls_tot / (model_ms_tot * 1e-3),
                 "pcg_iters_per_s": kit_tot / (kms_tot * 1e-3),
                 "steady_plastic": ({"steps": f"10..{args.solve_steps}", "qpt_updates_per_s_in_kernel": P_global * calls_pl / (ms_pl * 1e-3),
                                     "kernel_ms_per_call": ms_pl / calls_pl,
                                     "last_step": {"step": rows[-1]["step"], "kernel_ms_per_call": rows[-1]["kernel_ms_per_call"],
                                                   "qpt_updates_per_s_in_kernel": rows[-1]["qpt_updates_per_s_in_kernel"],
                                                   "note": "all in-solve launches of the last step (every Newton iterate, not only the converged one); the per-step rate is "
                                                           "still rising towards it over steps 10.. (per_step): the evaluation counts left by the elastic-plastic transition "
                                                           "decay from step to step, the timed passes repeat the converged launch of this step"},
                                     "local_solver_evals": dict(hist_dict(hist_pl), note="converged launch of every step >= 10 (rank 0)")} if calls_pl else None),
                 "avg_stress_zz": [float(x) for x in drv.avgs(0, 6)[:, 2]],      # committed steps
                 "model_failed_points": dg["model_failed_points"], "pcg_solves_at_iteration_cap": dg["pcg_not_converged"],
                 "pcg_worst_residual_reduction_at_cap": dg["pcg_worst_capped_reduction"], "pcg_rel_tol": 1e-7,
                 "note": "reference schedule (test/data/custom_dt.txt) and solver settings (NR rel 5e-5 / abs 5e-10, PCG rel 1e-7 / 1000 iterations, identity 'Jacobi'); at this "
                         "size the linear solves stop at the iteration cap: pcg_worst_residual_reduction_at_cap = |r|/|r0| they reached (Newton converges regardless)"}
        P_local = L.exa_driver_local_qpts(drv.h)
    # ---- timed region 1: constitutive passes at the benchmark state --------------------------------------------------------------------
    # untimed passes before the timed region: exactly the --warmup passes of the contract (after the real solve; see `settle` above); the line's
    # `warmup` is the number that ran
    if args.solve_steps > 0:
        t_model, kern_ms, failed, nfev = timed_passes(drv, args.steps, settle)
    else:
        t_model, kern_ms, failed, nfev = t_kin, kin_kern_ms, kin_failed, kin_hist
    m = {"failed": failed}
    value = P_global * args.steps / t_model
    # ---- timed region 2: PCG iterations (same state) ----------------------------------------------------------------------------------
    drv.bench_pcg(max(2, args.pcg_iters // 10))   # warm-up
    barrier()
    t0 = time.perf_counter()
    pc = drv.bench_pcg(args.pcg_iters)
    barrier()
    t_pcg_wall = max_over_ranks(time.perf_counter() - t0)
    pcg_ms = max_over_ranks(pc["pcg_ms"]); apply_ms = max_over_ranks(pc["apply_ms"]) / args.pcg_iters
    pcg_it_s = pc["iters"] / (pcg_ms * 1e-3)
    comm_ranks, comm_transport = drv.comm_info()
    comm_details = drv.comm_details()
    per_rank = None
    if world > 1:      # every rank's share: elements, neighbours, bytes per halo exchange, overlap setting
        per_rank = [None] * world
        dist.all_gather_object(per_rank, dict(rank=rank, **comm_details))
    # ---- the drop-in route at the same state: what include/exaconstit_mfem_adapters.hpp calls (never `value`) --------------------------------------------
    adapter = None
    if world == 1 and not args.no_adapter_route and args.assembly.upper() == "PA" and N <= 160:
        try:
            ar = drv.bench_adapter_route(max(5, min(args.steps, 40)), max(5, min(args.pcg_iters, 40)))
            Pq = float(P_local)
            adapter = {
                "what": "the SAME state through exactly the calls HipExaModel::ModelSetup and HipExaNLFIntegrator::{AssembleGradPA, AddMultGradPA} make "
                        "(include/exaconstit_mfem_adapters.hpp): exa_model_setup on the reference's (vdim, Q, E) quadrature functions with a Jacobian field and a velocity "
                        "E-vector, exa_grad_setup, exa_grad_apply on E-vectors between exa_restrict and exa_restrict_transpose_add; a second context in the AOS layout "
                        "that is given the driver's begin-of-step state (de-blocked on the device), coordinates and velocity",
                "aos_staging_through_lds": ar["aos_staging"],
                "model_setup": {"avg_kernel_ms": ar["model_ms"], "qpt_updates_per_s": Pq / (ar["model_ms"] * 1e-3), "bytes_per_qpt": MODEL_BYTES_PER_QPT,
                                "frac": MODEL_BYTES_PER_QPT * Pq / (ar["model_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "pass_ms_with_velocity_restriction": ar["pass_ms"], "nonconverged_points": ar["failed"],
                                "driver_route_same_loop_ms": ar["driver_route_model_ms"], "ratio_to_driver_route": ar["model_ms"] / ar["driver_route_model_ms"],
                                "note": "frac = 928 B x qpts / kernel time / 8 TB/s (SURVEY 8(d)); this launch moves all of them (Jacobian field in, 36 tangent entries out); "
                                        "driver_route = exa_model_setup_lvec_records on the element-blocked layout (`value`)"},
                "geometry_ms": ar["geometry_ms"],
                "grad_setup": {"avg_kernel_ms": ar["grad_setup_ms"], "bytes_per_qpt": 8.0 * (36 + 9 + 36),
                               "frac": 8.0 * (36 + 9 + 36) * Pq / (ar["grad_setup_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "AssembleGradPA: reads tangent 36 + Jacobian 9, writes the 36-double record (compact tangent 26 + adj(J) 9 + W detJ; the 46-double one "
                                       "on demand for the diagonal; the reference writes 81; the record route has no such pass)"},
                "grad_apply": {"avg_kernel_ms": ar["grad_apply_ms"], "bytes_per_qpt": APPLY_BYTES_PER_QPT,
                               "frac": APPLY_BYTES_PER_QPT * Pq / (ar["grad_apply_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "bytes_moved_per_qpt": 8.0 * (36 + 3 + 6), "frac_on_bytes_moved": 8.0 * 45 * Pq / (ar["grad_apply_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "action_ms_with_restriction_and_transpose": ar["action_ms"],
                               "driver_route_action_ms": ar["driver_route_apply_ms"], "ratio_to_driver_route": ar["action_ms"] / ar["driver_route_apply_ms"],
                               "note": "AddMultGradPA on E-vectors: 36-double record (compact tangent + geometry) + x (24 / element) + y read and written; frac priced at SURVEY 8(d)'s 408 B/qpt; "
                                       "action = L->E + kernel + E->L, what an MFEM caller pays per PCG iteration"},
                "lvec_pair": {"what": "HipExaModelLVec / HipExaNLFIntegratorLVec on the same AOS quadrature functions: exa_model_setup_lvec (node gathers + Jacobians + update in "
                                      "one launch, rows staged), exa_grad_setup with the compact tangent form, exa_grad_apply_lvec (gather + action + scatter-add), exa_residual_lvec",
                              "model_setup": {"avg_kernel_ms": ar["lvec_model_ms"], "qpt_updates_per_s": Pq / (ar["lvec_model_ms"] * 1e-3),
                                              "frac": MODEL_BYTES_PER_QPT * Pq / (ar["lvec_model_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "ratio_to_driver_route": ar["lvec_model_ms"] / ar["driver_route_model_ms"]},
                              "grad_setup_ms": ar["lvec_grad_setup_ms"],
                              "grad_apply": {"avg_kernel_ms": ar["lvec_apply_ms"], "bytes_per_qpt": APPLY_MOVED_BYTES_COMPACT,
                                             "frac": APPLY_MOVED_BYTES_COMPACT * Pq / (ar["lvec_apply_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "ratio_to_driver_route": ar["lvec_apply_ms"] / ar["driver_route_apply_ms"]},
                              "residual_ms": ar["lvec_residual_ms"],
                              "fused_records": {"what": "HipExaModelLVec(.., fused_records = true): exa_model_setup_lvec_records on the reference layout - state and stress rows staged, the "
                                                        "compact records of the action written by the launch instead of the tangent field; no exa_grad_setup pass",
                                                "avg_kernel_ms": ar["lvec_records_model_ms"], "qpt_updates_per_s": Pq / (ar["lvec_records_model_ms"] * 1e-3),
                                                "frac": MODEL_BYTES_PER_QPT * Pq / (ar["lvec_records_model_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "ratio_to_driver_route": ar["lvec_records_model_ms"] / ar["driver_route_model_ms"],
                                                "stress_max_rel_diff": ar["lvec_records_stress_rel_diff"], "action_max_rel_diff": ar["lvec_records_action_rel_diff"]},
                              "stress_max_rel_diff": ar["lvec_stress_rel_diff"], "action_max_rel_diff": ar["lvec_action_rel_diff"]},
                "parity_with_driver_route": {"stress_max_rel_diff": ar["stress_rel_diff"], "state_max_rel_diff": ar["state_rel_diff"], "action_max_rel_diff": ar["action_rel_diff"],
                                             "points_with_another_evaluation_count": ar["nfev_differing"],
                                             "note": "max |a - b| / max |a| over all points of the RVE: end-of-step stress and state (slot 3, the local solver's evaluation "
                                                     "count, counted separately) of the AOS launch against the record route's, K x through the E-vector action against the "
                                                     "L-vector action"},
            }
        except Exception as e:      # a reported block, never a dependency of the headline
            adapter = {"failed": str(e)}
    # ---- the solve goes on: in-solve kernel rates up to the plateau (qpt updates/s = sum of P over ModelSetup calls / kernel time) ----------
    in_solve = None
    if solve is not None and args.solve_steps_total > args.solve_steps:
        total = len(sched)
        barrier(); t0 = time.perf_counter()
        drv.commit_step()                         # end-of-step update of the step whose converged launch was timed
        for ti in range(args.solve_steps + 1, total + 1):
            rows.append(solve_step(ti, commit=True))
        barrier(); wall2 = max_over_ranks(time.perf_counter() - t0)

        def seg(lo, hi):
            rr = [r for r in rows if lo <= r["step"] <= hi]
            if not rr:
                return None
            ms = sum(r["_model_ms"] for r in rr); calls = sum(r["model_calls"] for r in rr)
            return {"steps": f"{lo}..{min(hi, rows[-1]['step'])}", "model_calls": calls, "kernel_ms_per_call": ms / calls,
                    "qpt_updates_per_s_in_kernel": P_global * calls / (ms * 1e-3), "frac": MODEL_BYTES_PER_QPT * P_local / (ms / calls * 1e-3) / 1e9 / HBM_PEAK_GBS}
        seg01_end = 21      # last step of the dt = 0.1 segment of test/data/custom_dt.txt
        in_solve = {"whole_solve": seg(1, total), "steps_ge_5": seg(5, total), "steps_ge_10": seg(10, total),
                    "plateau": seg(max(10, min(seg01_end, total) - 2), min(seg01_end, total)),
                    "dt_0p2_segment": seg(seg01_end + 1, total) if total > seg01_end else None,
                    "solved_to_step": total, "wall_s_of_the_continuation": wall2,
                    "per_step": [{k: v for k, v in r.items() if not k.startswith("_")} for r in rows[args.solve_steps:]],
                    "definition": "SURVEY 8(d): (sum over ModelSetup calls of P) / (time inside the fused constitutive launch), every launch of every Newton iterate of "
                                  "the steps named, HIP events around each launch; frac = 928 B x qpts / kernel time / 8 TB/s.  plateau = the last three steps of "
                                  "the schedule's dt = 0.1 segment (the evaluation counts left by the elastic-plastic transition have decayed); the dt = 0.2 segment "
                                  "that follows starts a new, shorter transition",
                    "claim_40pct_rests_on": "plateau (in-solve) and the timed passes of `value` (the converged launch of step "
                                            f"{args.solve_steps}); steps_ge_5 / whole_solve include the transition launches, whose evaluation counts (mean up to 10, "
                                            "wave maximum 15) set their time - see profiles/r05_transition_study.txt"}
    if rank == 0:
        ndof_local = L.exa_driver_local_dofs(drv.h)
        # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside the timed bench): bytes/qpt x local qpts.  The counter
        # files carry the kernel build id of the library they were measured on (scripts/profile_round4.sh); a file of another build is refused
        kid = L.exa_kernel_build_id().decode()
        traffic = {}

        def profile_json(suffix):
            """newest profiles/*<suffix> whose kernel_build_id is this library's; (None, reason) otherwise"""
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix))
            for f in reversed(cands):
                try:
                    d = json.load(open(os.path.join(ROOT, "profiles", f)))
                except Exception:
                    continue
                if d.get("kernel_build_id") == kid:
                    return d, "profiles/" + f
            return None, (f"no profiles/*{suffix} of kernel build {kid} (newest: {cands[-1]})" if cands else f"no profiles/*{suffix}")
        try:
            tj, tsrc = profile_json("_pmc_traffic.json")
            if tj is not None:
                raw = tj["bytes_per_qpt"]
                # per kernel either one number (measured on the fcc_voce instantiation) or {model: bytes}: never quote one model's counters for another
                for k, v in raw.items():
                    if isinstance(v, dict):
                        if args.model in v:
                            traffic[k] = v[args.model]
                    elif args.model == "fcc_voce" or k != "k_model_setup":
                        traffic[k] = v
            traffic["_file"] = tsrc
        except Exception as e:
            traffic = {"_file": f"unreadable: {e}"}
        flops = None; valu_per_wave = None
        try:
            fj, _ = profile_json("_pmc_flops.json")
            if fj is not None and args.model == "fcc_voce":      # the instruction counts were collected on the Voce instantiation: never quoted for another kernel
                pl = fj["k_model_setup"]["plastic"]
                flops = pl["flop_per_qpt"]; valu_per_wave = pl["valu_insts_per_wave"]
        except Exception:
            flops = None
        model_gbs = MODEL_BYTES_PER_QPT * P_local / (kern_ms * 1e-3) / 1e9
        records = args.assembly.upper() in ("PA", "EA") and os.environ.get("EXA_TANGENT_RECORDS") != "off" and not args.jacobi
        ea_streamed = args.assembly.upper() == "EA" and os.environ.get("EXA_EA_ASSEMBLED") == "1"   # 24 x 24 matrices from HBM
        geo = os.environ.get("EXA_APPLY_GEO", "on") != "off" and not ea_streamed
        compact = geo and os.environ.get("EXA_TANGENT_FORM", "compact") != "full"
        moved = 616.0 if ea_streamed else (APPLY_MOVED_BYTES_COMPACT if compact else (APPLY_MOVED_BYTES_PER_QPT if geo else APPLY_BYTES_PER_QPT))
        apply_gbs = APPLY_BYTES_PER_QPT * P_local / (apply_ms * 1e-3) / 1e9
        iter_bytes = APPLY_BYTES_PER_QPT * P_local + PCG_VEC_BYTES_PER_DOF * ndof_local
        out = {
            "metric": "quadrature-point constitutive updates/s + Newton-PCG iter/s, 128^3 hex RVE",
            "value": value, "unit": "qpt-updates/s", "n_gpus": world, "steps": args.steps, "warmup": settle, "warmup_requested": args.warmup,
            "ms_per_step": t_model / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}^3 hex RVE p=1, {MODEL_NAMES[args.model]} (ExaCMech evptn), plastic regime; "
                                   f"{'partial' if args.assembly.upper() == 'PA' else 'element'}-assembly PCG", "elements": N ** 3,
                       "state": (f"real Newton/PCG solve of the reference schedule, {args.solve_steps} steps; timed passes = the converged (last) residual evaluation of step {args.solve_steps}"
                                 if args.solve_steps > 0 else "kinematically driven (10 prescribed-velocity passes)"),
                       "qpts": P_global, "decomposition": f"{world} block(s)"},
            "library": {"path": L.LIB_PATH, "build_id": L.exa_build_id().decode(), "kernel_build_id": L.exa_kernel_build_id().decode()},
            "comm": {"transport": comm_transport, "ranks_reported_by_transport": comm_ranks, "rank0": comm_details, "per_rank": per_rank,
                     "note": "rccl: ncclCommCount of the library's communicator; ipc: shared-device inter-process transport (more ranks than devices)"},
            "pcg_iters_per_s": pcg_it_s, "pcg_iters": pc["iters"], "pcg_ms_per_iter": pcg_ms / max(pc["iters"], 1),
            "pcg_wall_s": t_pcg_wall, "nonconverged_points": m["failed"],
            "local_solver_evals": dict(hist_dict(nfev), note="residual/Jacobian evaluations of the 8-unknown point solve per quadrature point (rank 0), last timed pass"),
            "elastic_regime": elastic,
            "kinematic_state": kinematic,
            "roofline": {"kernel": f"k_model_setup<{'KM-DD' if 'kmdd' in args.model else 'Voce'}> (fused node gather + grad_calc + ExaCMech update + tangent)", "bound": "hbm",
                         "achieved": model_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": model_gbs / HBM_PEAK_GBS,
                         "traffic": traffic["k_model_setup"] * P_local if "k_model_setup" in traffic else None, "traffic_source": traffic.get("_file"), "kernel_build_id": kid,
                         "bytes_per_qpt": MODEL_BYTES_PER_QPT, "avg_kernel_ms": kern_ms,
                         # the same launch priced at what it moves by construction on the record route (648 B/qpt): the figure to compare with PMC traffic
                         "frac_on_bytes_moved": (648.0 * P_local / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (records and args.model in ("fcc_voce", "bcc_voce", "fcc_voce_nl")) else None,
                         "state_key": "value / roofline: converged launch of the last --solve-steps step of a real Newton/PCG solve (since round 4); `kinematic_state` "
                                      "(prescribed-velocity passes, the benchmark state of rounds 1-3) is reported under its own key every round: compare rounds on like keys",
                         "bytes_written_per_qpt_note": ("this launch writes the 26-number gradient record instead of the 36 tangent entries (AssembleGradPA fused in)") if records else None,
                         "fp64_flop_per_qpt": flops, "fp64_tflops": (flops * P_local / (kern_ms * 1e-3) / 1e12) if flops else None,
                         "fp64_vector_frac": (flops * P_local / (kern_ms * 1e-3) / 1e12 / FP64_VEC_PEAK_TFLOPS) if flops else None,
                         # FP64 issue roofline: a wave64 FP64 VALU instruction occupies its SIMD for 4 cycles (16 FMA lanes per clock and SIMD = the
                         # 78.6 TFLOP/s vector peak); issue time = instructions per wave x 4 cycles x waves per SIMD / 2.4 GHz
                         "fp64_issue": ({"valu_insts_per_wave": valu_per_wave, "ms_at_full_issue": valu_per_wave * 4.0 * (P_local / 64.0 / 1024.0) / 2.4e9 * 1e3,
                                         "frac": valu_per_wave * 4.0 * (P_local / 64.0 / 1024.0) / 2.4e9 * 1e3 / kern_ms} if valu_per_wave else None),
                         "bytes_moved_per_qpt_note": "by construction the launch moves 648 B/qpt (read ~6 velocity / coordinate, 15 state, 6 stress doubles; write 28 state, 6 stress, "
                                                      "26 record doubles: no Jacobian field, effective shear rate from state slot 0); frac stays priced at SURVEY 8(d)'s 928 B/qpt",
                         "note": "the contract's HBM fraction of the ALGORITHMIC bytes (928 B/qpt) is reported in frac; the launch is bound by FP64 VALU issue at the clock "
                                 "the chip sustains at its 1 400 W package limit (~1.93 GHz, not the nominal 2.4 GHz the fp64_issue figures are priced at): SQ counters "
                                 + ("(profiles/r06_sq_fcc_voce.txt) show the VALU busy 90 % of the wave cycles and waiting 12 % of them, 6 943 VALU instructions per wave, "
                                    "no scratch (round 4: 86 % / 19.5 % / 6 845; profiles/r05_kernel_experiments.txt); " if args.model == "fcc_voce" else
                                    f"of the Kocks-Mecking launches (main + tail) in profiles/r06_sq_{args.model}.txt; fp64_* figures are quoted for the Voce kernel only; ")
                                 + "traffic = L2-boundary bytes from the PMC "
                                 "passes of THIS kernel build and instantiation (profiles/*_pmc_traffic.json with the library's kernel_build_id; null otherwise); "
                                 "roofline_pcg_apply is the HBM-bound half of the metric"},
            "roofline_pcg_apply": {"kernel": ("k_ea_apply_p1 (element mat-vec)" if ea_streamed else "k_grad_apply_p1<LVEC,GEO,CMP> (AddMultGradPA / matrix-free element-assembly action + gather/scatter)"), "bound": "hbm",
                                   "achieved": moved * P_local / (apply_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": moved * P_local / (apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": traffic["k_grad_apply_p1"] * P_local if "k_grad_apply_p1" in traffic else None,
                                   "bytes_per_qpt": moved, "avg_kernel_ms": apply_ms,
                                   "geometry_recomputed": geo, "compact_tangent": compact,
                                   "survey_8d_bytes_per_qpt": APPLY_BYTES_PER_QPT, "survey_8d_equivalent_gbs": apply_gbs,
                                   "note": "bytes_per_qpt = what this kernel has to move (tangent 26 or 36 doubles [+ 10 geometry doubles when streamed] + "
                                           "L-vector gather/scatter); SURVEY 8(d) prices the reference's algorithm at 408 B/qpt (tangent 36 + Jacobian 9 + "
                                           "E-vector x 3 + y 3), survey_8d_equivalent_gbs is the rate a kernel streaming those bytes would need for the same time",
                                   "pcg_iteration_frac": (moved * P_local + PCG_VEC_BYTES_PER_DOF * ndof_local) / (pcg_ms * 1e-3 / max(pc["iters"], 1)) / 1e9 / HBM_PEAK_GBS},
        }
        if solve is not None:
            out["newton_pcg_solve"] = solve
        if in_solve is not None:
            # the in-solve fractions first, as plain numbers of the roofline object (a truncated record of this line still carries them); the segments follow
            for key, name in (("whole_solve", "in_solve_frac_whole_solve"), ("steps_ge_5", "in_solve_frac_steps_ge_5"), ("steps_ge_10", "in_solve_frac_steps_ge_10"),
                              ("plateau", "in_solve_frac_plateau"), ("dt_0p2_segment", "in_solve_frac_dt_0p2_segment")):
                if in_solve.get(key):
                    out["roofline"][name] = in_solve[key]["frac"]
            out["roofline"] = {k: out["roofline"][k] for k in (["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"] + [k for k in out["roofline"] if k.startswith("in_solve_frac_")]
                                                                + [k for k in out["roofline"] if k not in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic") and not k.startswith("in_solve_frac_")])}
            out["roofline"]["in_solve"] = in_solve
        if adapter is not None:
            # L2-boundary traffic of the adapter route's launches from the counter passes of this kernel build (Voce instantiations; per launch, like roofline.traffic)
            try:
                ab = (tj or {}).get("adapter_route_bytes_per_qpt") if args.model == "fcc_voce" else None
                if ab and "model_setup" in adapter:
                    adapter["model_setup"]["traffic"] = ab["k_model_setup_staged_aos_evec"] * P_local
                    adapter["grad_apply"]["traffic"] = ab["k_grad_apply_p1_evec_compact_geo"] * P_local
                    adapter["lvec_pair"]["model_setup"]["traffic"] = ab["k_model_setup_staged_aos_lvec"] * P_local
                    adapter["lvec_pair"]["fused_records"]["traffic"] = ab["k_model_setup_staged_aos_lvec_records"] * P_local
                    adapter["traffic_source"] = tsrc
            except Exception:
                pass
            out["adapter_route"] = adapter
        if world == 1 and not args.no_cpu_baseline:
            try:
                voce = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "props_cp_voce.txt")).ravel()
                out["cpu_baseline"] = cpu_baseline(props, gpu_step=gpu_config1_step(L, voce))
            except Exception as e:   # the baseline is a reported number, never a dependency of the product path
                out["cpu_baseline"] = {"value": None, "unit": "qpt-updates/s", "cores": host_cores(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    drv.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
