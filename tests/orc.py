"""ctypes binding of the CPU oracle (oracle/liboracle.so) + loader for the reference's regression cases.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The regression-case loader mirrors how the reference's driver interprets its options file
(reference src/option_parser.cpp:26-932, src/mechanics_driver.cpp:243-546) for the subset the golden cases use.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFDATA = os.path.join(ROOT, "tests", "golden", "refdata")

XTAL = {"fcc": 0, "bcc": 1}
KIN = {"powervoce": 0, "powervocenl": 1, "mtsdd": 2}
ASM = {"pa": 0, "ea": 1, "full": 1}   # FULL assembles the same operator as EA (sparse matrix + AMG in the reference)
NLS = {"nr": 0, "nrls": 1}

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _p(a):
    return a.ctypes.data_as(dp)


def _ip(a):
    return a.ctypes.data_as(ip)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def build_native(out_dir):
    """The oracle compiled for THIS host (-O3 -march=native -fopenmp) into out_dir: the CPU-baseline leg of bench.py times this build (a
    -march=native object must not travel between machines, so it is never written into the tree).  Returns the path."""
    so = os.path.join(out_dir, "liboracle_native.so")
    subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-Wno-unused-variable", "-shared", "-o", so,
                           os.path.join(ORACLE_DIR, "oracle_capi.cpp")])
    return so


def use_lib(path):
    """bind this module to another build of the oracle (bench.py: the -march=native one)"""
    global _lib
    _lib = C.CDLL(path)
    _lib.orc_ref_elem.restype = C.c_int
    return _lib


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build()
        _lib = C.CDLL(so)
        _lib.orc_ref_elem.restype = C.c_int
    return _lib


class OrcCase(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("p", C.c_int),
                ("sx", C.c_double), ("sy", C.c_double), ("sz", C.c_double),
                ("xtal", C.c_int), ("kin", C.c_int), ("nprops", C.c_int), ("props", dp), ("temp_k", C.c_double),
                ("ngrains", C.c_int), ("elem_grain", ip), ("quats", dp),
                ("nsteps", C.c_int), ("dts", dp),
                ("nbc", C.c_int), ("bc_step", ip), ("bc_nids", ip), ("bc_ids", ip), ("bc_comps", ip), ("bc_vals", dp), ("bc_vgrad", dp),
                ("assembly", C.c_int), ("nl_solver", C.c_int), ("precond", C.c_int), ("integ", C.c_int),
                ("newton_rel", C.c_double), ("newton_abs", C.c_double), ("newton_iter", C.c_int),
                ("krylov_rel", C.c_double), ("krylov_abs", C.c_double), ("krylov_iter", C.c_int),
                ("additional_avgs", C.c_int), ("second_order_terms", C.c_int), ("use_input_temperature", C.c_int),
                ("verbose", C.c_int),
                ("dt_auto", C.c_int), ("dt_start", C.c_double), ("dt_min", C.c_double), ("dt_scale", C.c_double), ("t_final", C.c_double)]


class OrcResult(C.Structure):
    _fields_ = [("avg_stress", dp), ("avg_def_grad", dp), ("avg_pl_work", dp), ("avg_dp_tensor", dp),
                ("newton_iters", ip), ("krylov_iters", ip), ("model_calls", ip),
                ("qpt_updates", C.c_int64), ("t_model", C.c_double), ("t_krylov", C.c_double), ("t_total", C.c_double),
                ("failed", C.c_int), ("steps_done", C.c_int), ("dts_used", dp)]


def load_case(toml_name, datadir=REFDATA):
    """Parse one of the reference's regression option files into a plain dict (auto mesh + ExaCMech subset)."""
    import tomli
    with open(os.path.join(datadir, toml_name), "rb") as f:
        t = tomli.load(f)
    props = np.loadtxt(os.path.join(datadir, t["Properties"]["Matl_Props"]["floc"])).ravel()
    g = t["Properties"]["Grain"]
    quats = np.loadtxt(os.path.join(datadir, g["ori_floc"]))[: g["num_grains"]]
    grains = np.loadtxt(os.path.join(datadir, g["grain_floc"])).astype(np.int64).ravel()
    mesh = t["Mesh"]
    ncuts = mesh["Auto"]["ncuts"]
    length = mesh["Auto"]["length"]
    ref = int(mesh.get("ref_ser", 0))
    nx, ny, nz = [int(c) * 2 ** ref for c in ncuts]
    # uniform refinement: children inherit the parent's grain id
    f = 2 ** ref
    eg = np.empty(nx * ny * nz, dtype=np.int32)
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                eg[i + nx * (j + ny * k)] = grains[(i // f) + ncuts[0] * ((j // f) + ncuts[1] * (k // f))] - 1
    tm = t["Time"]
    if "Custom" in tm:
        dts = np.loadtxt(os.path.join(datadir, tm["Custom"]["floc"])).ravel()[: tm["Custom"]["nsteps"]]
        auto = None
    elif "Auto" in tm:
        dts = None
        auto = tm["Auto"]
    else:
        dt, tf = tm["Fixed"]["dt"], tm["Fixed"]["t_final"]
        n = int(round(tf / dt))
        dts = np.full(n, dt)
        auto = None
    bcs = t["BCs"]
    if bcs.get("changing_ess_bcs", False):
        steps = bcs["update_steps"]
        ids, comps = bcs["essential_ids"], bcs["essential_comps"]
        vals = bcs.get("essential_vals", [[0.0] * (3 * len(i)) for i in ids])
        vgrad = bcs.get("essential_vel_grad", [[[0.0] * 3] * 3 for _ in steps])
    else:
        steps = [1]
        ids, comps = [bcs["essential_ids"]], [bcs["essential_comps"]]
        vals = [bcs.get("essential_vals", [0.0] * (3 * len(ids[0])))]
        vgrad = [bcs.get("essential_vel_grad", [[0.0] * 3] * 3)]
    vgrad = [[float(x) for row in m for x in row] for m in vgrad]
    sol = t["Solvers"]
    ecm = t["Model"]["ExaCMech"]
    vis = t.get("Visualizations", {})
    return dict(
        nx=nx, ny=ny, nz=nz, p=int(mesh.get("p_refinement", 1)), length=[float(x) for x in length],
        xtal=XTAL[ecm["xtal_type"].lower()], kin=KIN[ecm["slip_type"].lower()], props=props,
        temp_k=float(t["Properties"]["temperature"]), elem_grain=eg, quats=np.ascontiguousarray(quats, dtype=np.float64),
        dts=dts, auto=auto, bc_steps=steps, bc_ids=ids, bc_comps=comps, bc_vals=vals, bc_vgrad=vgrad,
        assembly=ASM[sol.get("assembly", "FULL").lower()], nl_solver=NLS[sol.get("NR", {}).get("nl_solver", "NR").lower()],
        newton_rel=sol["NR"]["rel_tol"], newton_abs=sol["NR"]["abs_tol"], newton_iter=sol["NR"]["iter"],
        krylov_rel=sol["Krylov"]["rel_tol"], krylov_abs=sol["Krylov"]["abs_tol"], krylov_iter=sol["Krylov"]["iter"],
        additional_avgs=bool(vis.get("additional_avgs", False)),
        integ=1 if sol.get("integ_model", "FULL").lower() == "bbar" else 0,
    )


def run_case(case, nsteps=None, precond=0, second_order_terms=False, use_input_temperature=False, verbose=0, replay_target33=None, replay_increments=False):
    """Run a regression case on the oracle; returns dict of arrays.
    replay_target33 (Time.Auto cases only): golden sigma_33 column; the step sizes are then chosen among the reference's admissible
    candidates so as to follow it (oracle/driver_port.hpp run_case_replay) and out["ks"] holds the Newton counts this implies."""
    L = lib()
    auto = case.get("auto") if case.get("dts") is None else None
    if auto is not None:      # Time.Auto: nsteps = row capacity (reference: ceil(t_final / dt_min), src/mechanics_driver.cpp:212)
        ns = int(nsteps) if nsteps is not None else int(np.ceil(auto["t_final"] / auto["dt_min"]))
        dts = np.zeros(ns)
    else:
        dts = np.ascontiguousarray(case["dts"], dtype=np.float64)
        if nsteps is not None:
            dts = dts[:nsteps]
        ns = len(dts)
    props = np.ascontiguousarray(case["props"], dtype=np.float64)
    eg = np.ascontiguousarray(case["elem_grain"], dtype=np.int32)
    quats = np.ascontiguousarray(case["quats"], dtype=np.float64)
    bc_step = np.array(case["bc_steps"], dtype=np.int32)
    bc_nids = np.array([len(x) for x in case["bc_ids"]], dtype=np.int32)
    bc_ids = np.array([i for x in case["bc_ids"] for i in x], dtype=np.int32)
    bc_comps = np.array([i for x in case["bc_comps"] for i in x], dtype=np.int32)
    bc_vals = np.array([v for x in case["bc_vals"] for v in x], dtype=np.float64)
    bc_vgrad = np.array([v for x in case.get("bc_vgrad", [[0.0] * 9] * len(bc_step)) for v in x], dtype=np.float64)
    c = OrcCase(case["nx"], case["ny"], case["nz"], case["p"], *case["length"],
                case["xtal"], case["kin"], len(props), _p(props), case["temp_k"],
                quats.shape[0], _ip(eg), _p(quats), ns, _p(dts),
                len(bc_step), _ip(bc_step), _ip(bc_nids), _ip(bc_ids), _ip(bc_comps), _p(bc_vals), _p(bc_vgrad),
                case["assembly"], case["nl_solver"], precond, int(case.get("integ", 0)),
                case["newton_rel"], case["newton_abs"], case["newton_iter"],
                case["krylov_rel"], case["krylov_abs"], case["krylov_iter"],
                int(case["additional_avgs"]), int(second_order_terms), int(use_input_temperature), verbose,
                int(auto is not None), *([float(auto.get(k, d)) for k, d in (("dt_start", 1.0), ("dt_min", 1.0), ("dt_scale", 0.25), ("t_final", 1.0))] if auto is not None else [0.0] * 4))
    out = dict(avg_stress=np.zeros((ns, 6)), avg_def_grad=np.zeros((ns, 9)), avg_pl_work=np.zeros(ns),
               avg_dp_tensor=np.zeros((ns, 6)), newton_iters=np.zeros(ns, np.int32), krylov_iters=np.zeros(ns, np.int32),
               model_calls=np.zeros(ns, np.int32))
    dts_used = np.zeros(ns)
    r = OrcResult(_p(out["avg_stress"]), _p(out["avg_def_grad"]), _p(out["avg_pl_work"]), _p(out["avg_dp_tensor"]),
                  _ip(out["newton_iters"]), _ip(out["krylov_iters"]), _ip(out["model_calls"]), 0, 0, 0, 0, 0, 0, _p(dts_used))
    if replay_target33 is not None:
        tgt = np.ascontiguousarray(replay_target33, dtype=np.float64)
        ks = np.zeros(ns, np.int32)
        L.orc_run_case_replay(C.byref(c), _p(tgt), len(tgt), C.byref(r), _ip(ks), int(replay_increments))
        out["ks"] = ks[: r.steps_done]
    else:
        L.orc_run_case(C.byref(c), C.byref(r))
    if auto is not None:
        for k in list(out):
            out[k] = out[k][: r.steps_done]
        out["dts"] = dts_used[: r.steps_done]
    out.update(qpt_updates=r.qpt_updates, t_model=r.t_model, t_krylov=r.t_krylov, t_total=r.t_total, failed=r.failed)
    return out


def golden(name):
    return np.loadtxt(os.path.join(REFDATA, name), ndmin=2)


def fmt6(a):
    """Round to the 6 significant digits the reference prints (default ostream precision)."""
    return np.array([float("%.6g" % v) for v in np.ravel(a)]).reshape(np.shape(a))
