"""ctypes binding of libexaconstit_hip.so (the C ABI of include/exaconstit_hip.h).

No CPU fallback: if the library is missing this module raises at import; if no GPU is present exa_create fails.
Device memory is passed as raw pointers (torch tensors' data_ptr() in the tests/bench — PyTorch is only the allocator).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXA_LIB") or os.path.join(_HERE, "libexaconstit_hip.so")   # EXA_LIB: a timing variant built by `make variant` (experiments only)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C exaconstit_amd/csrc` (or __graft_entry__.build()); "
        "there is no CPU fallback for the product path")

# PyTorch bundles its own HIP runtime (soname libamdhip64.so.7).  Two HIP runtimes in one process do not share
# devices, so when torch is the allocator it must be loaded FIRST: the dynamic loader then resolves this library's
# NEEDED libamdhip64.so.7 to the copy torch already mapped.
try:
    import torch  # noqa: F401
except ImportError:  # stand-alone use (C++ driver, plain ctypes): the system ROCm runtime is used
    torch = None

_lib = C.CDLL(LIB_PATH)

EXA_FCC_VOCE, EXA_FCC_VOCE_NL, EXA_BCC_VOCE, EXA_BCC_VOCE_NL, EXA_FCC_KMDD, EXA_BCC_KMDD = range(6)
EXA_ASSEMBLY_PA, EXA_ASSEMBLY_EA = 0, 1
EXA_INTEG_FULL, EXA_INTEG_BBAR = 0, 1

dptr = C.c_void_p


class ExaConfig(C.Structure):
    _fields_ = [("model", C.c_int), ("nprops", C.c_int), ("props", C.POINTER(C.c_double)), ("temp_k", C.c_double),
                ("order", C.c_int), ("nelems", C.c_int), ("assembly", C.c_int), ("integ", C.c_int), ("device", C.c_int)]


def _sig(name, restype, *argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


# every symbol the header declares (tests check the list against include/exaconstit_hip.h)
exa_create = _sig("exa_create", C.c_void_p, C.POINTER(ExaConfig), C.POINTER(C.c_int))
exa_destroy = _sig("exa_destroy", None, C.c_void_p)
exa_last_error = _sig("exa_last_error", C.c_char_p, C.c_void_p)
exa_build_id = _sig("exa_build_id", C.c_char_p)
exa_kernel_build_id = _sig("exa_kernel_build_id", C.c_char_p)
exa_num_state_vars = _sig("exa_num_state_vars", C.c_int, C.c_void_p)
exa_nodes_per_elem = _sig("exa_nodes_per_elem", C.c_int, C.c_void_p)
exa_qpts_per_elem = _sig("exa_qpts_per_elem", C.c_int, C.c_void_p)
exa_shape_table = _sig("exa_shape_table", C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
exa_set_quadrature_layout = _sig("exa_set_quadrature_layout", C.c_int, C.c_void_p, C.c_int)
exa_get_quadrature_layout = _sig("exa_get_quadrature_layout", C.c_int, C.c_void_p)
exa_set_aos_staging = _sig("exa_set_aos_staging", C.c_int, C.c_void_p, C.c_int)
exa_get_aos_staging = _sig("exa_get_aos_staging", C.c_int, C.c_void_p)
exa_qf_size = _sig("exa_qf_size", C.c_int64, C.c_void_p, C.c_int)
EXA_QLAYOUT_AOS, EXA_QLAYOUT_EB64 = 0, 1
EXA_OK, EXA_ERR_ARG, EXA_ERR_HIP, EXA_ERR_STATE, EXA_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
exa_init_state = _sig("exa_init_state", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_state_normalize = _sig("exa_state_normalize", C.c_int, C.c_void_p, dptr, C.c_void_p)
exa_model_setup = _sig("exa_model_setup", C.c_int, C.c_void_p, C.c_double, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.c_void_p)
exa_model_setup_checked = _sig("exa_model_setup_checked", C.c_int, C.c_void_p, C.c_double, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.c_void_p)
exa_model_setup_lvec_records = _sig("exa_model_setup_lvec_records", C.c_int, C.c_void_p, C.c_double, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.c_void_p)
exa_model_setup_lvec = _sig("exa_model_setup_lvec", C.c_int, C.c_void_p, C.c_double, dptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.c_void_p)
exa_set_newton_cap = _sig("exa_set_newton_cap", C.c_int, C.c_void_p, C.c_int)
exa_selftest_km_math = _sig("exa_selftest_km_math", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
exa_set_newton_caps = _sig("exa_set_newton_caps", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
exa_model_tail_count = _sig("exa_model_tail_count", C.c_int, C.c_void_p, C.c_void_p)
exa_set_newton_cap_auto = _sig("exa_set_newton_cap_auto", C.c_int, C.c_void_p, C.c_int, C.c_double)
exa_get_newton_cap = _sig("exa_get_newton_cap", C.c_int, C.c_void_p)
exa_model_nfev_hist = _sig("exa_model_nfev_hist", C.c_int, C.c_void_p, dptr, C.POINTER(C.c_int), C.c_void_p)
exa_model_status = _sig("exa_model_status", C.c_int, C.c_void_p, C.c_void_p)
exa_calc_dp = _sig("exa_calc_dp", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_jacobians = _sig("exa_jacobians", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_jacobians_from_geom = _sig("exa_jacobians_from_geom", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_grad_calc = _sig("exa_grad_calc", C.c_int, C.c_void_p, dptr, dptr, dptr, C.c_void_p)
exa_residual_setup = _sig("exa_residual_setup", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_residual_apply = _sig("exa_residual_apply", C.c_int, C.c_void_p, dptr, C.c_void_p)
exa_grad_setup = _sig("exa_grad_setup", C.c_int, C.c_void_p, C.c_double, dptr, dptr, C.c_void_p)
exa_grad_apply = _sig("exa_grad_apply", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_grad_diagonal = _sig("exa_grad_diagonal", C.c_int, C.c_void_p, dptr, C.c_void_p)
exa_grad_get_ea = _sig("exa_grad_get_ea", C.c_int, C.c_void_p, dptr, C.c_void_p)
exa_set_connectivity = _sig("exa_set_connectivity", C.c_int, C.c_void_p, dptr, C.c_int)
exa_restrict = _sig("exa_restrict", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_restrict_transpose_add = _sig("exa_restrict_transpose_add", C.c_int, C.c_void_p, dptr, dptr, C.c_void_p)
exa_grad_apply_lvec = _sig("exa_grad_apply_lvec", C.c_int, C.c_void_p, dptr, dptr, dptr, C.c_void_p)
exa_set_tangent_form = _sig("exa_set_tangent_form", C.c_int, C.c_void_p, C.c_int)
exa_set_deterministic = _sig("exa_set_deterministic", C.c_int, C.c_void_p, C.c_int)
exa_grad_tangent_defect = _sig("exa_grad_tangent_defect", C.c_int, C.c_void_p, dptr, C.POINTER(C.c_double), C.c_void_p)
EXA_TANGENT_FULL, EXA_TANGENT_DEV5_BULK, EXA_TANGENT_DEV5_BULK_GEO = 0, 1, 2
exa_set_ea_matrix_free = _sig("exa_set_ea_matrix_free", C.c_int, C.c_void_p, C.c_int)
exa_grad_set_coords = _sig("exa_grad_set_coords", C.c_int, C.c_void_p, dptr)
exa_residual_lvec = _sig("exa_residual_lvec", C.c_int, C.c_void_p, dptr, dptr, dptr, C.c_void_p)
exa_vol_avg = _sig("exa_vol_avg", C.c_int, C.c_void_p, dptr, dptr, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p)

MODEL_IDS = {("fcc", "powervoce"): EXA_FCC_VOCE, ("fcc", "powervocenl"): EXA_FCC_VOCE_NL, ("bcc", "powervoce"): EXA_BCC_VOCE,
             ("bcc", "powervocenl"): EXA_BCC_VOCE_NL, ("fcc", "mtsdd"): EXA_FCC_KMDD, ("bcc", "mtsdd"): EXA_BCC_KMDD}


class Context:
    """Owns one exa_ctx.  Mirrors how the reference's operator owns its model + integrator (mechanics_operator.cpp:49-225)."""

    def __init__(self, model, props, temp_k, order, nelems, assembly=EXA_ASSEMBLY_PA, integ=EXA_INTEG_FULL, device=-1):
        import numpy as np
        self._props = np.ascontiguousarray(props, dtype=np.float64)
        cfg = ExaConfig(model, len(self._props), self._props.ctypes.data_as(C.POINTER(C.c_double)), float(temp_k),
                        order, nelems, assembly, integ, device)
        err = C.c_int(0)
        self.h = exa_create(C.byref(cfg), C.byref(err))
        if not self.h:
            raise RuntimeError(f"exa_create failed with code {err.value} (is a HIP device present and the model/props valid?)")
        self.n = exa_nodes_per_elem(self.h)
        self.Q = exa_qpts_per_elem(self.h)
        self.E = nelems
        self.nstatev = exa_num_state_vars(self.h)

    def check(self, rc, what=""):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {exa_last_error(self.h).decode()}")
        return rc

    def shape_table(self):
        import numpy as np
        G = np.zeros(self.n * 3 * self.Q)
        W = np.zeros(self.Q)
        self.check(exa_shape_table(self.h, G.ctypes.data_as(C.POINTER(C.c_double)), W.ctypes.data_as(C.POINTER(C.c_double))))
        return G, W

    def close(self):
        if self.h:
            exa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- stand-alone driver (include/exaconstit_driver.h) ---------------------------------------------------
class ExaSynthConfig(C.Structure):
    _fields_ = [("N", C.c_int), ("bcc", C.c_int), ("slip", C.c_int), ("nprops", C.c_int), ("props", C.POINTER(C.c_double)),
                ("temp_k", C.c_double), ("quats", C.POINTER(C.c_double)), ("assembly", C.c_int), ("nrls", C.c_int), ("jacobi", C.c_int),
                ("newton_iter", C.c_int), ("newton_rel", C.c_double), ("newton_abs", C.c_double),
                ("krylov_iter", C.c_int), ("krylov_rel", C.c_double), ("krylov_abs", C.c_double),
                ("nsteps", C.c_int), ("dts", C.POINTER(C.c_double)), ("vz", C.c_double), ("order", C.c_int), ("bbar", C.c_int),
                ("nrev", C.c_int), ("rev_steps", C.POINTER(C.c_int))]


exa_rccl_unique_id = _sig("exa_rccl_unique_id", C.c_int, C.c_void_p)
exa_comm_unique_id = _sig("exa_comm_unique_id", C.c_int, C.c_void_p, C.c_int)
exa_device_identity = _sig("exa_device_identity", C.c_int, C.c_char_p, C.c_int)
exa_driver_comm_info = _sig("exa_driver_comm_info", C.c_int, C.c_void_p, C.POINTER(C.c_int))
exa_loopback_group_create = _sig("exa_loopback_group_create", C.c_int, C.c_int, C.c_void_p)
exa_loopback_group_destroy = _sig("exa_loopback_group_destroy", None, C.c_void_p)
exa_driver_comm_details = _sig("exa_driver_comm_details", C.c_int, C.c_void_p, C.POINTER(C.c_int64))
exa_driver_create = _sig("exa_driver_create", C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int)
exa_driver_create_synthetic = _sig("exa_driver_create_synthetic", C.c_void_p, C.POINTER(ExaSynthConfig), C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int)
exa_driver_destroy = _sig("exa_driver_destroy", None, C.c_void_p)
exa_driver_num_steps = _sig("exa_driver_num_steps", C.c_int, C.c_void_p)
exa_driver_local_qpts = _sig("exa_driver_local_qpts", C.c_int64, C.c_void_p)
exa_driver_local_dofs = _sig("exa_driver_local_dofs", C.c_int64, C.c_void_p)
exa_driver_step = _sig("exa_driver_step", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_int)
exa_driver_step_nocommit = _sig("exa_driver_step_nocommit", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_int)
exa_driver_commit_step = _sig("exa_driver_commit_step", C.c_int, C.c_void_p, C.c_char_p, C.c_int)
exa_driver_run = _sig("exa_driver_run", C.c_int, C.c_void_p, C.c_char_p, C.c_int)
exa_driver_get_avgs = _sig("exa_driver_get_avgs", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int)
exa_driver_get_stats = _sig("exa_driver_get_stats", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int)
exa_driver_get_timers = _sig("exa_driver_get_timers", None, C.c_void_p, C.POINTER(C.c_double))
exa_driver_reset_timers = _sig("exa_driver_reset_timers", None, C.c_void_p)
exa_driver_nfev_hist = _sig("exa_driver_nfev_hist", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_int)
exa_driver_get_qf_component = _sig("exa_driver_get_qf_component", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int)
exa_driver_nfev_hist_of = _sig("exa_driver_nfev_hist_of", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_int)
exa_driver_get_diagnostics = _sig("exa_driver_get_diagnostics", None, C.c_void_p, C.POINTER(C.c_int64))
exa_rccl_microbench = _sig("exa_rccl_microbench", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int)
exa_bootstrap_env = _sig("exa_bootstrap_env", C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
exa_bootstrap_reply_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p)
exa_bootstrap_gather_reply = _sig("exa_bootstrap_gather_reply", C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, exa_bootstrap_reply_fn, C.c_void_p,
                                  C.c_double, C.c_char_p, C.c_int)
exa_transport_from_identities = _sig("exa_transport_from_identities", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_int)
exa_bootstrap_bcast = _sig("exa_bootstrap_bcast", C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_char_p, C.c_int)
exa_bootstrap = _sig("exa_bootstrap", C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_char_p, C.c_int)
exa_driver_get_pcg_reduction = _sig("exa_driver_get_pcg_reduction", None, C.c_void_p, C.POINTER(C.c_double))
exa_driver_bench_prepare = _sig("exa_driver_bench_prepare", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_char_p, C.c_int)
exa_driver_bench_model = _sig("exa_driver_bench_model", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int)
exa_driver_bench_adapter_route = _sig("exa_driver_bench_adapter_route", C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int)
exa_driver_bench_pcg = _sig("exa_driver_bench_pcg", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int)
exa_choose_newton_cap = _sig("exa_choose_newton_cap", C.c_int, C.POINTER(C.c_int), C.c_double)
exa_choose_newton_caps = _sig("exa_choose_newton_caps", C.c_int, C.POINTER(C.c_int), C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int))
exa_options_query = _sig("exa_options_query", C.c_int, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_int)
exa_mesh_partition_query_order = _sig("exa_mesh_partition_query_order", C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int)
exa_mesh_partition_query = _sig("exa_mesh_partition_query", C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int)
exa_partition_query_boundary_first = _sig("exa_partition_query_boundary_first", C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p)
exa_partition_query = _sig("exa_partition_query", C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


class Driver:
    """SystemDriver of the reference (src/system_driver.hpp:101-143) running on the GPU behind the C ABI."""

    def __init__(self, handle, errbuf):
        if not handle:
            raise RuntimeError("driver creation failed: " + errbuf.value.decode())
        self.h = handle
        self._err = C.create_string_buffer(512)

    @classmethod
    def from_toml(cls, path, out_dir=".", rank=0, nranks=1, uid=None, jacobi=False, write_files=True):
        err = C.create_string_buffer(512)
        h = exa_driver_create(path.encode(), out_dir.encode(), rank, nranks, uid, int(jacobi), int(write_files), err, 512)
        return cls(h, err)

    @classmethod
    def synthetic(cls, N, props, quats, dts, bcc=False, slip=0, temp_k=298.0, assembly=0, nrls=False, jacobi=False,
                  newton=(25, 5e-5, 5e-10), krylov=(1000, 1e-7, 1e-27), vz=1.0e-3, rank=0, nranks=1, uid=None, order=1, bbar=False, reversals=()):
        import numpy as np
        props = np.ascontiguousarray(props, dtype=np.float64)
        quats = np.ascontiguousarray(quats, dtype=np.float64)
        dts = np.ascontiguousarray(dts, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        cfg = ExaSynthConfig(N, int(bcc), slip, len(props), props.ctypes.data_as(dp), temp_k, quats.ctypes.data_as(dp), assembly, int(nrls),
                             int(jacobi), newton[0], newton[1], newton[2], krylov[0], krylov[1], krylov[2], len(dts), dts.ctypes.data_as(dp), vz, order, int(bbar),
                             len(reversals), (C.c_int * max(len(reversals), 1))(*[int(r) for r in reversals]))
        err = C.create_string_buffer(512)
        h = exa_driver_create_synthetic(C.byref(cfg), rank, nranks, uid, err, 512)
        return cls(h, err)

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self._err.value.decode())
        return rc

    def step(self, ti, commit=True):
        rc = (exa_driver_step if commit else exa_driver_step_nocommit)(self.h, ti, self._err, 512)
        if rc < 0:
            raise RuntimeError(self._err.value.decode())
        return rc == 1

    def commit_step(self):
        self._chk(exa_driver_commit_step(self.h, self._err, 512))

    def run(self):
        rc = exa_driver_run(self.h, self._err, 512)
        if rc == -1000000:
            raise RuntimeError(self._err.value.decode())
        return rc

    def avgs(self, which, width, maxrows=4096):
        import numpy as np
        out = np.zeros((maxrows, width))
        rows = exa_driver_get_avgs(self.h, which, out.ctypes.data_as(C.POINTER(C.c_double)), maxrows)
        return out[:rows].copy()

    def stats(self, maxrows=4096):
        import numpy as np
        a = [np.zeros(maxrows, np.int32) for _ in range(3)]
        ip = C.POINTER(C.c_int)
        rows = exa_driver_get_stats(self.h, a[0].ctypes.data_as(ip), a[1].ctypes.data_as(ip), a[2].ctypes.data_as(ip), maxrows)
        return [x[:rows].copy() for x in a]

    def timers(self):
        import numpy as np
        t = np.zeros(5)
        exa_driver_get_timers(self.h, t.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(model_ms=t[0], krylov_ms=t[1], solve_ms=t[2], qpt_updates=int(t[3]), krylov_iters=int(t[4]))

    def diagnostics(self):
        import numpy as np
        o = np.zeros(4, np.int64)
        exa_driver_get_diagnostics(self.h, o.ctypes.data_as(C.POINTER(C.c_int64)))
        r = np.zeros(2)
        exa_driver_get_pcg_reduction(self.h, r.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(model_failed_points=int(o[0]), pcg_not_converged=int(o[1]), pcg_indefinite_iters=int(o[2]), pcg_last_flag=int(o[3]),
                    pcg_last_reduction=float(r[0]), pcg_worst_capped_reduction=float(r[1]))

    def bench_prepare(self, dts, perturb=1.0, advance=True):
        """Kinematic drive to the state the timed passes start from; advance=False keeps the virgin state (elastic first step, dt = dts[0])."""
        import numpy as np
        dts = np.ascontiguousarray(dts, dtype=np.float64)
        self._chk(exa_driver_bench_prepare(self.h, len(dts) if advance else 0, dts.ctypes.data_as(C.POINTER(C.c_double)), perturb, self._err, 512))

    def nfev_hist(self, which=1):
        """Histogram (64 bins) of the local-solver evaluation counts of the last constitutive launch (which = 1: end-of-step state) or,
        after a completed step, of the launch that step converged with (which = 0: begin-of-step state)."""
        import numpy as np
        h = np.zeros(64, dtype=np.int32)
        self._chk(exa_driver_nfev_hist_of(self.h, which, h.ctypes.data_as(C.POINTER(C.c_int)), self._err, 512))
        return h

    def comm_info(self):
        """(rank count the transport reports - ncclCommCount for RCCL -, transport name)"""
        o = (C.c_int * 2)()
        assert exa_driver_comm_info(self.h, o) == 0
        return int(o[0]), ("none", "rccl", "ipc", "loopback")[o[1]]

    def comm_details(self):
        out = (C.c_int64 * 8)()
        assert exa_driver_comm_details(self.h, out) == 0
        return {"elements": int(out[0]), "boundary_block_elements": int(out[1]), "neighbours": int(out[2]), "halo_bytes_per_exchange": 8 * int(out[3]),
                "halo_overlap": bool(out[4])}

    def reset_timers(self):
        exa_driver_reset_timers(self.h)

    def qf_component(self, which, comp):
        """One component of a quadrature function ([element][point]); which: 0/1 begin/end state, 2/3 begin/end stress."""
        import numpy as np
        out = np.zeros(exa_driver_local_qpts(self.h))
        self._chk(exa_driver_get_qf_component(self.h, which, comp, out.ctypes.data_as(C.c_void_p), self._err, 512))
        return out

    def bench_model(self, steps):
        import numpy as np
        o = np.zeros(3)
        self._chk(exa_driver_bench_model(self.h, steps, o.ctypes.data_as(C.POINTER(C.c_double)), self._err, 512))
        return dict(loop_ms=o[0], kernel_ms=o[1], failed=int(o[2]))

    def bench_adapter_route(self, steps, iters):
        """The calls the MFEM adapters make (AOS exa_model_setup, exa_grad_setup, E-vector exa_grad_apply between L->E and E->L) at this driver's state."""
        import numpy as np
        o = np.zeros(24)
        self._chk(exa_driver_bench_adapter_route(self.h, steps, iters, o.ctypes.data_as(C.POINTER(C.c_double)), self._err, 512))
        return dict(model_ms=o[0], pass_ms=o[1], geometry_ms=o[2], grad_setup_ms=o[3], grad_apply_ms=o[4], action_ms=o[5], stress_rel_diff=o[6], state_rel_diff=o[7],
                    action_rel_diff=o[8], failed=int(o[9]), driver_route_model_ms=o[10], driver_route_apply_ms=o[11], aos_staging=bool(o[12]), nfev_differing=int(o[13]),
                    lvec_model_ms=o[14], lvec_apply_ms=o[15], lvec_stress_rel_diff=o[16], lvec_grad_setup_ms=o[17], lvec_action_rel_diff=o[18], lvec_residual_ms=o[19],
                    lvec_records_model_ms=o[20], lvec_records_stress_rel_diff=o[21], lvec_records_action_rel_diff=o[22])

    def bench_pcg(self, iters):
        import numpy as np
        o = np.zeros(3)
        self._chk(exa_driver_bench_pcg(self.h, iters, o.ctypes.data_as(C.POINTER(C.c_double)), self._err, 512))
        return dict(pcg_ms=o[0], iters=int(o[1]), apply_ms=o[2])

    def close(self):
        if self.h:
            exa_driver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
