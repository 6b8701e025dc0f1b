// C entry points of the stand-alone driver (used by the `mechanics` executable, the tests and bench.py through ctypes).
// They expose the reference's run-time surface: `mechanics -opt options.toml` (reference src/mechanics_driver.cpp:139-144),
// the per-step loop, the avg_* outputs (reference src/system_driver.cpp:444-553) and the timing regions the reference marks
// with Caliper (ecmech_kernel, krylov_solver: src/mechanics_ecmech.cpp:237-257, src/mechanics_solver.cpp:99-103).
#include "driver.hpp"
#include "roctx.hpp"
#include "../../../include/exaconstit_driver.h"
#include <unistd.h>
#include <cmath>
#include <cstring>
#include <string>

using namespace exa_host;

struct exa_driver {
   std::unique_ptr<SystemDriver> sd;
   std::vector<double> v_kin;   // kinematic velocity field of the bench (host copy)
};

namespace {
void set_err(char* err, int n, const std::string& m) { if (err && n > 0) { std::strncpy(err, m.c_str(), n - 1); err[n - 1] = 0; } }
}

extern "C" {

// Transport from the identity of the ranks' devices (96-byte records: host name[64], PCI bus id[32]).  RCCL needs one device per rank; the
// shared-device transport (POSIX shm + hipIpc, host/driver.hip) needs every rank on ONE host.  Counting ranks against visible devices cannot
// tell the two apart (16 ranks on 2 x 8 GPUs, or one visible device per rank under srun --gpus-per-task=1), identities can.
int exa_transport_from_identities(const void* ids, int nranks, char* err, int errlen) {
   const char* t = (const char*)ids;
   bool one_host = true, shared = false;
   for (int i = 0; i < nranks; i++) {
      if (std::strncmp(t, t + 96 * (size_t)i, 64) != 0) one_host = false;
      for (int j = 0; j < i; j++) if (std::memcmp(t + 96 * (size_t)i, t + 96 * (size_t)j, 96) == 0) shared = true;
   }
   const char* e = std::getenv("EXA_TRANSPORT");
   const std::string force = e ? e : "";
   if (force == "ipc" || (force != "rccl" && shared)) {
      if (!one_host) { set_err(err, errlen, "exa_bootstrap: the shared-device transport serves ranks of one host only; this group spans hosts (one GPU per rank and RCCL is the multi-node configuration)"); return -1; }
      return 2;
   }
   if (shared) { set_err(err, errlen, "exa_bootstrap: EXA_TRANSPORT=rccl, but two ranks sit on the same device (RCCL refuses that)"); return -1; }
   return 1;
}

namespace {
int bootstrap_reply(const void* all, int nranks, int nbytes, void* reply, int reply_bytes, void* user) {
   if (nbytes != 96 || reply_bytes != 132) return -1;
   char* err = (char*)user;
   const int kind = exa_transport_from_identities(all, nranks, err, 256);
   if (kind < 0) { std::memset(reply, 0, 132); int32_t k = -1; std::memcpy((char*)reply + 128, &k, 4); std::snprintf((char*)reply, 128, "%s", err); return 0; }   // every rank learns why
   try { if (kind == 2) Comm::ipc_unique_id(reply); else Comm::get_unique_id(reply); }
   catch (const std::exception& e) { std::snprintf(err, 256, "exa_bootstrap: %s", e.what()); return -1; }
   const int32_t k = kind; std::memcpy((char*)reply + 128, &k, 4);
   return 0;
}
}

// rank / size from the launcher's environment, device = local rank mod visible devices, then ONE TCP rendez-vous (host/bootstrap.cpp): every
// rank sends the identity of its device, rank 0 decides the transport from them and answers with the unique id of that transport - a RCCL
// id, or the shared-device transport's when ranks of one host share a GPU.  Everything `mechanics` needs before exa_driver_create (reference:
// MPI_Init + MPI_Comm_rank/size, src/mechanics_driver.cpp:119-150)
int exa_bootstrap(int* rank, int* nranks, void* uid128, char* err, int errlen) {
   int lr = 0;
   if (exa_bootstrap_env(rank, nranks, &lr) != 0) { set_err(err, errlen, "exa_bootstrap: inconsistent rank / size in the environment"); return -1; }
   int nd = 0;
   if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { set_err(err, errlen, "exa_bootstrap: no HIP device"); return -1; }
   if (hipSetDevice(lr % nd) != hipSuccess) { set_err(err, errlen, "exa_bootstrap: hipSetDevice failed"); return -1; }
   std::memset(uid128, 0, 128);
   if (*nranks > 1) {
      char mine[96]; std::memset(mine, 0, sizeof(mine));
      if (::gethostname(mine, 63) != 0) std::snprintf(mine, 64, "unknown-host");
      if (exa_device_identity(mine + 64, 32) != 0) std::snprintf(mine + 64, 32, "device-index-%d-of-%d", lr % nd, nd);
      char reply[132]; char cb_err[256] = "";
      double tmo = 60.0; if (const char* t = std::getenv("EXA_RENDEZVOUS_TIMEOUT")) { const double v = std::atof(t); if (v > 0) tmo = v; }
      if (exa_bootstrap_gather_reply(*rank, *nranks, mine, 96, reply, 132, bootstrap_reply, cb_err, tmo, err, errlen) != 0) {
         if (cb_err[0]) set_err(err, errlen, cb_err);
         return -1;
      }
      int32_t kind = 0; std::memcpy(&kind, reply + 128, 4);
      if (kind < 0) { reply[127] = 0; set_err(err, errlen, reply); return -1; }
      std::memcpy(uid128, reply, 128);
      if (*rank == 0) std::fprintf(stderr, "exa_bootstrap: %d ranks, transport %s (decided from host names and PCI bus ids)\n", *nranks, kind == 2 ? "ipc (ranks share a device)" : "rccl");
   }
   return 0;
}

// latency floor of the two RCCL calls of a PCG iteration on this device (one-rank communicator): out2 = { us per 16-byte all-reduce,
// us per grouped send/recv of n doubles to the own rank }; input of the scaling prediction in DESIGN.md section 7
int exa_rccl_microbench(int iters, int n, double* out2, char* err, int errlen) {
   try {
      Comm c; c.init(0, 1, nullptr, /*force_rccl=*/true);   // (no environment change: later drivers of the process keep their own setting)
      c.microbench(iters, n, out2, out2 + 1);
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

int exa_rccl_unique_id(void* out128) {
   try { Comm::get_unique_id(out128); return 0; } catch (const std::exception& e) { std::fprintf(stderr, "exa_rccl_unique_id: %s\n", e.what()); return -1; }
}

// PCI bus id ("0000:c1:00.0") of the calling thread's current device: what the launchers compare to tell one rank per physical GPU from several
// ranks on one GPU, whatever device numbering each rank sees
int exa_device_identity(char* out, int len) {
   int dev = 0;
   if (!out || len < 16 || hipGetDevice(&dev) != hipSuccess) return -1;
   return hipDeviceGetPCIBusId(out, len, dev) == hipSuccess ? 0 : -1;
}
// the id rank 0 hands out for a group of `nranks` when the caller has NOT compared device identities itself: a RCCL unique id, or - with
// EXA_TRANSPORT=ipc, or when the ranks of this node outnumber its visible devices - the id of the inter-process transport of host/driver.hip
// (class Comm).  bench.py compares identities and sets EXA_TRANSPORT; exa_bootstrap decides from identities (exa_transport_from_identities).
int exa_comm_unique_id(void* out128, int nranks) {
   try { if (Comm::want_ipc_transport(nranks)) Comm::ipc_unique_id(out128); else Comm::get_unique_id(out128); return 0; }
   catch (const std::exception& e) { std::fprintf(stderr, "exa_comm_unique_id: %s\n", e.what()); return -1; }
}
// out2 = { ranks the transport itself reports (ncclCommCount for RCCL), kind: 0 none (one rank), 1 rccl, 2 ipc (shared device), 3 in-process loopback }
int exa_driver_comm_info(exa_driver* d, int* out2) {
   try {
      const Comm& c = d->sd->comm; const std::string t = c.transport();
      out2[0] = c.reported_ranks(); out2[1] = t == "rccl" ? 1 : (t == "ipc" ? 2 : (t == "loopback" ? 3 : 0));
      return 0;
   } catch (const std::exception& e) { std::fprintf(stderr, "exa_driver_comm_info: %s\n", e.what()); return -1; }
}

// out8 = { local elements, elements in blocks that touch shared nodes, neighbours, doubles sent (= received) per halo exchange, halo exchange
//          overlapped with the interior blocks (0 / 1), transport kind (exa_driver_comm_info), ranks the transport reports, 0 }
int exa_driver_comm_details(exa_driver* d, int64_t* out8) {
   try {
      const SystemDriver& sd = *d->sd; int info[2] = { 1, 0 };
      (void)exa_driver_comm_info(d, info);
      out8[0] = sd.part.E; out8[1] = sd.part.E_bdr; out8[2] = (int64_t)sd.part.nbrs.size(); out8[3] = (int64_t)sd.comm.halo_dofs();
      out8[4] = d->sd->oper().halo_overlap() ? 1 : 0; out8[5] = info[1]; out8[6] = info[0]; out8[7] = 0;
      return 0;
   } catch (const std::exception& e) { std::fprintf(stderr, "exa_driver_comm_details: %s\n", e.what()); return -1; }
}

int exa_loopback_group_create(int nranks, void* out128) { try { Comm::loopback_create(nranks, out128); return 0; } catch (...) { return -1; } }
void exa_loopback_group_destroy(const void* id128) { Comm::loopback_destroy(id128); }

exa_driver* exa_driver_create(const char* toml_path, const char* out_dir, int rank, int nranks, const void* uid, int jacobi, int write_files, char* err, int errlen) {
   try {
      ExaOptions opt; opt.parse_options(toml_path);
      auto d = new exa_driver();
      d->sd.reset(new SystemDriver(opt, rank, nranks, uid));
      d->sd->out_dir = out_dir ? out_dir : "."; d->sd->write_files = write_files != 0;
      d->sd->precond = jacobi ? Precond::JACOBI : Precond::IDENTITY; d->sd->oper().precond = d->sd->precond;
      return d;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return nullptr; }
}

exa_driver* exa_driver_create_synthetic(const exa_synth_config* c, int rank, int nranks, const void* uid, char* err, int errlen) {
   try {
      ExaOptions opt;
      opt.temp_k = c->temp_k;
      opt.xtal = c->bcc ? XtalType::BCC : XtalType::FCC;
      opt.slip = c->slip == 0 ? SlipType::POWERVOCE : (c->slip == 1 ? SlipType::POWERVOCENL : SlipType::MTSDD);
      for (int i = 0; i < 3; i++) { opt.ncuts[i] = c->N; opt.length[i] = 1.0; }
      opt.assembly = c->assembly == 0 ? Assembly::PA : Assembly::EA;
      opt.order = c->order > 0 ? c->order : 1; opt.integ_model = c->bbar ? "BBAR" : "FULL";
      opt.nl_solver = c->nrls ? NLSolver::NRLS : NLSolver::NR;
      opt.newton_iter = c->newton_iter; opt.newton_rel = c->newton_rel; opt.newton_abs = c->newton_abs;
      opt.krylov_iter = c->krylov_iter; opt.krylov_rel = c->krylov_rel; opt.krylov_abs = c->krylov_abs;
      opt.dt_cust = true; opt.nsteps = c->nsteps; opt.cust_dt.assign(c->dts, c->dts + c->nsteps);
      // uniaxial z-tension with three symmetry planes (reference test/data/voce_pa.toml [BCs])
      BCEntry bc; bc.step = 1; bc.ids = { 1, 2, 3, 4 }; bc.comps = { 3, 1, 2, 3 }; bc.vals = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, c->vz };
      opt.bcs.push_back(bc);
      double vz = c->vz;
      for (int i = 0; i < c->nrev; i++) {   // cyclic loading (reference test/data/voce_full_cyclic.toml: changing_ess_bcs with update_steps)
         vz = -vz;
         BCEntry rb = bc; rb.step = c->rev_steps[i]; rb.vals.back() = vz;
         opt.bcs.push_back(rb);
      }
      std::vector<double> props(c->props, c->props + c->nprops);
      const size_t Eg = (size_t)c->N * c->N * c->N;
      std::vector<double> quats(c->quats, c->quats + 4 * Eg);
      auto d = new exa_driver();
      d->sd.reset(new SystemDriver(opt, props, quats, rank, nranks, uid));
      d->sd->write_files = false;
      d->sd->precond = c->jacobi ? Precond::JACOBI : Precond::IDENTITY; d->sd->oper().precond = d->sd->precond;
      return d;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return nullptr; }
}

void exa_driver_destroy(exa_driver* d) { delete d; }

int exa_driver_num_steps(exa_driver* d) { return d->sd->options().nsteps; }
int64_t exa_driver_local_qpts(exa_driver* d) { return (int64_t)d->sd->part.E * d->sd->part.n; }
int64_t exa_driver_local_dofs(exa_driver* d) { return (int64_t)d->sd->part.NN * 3; }

int exa_driver_step(exa_driver* d, int ti, char* err, int errlen) {
   try { return d->sd->Step(ti) ? 1 : 0; } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// the step's Newton/PCG solve without the end-of-step update (bench: the timed constitutive passes that follow repeat the step's converged launch)
int exa_driver_step_nocommit(exa_driver* d, int ti, char* err, int errlen) {
   try { return d->sd->Step(ti, false) ? 1 : 0; } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// ... and its end-of-step update afterwards (bench: the solve goes on from the step whose converged launch was timed)
int exa_driver_commit_step(exa_driver* d, char* err, int errlen) {
   try { d->sd->CommitStep(); return 0; } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

int exa_driver_run(exa_driver* d, char* err, int errlen) {
   try { return d->sd->RunAll(); } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1000000; }
}

// which: 0 stress (6), 1 def_grad (9), 2 pl_work (1), 3 dp_tensor (6); returns rows copied
int exa_driver_get_avgs(exa_driver* d, int which, double* out, int maxrows) {
   const std::vector<double>* v = which == 0 ? &d->sd->avg_stress : (which == 1 ? &d->sd->avg_def_grad : (which == 2 ? &d->sd->avg_pl_work : &d->sd->avg_dp_tensor));
   const int w = which == 0 ? 6 : (which == 1 ? 9 : (which == 2 ? 1 : 6));
   int rows = (int)(v->size() / w); if (rows > maxrows) rows = maxrows;
   std::memcpy(out, v->data(), sizeof(double) * rows * w);
   return rows;
}

int exa_driver_get_stats(exa_driver* d, int* newton, int* krylov, int* model_calls, int maxrows) {
   int rows = (int)d->sd->stats.size(); if (rows > maxrows) rows = maxrows;
   for (int i = 0; i < rows; i++) { newton[i] = d->sd->stats[i].newton_iters; krylov[i] = d->sd->stats[i].krylov_iters; model_calls[i] = d->sd->stats[i].model_calls; }
   return rows;
}

// out[0] ms in the fused constitutive kernel, out[1] ms in PCG, out[2] ms in Solve (+SolveInit), out[3] qpt updates, out[4] PCG iterations
void exa_driver_get_timers(exa_driver* d, double* out) {
   d->sd->oper().FlushModelTimers();
   const Timers& t = d->sd->oper().timers;
   out[0] = t.t_model_ms; out[1] = t.t_krylov_ms; out[2] = t.t_solve_ms; out[3] = (double)t.qpt_updates; out[4] = (double)t.krylov_iters;
}
void exa_driver_reset_timers(exa_driver* d) { d->sd->oper().FlushModelTimers(); d->sd->oper().timers = Timers(); }   // pending event pairs belong to the old totals
// out[0] quadrature points whose local solve failed (sum over all constitutive launches, this rank), out[1] linear solves that
// did not converge, out[2] PCG iterations that saw (Ad, d) < 0, out[3] flag of the last PCG solve (1 converged, 2 max_iter, -1 den == 0)
void exa_driver_get_diagnostics(exa_driver* d, int64_t* out) {
   d->sd->oper().ReadModelStatus();
   out[0] = d->sd->oper().model_fail_total; out[1] = d->sd->cg_not_converged; out[2] = d->sd->cg_indefinite_iters; out[3] = d->sd->last_cg_flag;
}

// out[0] = sqrt((r, M^-1 r) / (r0, M^-1 r0)) reached by the last PCG solve, out[1] = the worst value among the solves that stopped at max_iter
void exa_driver_get_pcg_reduction(exa_driver* d, double* out2) { out2[0] = d->sd->last_cg_reduction; out2[1] = d->sd->worst_capped_cg_reduction; }

// which: 1 = end-of-step state of the last constitutive launch, 0 = begin-of-step state (after a completed step: that step's converged launch)
int exa_driver_nfev_hist_of(exa_driver* d, int which, int* hist64, char* err, int errlen) {
   try {
      NonlinearMechOperator& op = d->sd->oper();
      const double* st = which == 0 ? op.matVars0.p : op.matVars1.p;
      if (exa_model_nfev_hist(op.GetModel()->ctx(), st, hist64, op.stream()) != EXA_OK) throw std::runtime_error(exa_last_error(op.GetModel()->ctx()));
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}
int exa_driver_nfev_hist(exa_driver* d, int* hist64, char* err, int errlen) { return exa_driver_nfev_hist_of(d, 1, hist64, err, errlen); }

int exa_driver_get_qf_component(exa_driver* d, int which, int comp, double* out, char* err, int errlen) {
   try {
      NonlinearMechOperator& op = d->sd->oper();
      const DevBuf<double>* b = which == 0 ? &op.matVars0 : (which == 1 ? &op.matVars1 : (which == 2 ? &op.stress0 : &op.stress1));
      const int W = which < 2 ? 28 : 6;
      if (comp < 0 || comp >= W) throw std::runtime_error("exa_driver_get_qf_component: component out of range");
      EXA_HC(hipStreamSynchronize(op.stream()));
      const std::vector<double> h = b->to_host();
      const exa_ctx* ctx = op.GetModel()->ctx();
      const int Q = exa_qpts_per_elem(ctx); const int64_t E = d->sd->part.E;
      const bool blk = exa_get_quadrature_layout(ctx) == EXA_QLAYOUT_EB64;
      for (int64_t e = 0; e < E; e++)
         for (int q = 0; q < Q; q++)
            out[e * Q + q] = blk ? h[((((e >> 6) * Q + q) * (int64_t)W + comp) << 6) + (e & 63)] : h[(int64_t)W * (q + (int64_t)Q * e) + comp];
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// ---- benchmark hooks --------------------------------------------------------------------------------------------------
// Kinematically drive the RVE into the plastic regime: nodal velocity v = L0 x (+ seeded perturbation), `nsteps` constitutive
// passes with state/coordinate updates and no equilibrium solve (SURVEY 8(d) kernel micro-benchmark).
int exa_driver_bench_prepare(exa_driver* d, int nsteps, const double* dts, double perturb, char* err, int errlen) {
   try {
      SystemDriver& sd = *d->sd; NonlinearMechOperator& op = sd.oper();
      const Partition& part = sd.part; const int nn = part.NN;
      d->v_kin.assign((size_t)3 * nn, 0.0);
      const double L0[3][3] = { { -0.45e-3, 0.3e-4, -0.2e-4 }, { -0.3e-4, -0.45e-3, 0.1e-4 }, { 0.2e-4, -0.1e-4, 1.0e-3 } };
      for (int g = 0; g < nn; g++) {
         const double x[3] = { part.X[g], part.X[g + nn], part.X[g + 2 * (size_t)nn] };
         // deterministic perturbation from the GLOBAL node coordinates so that duplicated interface nodes agree across ranks
         for (int c = 0; c < 3; c++) {
            const double ph = std::sin(12.9898 * x[0] * part.N[0] + 78.233 * x[1] * part.N[1] + 37.719 * x[2] * part.N[2] + 4.0 * c) * 43758.5453;
            const double u = 2.0 * (ph - std::floor(ph)) - 1.0;
            d->v_kin[g + (size_t)nn * c] = L0[c][0] * x[0] + L0[c][1] * x[1] + L0[c][2] * x[2] + perturb * 1.0e-4 / part.N[0] * u;
         }
      }
      sd.v_sol.upload(d->v_kin);
      for (int i = 0; i < nsteps; i++) {
         op.SetDt(dts[i]);
         op.Setup<true>(sd.v_sol.p);
         op.UpdateModel(); op.SwapCoords();
      }
      op.SetDt(dts[nsteps > 0 ? nsteps - 1 : 0]);   // nsteps == 0: keep the virgin state, passes are then the elastic first step with dt = dts[0]
      EXA_HC(hipStreamSynchronize(op.stream()));
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// `steps` constitutive passes (restriction + geometric factors + fused model kernel) from the fixed begin-of-step state, as the
// residual evaluations of one Newton solve do.  out[0] = wall ms of the whole loop (HIP events), out[1] = ms inside the fused
// constitutive kernel only, out[2] = non-converged points of the last pass.
int exa_driver_bench_model(exa_driver* d, int steps, double* out, char* err, int errlen) {
   try {
      SystemDriver& sd = *d->sd; NonlinearMechOperator& op = sd.oper();
      hipStream_t s = op.stream();
      hipEvent_t e0, e1; EXA_HC(hipEventCreate(&e0)); EXA_HC(hipEventCreate(&e1));
      op.FlushModelTimers();
      const double t0 = op.timers.t_model_ms;
      const std::string rname = "timed_region_model[passes=" + std::to_string(steps) + "]";   // the bench's warm-up / elastic / plastic loops differ in length
      ProfRegion prof(rname.c_str());
      EXA_HC(hipEventRecord(e0, s));
      for (int i = 0; i < steps; i++) op.Setup<true>(sd.v_sol.p);
      EXA_HC(hipEventRecord(e1, s)); EXA_HC(hipEventSynchronize(e1));
      float ms = 0; EXA_HC(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      op.FlushModelTimers(); op.ReadModelStatus();
      out[0] = ms; out[1] = op.timers.t_model_ms - t0; out[2] = (double)op.model_fail;
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// Fixed-length PCG: residual + Jacobian set-up at the prepared state, then exactly `iters` PCG iterations (tolerances disabled).
// out[0] = ms of the PCG loop, out[1] = iterations executed, out[2] = ms of `iters` back-to-back gradient-action launches alone.
int exa_driver_bench_pcg(exa_driver* d, int iters, double* out, char* err, int errlen) {
   try {
      SystemDriver& sd = *d->sd; NonlinearMechOperator& op = sd.oper();
      hipStream_t s = op.stream(); const int nd = op.Height();
      DevBuf<double> r(nd), c(nd);
      op.Mult(sd.v_sol.p, r.p);
      op.GetGradient();
      ExaOptions& o = const_cast<ExaOptions&>(sd.options());
      const double rel = o.krylov_rel, ab = o.krylov_abs; const int mi = o.krylov_iter;
      o.krylov_rel = 0.0; o.krylov_abs = 0.0; o.krylov_iter = iters;
      const double t0 = op.timers.t_krylov_ms;
      const std::string rname = "timed_region_pcg[iters=" + std::to_string(iters) + "]";
      ProfRegion prof(rname.c_str());
      const int it = sd.CGSolve(r.p, c.p);
      out[0] = op.timers.t_krylov_ms - t0; out[1] = it;
      o.krylov_rel = rel; o.krylov_abs = ab; o.krylov_iter = mi;
      sd.drop_cg_graph();   // the captured chunk refers to the local solution buffer c
      hipEvent_t e0, e1; EXA_HC(hipEventCreate(&e0)); EXA_HC(hipEventCreate(&e1));
      EXA_HC(hipEventRecord(e0, s));
      for (int i = 0; i < iters; i++) exa_grad_apply_lvec(op.GetModel()->ctx(), r.p, c.p, op.ess_mask.p, s);
      EXA_HC(hipEventRecord(e1, s)); EXA_HC(hipEventSynchronize(e1));
      float ms = 0; EXA_HC(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      out[2] = ms;
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// The drop-in route at the driver's current state: exactly the calls the MFEM adapters of include/exaconstit_mfem_adapters.hpp make
// (HipExaModel::ModelSetup -> exa_model_setup on the reference's (vdim, Q, E) quadrature functions with a Jacobian field and a velocity
// E-vector; HipExaNLFIntegrator::AssembleGradPA -> exa_grad_setup; AddMultGradPA -> exa_grad_apply on E-vectors between the element
// restriction and its transpose, which MFEM's nonlinear form performs around it: spec reference src/mechanics_operator_ext.cpp:149-157),
// on a second context in the AOS layout that is given this driver's begin-of-step state, coordinates and velocity.  `steps` timed
// constitutive launches, `iters` timed gradient actions; results compared with what the driver's own route produced from the same state.
//   out[0]  ms per exa_model_setup launch                     out[1]  ms per pass = velocity L->E + exa_model_setup
//   out[2]  ms of coordinates L->E + exa_jacobians            out[3]  ms per exa_grad_setup
//   out[4]  ms per exa_grad_apply (E-vector kernel alone)     out[5]  ms per action = L->E + exa_grad_apply + E->L
//   out[6]  max |stress1 - driver's| / max |stress1|          out[7]  max |state1 - driver's| / max |state1|
//   out[8]  max |K x - driver's K x| / max |K x|              out[9]  non-converged points
//   out[10] ms per launch of the driver's own route at this state (same loop)      out[11] ms of the driver's own gradient action
//   out[12] AOS staging through LDS in effect (1 / 0)         out[13] points whose evaluation count (state slot 3, left out of out[7]) differs
// and the L-vector pair (HipExaModelLVec / HipExaNLFIntegratorLVec) on the same context and quadrature functions:
//   out[14] ms per exa_model_setup_lvec launch (gathers + Jacobians + update, AOS rows staged)      out[15] ms per exa_grad_apply_lvec (gather + action + scatter)
//   out[16] max |stress1 - driver's| / max |stress1|          out[17] ms per exa_grad_setup with the compact tangent form
//   out[18] max |K x - driver's K x| / max |K x|              out[19] ms per exa_residual_lvec
//   out[20] ms per exa_model_setup_lvec_records on the reference layout (records instead of the tangent field: no exa_grad_setup)
//   out[21] its max |stress1 - driver's| / max |stress1|      out[22] max |K x - driver's K x| / max |K x| of the action on its records      out[23] 0
int exa_driver_bench_adapter_route(exa_driver* d, int steps, int iters, double* out, char* err, int errlen) {
   try {
      SystemDriver& sd = *d->sd; NonlinearMechOperator& op = sd.oper();
      if (sd.part.p != 1 || sd.comm.nranks != 1) throw std::runtime_error("exa_driver_bench_adapter_route: one rank, p = 1");
      hipStream_t s = op.stream(); const int nd = op.Height(); const int nn = nd / 3;
      const int64_t E = sd.part.E; const int Q = 8; const int64_t P = E * Q;
      for (int i = 0; i < 24; i++) out[i] = 0.0;
      // the driver's own residual evaluation + gradient data at this state: the outputs the adapter route is compared with
      DevBuf<double> r(nd), yC(nd), yA(nd);
      op.Mult(sd.v_sol.p, r.p);
      op.GetGradient();
      exa_ctx* ctxC = op.GetModel()->ctx();
      const bool blocked = exa_get_quadrature_layout(ctxC) == EXA_QLAYOUT_EB64;
      auto timed = [&](int n, auto&& body) {
         hipEvent_t e0, e1; EXA_HC(hipEventCreate(&e0)); EXA_HC(hipEventCreate(&e1));
         EXA_HC(hipEventRecord(e0, s));
         for (int i = 0; i < n; i++) body();
         EXA_HC(hipEventRecord(e1, s)); EXA_HC(hipEventSynchronize(e1));
         float ms = 0; EXA_HC(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
         return (double)ms / (n > 0 ? n : 1);
      };
      {  ProfRegion prof("adapter_route_reference_launches");
         op.Setup<true>(sd.v_sol.p);
         out[10] = timed(steps, [&] { op.Setup<true>(sd.v_sol.p); });
         op.FlushModelTimers();
         EXA_HC(hipMemsetAsync(yC.p, 0, sizeof(double) * nd, s));
         out[11] = timed(iters, [&] { if (exa_grad_apply_lvec(ctxC, r.p, yC.p, nullptr, s) != EXA_OK) throw std::runtime_error(exa_last_error(ctxC)); });
         EXA_HC(hipMemsetAsync(yC.p, 0, sizeof(double) * nd, s));
         if (exa_grad_apply_lvec(ctxC, r.p, yC.p, nullptr, s) != EXA_OK) throw std::runtime_error(exa_last_error(ctxC));
      }
      // second context: the reference's quadrature-function layout, E-vector entry points
      exa_config cfg = op.cfg_used; cfg.props = op.props.data(); cfg.nprops = (int)op.props.size(); cfg.assembly = EXA_ASSEMBLY_PA; cfg.integ = EXA_INTEG_FULL;
      int ec = 0; exa_ctx* ctxA = exa_create(&cfg, &ec);
      if (!ctxA) throw std::runtime_error("exa_driver_bench_adapter_route: exa_create failed");
      struct Guard { exa_ctx* c; ~Guard() { exa_destroy(c); } } guard{ ctxA };
      auto chk = [&](int rc, const char* what) { if (rc < 0) throw std::runtime_error(std::string(what) + ": " + exa_last_error(ctxA)); };
      chk(exa_set_connectivity(ctxA, op.conn.p, nn), "exa_set_connectivity");
      chk(exa_set_newton_cap_auto(ctxA, 1, 0.0), "exa_set_newton_cap_auto");   // as the adapters' constructors do
      DevBuf<double> sv0((size_t)28 * P), s0((size_t)6 * P), sv1((size_t)28 * P), s1((size_t)6 * P), cm((size_t)36 * P), J((size_t)9 * P), tmp((size_t)28 * P);
      DevBuf<double> elx((size_t)24 * E), elv((size_t)24 * E), ely((size_t)24 * E), sc(3);
      auto to_aos = [&](int W, const DevBuf<double>& src, DevBuf<double>& dst) {
         if (blocked) vk_qf_eb64_to_aos(W, Q, E, src.p, dst.p, s);
         else EXA_HC(hipMemcpyAsync(dst.p, src.p, sizeof(double) * W * P, hipMemcpyDeviceToDevice, s));
      };
      to_aos(28, op.matVars0, sv0); to_aos(6, op.stress0, s0);
      const double dt = op.dt();
      // geometric factors of the end-of-step configuration (MFEM: GetGeometricFactors + the re-layout of exa_jacobians_from_geom)
      out[2] = timed(2, [&] { chk(exa_restrict(ctxA, op.x_cur.p, elx.p, s), "exa_restrict"); chk(exa_jacobians(ctxA, elx.p, J.p, s), "exa_jacobians"); });
      chk(exa_restrict(ctxA, sd.v_sol.p, elv.p, s), "exa_restrict");
      auto model = [&] { chk(exa_model_setup(ctxA, dt, J.p, elv.p, s0.p, sv0.p, s1.p, sv1.p, cm.p, s), "exa_model_setup"); };
      model(); model();
      {  ProfRegion prof(("adapter_route_model[passes=" + std::to_string(steps) + "]").c_str());
         out[0] = timed(steps, model); }
      out[1] = timed(steps, [&] { chk(exa_restrict(ctxA, sd.v_sol.p, elv.p, s), "exa_restrict"); model(); });
      out[9] = (double)exa_model_status(ctxA, s);
      out[12] = (double)exa_get_aos_staging(ctxA);
      unsigned long long differing = 0;
      auto rel_diff = [&](int64_t n, const double* a, const double* b, int W = 0, int skip = -1) {
         vk_max_abs_diff(n, a, b, sc.p, s, W, skip);
         double h[3]; EXA_HC(hipMemcpyAsync(h, sc.p, sizeof(h), hipMemcpyDeviceToHost, s)); EXA_HC(hipStreamSynchronize(s));
         std::memcpy(&differing, &h[2], sizeof(differing));
         return h[1] > 0.0 ? h[0] / h[1] : h[0];
      };
      to_aos(6, op.stress1, tmp); out[6] = rel_diff(6 * P, s1.p, tmp.p);
      to_aos(28, op.matVars1, tmp); out[7] = rel_diff(28 * P, sv1.p, tmp.p, 28, 3); out[13] = (double)differing;   // slot 3 = the local solver's evaluation count
      // AssembleGradPA (HipExaNLFIntegrator: the compact tangent + adj(J) records after its defect check)
      {  double defect = 1.0; chk(exa_set_tangent_form(ctxA, EXA_TANGENT_DEV5_BULK_GEO), "exa_set_tangent_form");
         chk(exa_grad_tangent_defect(ctxA, cm.p, &defect, s), "exa_grad_tangent_defect");
         if (!(defect < 1e-11)) throw std::runtime_error("exa_driver_bench_adapter_route: tangent not of the compact form"); }
      chk(exa_grad_setup(ctxA, dt, J.p, cm.p, s), "exa_grad_setup");
      {  ProfRegion prof("adapter_route_grad_setup");
         out[3] = timed(3, [&] { chk(exa_grad_setup(ctxA, dt, J.p, cm.p, s), "exa_grad_setup"); }); }
      // AddMultGradPA between the element restriction and its transpose
      chk(exa_restrict(ctxA, r.p, elx.p, s), "exa_restrict");
      ely.zero(s);
      chk(exa_grad_apply(ctxA, elx.p, ely.p, s), "exa_grad_apply");
      {  ProfRegion prof(("adapter_route_grad_apply[iters=" + std::to_string(iters) + "]").c_str());
         out[4] = timed(iters, [&] { chk(exa_grad_apply(ctxA, elx.p, ely.p, s), "exa_grad_apply"); }); }
      auto action = [&] {
         chk(exa_restrict(ctxA, r.p, elx.p, s), "exa_restrict");
         ely.zero(s);
         chk(exa_grad_apply(ctxA, elx.p, ely.p, s), "exa_grad_apply");
         chk(exa_restrict_transpose_add(ctxA, ely.p, yA.p, s), "exa_restrict_transpose_add");
      };
      {  ProfRegion prof(("adapter_route_action[iters=" + std::to_string(iters) + "]").c_str());
         out[5] = timed(iters, action); }
      EXA_HC(hipMemsetAsync(yA.p, 0, sizeof(double) * nd, s));
      action();
      out[8] = rel_diff(nd, yA.p, yC.p);
      // ---- the L-vector pair on the same quadrature functions
      auto model_lv = [&] { chk(exa_model_setup_lvec(ctxA, dt, op.x_cur.p, sd.v_sol.p, s0.p, sv0.p, s1.p, sv1.p, cm.p, J.p, s), "exa_model_setup_lvec"); };
      model_lv(); model_lv();
      {  ProfRegion prof(("adapter_route_lvec_model[passes=" + std::to_string(steps) + "]").c_str());
         out[14] = timed(steps, model_lv); }
      if (exa_model_status(ctxA, s) != 0) throw std::runtime_error("exa_driver_bench_adapter_route: the L-vector launch left unconverged points");
      to_aos(6, op.stress1, tmp); out[16] = rel_diff(6 * P, s1.p, tmp.p);
      chk(exa_set_tangent_form(ctxA, EXA_TANGENT_DEV5_BULK), "exa_set_tangent_form");
      chk(exa_grad_setup(ctxA, dt, J.p, cm.p, s), "exa_grad_setup");
      out[17] = timed(3, [&] { chk(exa_grad_setup(ctxA, dt, J.p, cm.p, s), "exa_grad_setup"); });
      chk(exa_grad_set_coords(ctxA, op.x_cur.p), "exa_grad_set_coords");
      EXA_HC(hipMemsetAsync(yA.p, 0, sizeof(double) * nd, s));
      chk(exa_grad_apply_lvec(ctxA, r.p, yA.p, nullptr, s), "exa_grad_apply_lvec");
      out[18] = rel_diff(nd, yA.p, yC.p);
      {  ProfRegion prof(("adapter_route_lvec_apply[iters=" + std::to_string(iters) + "]").c_str());
         out[15] = timed(iters, [&] { chk(exa_grad_apply_lvec(ctxA, r.p, yA.p, nullptr, s), "exa_grad_apply_lvec"); }); }
      out[19] = timed(iters, [&] { chk(exa_residual_lvec(ctxA, J.p, s1.p, yA.p, s), "exa_residual_lvec"); });
      // ... and with AssembleGradPA fused into ModelSetup (HipExaModelLVec(.., fused_records = true)): the compact records instead of the tangent field, no exa_grad_setup
      auto model_rec = [&] { chk(exa_model_setup_lvec_records(ctxA, dt, op.x_cur.p, sd.v_sol.p, s0.p, sv0.p, s1.p, sv1.p, J.p, s), "exa_model_setup_lvec_records"); };
      model_rec(); model_rec();
      {  ProfRegion prof(("adapter_route_lvec_records[passes=" + std::to_string(steps) + "]").c_str());
         out[20] = timed(steps, model_rec); }
      if (exa_model_status(ctxA, s) != 0) throw std::runtime_error("exa_driver_bench_adapter_route: the record launch left unconverged points");
      to_aos(6, op.stress1, tmp); out[21] = rel_diff(6 * P, s1.p, tmp.p);
      EXA_HC(hipMemsetAsync(yA.p, 0, sizeof(double) * nd, s));
      chk(exa_grad_apply_lvec(ctxA, r.p, yA.p, nullptr, s), "exa_grad_apply_lvec");
      out[22] = rel_diff(nd, yA.p, yC.p);
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

int exa_choose_newton_cap(const int* hist64, double tail_cost) { return choose_newton_cap(hist64, tail_cost); }
int exa_choose_newton_caps(const int* hist64, double tail_cost, int* k1, int* k2) { if (!hist64 || !k1 || !k2) return -1; choose_newton_caps_resume(hist64, tail_cost, *k1, *k2); return 0; }

int exa_options_query(const char* toml_path, double* out, char* err, int errlen) {
   try {
      ExaOptions o; o.parse_options(toml_path);
      const double v[20] = { o.temp_k, (double)o.nprops, (double)o.num_grains, o.xtal == XtalType::BCC ? 1.0 : 0.0, (double)(int)o.slip, o.dt_cust ? 1.0 : 0.0,
                             o.dt_auto ? 1.0 : 0.0, (double)o.nsteps, o.assembly == Assembly::PA ? 0.0 : 1.0, o.nl_solver == NLSolver::NRLS ? 1.0 : 0.0,
                             (double)o.newton_iter, o.newton_rel, o.newton_abs, (double)o.krylov_iter, o.krylov_rel, o.krylov_abs, (double)o.ref_ser,
                             (double)o.ncuts[0], o.additional_avgs ? 1.0 : 0.0, (double)o.bcs.size() };
      std::memcpy(out, v, sizeof(v));
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

static void export_partition(const Partition& p, int64_t* info, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                             int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs) {
   int64_t shared = 0; for (auto& nb : p.nbrs) shared += (int64_t)nb.dofs.size();
   info[0] = p.E; info[1] = p.NN; info[2] = (int64_t)p.nbrs.size(); info[3] = p.pg[0]; info[4] = p.pg[1]; info[5] = p.pg[2]; info[6] = shared; info[7] = p.n;
   if (conn) std::memcpy(conn, p.conn.data(), sizeof(int32_t) * p.conn.size());
   if (X) std::memcpy(X, p.X.data(), sizeof(double) * p.X.size());
   if (elem_gid) std::memcpy(elem_gid, p.elem_gid.data(), sizeof(int64_t) * p.elem_gid.size());
   if (weight) std::memcpy(weight, p.weight.data(), sizeof(double) * p.weight.size());
   size_t off = 0;
   for (size_t i = 0; i < p.nbrs.size(); i++) {
      if (nbr_rank) nbr_rank[i] = p.nbrs[i].rank;
      if (nbr_count) nbr_count[i] = (int32_t)p.nbrs[i].dofs.size();
      if (nbr_dofs) std::memcpy(nbr_dofs + off, p.nbrs[i].dofs.data(), sizeof(int32_t) * p.nbrs[i].dofs.size());
      off += p.nbrs[i].dofs.size();
   }
}

int exa_partition_query(const int* N, int rank, int nranks, int64_t* info, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                        int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs) {
   Partition p; const double L[3] = { 1.0, 1.0, 1.0 };
   const int order = (info[7] >= 2 && info[7] <= 6) ? (int)info[7] : 1;   // info[7] is in/out: H1 order on input (0/1 -> 1), nodes per element on output
   p.build(N, L, rank, nranks, order);
   export_partition(p, info, conn, X, elem_gid, weight, nbr_rank, nbr_count, nbr_dofs);
   return 0;
}

// the element order the driver runs with on several ranks (Partition::order_boundary_first): out2 = { E, E_bdr }; conn (n, E) and gid (E) may be null
int exa_partition_query_boundary_first(const int* N, int rank, int nranks, int order, int64_t* out2, int32_t* conn, int64_t* elem_gid) {
   Partition p; const double L[3] = { 1.0, 1.0, 1.0 };
   p.build(N, L, rank, nranks, order == 2 ? 2 : 1);
   p.order_boundary_first();
   out2[0] = p.E; out2[1] = p.E_bdr;
   if (conn) std::memcpy(conn, p.conn.data(), sizeof(int32_t) * p.conn.size());
   if (elem_gid) std::memcpy(elem_gid, p.elem_gid.data(), sizeof(int64_t) * p.elem_gid.size());
   return 0;
}

int exa_mesh_partition_query(const char* mesh_path, int rank, int nranks, int64_t* info, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                             int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs, char* err, int errlen) {
   try {
      Partition p; p.build_from_mfem_mesh(mesh_path, rank, nranks);
      export_partition(p, info, conn, X, elem_gid, weight, nbr_rank, nbr_count, nbr_dofs);
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

// the same at p_refinement = order (1 or 2: edge / face / element nodes added to the trilinear file mesh)
int exa_mesh_partition_query_order(const char* mesh_path, int rank, int nranks, int order, int64_t* info, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                                   int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs, char* err, int errlen) {
   try {
      Partition p; p.build_from_mfem_mesh(mesh_path, rank, nranks, order);
      export_partition(p, info, conn, X, elem_gid, weight, nbr_rank, nbr_count, nbr_dofs);
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, e.what()); return -1; }
}

}  // extern "C"
