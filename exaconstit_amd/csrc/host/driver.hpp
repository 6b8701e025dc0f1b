// Stand-alone host driver above the C ABI: the reference's operator / solver / system-driver classes re-expressed for
// device-resident data (MFEM is not available in this image, so the data containers are plain device buffers; the class
// and method names, argument meaning and control flow follow the reference so that its regression cases run unchanged).
//   ExaModel / ExaCMechModel            reference src/mechanics_model.hpp:17-241, src/mechanics_ecmech.hpp:12-109
//   ExaNLFIntegrator                     reference src/mechanics_integrators.hpp:14-76
//   NonlinearMechOperator                reference src/mechanics_operator.hpp:18-100, src/mechanics_operator.cpp:288-483
//   MechOperatorJacobiSmoother           reference src/mechanics_operator_ext.cpp:11-55
//   ExaNewtonSolver / ExaNewtonLSSolver  reference src/mechanics_solver.cpp:39-281
//   SystemDriver                         reference src/system_driver.cpp:221-558
//   time-step loop                       reference src/mechanics_driver.cpp:837-907
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "../../../include/exaconstit_hip.h"
#include "device_utils.hpp"
#include "mesh.hpp"
#include "options.hpp"

namespace exa_host {

// ---- communication: RCCL over xGMI when nranks > 1 (loaded at run time), no-ops on one rank -----------------------
class Comm {
 public:
   int rank = 0, nranks = 1;
   ~Comm();
   // force_rccl: run the RCCL calls on a one-rank communicator too (what EXA_FORCE_RCCL=1 selects from the environment)
   void init(int rank_, int nranks_, const void* nccl_unique_id /*128 bytes, identical on all ranks*/, bool force_rccl = false);
   static void get_unique_id(void* out128);
   // inter-process transport for ranks that share a device (driver.hip): an id of that kind, and whether this launch needs it
   static void ipc_unique_id(void* out128);
   static bool want_ipc_transport(int nranks);
   int reported_ranks() const;            // what the transport itself says (ncclCommCount for RCCL)
   const char* transport() const { return ipc_ ? "ipc" : (loop_ ? "loopback" : (comm_ ? "rccl" : "none")); }
   // in-process loopback transport for tests: several ranks (one host thread each) on ONE device, host-synchronous exchanges
   static void loopback_create(int nranks, void* out128);
   static void loopback_destroy(const void* id128);
   void allreduce_sum(double* dev, int n, hipStream_t s);
   void allreduce_min(double* dev, int n, hipStream_t s);
   // y(shared dofs) <- sum over all ranks holding them
   void halo_sum(const Partition& part, double* y, hipStream_t s);
   // the same in two halves for overlap with work on stream s: begin = pack + exchange on the communication stream once everything
   // enqueued on s so far has finished; end = s waits for the exchange, then adds the received segments
   void halo_begin(const Partition& part, double* y, hipStream_t s);
   void halo_end(const Partition& part, double* y, hipStream_t s);
   void setup_halo(const Partition& part);
   double max_over_ranks(double v);
   // us per call of the fused 16-byte all-reduce and of a grouped send/recv of `n` doubles to the own rank (one-rank communicator: latency floor of the RCCL calls)
   void microbench(int iters, int n, double* us_allreduce, double* us_sendrecv);
   bool deterministic = false;               // halo contributions added segment by segment (fixed order) instead of one atomic pass
   size_t halo_dofs() const { return seg_off_.back(); }   // doubles this rank sends (= receives) per exchange, all neighbours
   bool selftest() const { return selftest_zero_; }   // EXA_HALO_SELFTEST: the forced one-rank communicator exchanges zeros with itself (driver.hip, Comm::init)
   bool forced() const { return force_; }   // EXA_FORCE_RCCL=1: the one-rank communicator runs the multi-rank code paths and every RCCL call
 private:
   void unpack(double* y, hipStream_t s);
   void loopback_reduce(double* dev, int n, int op, hipStream_t s);
   void* comm_ = nullptr; void* loop_ = nullptr; void* ipc_ = nullptr; bool force_ = false;
   bool selftest_zero_ = false;
   bool loop_async_ = false;   // loopback: exchanges ordered by events only, no stream is drained (driver.hip, Comm::exchange)
   DevBuf<int32_t> idx_all_; DevBuf<double> sbuf_all_, rbuf_all_; std::vector<size_t> seg_off_{ 0 };   // concatenated neighbour segments
   DevBuf<double> tmp_;
   hipStream_t cs_ = nullptr; hipEvent_t ev_ready_ = nullptr, ev_done_ = nullptr;   // communication stream of halo_begin / halo_end
   void exchange(const Partition& part, hipStream_t s);   // send buffers -> neighbours' receive buffers (RCCL grouped send/recv or loopback copies) on stream s
};

enum class Precond { IDENTITY, JACOBI };

struct SolverStats { int newton_iters = 0; int krylov_iters = 0; int model_calls = 0; bool converged = false; };

struct Timers {
   double t_model_ms = 0, t_krylov_ms = 0, t_solve_ms = 0; int64_t qpt_updates = 0; int64_t krylov_iters = 0;
};

// Per-quadrature-point model seam (ExaModel): owns nothing but scratch; the driver owns the quadrature functions.
class ExaCMechModel {
 public:
   ExaCMechModel(exa_ctx* ctx, DevBuf<double>* stress0, DevBuf<double>* stress1, DevBuf<double>* matGrad, DevBuf<double>* matVars0, DevBuf<double>* matVars1)
      : ctx_(ctx), stress0_(stress0), stress1_(stress1), matGrad_(matGrad), matVars0_(matVars0), matVars1_(matVars1) {}
   void SetModelDt(double dt) { dt_ = dt; }
   double GetModelDt() const { return dt_; }
   // ExaCMechModel::ModelSetup (reference src/mechanics_ecmech.cpp:192-258)
   void ModelSetup(const double* jacobian, const double* vel_evec, hipStream_t s);
   // the same with the operator's L->E restrictions and Jacobian refresh fused in (writes the Jacobians)
   void ModelSetupLVec(const double* x_lvec, const double* v_lvec, double* jacobian_out, hipStream_t s);
   // ... and with AssembleGradPA fused in too: writes the compact gradient records instead of the tangent field (p = 1 fast path)
   void ModelSetupLVecRecords(const double* x_lvec, const double* v_lvec, double* jacobian_out, hipStream_t s);
   void UpdateModelVars() {}
   void UpdateStress() { stress0_->swap(*stress1_); }        // reference src/mechanics_model.cpp:435-438
   void UpdateStateVars() { matVars0_->swap(*matVars1_); }   // reference src/mechanics_model.cpp:440-443
   void calcDpMat(double* dp, hipStream_t s) const;          // reads matVars1 (reference src/mechanics_ecmech.hpp:309)
   DevBuf<double>* GetStress0() { return stress0_; } DevBuf<double>* GetStress1() { return stress1_; }
   DevBuf<double>* GetMatVars0() { return matVars0_; } DevBuf<double>* GetMatVars1() { return matVars1_; } DevBuf<double>* GetMatGrad() { return matGrad_; }
   exa_ctx* ctx() const { return ctx_; }
 private:
   exa_ctx* ctx_; double dt_ = 1.0;
   DevBuf<double>*stress0_, *stress1_, *matGrad_, *matVars0_, *matVars1_;
};

// tail-split controller (see driver.hip): cap on local-solver evaluations from a 64-bin histogram of their counts; 0 = no cap
int choose_newton_cap(const int* hist64, double tail_cost);
void choose_newton_caps_resume(const int* hist64, double tail_cost, int& k1, int& k2);

class NonlinearMechOperator {
 public:
   NonlinearMechOperator(const ExaOptions& opt, const Partition& part, Comm& comm, const std::vector<double>& props, const std::vector<double>& quats_per_elem);
   ~NonlinearMechOperator();
   int Height() const { return nd_; }
   void SetDt(double dt) { dt_ = dt; model_->SetModelDt(dt); }
   void UpdateEssTDofs(const std::vector<uint8_t>& mask);
   // y = F(k): residual with essential rows zeroed (reference src/mechanics_operator.cpp:288-308)
   void Mult(const double* k, double* y);
   template <bool upd_crds> void Setup(const double* k);
   // Jacobian set-up + Jacobi diagonal (reference src/mechanics_operator.cpp:436-443)
   void GetGradient();
   // y = K x with essential columns/rows masked (constrained) or the plain local action
   // y_prezeroed / skip_out_mask: the PCG loop folds the zero fill into its direction update and the output mask into its dot product
   void GradMult(const double* x, double* y, bool constrained, const double* done_flag = nullptr, bool y_prezeroed = false, bool skip_out_mask = false);
   // reference src/mechanics_operator.cpp:446-483
   void GetUpdateBCsAction(const double* k, const double* x, double* y);
   void ResidualAction(double* y);
   void RefreshJacobians();                // el_jac of x_cur when the record route left it unwritten (volume averages)
   void UpdateModel();                     // swap begin/end state, x_beg <- x_cur
   void SwapCoords();
   ExaCMechModel* GetModel() { return model_.get(); }
   hipStream_t stream() const { return stream_; }
   const Partition& part() const { return part_; }
   Comm& comm() { return comm_; }
   bool halo_overlap() const { return overlap_; }      // the gradient action overlaps the halo exchange with its interior blocks
   // data (device)
   DevBuf<double> x_ref, x_beg, x_cur, el_x, el_v, el_jac, diag, dinv, weight;
   DevBuf<double> stress0, stress1, matVars0, matVars1, matGrad;
   DevBuf<uint8_t> ess_mask;
   DevBuf<int32_t> conn;
   Precond precond = Precond::IDENTITY;
   Timers timers;
   int model_calls = 0;
   std::vector<double> props;     // material parameters the context was created with (the adapter-route bench creates a second, AOS context from them)
   exa_config cfg_used;           // ... and its configuration (props pointer not valid after construction)
   double dt() const { return dt_; }
   // quadrature points whose local (ExaCMech) solve did not converge in the last constitutive launch.  The library fails the run
   // in that case (ECMECH_FAIL in getResponseSngl); here a non-zero count poisons the next residual norm on every rank, so that
   // Newton reports non-convergence: Time.Auto then cuts dt, otherwise the run stops.
   int model_fail = 0; int64_t model_fail_total = 0;
   bool model_status_pending_ = false;       // a constitutive launch whose failure count has not reached the host yet (ResidualNorm / ReadModelStatus)
   struct EvPair { hipEvent_t a = nullptr, b = nullptr; bool pending = false; long call = 0; };
   void ReadTimer(EvPair& e);   // adds a finished launch to timers.t_model_ms (EXA_MODEL_TIMES=1: and prints it)
   EvPair& NextModelTimer(); void FlushModelTimers(); void ReadModelStatus();
   double ResidualNorm(const double* r);
   double dot(const double* a, const double* b);   // weighted, all-reduced, synchronising
   DevBuf<double> partial, scal;
 private:
   ExaOptions opt_; const Partition& part_; Comm& comm_;
   exa_ctx* ctx_ = nullptr; std::unique_ptr<ExaCMechModel> model_;
   hipStream_t stream_ = nullptr; hipEvent_t ev0_, ev1_;
   std::vector<EvPair> ev_ring_; int ev_head_ = 0;   // event pairs around the constitutive launches, read back lazily
   int nn_, nd_, E_, npe_ = 8; double dt_ = 1.0;
   bool records_setup_ = false;   // gradient records written by the constitutive launch (p = 1 fast path, identity preconditioner)
   bool use_records() const { return records_setup_ && precond == Precond::IDENTITY; }
   bool geo_resid_ = true, jac_stale_ = false;   // record route + L-vector residual: no Jacobian field is written, both actions recompute the geometry (EXA_JAC_FIELD=on keeps it)
   bool geo_resid() const { return geo_resid_ && lvec_resid_ && fast_p1_; }      // (p = 2: the Jacobian field is written by the geometry pre-pass and read by the residual)
   void ensure_mat_grad();
   bool overlap_ = false; int nblk_bdr_ = 0;   // halo exchange overlapped with the interior element blocks (several ranks, atomic p = 1 record action)
   bool fast_p1_ = true, lvec_grad_ = true, fused_setup_ = true; bool lvec_resid_ = false; bool compact_tangent_ = false;
   bool cap_auto_ = true; int newton_cap_ = 0, newton_cap2_ = 0; bool tail_resume_ = true; double tail_cost_ = 4.0;
   DevBuf<double> tmp_l_, tmp_r_, el_y_, el_x2_;
};

class SystemDriver {
 public:
   SystemDriver(const ExaOptions& opt, int rank, int nranks, const void* nccl_uid);
   ~SystemDriver();
   // synthetic RVE without files (bench): N^3 elements, one grain per element, seeded orientations
   SystemDriver(const ExaOptions& opt, const std::vector<double>& props, const std::vector<double>& quats_per_global_elem, int rank, int nranks, const void* nccl_uid);
   void UpdateEssBdr(const BCEntry& bc);
   void UpdateVelocity(double* v);
   void SolveInit(const double* xprev, double* x);
   bool Solve(double* x);
   void UpdateModel();
   // one time step of the reference's loop (src/mechanics_driver.cpp:837-907); returns false if Newton failed
   // commit = false (bench): solve the step but leave begin-of-step state, coordinates and outputs untouched
   bool Step(int ti, bool commit = true);
   void CommitStep();                      // end-of-step update of a step solved with commit = false
   int RunAll();
   bool NewtonSolve(double* x, SolverStats& st);
   int CGSolve(const double* b, double* x);   // device PCG, returns iterations
   int CGSolveSingleReduction(const double* b, double* x);   // more than one rank: one fused 16-byte all-reduce per iteration
   void note_cg_reduction(const double* hS);
   double last_cg_reduction = 0.0, worst_capped_cg_reduction = 0.0;   // |r|_M / |r0|_M of the last PCG solve / the worst among the solves that stopped at max_iter
   void drop_cg_graph();                      // forget the captured PCG chunk (its solution buffer is about to go away)
   void report_cg(const double* hS, int iters) const;   // MFEM CGSolver::Mult diagnostics (verbose / EXA_VERBOSE)
   NonlinearMechOperator& oper() { return *oper_; }
   const ExaOptions& options() const { return opt_; }
   std::vector<double> avg_stress, avg_def_grad, avg_pl_work, avg_dp_tensor;   // one row per completed step (rank 0 view, all ranks identical)
   std::vector<SolverStats> stats;
   DevBuf<double> v_sol;
   double time = 0.0, dt_class = 0.0; int steps_done = 0;
   bool write_files = true; std::string out_dir = ".";
   std::vector<double> step_wall_s;            // wall time of each step (solve part), written to time/time_solve.<rank>.txt by RunAll
   void WriteStepTimes();
   Precond precond = Precond::IDENTITY;
   int cg_check_every = 16;
   int64_t cg_graph_max_dofs = 3 * 33 * 33 * 33;   // PCG iterations replayed from a hipGraph up to this many local dofs (32^3 elements at p = 1: +9 % at 16^3, +3 % at 32^3, a loss from 48^3 on); EXA_PCG_GRAPH=0 | all
   // linear-solver diagnostics (MFEM's CGSolver prints these): flag of the last solve (1 converged, 2 max_iter, -1 den == 0),
   // number of solves that did not converge, iterations that saw (Ad, d) < 0
   int last_cg_flag = 1; int64_t cg_not_converged = 0, cg_indefinite_iters = 0;
   bool verbose = false;
   Partition part;
   Comm comm;
 private:
   void init(const std::vector<double>& props, const std::vector<double>& quats_local);
   ExaOptions opt_;
   std::unique_ptr<NonlinearMechOperator> oper_;
   DevBuf<double> r_, c_, xt_, cg_r_, cg_z_, cg_d_, cg_s_, cg_q_, ess_val_;
   std::vector<uint8_t> ess_host_; std::vector<double> ess_val_host_;
   DevBuf<uint8_t> vel_mask_, vg_mask_; bool have_vel_ = false, have_vgrad_ = false; double vgrad_[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
   double last_dt_ = 0.0;
   void* cg_graph_ = nullptr; const double* cg_graph_x_ = nullptr; int64_t cg_graph_key_ = -1;   // captured PCG chunk (hipGraphExec_t) and what it was captured for
};

}  // namespace exa_host
