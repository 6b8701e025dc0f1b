/* exaconstit_driver.h — C entry points of the stand-alone driver inside libexaconstit_hip.so.
 *
 * Run-time surface of the reference's `mechanics -opt options.toml` executable (reference src/mechanics_driver.cpp:112-1022) and
 * of its SystemDriver (reference src/system_driver.hpp:101-143) for callers without MFEM: the `mechanics` binary of this repo,
 * the tests and bench.py.  Multi-GPU: one process per GPU; rank 0 obtains a RCCL unique id (exa_rccl_unique_id), the launcher
 * distributes the 128 bytes (bench.py uses torch.distributed), every rank passes them to exa_driver_create*.
 */
#ifndef EXA_DRIVER_CAPI_H
#define EXA_DRIVER_CAPI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct exa_driver exa_driver;

typedef struct {
   int N;                      /* N^3 elements on the unit cube, p = 1 */
   int bcc;                    /* 0 fcc, 1 bcc */
   int slip;                   /* 0 powervoce, 1 powervocenl, 2 mtsdd */
   int nprops; const double* props; double temp_k;
   const double* quats;        /* (4, N^3) one orientation per element, x fastest */
   int assembly;               /* 0 PA, 1 EA */
   int nrls, jacobi;
   int newton_iter; double newton_rel, newton_abs;
   int krylov_iter; double krylov_rel, krylov_abs;
   int nsteps; const double* dts;
   double vz;                  /* z-velocity of the top face */
   int order;                  /* H1 order p (1 or 2; 0 = 1) */
   int bbar;                   /* 1: B-bar integrator (element assembly only) */
   int nrev; const int* rev_steps;   /* load reversals: the top-face velocity changes sign at these steps (the cyclic schedule of the reference's
                                * voce_full_cyclic.toml: update_steps 11, 31, 51, 71); each one is a boundary-condition change with its corrector solve */
} exa_synth_config;

int exa_rccl_unique_id(void* out128);
/* the id rank 0 hands to a group of nranks: a RCCL unique id, or the id of the inter-process shared-device transport when RCCL cannot serve
 * the launch (more ranks than visible devices - RCCL refuses two ranks on one device - or EXA_TRANSPORT=ipc).  bench.py and exa_bootstrap use it. */
int exa_comm_unique_id(void* out128, int nranks);
/* PCI bus id of the current device ("0000:c1:00.0"): lets a launcher tell one rank per physical GPU from several ranks on one GPU */
int exa_device_identity(char* out, int len);
/* out2 = { rank count the transport itself reports (ncclCommCount for RCCL), kind: 0 none, 1 rccl, 2 ipc, 3 in-process loopback } */
int exa_driver_comm_info(exa_driver* d, int* out2);
/* out8 = { local elements, elements in 64-blocks that touch shared nodes, neighbour ranks, doubles sent (= received) per halo exchange,
 *          halo exchange overlapped with the interior blocks (0 / 1; over RCCL opt-in with EXA_HALO_OVERLAP=on), transport kind, ranks the
 *          transport reports, 0 } - what bench.py prints per rank of a multi-rank run */
int exa_driver_comm_details(exa_driver* d, int64_t* out8);
/* latency floor of the RCCL calls of one PCG iteration on this device (one-rank communicator): out2 = { us per 16-byte all-reduce,
 * us per grouped send/recv of n doubles to the own rank } */
int exa_rccl_microbench(int iters, int n, double* out2, char* err, int errlen);
/* Launcher-agnostic process-group bootstrap of the `mechanics` executable (reference: MPI_Init / MPI_Comm_rank / MPI_Comm_size,
 * src/mechanics_driver.cpp:119-150).  exa_bootstrap_env reads rank / size / local rank from the environment of mpirun (MPICH PMI_*,
 * Open MPI OMPI_COMM_WORLD_*), srun (SLURM_*), torchrun-style launchers (RANK / WORLD_SIZE / LOCAL_RANK) or EXA_RANK / EXA_NRANKS;
 * exa_bootstrap_bcast copies rank 0's buffer to every rank over a TCP rendez-vous on [EXA_]MASTER_ADDR:[EXA_]MASTER_PORT (default
 * 127.0.0.1:29517); exa_bootstrap = both + device selection (local rank mod visible devices) + the RCCL unique id in uid128. */
int exa_bootstrap_env(int* rank, int* nranks, int* local_rank);
int exa_bootstrap_bcast(int rank, int nranks, void* buf, int nbytes, double timeout_s, char* err, int errlen);
/* The rendez-vous in its general form: every rank contributes nbytes, rank 0 turns the table of all contributions (rank order) into the
 * reply every rank receives (fn runs on rank 0 only; 0 = ok).  exa_bootstrap uses it to decide the transport from the IDENTITY of the
 * ranks' devices (host name + PCI bus id): RCCL when every rank has a GPU of its own - on one node or several -, the shared-device
 * inter-process transport only when ranks of ONE host share a device; ranks that share a device across hosts cannot exist, and the
 * shared-device transport is refused for a group that spans hosts. */
typedef int (*exa_bootstrap_reply_fn)(const void* all, int nranks, int nbytes, void* reply, int reply_bytes, void* user);
int exa_bootstrap_gather_reply(int rank, int nranks, const void* mine, int nbytes, void* reply, int reply_bytes, exa_bootstrap_reply_fn fn, void* user,
                               double timeout_s, char* err, int errlen);
/* the decision itself, exposed for the tests: ids = nranks records of 96 bytes (host name[64], PCI bus id[32], zero padded);
 * returns 1 = RCCL, 2 = shared-device transport, -1 = impossible (reason in err) - EXA_TRANSPORT=rccl|ipc overrides where it can */
int exa_transport_from_identities(const void* ids, int nranks, char* err, int errlen);
int exa_bootstrap(int* rank, int* nranks, void* uid128, char* err, int errlen);
/* Test transport: `nranks` drivers on ONE device, one host thread each, exchanging through an in-process group instead of RCCL
 * (RCCL refuses two ranks on one device).  Pass the 128 bytes as the unique id of every rank; destroy after the drivers. */
int exa_loopback_group_create(int nranks, void* out128);
void exa_loopback_group_destroy(const void* id128);
exa_driver* exa_driver_create(const char* toml_path, const char* out_dir, int rank, int nranks, const void* uid, int jacobi, int write_files, char* err, int errlen);
exa_driver* exa_driver_create_synthetic(const exa_synth_config* c, int rank, int nranks, const void* uid, char* err, int errlen);
void exa_driver_destroy(exa_driver* d);
int exa_driver_num_steps(exa_driver* d);
int64_t exa_driver_local_qpts(exa_driver* d);
int64_t exa_driver_local_dofs(exa_driver* d);
int exa_driver_step(exa_driver* d, int ti, char* err, int errlen);
/* solve step ti but do not commit it (no begin/end swap, no coordinate update, no output row): the state bench.py times its passes on */
int exa_driver_step_nocommit(exa_driver* d, int ti, char* err, int errlen);
/* end-of-step update of a step solved by exa_driver_step_nocommit, valid while only residual evaluations at the converged velocity
 * (exa_driver_bench_model / exa_driver_bench_pcg) have run since */
int exa_driver_commit_step(exa_driver* d, char* err, int errlen);
int exa_driver_run(exa_driver* d, char* err, int errlen);
int exa_driver_get_avgs(exa_driver* d, int which, double* out, int maxrows);
int exa_driver_get_stats(exa_driver* d, int* newton, int* krylov, int* model_calls, int maxrows);
void exa_driver_get_timers(exa_driver* d, double* out5);
void exa_driver_reset_timers(exa_driver* d);
/* Solver diagnostics the reference prints or aborts on: out4[0] quadrature points whose ExaCMech solve did not converge (the
 * library's ECMECH_FAIL; here Newton reports non-convergence), [1] PCG solves without convergence, [2] PCG iterations with
 * (Ad, d) < 0 (MFEM: "The operator is not positive definite"), [3] flag of the last PCG solve (1 ok, 2 max_iter, -1 (Ad, d) = 0). */
void exa_driver_get_diagnostics(exa_driver* d, int64_t* out4);
/* Residual reduction |r|_M / |r0|_M the PCG reached: out2[0] last solve, out2[1] the worst among the solves that stopped at max_iter
 * (MFEM's CGSolver prints "No convergence!" with the final norms, linalg/solvers.cpp; the reference's Newton loop goes on regardless). */
void exa_driver_get_pcg_reduction(exa_driver* d, double* out2);
/* 64-bin histogram of the local-solver evaluation counts (ExaCMech's nFEval state variable) of the last constitutive launch */
int exa_driver_nfev_hist(exa_driver* d, int* hist64, char* err, int errlen);
/* the same of the begin-of-step state (which = 0: after a completed step, the launch that step converged with) or the end-of-step state (1) */
int exa_driver_nfev_hist_of(exa_driver* d, int which, int* hist64, char* err, int errlen);
/* one component of a quadrature function of the operator, de-blocked on the host: which = 0 begin-of-step state, 1 end-of-step state (28),
 * 2 begin stress, 3 end stress (6); out receives E * Q doubles ordered [element][point] (diagnostics: nFEval maps, parity tools) */
int exa_driver_get_qf_component(exa_driver* d, int which, int comp, double* out, char* err, int errlen);
int exa_driver_bench_prepare(exa_driver* d, int nsteps, const double* dts, double perturb, char* err, int errlen);
int exa_driver_bench_model(exa_driver* d, int steps, double* out3, char* err, int errlen);
int exa_driver_bench_pcg(exa_driver* d, int iters, double* out3, char* err, int errlen);
/* the drop-in route (the calls of include/exaconstit_mfem_adapters.hpp: AOS exa_model_setup, exa_grad_setup, E-vector exa_grad_apply between the element
 * restriction and its transpose - and the L-vector pair's exa_model_setup_lvec / exa_grad_apply_lvec) timed on a second context that is given this driver's state;
 * out24 documented at the definition (host/driver_capi.hip) */
int exa_driver_bench_adapter_route(exa_driver* d, int steps, int iters, double* out24, char* err, int errlen);

/* host-logic queries that need no GPU (used by the CPU tests) ------------------------------------------------------------ */
/* options.toml reader (reference src/option_parser.cpp:26-932): fills out[0..19] =
 * {temp_k, nprops, num_grains, xtal, slip, dt_cust, dt_auto, nsteps, assembly(0 PA,1 EA), nl_solver(0 NR,1 NRLS), newton_iter, newton_rel,
 *  newton_abs, krylov_iter, krylov_rel, krylov_abs, ref_ser, ncuts0, additional_avgs, number of BC change steps}; returns 0 or -1 (err) */
int exa_options_query(const char* toml_path, double* out20, char* err, int errlen);
/* the driver's tail-split controller as a pure function (host logic; tests): cap on local-solver evaluations chosen from a 64-bin
 * histogram of evaluation counts, 0 = leave the launch uncapped.  tail_cost = relative cost of a point in the second launch. */
int exa_choose_newton_cap(const int* hist64, double tail_cost);
/* the same for resumed tail points (exa_set_newton_caps): first cap and second cap (0 = one dense launch) */
int exa_choose_newton_caps(const int* hist64, double tail_cost, int* k1, int* k2);

/* block decomposition of an N0 x N1 x N2 element grid (reference: ParMesh/METIS, src/mechanics_driver.cpp:312): sizes first
 * (info[0..7] = {E, NN, nneighbors, pg0, pg1, pg2, total shared dofs, n}; info[7] is in/out: H1 order p on input (0 or 1 -> 1, 2 -> 2),
 * nodes per element n = (p+1)^3 on output), then the arrays when the pointers are non-null:
 * conn (n,E) native node order, X (NN,3 byNODES), elem_gid (E), weight (NN), nbr_rank (nneighbors), nbr_count (nneighbors), nbr_dofs (concatenated) */
int exa_partition_query(const int* N, int rank, int nranks, int64_t* info8, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                        int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs);
/* the same view of the partition of an MFEM mesh v1.0 file (Mesh.type = "other"): every rank reads the file, elements are split by
 * recursive coordinate bisection of their centroids (reference: METIS through ParMesh, src/mechanics_driver.cpp:312), elem_gid = index
 * of the element in the file, nodes renumbered per rank in ascending global order.  Returns 0 or -1 (err). */
/* The element order the driver runs with on several ranks: elements touching a node shared with another rank first (their 64-element
 * blocks are computed before the halo exchange starts, the interior ones while it is on the wire).  out2 = { E, E_bdr }. */
int exa_partition_query_boundary_first(const int* N, int rank, int nranks, int order, int64_t* out2, int32_t* conn, int64_t* elem_gid);
int exa_mesh_partition_query(const char* mesh_path, int rank, int nranks, int64_t* info8, int32_t* conn, double* X, int64_t* elem_gid, double* weight,
                             int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs, char* err, int errlen);
/* ... at p_refinement = order: 1, or 2 = one node added per edge, face and element of the trilinear file mesh (what the reference's order
 * elevation of the nodal space gives for straight-sided hexahedra, src/mechanics_driver.cpp:300-306); higher orders: generated meshes only. */
int exa_mesh_partition_query_order(const char* mesh_path, int rank, int nranks, int order, int64_t* info8, int32_t* conn, double* X, int64_t* elem_gid,
                                   double* weight, int32_t* nbr_rank, int32_t* nbr_count, int32_t* nbr_dofs, char* err, int errlen);
#ifdef __cplusplus
}
#endif
#endif
