/* C entry points of the stand-alone driver inside libexaconstit_hip.so (run-time surface of the reference's `mechanics`
 * executable: reference src/mechanics_driver.cpp:112-1022).  Used by the `mechanics` binary, tests and bench.py. */
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
} exa_synth_config;

int exa_rccl_unique_id(void* out128);
exa_driver* exa_driver_create(const char* toml_path, const char* out_dir, int rank, int nranks, const void* uid, int jacobi, int write_files, char* err, int errlen);
exa_driver* exa_driver_create_synthetic(const exa_synth_config* c, int rank, int nranks, const void* uid, char* err, int errlen);
void exa_driver_destroy(exa_driver* d);
int exa_driver_num_steps(exa_driver* d);
int64_t exa_driver_local_qpts(exa_driver* d);
int64_t exa_driver_local_dofs(exa_driver* d);
int exa_driver_step(exa_driver* d, int ti, char* err, int errlen);
int exa_driver_run(exa_driver* d, char* err, int errlen);
int exa_driver_get_avgs(exa_driver* d, int which, double* out, int maxrows);
int exa_driver_get_stats(exa_driver* d, int* newton, int* krylov, int* model_calls, int maxrows);
void exa_driver_get_timers(exa_driver* d, double* out5);
void exa_driver_reset_timers(exa_driver* d);
int exa_driver_bench_prepare(exa_driver* d, int nsteps, const double* dts, double perturb, char* err, int errlen);
int exa_driver_bench_model(exa_driver* d, int steps, double* out3, char* err, int errlen);
int exa_driver_bench_pcg(exa_driver* d, int iters, double* out3, char* err, int errlen);
#ifdef __cplusplus
}
#endif
#endif
