// `mechanics -opt options.toml` — stand-alone equivalent of the reference executable (reference src/mechanics_driver.cpp:112-1022)
// for its hot-path subset.  One process per GPU, like the reference's one MPI rank per device:
//    mechanics -opt case.toml                      one MI355X
//    mpirun -np 8 mechanics -opt case.toml         eight ranks (any launcher that exports a rank number: host/bootstrap.cpp)
// The ranks find each other through exa_bootstrap (rank / size from the launcher's environment, RCCL unique id over a TCP rendez-vous);
// every collective afterwards is RCCL over xGMI.  Rank 0 prints and writes the avg_* files, every rank writes time/time_solve.<rank>.txt.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../../include/exaconstit_driver.h"

int main(int argc, char** argv) {
   // Several ranks on hosts whose kernel driver offers dmabuf IPC only (this pool's): without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL and hipIpcGetMemHandle fail between
   // processes.  Set before the HIP runtime comes up (first HIP call: exa_bootstrap), for multi-rank launches only, never overriding what the user exported, and
   // said once on stderr (a host where only the legacy mode works wants HSA_ENABLE_IPC_MODE_LEGACY=1 exported instead: INTEGRATION.md, "Without MFEM").
   {
      int r0 = 0, n0 = 1, l0 = 0;
      if (exa_bootstrap_env(&r0, &n0, &l0) == 0 && n0 > 1 && std::getenv("HSA_ENABLE_IPC_MODE_LEGACY") == nullptr) {
         ::setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
         if (r0 == 0) std::fprintf(stderr, "mechanics: %d ranks, HSA_ENABLE_IPC_MODE_LEGACY was unset: running with 0 (dmabuf IPC)\n", n0);
      }
   }
   std::string opt = "options.toml";
   for (int i = 1; i < argc; i++) if ((!std::strcmp(argv[i], "-opt") || !std::strcmp(argv[i], "--option")) && i + 1 < argc) opt = argv[++i];
   char err[512] = { 0 };
   int rank = 0, nranks = 1; unsigned char uid[128];
   if (exa_bootstrap(&rank, &nranks, uid, err, sizeof(err)) != 0) { std::fprintf(stderr, "mechanics: %s\n", err); return 1; }
   const auto t0 = std::chrono::steady_clock::now();
   exa_driver* d = exa_driver_create(opt.c_str(), ".", rank, nranks, nranks > 1 ? uid : nullptr, 0, 1, err, sizeof(err));
   if (!d) { std::fprintf(stderr, "mechanics (rank %d): %s\n", rank, err); return 1; }
   const int rc = exa_driver_run(d, err, sizeof(err));
   if (rc < 0) { std::fprintf(stderr, "mechanics (rank %d): run failed (%d) %s\n", rank, rc, err); exa_driver_destroy(d); return 2; }
   double t[5]; exa_driver_get_timers(d, t);
   const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   if (rank == 0) {
      std::printf("The process took %lf seconds to run\n", wall);
      std::printf("ranks %d | steps %d | constitutive kernel %.3f s for %.0f qpt updates on rank 0 (%.3e qpt/s) | krylov %.3f s for %.0f iterations (%.1f it/s)\n", nranks, rc,
                  t[0] * 1e-3, t[3], t[3] / (t[0] * 1e-3), t[1] * 1e-3, t[4], t[4] / (t[1] * 1e-3));
   }
   exa_driver_destroy(d);
   return 0;
}
