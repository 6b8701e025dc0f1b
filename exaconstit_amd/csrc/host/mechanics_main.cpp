// `mechanics -opt options.toml` — stand-alone equivalent of the reference executable (reference src/mechanics_driver.cpp:112-1022)
// for its hot-path subset, running on one MI355X (multi-GPU runs are launched through bench.py / the Python binding, which
// distribute the RCCL unique id).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../../include/exaconstit_driver.h"

int main(int argc, char** argv) {
   std::string opt = "options.toml";
   for (int i = 1; i < argc; i++) if ((!std::strcmp(argv[i], "-opt") || !std::strcmp(argv[i], "--option")) && i + 1 < argc) opt = argv[++i];
   char err[512] = { 0 };
   const auto t0 = std::chrono::steady_clock::now();
   exa_driver* d = exa_driver_create(opt.c_str(), ".", 0, 1, nullptr, 0, 1, err, sizeof(err));
   if (!d) { std::fprintf(stderr, "mechanics: %s\n", err); return 1; }
   const int rc = exa_driver_run(d, err, sizeof(err));
   if (rc < 0) { std::fprintf(stderr, "mechanics: run failed (%d) %s\n", rc, err); exa_driver_destroy(d); return 2; }
   double t[5]; exa_driver_get_timers(d, t);
   const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   std::printf("The process took %lf seconds to run\n", wall);
   std::printf("steps %d | constitutive kernel %.3f s for %.0f qpt updates (%.3e qpt/s) | krylov %.3f s for %.0f iterations (%.1f it/s)\n", rc,
               t[0] * 1e-3, t[3], t[3] / (t[0] * 1e-3), t[1] * 1e-3, t[4], t[4] / (t[1] * 1e-3));
   exa_driver_destroy(d);
   return 0;
}
