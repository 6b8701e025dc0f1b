// Process-group bootstrap of the `mechanics` executable: who am I, how many are we, and the 128 bytes of the RCCL unique id on every rank.
//
// The reference is `mpirun -np N mechanics -opt options.toml` (src/mechanics_driver.cpp:119-150: MPI_Init, MPI_Comm_rank/size) and uses
// MPI for every collective.  Here the collectives are RCCL over xGMI (host/driver.hip, class Comm); the only thing a launcher has to
// provide is a rank number, so the executable runs unchanged under
//    mpirun / mpiexec.hydra (MPICH: PMI_RANK, PMI_SIZE; Open MPI: OMPI_COMM_WORLD_RANK/SIZE/LOCAL_RANK), srun (SLURM_PROCID, SLURM_NTASKS,
//    SLURM_LOCALID), torchrun-style environments (RANK, WORLD_SIZE, LOCAL_RANK) or a shell loop (EXA_RANK, EXA_NRANKS)
// without linking any of them.  The unique id travels over a plain TCP rendez-vous: rank 0 listens on MASTER_ADDR:MASTER_PORT
// (EXA_MASTER_ADDR / EXA_MASTER_PORT first; default 127.0.0.1:29517, i.e. one node), the others connect (with retries while rank 0
// is still starting) and read the payload.  No Python, no MPI library in the path.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/select.h>
#include <sys/socket.h>
#include <unistd.h>
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/exaconstit_driver.h"

namespace {

bool env_int(const char* name, int& out) {
   const char* e = std::getenv(name);
   if (!e || !*e) return false;
   char* end = nullptr; const long v = std::strtol(e, &end, 10);
   if (end == e) return false;
   while (*end == ' ' || *end == '\t') end++;
   if (*end != 0) return false;      // "8x", "2,3": not a number
   out = (int)v; return true;
}
bool env_set(const char* name) { const char* e = std::getenv(name); return e && *e; }

void set_err(char* err, int errlen, const std::string& m) { if (err && errlen > 0) { std::snprintf(err, (size_t)errlen, "%s", m.c_str()); } }

void send_all(int fd, const void* buf, size_t n) {
   const char* p = (const char*)buf;
   while (n) { const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL); if (k <= 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("send: ") + std::strerror(errno)); } p += k; n -= (size_t)k; }
}
void recv_all(int fd, void* buf, size_t n) {
   char* p = (char*)buf;
   while (n) { const ssize_t k = ::recv(fd, p, n, 0); if (k <= 0) { if (k < 0 && errno == EINTR) continue; throw std::runtime_error(k == 0 ? "recv: peer closed the connection" : std::string("recv: ") + std::strerror(errno)); } p += k; n -= (size_t)k; }
}

struct Closer { int fd; ~Closer() { if (fd >= 0) ::close(fd); } };   // closes a socket on every way out of a scope

struct Endpoint { std::string addr; int port; };
Endpoint endpoint() {
   Endpoint e{ "127.0.0.1", 29517 };
   for (const char* k : { "EXA_MASTER_ADDR", "MASTER_ADDR" }) if (const char* v = std::getenv(k)) if (*v) { e.addr = v; break; }
   for (const char* k : { "EXA_MASTER_PORT", "MASTER_PORT" }) { int p; if (env_int(k, p) && p > 0 && p < 65536) { e.port = p; break; } }
   return e;
}

}  // namespace

extern "C" {

// rank / size / local rank from the launcher's environment; 0 / 1 / 0 when there is none (plain `mechanics -opt ...`).
// Rank and size always come from the SAME launcher family.  Families whose variables also exist outside a launch are opt-in: SLURM_PROCID /
// SLURM_NTASKS count only inside an srun step (SLURM_STEP_ID; an sbatch script without srun exports them too), RANK / WORLD_SIZE only with
// a rendez-vous address (MASTER_ADDR, which torchrun-style launchers always export) - otherwise a plain `mechanics -opt x.toml` in such
// an environment would wait for peers that never come.  The source is reported on stderr whenever more than one rank is found.
int exa_bootstrap_env(int* rank, int* nranks, int* local_rank) {
   struct Family { const char* name; const char* rk; const char* nk; const char* lk; bool enabled; };
   const Family fam[] = {
      { "EXA_RANK / EXA_NRANKS", "EXA_RANK", "EXA_NRANKS", "EXA_LOCAL_RANK", true },
      { "MPICH / hydra (PMI_RANK, PMI_SIZE)", "PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", true },
      { "Open MPI (OMPI_COMM_WORLD_*)", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK", true },
      { "srun (SLURM_PROCID, SLURM_NTASKS)", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID", env_set("SLURM_STEP_ID") },
      { "torchrun-style (RANK, WORLD_SIZE)", "RANK", "WORLD_SIZE", "LOCAL_RANK", env_set("MASTER_ADDR") || env_set("EXA_MASTER_ADDR") },
   };
   int r = 0, n = 1, l = -1; const char* src = nullptr;
   for (const Family& f : fam) {
      int fr, fn;
      const bool hr = env_int(f.rk, fr), hn = env_int(f.nk, fn);
      if (!hr && !hn) continue;
      if (!f.enabled) continue;
      if (hr != hn) { std::fprintf(stderr, "exa_bootstrap: %s: only one of rank / size is set\n", f.name); return -1; }
      r = fr; n = fn; src = f.name;
      if (!env_int(f.lk, l)) l = -1;
      break;
   }
   if (n < 1 || r < 0 || r >= n) { std::fprintf(stderr, "exa_bootstrap: %s: rank %d of %d is not a valid rank\n", src ? src : "?", r, n); return -1; }
   if (l < 0) l = r;   // one node: the local rank is the rank
   if (n > 1 || env_set("EXA_VERBOSE")) std::fprintf(stderr, "exa_bootstrap: rank %d of %d (local rank %d) from %s\n", r, n, l, src ? src : "no launcher: one rank");
   *rank = r; *nranks = n; *local_rank = l;
   return 0;
}

// Gather-and-reply over TCP: every rank contributes `nbytes` (rank 0: listen + accept nranks-1 peers; others: connect with retries, send
// (magic, rank) - so that a stray connection cannot consume a slot - and their contribution); once everybody has arrived rank 0 calls
// fn(table of all contributions in rank order) to build the `reply_bytes` answer every rank receives.  nbytes may be 0 and fn null (plain
// broadcast of rank 0's `reply`).  timeout_s bounds the whole exchange.  A peer that reconnects replaces its earlier connection.
int exa_bootstrap_gather_reply(int rank, int nranks, const void* mine, int nbytes, void* reply, int reply_bytes, exa_bootstrap_reply_fn fn, void* user,
                               double timeout_s, char* err, int errlen) {
   if (nranks <= 1) {
      if (fn && fn(mine, 1, nbytes, reply, reply_bytes, user) != 0) { set_err(err, errlen, "exa_bootstrap_gather_reply: the reply callback failed"); return -1; }
      return 0;
   }
   static const uint32_t kMagic = 0x45584132u;   // "EXA2"
   const Endpoint ep = endpoint();
   const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 120.0);
   std::vector<int> peer_fd;      // rank 0: the open connection of every peer (closed on every way out)
   struct CloseAll { std::vector<int>& v; ~CloseAll() { for (int fd : v) if (fd >= 0) ::close(fd); } } close_all{ peer_fd };
   try {
      if (rank == 0) {
         // listen on the rendez-vous address itself (loopback by default), not on every interface
         addrinfo hints{}; hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM; hints.ai_flags = AI_PASSIVE;
         addrinfo* res = nullptr;
         sockaddr_in sa{}; sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)ep.port); sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
         if (::getaddrinfo(ep.addr.c_str(), nullptr, &hints, &res) == 0 && res) { sa.sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr; ::freeaddrinfo(res); }
         const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
         if (ls < 0) throw std::runtime_error(std::string("socket: ") + std::strerror(errno));
         Closer ls_guard{ ls };
         int one = 1; ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
         if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) {
            sa.sin_addr.s_addr = htonl(INADDR_ANY);      // the address is not local to rank 0 (NAT, alias): fall back to all interfaces
            if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) throw std::runtime_error(std::string("bind ") + ep.addr + ":" + std::to_string(ep.port) + ": " + std::strerror(errno));
         }
         if (::listen(ls, nranks) != 0) throw std::runtime_error(std::string("listen: ") + std::strerror(errno));
         peer_fd.assign((size_t)nranks, -1);
         std::vector<char> table((size_t)nranks * (size_t)nbytes);
         if (nbytes > 0) std::memcpy(table.data(), mine, (size_t)nbytes);
         int arrived = 0;
         while (arrived < nranks - 1) {
            timeval tv{}; const double left = std::chrono::duration<double>(t_end - std::chrono::steady_clock::now()).count();
            if (left <= 0) throw std::runtime_error("rendez-vous on " + ep.addr + ":" + std::to_string(ep.port) + " timed out: " + std::to_string(arrived) + " of " + std::to_string(nranks - 1) +
                                                    " peers connected (is every rank started, and with the same [EXA_]MASTER_ADDR / PORT?)");
            tv.tv_sec = (long)left; tv.tv_usec = (long)((left - (long)left) * 1e6);
            fd_set fds; FD_ZERO(&fds); FD_SET(ls, &fds);
            if (::select(ls + 1, &fds, nullptr, nullptr, &tv) <= 0) continue;
            const int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) continue;
            Closer fd_guard{ fd };
            timeval rt{ 10, 0 }; ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &rt, sizeof(rt)); ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &rt, sizeof(rt));
            uint32_t hello[3] = { 0, 0, 0 };
            std::vector<char> theirs((size_t)nbytes);
            try {
               recv_all(fd, hello, sizeof(hello));
               if (hello[0] != kMagic || hello[1] == 0 || hello[1] >= (uint32_t)nranks) continue;      // not one of ours
               if (hello[2] != (uint32_t)nbytes)      // one of ours, but it speaks another record size (another build of the library): say so instead of letting it time out
                  throw std::logic_error("rank " + std::to_string(hello[1]) + " contributes " + std::to_string(hello[2]) + " bytes where rank 0 expects " + std::to_string(nbytes) +
                                         " (ranks running different builds of libexaconstit_hip?)");
               if (nbytes > 0) recv_all(fd, theirs.data(), (size_t)nbytes);
            } catch (const std::logic_error&) { throw; }
            catch (...) { continue; }                // a peer that went away does not end the rendez-vous
            if (nbytes > 0) std::memcpy(table.data() + (size_t)hello[1] * (size_t)nbytes, theirs.data(), (size_t)nbytes);
            if (peer_fd[hello[1]] >= 0) ::close(peer_fd[hello[1]]); else arrived++;      // (a rank that retries is kept once)
            peer_fd[hello[1]] = fd; fd_guard.fd = -1;
         }
         if (fn && fn(table.data(), nranks, nbytes, reply, reply_bytes, user) != 0) throw std::runtime_error("the reply callback failed");
         for (int r = 1; r < nranks; r++) send_all(peer_fd[r], reply, (size_t)reply_bytes);
      } else {
         addrinfo hints{}; hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
         addrinfo* res = nullptr;
         if (::getaddrinfo(ep.addr.c_str(), std::to_string(ep.port).c_str(), &hints, &res) != 0 || !res) throw std::runtime_error("cannot resolve " + ep.addr);
         int fd = -1;
         for (;;) {
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
            if (fd >= 0) ::close(fd);
            fd = -1;
            if (std::chrono::steady_clock::now() > t_end) { ::freeaddrinfo(res); throw std::runtime_error("rendez-vous timed out connecting to " + ep.addr + ":" + std::to_string(ep.port)); }
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
         }
         ::freeaddrinfo(res);
         Closer fd_guard{ fd };
         timeval rt{ (long)std::max(10.0, timeout_s > 0 ? timeout_s : 120.0), 0 }; ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &rt, sizeof(rt));   // a rank 0 that dies after accept() does not hang its peers
         const uint32_t hello[3] = { kMagic, (uint32_t)rank, (uint32_t)nbytes };
         send_all(fd, hello, sizeof(hello));
         if (nbytes > 0) send_all(fd, mine, (size_t)nbytes);
         recv_all(fd, reply, (size_t)reply_bytes);
      }
      return 0;
   } catch (const std::exception& e) { set_err(err, errlen, std::string("exa_bootstrap (rank ") + std::to_string(rank) + "): " + e.what()); return -1; }
}

// Broadcast `nbytes` from rank 0 to all ranks (the gather-and-reply exchange with empty contributions)
int exa_bootstrap_bcast(int rank, int nranks, void* buf, int nbytes, double timeout_s, char* err, int errlen) {
   return exa_bootstrap_gather_reply(rank, nranks, nullptr, 0, buf, nbytes, nullptr, nullptr, timeout_s, err, errlen);
}

}  // extern "C"
