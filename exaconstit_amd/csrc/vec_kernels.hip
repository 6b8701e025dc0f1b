// Vector-level kernels of the Newton / PCG loop (gfx950), all on device-resident scalars so that one PCG iteration is a
// fixed sequence of launches with no host synchronisation:
//   MFEM CGSolver vector ops + dots (used at reference src/system_driver.cpp:166-177, src/mechanics_solver.cpp:62-121)
//   MechOperatorJacobiSmoother::Mult  (reference src/mechanics_operator_ext.cpp:38-55)            -> fused into cg_step1
//   essential-dof masking             (reference src/mechanics_operator_ext.cpp:146,172)
//   ExaModel::UpdateEndCoords         (reference src/mechanics_model.cpp:472-474)
// Dot products are weighted by 1/multiplicity of a node across ranks so that duplicated interface nodes count once.
#include "host/device_utils.hpp"

namespace {

constexpr int RBLK = 256;

__device__ __forceinline__ double block_sum(double v, double* sm) {
   sm[threadIdx.x] = v; __syncthreads();
   for (int s = RBLK / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
   const double r = sm[0]; __syncthreads();
   return r;
}

__global__ void k_update_coords(int64_t n, const double* __restrict__ xb, const double* __restrict__ v, double dt, double* __restrict__ xe) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) xe[i] = xb[i] + v[i] * dt;
}

__global__ void k_mask_zero(int64_t n, const uint8_t* __restrict__ m, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n && m[i]) y[i] = 0.0;
}

__global__ void k_mask_set(int64_t n, const uint8_t* __restrict__ m, const double* __restrict__ val, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n && m[i]) y[i] = val[i];
}

// y = a*x + b*y
__global__ void k_axpby(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) y[i] = a * x[i] + b * y[i];
}

// dinv = mask ? 1 : 1/diag   (MechOperatorJacobiSmoother::Setup); identity mode: all ones
__global__ void k_jacobi_setup(int64_t n, const uint8_t* __restrict__ m, const double* __restrict__ diag, int identity, double* __restrict__ dinv) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) dinv[i] = (identity || m[i]) ? 1.0 : 1.0 / diag[i];
}

__global__ void k_mask_one(int64_t n, const uint8_t* __restrict__ m, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n && m[i]) y[i] = 1.0;
}

// node of dof i of a byNODES vector of three components (i in [0, 3 nn)): i % nn without the 64-bit division
__device__ __forceinline__ int64_t node_of(const int64_t i, const int64_t nn) { return i - (i >= 2 * nn ? 2 * nn : (i >= nn ? nn : 0)); }

// partial weighted dot: partial[b] = sum_i w[i % nn] a[i] b[i]
__global__ void k_dot_partial(int64_t n, int64_t nn, const double* __restrict__ w, const double* __restrict__ a, const double* __restrict__ b,
                              const double* __restrict__ flag, double* __restrict__ partial) {
   __shared__ double sm[RBLK];
   if (flag && flag[0] != 0.0) return;
   double acc = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += w[node_of(i, nn)] * a[i] * b[i];
   const double s = block_sum(acc, sm);
   if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[slot] = sum partial
__global__ void k_reduce(int nb, const double* __restrict__ partial, const double* __restrict__ flag, double* __restrict__ out) {
   __shared__ double sm[RBLK];
   if (flag && flag[0] != 0.0) return;
   double acc = 0;
   for (int i = threadIdx.x; i < nb; i += RBLK) acc += partial[i];
   const double s = block_sum(acc, sm);
   if (threadIdx.x == 0) out[0] = s;
}

// PCG scalars: S[0]=nom S[1]=den S[2]=betanom S[3]=r0 S[4]=alpha S[5]=beta S[6]=done flag (0 run, 1 converged, -1 breakdown) S[7]=iterations
// S[8]=scratch for reductions  S[9]=scratch of the operator's dot  S[10]=number of iterations with (Ad, d) < 0  S[16], S[17]=nom / iterations in flight (consumer-side reductions)
__global__ void k_cg_init(double* S, double rel, double abs_) {       // after nom was reduced into S[8]
   const double nom = S[8];
   S[0] = nom; S[3] = fmax(nom * rel * rel, abs_ * abs_); S[7] = 0.0; S[2] = nom; S[11] = nom;   // S[11]: (r0, z0), S[2]: latest (r, z) - the achieved reduction is reported
   S[16] = nom; S[17] = 0.0;   // (nom, iterations) as the consumer-side reductions hand them from k_cg_step2z to k_cg_step1 (below)
   S[6] = (nom < 0.0) ? -1.0 : ((nom <= S[3]) ? 1.0 : 0.0);
}
__device__ __forceinline__ void cg_den_update(double* S) {            // den reduced into S[8]
   const double den = S[8];
   S[1] = den;
   // MFEM's CGSolver only warns when (Ad, d) < 0 ("The operator is not positive definite") and keeps iterating; it stops on den == 0
   if (den == 0.0) { S[6] = -1.0; return; }
   if (den < 0.0) S[10] += 1.0;      // S[10]: iterations with a negative denominator (reported by the driver)
   S[4] = S[0] / den;
}
__device__ __forceinline__ void cg_beta_update(double* S, double max_iter) {   // betanom reduced into S[8]
   const double bn = S[8];
   S[2] = bn; S[7] += 1.0;
   if (bn <= S[3]) { S[6] = 1.0; return; }
   if (S[7] >= max_iter) { S[6] = 2.0; return; }
   S[5] = bn / S[0]; S[0] = bn;
}
__global__ void k_cg_den(double* S) { if (S[6] == 0.0) cg_den_update(S); }
__global__ void k_cg_beta(double* S, double max_iter) { if (S[6] == 0.0) cg_beta_update(S, max_iter); }
// one rank: no all-reduce between the reduction of the partial sums and the scalar update, so they are one launch
// (MODE 1: betanom -> beta, MODE 2: den -> alpha)
template <int MODE>
__global__ void k_reduce_cg(int nb, const double* __restrict__ partial, double* __restrict__ S, double max_iter) {
   __shared__ double sm[RBLK];
   if (S[6] != 0.0) return;
   double acc = 0;
   for (int i = threadIdx.x; i < nb; i += RBLK) acc += partial[i];
   const double s = block_sum(acc, sm);
   if (threadIdx.x == 0) { S[8] = s; if (MODE == 1) cg_beta_update(S, max_iter); else cg_den_update(S); }
}

// x += alpha d; r -= alpha z; z = dinv r; partial of (r, z)_w
// IDENT: the preconditioner is the identity (the reference's Jacobi smoother with its never-refreshed dinv = 1): z == r, so dinv is
// not read and z is not written (k_cg_step2z<true> takes r instead)
#ifndef EXA_CG_X_NT
#define EXA_CG_X_NT 1
#endif
// Consumer-side reductions (one rank, fused loop, small systems): the two one-block launches that turned partial sums into alpha / beta (k_reduce_cg) are gone;
// every block of the NEXT kernel sums the partial sums itself - same order as k_reduce_cg, so the same bits in every block and as before - and block 0 keeps the
// scalar record.  What all blocks read in a kernel is never written in that kernel: k_cg_step2z reads (nom, iterations) from S[0], S[7] and leaves the new pair in
// S[16], S[17]; k_cg_step1 reads S[16] and commits both.  The done flag S[6] may be set by block 0 while other blocks start: a block takes it through one thread
// (no divergence across a barrier), and a block that still reads 0 reaches the same decision from the sum it computes.
// block_sum's tree with fewer barriers: the steps 128 and 64 through LDS, the steps 32 ... 1 inside the first wave (the same pairs are added: the same bits)
__device__ __forceinline__ double block_sum_w(double v, double* sm) {
   static_assert(RBLK == 256, "two LDS steps, then one wave");
   sm[threadIdx.x] = v; __syncthreads();
   if (threadIdx.x < 128) sm[threadIdx.x] += sm[threadIdx.x + 128];
   __syncthreads();
   if (threadIdx.x < 64) {
      double x = sm[threadIdx.x] + sm[threadIdx.x + 64];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) x += __shfl_down(x, s);      // lane t < s adds lane t + s, as sm[t] += sm[t + s] does; the other lanes' sums are not used
      if (threadIdx.x == 0) sm[0] = x;
   }
   __syncthreads();
   const double r = sm[0]; __syncthreads();
   return r;
}
// sum of the partial sums and the done flag in one go: the flag is taken by one thread (no divergence across the barriers) and published by the tree's first barrier,
// so its round trip to L2 runs beside the loads of the partial sums
__device__ __forceinline__ double sum_partials(int nb, const double* __restrict__ partial, const double* S, double* sm, double* sflag, bool& done) {
   double acc = 0;
   for (int i = threadIdx.x; i < nb; i += RBLK) acc += partial[i];
   if (threadIdx.x == 0) *sflag = S[6];
   const double r = block_sum_w(acc, sm);
   done = *sflag != 0.0;
   return r;
}
// RED: alpha from the partial sums of the denominator (k_mask_dot_partial / k_dot_partial wrote partialD[0 .. nbD))
template <bool IDENT, bool XNT = (EXA_CG_X_NT != 0), bool RED = false>
__global__ void k_cg_step1(int64_t n, int64_t nn, double* S, const double* __restrict__ w, const double* __restrict__ dinv,
                           const double* __restrict__ d, double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, double* __restrict__ partial,
                           int nbD, const double* __restrict__ partialD) {
   __shared__ double sm[RBLK];
   double alpha;
   if constexpr (RED) {
      __shared__ double sflag;
      const double nom = S[16], its = S[17];      // (requested before the reduction: nothing below waits for them alone)
      bool done;
      const double den = sum_partials(nbD, partialD, S, sm, &sflag, done);      // cg_den_update, in every block
      if (done) return;
      const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
      if (den == 0.0) { if (lead) { S[8] = den; S[1] = den; S[6] = -1.0; } return; }
      alpha = nom / den;
      if (lead) { S[8] = den; S[1] = den; if (den < 0.0) S[10] += 1.0; S[4] = alpha; S[0] = nom; S[7] = its; }
   } else {
      if (S[6] != 0.0) return;
      alpha = S[4];
   }
   double acc = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      if constexpr (XNT) __builtin_nontemporal_store(__builtin_nontemporal_load(&x[i]) + alpha * d[i], &x[i]);      // x is touched once per iteration: it need not displace d, r, z from the caches
      else x[i] += alpha * d[i];      // (small systems: every vector stays cached from one iteration to the next, vk_x_nt)
      const double ri = r[i] - alpha * z[i];
      r[i] = ri;
      const double zi = IDENT ? ri : dinv[i] * ri;
      if (!IDENT) z[i] = zi;
      acc += w[node_of(i, nn)] * ri * zi;
   }
   const double s = block_sum(acc, sm);
   if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// d = z + beta d
__global__ void k_cg_step2(int64_t n, const double* __restrict__ S, const double* __restrict__ z, double* __restrict__ d) {
   if (S[6] != 0.0) return;
   const double beta = S[5];
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = z[i] + beta * d[i];
}

// d = z + beta d, then z = 0: the operator action that follows accumulates into z with atomics, so the separate fill pass is folded in
// RED: beta from the partial sums of (r, z) (k_cg_step1 wrote partialN[0 .. nbN)); cg_beta_update in every block
template <bool IDENT, bool RED = false>
__global__ void k_cg_step2z(int64_t n, double* S, double* __restrict__ z, const double* __restrict__ r, double* __restrict__ d,
                            int nbN, const double* __restrict__ partialN, double max_iter) {
   double beta;
   if constexpr (RED) {
      __shared__ double sm[RBLK]; __shared__ double sflag;
      const double nom = S[0], thr = S[3], it = S[7] + 1.0;
      bool done;
      const double bn = sum_partials(nbN, partialN, S, sm, &sflag, done);
      if (done) return;
      const bool conv = bn <= thr, capped = !conv && it >= max_iter;
      if (blockIdx.x == 0 && threadIdx.x == 0) {
         S[8] = bn; S[2] = bn; S[17] = it;
         if (conv) S[6] = 1.0; else if (capped) S[6] = 2.0; else { S[5] = bn / nom; S[16] = bn; }
      }
      if (conv || capped) return;
      beta = bn / nom;
   } else {
      if (S[6] != 0.0) return;
      beta = S[5];
   }
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { d[i] = (IDENT ? r[i] : z[i]) + beta * d[i]; z[i] = 0.0; }
}

// ---- single-reduction PCG (Chronopoulos & Gear), used on more than one rank: one 16-byte all-reduce per iteration instead of two 8-byte ones
// scalars as above; S[8], S[9] hold the reduced pair gamma = (r, u), delta = (A u, u)
__global__ void k_cg2_init(double* S, double rel, double abs_) {          // after (gamma, delta) were reduced into S[8], S[9]
   const double g = S[8], dl = S[9];
   S[0] = g; S[1] = dl; S[3] = fmax(g * rel * rel, abs_ * abs_); S[7] = 0.0; S[5] = 0.0; S[2] = g; S[11] = g;
   S[6] = (g < 0.0) ? -1.0 : ((g <= S[3]) ? 1.0 : 0.0);
   if (S[6] == 0.0) { if (dl == 0.0) S[6] = -1.0; else { if (dl < 0.0) S[10] += 1.0; S[4] = g / dl; } }
}
__global__ void k_cg2_scalars(double* S, double max_iter) {               // new (gamma, delta) in S[8], S[9]
   if (S[6] != 0.0) return;
   const double gn = S[8], dl = S[9];
   S[2] = gn; S[7] += 1.0;
   if (gn <= S[3]) { S[6] = 1.0; return; }
   if (S[7] >= max_iter) { S[6] = 2.0; return; }
   const double beta = gn / S[0];
   const double den = dl - beta * gn / S[4];                              // = (A p, p) of the next direction
   S[1] = den;
   if (den == 0.0) { S[6] = -1.0; return; }
   if (den < 0.0) S[10] += 1.0;
   S[5] = beta; S[4] = gn / den; S[0] = gn;
}
// p = u + beta p; q = s + beta q; x += alpha p; r -= alpha q; u = dinv r; s = 0 (the operator action that follows accumulates into it)
template <bool IDENT, bool XNT = (EXA_CG_X_NT != 0)>
__global__ void k_cg2_update(int64_t n, const double* __restrict__ S, const double* __restrict__ dinv, double* __restrict__ x, double* __restrict__ r,
                             double* __restrict__ u, double* __restrict__ p, double* __restrict__ sv, double* __restrict__ q) {
   if (S[6] != 0.0) return;
   const double alpha = S[4], beta = S[5];
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const double ui = IDENT ? r[i] : u[i];
      const double pi = ui + beta * p[i], qi = sv[i] + beta * q[i];
      p[i] = pi; q[i] = qi;
      if constexpr (XNT) __builtin_nontemporal_store(__builtin_nontemporal_load(&x[i]) + alpha * pi, &x[i]);      // (x and q are touched once per iteration: see k_cg_step1)
      else x[i] += alpha * pi;
      const double ri = r[i] - alpha * qi;
      r[i] = ri;
      if (!IDENT) u[i] = dinv[i] * ri;
      sv[i] = 0.0;
   }
}
// essential rows of s zeroed in place + the two partial weighted dots (r, u) and (s, u) in one pass (partial[0..nb) and partial[nb..2nb))
template <bool IDENT>
__global__ void k_cg2_dots(int64_t n, int64_t nn, const double* __restrict__ w, const uint8_t* __restrict__ m, const double* __restrict__ r, const double* __restrict__ u,
                           double* __restrict__ sv, const double* __restrict__ flag, double* __restrict__ partial) {
   __shared__ double sm[RBLK];
   if (flag[0] != 0.0) return;
   double a0 = 0, a1 = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const double ui = IDENT ? r[i] : u[i], wi = w[node_of(i, nn)], si = sv[i]; const uint8_t mi = m[i];      // (all requested before the mask byte is looked at)
      a0 += wi * r[i] * ui;
      if (mi) sv[i] = 0.0;
      a1 += mi ? 0.0 : wi * si * ui;
   }
   const double s0 = block_sum(a0, sm); __syncthreads();
   const double s1 = block_sum(a1, sm);
   if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s1; }
}
__global__ void k_reduce2(int nb, const double* __restrict__ partial, const double* __restrict__ flag, double* __restrict__ out2) {
   __shared__ double sm[RBLK];
   // solver finished: the remaining iterations of the chunk are no-ops, but their all-reduce still runs - feed it zeros instead of the stale
   // pair (which every further all-reduce would multiply by the number of ranks)
   if (flag[0] != 0.0) { if (threadIdx.x == 0) { out2[0] = 0.0; out2[1] = 0.0; } return; }
   double a0 = 0, a1 = 0;
   for (int i = threadIdx.x; i < nb; i += RBLK) { a0 += partial[i]; a1 += partial[nb + i]; }
   const double s0 = block_sum(a0, sm); __syncthreads();
   const double s1 = block_sum(a1, sm);
   if (threadIdx.x == 0) { out2[0] = s0; out2[1] = s1; }
}

// essential rows of b zeroed in place + partial weighted dot (a, b): the operator's output mask and the PCG denominator in one pass
__global__ void k_mask_dot_partial(int64_t n, int64_t nn, const double* __restrict__ w, const uint8_t* __restrict__ m, const double* __restrict__ a,
                                   double* __restrict__ b, const double* __restrict__ flag, double* __restrict__ partial) {
   __shared__ double sm[RBLK];
   if (flag && flag[0] != 0.0) return;
   // Four dofs per thread and pass, every operand requested before the mask byte is looked at: written as `if (m[i]) ... else acc += w a b` the three
   // loads wait for the mask byte (two dependent round trips per pass, 17 bytes in flight per thread: 33 us for 110 MB at 128^3)
   double acc = 0;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
      double av[4], bv[4], wv[4]; uint8_t mv[4]; bool in[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int64_t i = i0 + u * stride; in[u] = i < n;
         const int64_t j = in[u] ? i : i0;      // (clamped: the loads below are unconditional)
         mv[u] = m[j]; av[u] = a[j]; bv[u] = b[j]; wv[u] = w[node_of(j, nn)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         if (in[u] && mv[u]) b[i0 + u * stride] = 0.0;
         acc += (in[u] && !mv[u]) ? wv[u] * av[u] * bv[u] : 0.0;
      }
   }
   const double s = block_sum(acc, sm);
   if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void k_fill_if(int64_t n, const double* __restrict__ flag, double val, double* __restrict__ y) {
   if (flag && flag[0] != 0.0) return;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = val;
}

// z = dinv .* r
__global__ void k_pointwise(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) y[i] = a[i] * b[i];
}

// halo pack / unpack-add on lists of local dof indices
__global__ void k_pack(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ y, double* __restrict__ buf) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) buf[i] = y[idx[i]];
}
__global__ void k_unpack_add(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ y) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) atomicAdd(&y[idx[i]], buf[i]);   // a dof shared with several neighbours occurs once per neighbour segment
}

inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// coordinate minima for the velocity-gradient BC origin (reference src/system_driver.cpp:355-396): out[d] = min_g x(g,d)
struct double9 { double a[9]; };
__global__ void __launch_bounds__(RBLK) k_min3_partial(int64_t nn, const double* __restrict__ x, double* __restrict__ partial) {
   __shared__ double sm[RBLK];
   for (int d = 0; d < 3; d++) {
      double m = 1.7976931348623157e308;
      for (int64_t i = (int64_t)blockIdx.x * RBLK + threadIdx.x; i < nn; i += (int64_t)gridDim.x * RBLK) m = fmin(m, x[i + nn * d]);
      sm[threadIdx.x] = m; __syncthreads();
      for (int st = RBLK / 2; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + st]); __syncthreads(); }
      if (threadIdx.x == 0) partial[blockIdx.x + (int64_t)gridDim.x * d] = sm[0];
      __syncthreads();
   }
}
__global__ void __launch_bounds__(RBLK) k_min3_final(int nb, const double* __restrict__ partial, double* __restrict__ out) {
   __shared__ double sm[RBLK];
   for (int d = 0; d < 3; d++) {
      double m = 1.7976931348623157e308;
      for (int i = threadIdx.x; i < nb; i += RBLK) m = fmin(m, partial[i + (int64_t)nb * d]);
      sm[threadIdx.x] = m; __syncthreads();
      for (int st = RBLK / 2; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + st]); __syncthreads(); }
      if (threadIdx.x == 0) out[d] = sm[0];
      __syncthreads();
   }
}
// v(g,i) = sum_j L(i,j) (x(g,j) - x0(j)) on the dofs flagged in m (reference src/system_driver.cpp:400-424)
__global__ void k_vgrad_velocity(int64_t nn, const uint8_t* __restrict__ m, const double* __restrict__ x, const double* __restrict__ org, double9 L, double* __restrict__ v) {
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= nn) return;
   const double d0 = x[g] - org[0], d1 = x[g + nn] - org[1], d2 = x[g + 2 * nn] - org[2];
   for (int i = 0; i < 3; i++) if (m[g + nn * i]) v[g + nn * i] = L.a[3 * i] * d0 + L.a[3 * i + 1] * d1 + L.a[3 * i + 2] * d2;
}

// quadrature function in the element-blocked layout [block of 64 elements][q][W][lane] -> the reference's (W, Q, E) layout (bench / parity tooling of
// the adapter route: host/driver_capi.hip, exa_driver_bench_adapter_route)
__global__ void k_qf_eb64_to_aos(const int W, const int Q, const int64_t n, const double* __restrict__ src, double* __restrict__ dst) {
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % W); const int64_t pt = i / W; const int q = (int)(pt % Q); const int64_t e = pt / Q;
      dst[i] = src[((((e >> 6) * Q + q) * (int64_t)W + c) << 6) + (e & 63)];
   }
}
// out[0] = max |a - b|, out[1] = max |a| (bit patterns of non-negative doubles order like the numbers; out zeroed by the launcher); a NaN on either side -> out[0] = +inf.
// W > 0: the arrays hold W-vectors and component `skip` is left out of the two maxima; out[2] counts its entries that differ (the evaluation counter of a state array)
__global__ void k_max_abs_diff(const int64_t n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out, const int W, const int skip) {
   double d = 0.0, m = 0.0; unsigned long long cnt = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const double x = a[i], y = b[i]; const double t = fabs(x - y);
      if (W > 0 && (int)(i % W) == skip) { cnt += (x != y) ? 1ull : 0ull; continue; }
      d = (t == t) ? fmax(d, t) : __longlong_as_double(0x7ff0000000000000ll); m = fmax(m, fabs(x));
   }
   for (int o = 32; o > 0; o >>= 1) { d = fmax(d, __shfl_xor(d, o)); m = fmax(m, __shfl_xor(m, o)); }
   if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], (unsigned long long)__double_as_longlong(d)); atomicMax(&out[1], (unsigned long long)__double_as_longlong(m)); }
   if (cnt) atomicAdd(&out[2], cnt);
}

inline unsigned gblk(int64_t n) { const int64_t b = (n + RBLK - 1) / RBLK; return (unsigned)(b < exa_host::DOT_BLOCKS ? (b > 0 ? b : 1) : exa_host::DOT_BLOCKS); }

}  // namespace

namespace exa_host {

void vk_update_coords(int64_t n, const double* xb, const double* v, double dt, double* xe, hipStream_t s) { hipLaunchKernelGGL(k_update_coords, dim3(nblk(n)), dim3(256), 0, s, n, xb, v, dt, xe); }
void vk_mask_zero(int64_t n, const uint8_t* m, double* y, hipStream_t s) { hipLaunchKernelGGL(k_mask_zero, dim3(nblk(n)), dim3(256), 0, s, n, m, y); }
void vk_mask_set(int64_t n, const uint8_t* m, const double* val, double* y, hipStream_t s) { hipLaunchKernelGGL(k_mask_set, dim3(nblk(n)), dim3(256), 0, s, n, m, val, y); }
void vk_mask_one(int64_t n, const uint8_t* m, double* y, hipStream_t s) { hipLaunchKernelGGL(k_mask_one, dim3(nblk(n)), dim3(256), 0, s, n, m, y); }
void vk_axpby(int64_t n, double a, const double* x, double b, double* y, hipStream_t s) { hipLaunchKernelGGL(k_axpby, dim3(nblk(n)), dim3(256), 0, s, n, a, x, b, y); }
void vk_jacobi_setup(int64_t n, const uint8_t* m, const double* diag, int identity, double* dinv, hipStream_t s) { hipLaunchKernelGGL(k_jacobi_setup, dim3(nblk(n)), dim3(256), 0, s, n, m, diag, identity, dinv); }
void vk_pointwise(int64_t n, const double* a, const double* b, double* y, hipStream_t s) { hipLaunchKernelGGL(k_pointwise, dim3(nblk(n)), dim3(256), 0, s, n, a, b, y); }
void vk_fill_if(int64_t n, const double* flag, double val, double* y, hipStream_t s) { hipLaunchKernelGGL(k_fill_if, dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, flag, val, y); }
// local part of the weighted dot: result in out[0] (device)
// non-temporal update of the solution vector only when the six vectors of an iteration do not stay cached anyway (exa_internal.hpp, exa_stream_nt: 64^3 and up;
// measured with the record stream's hint in one A/B, see there)
static inline bool vk_x_nt(int64_t n) {
   static const double min_mb = [] { const char* e = std::getenv("EXA_NT_MIN_MB"); return e ? std::atof(e) : 128.0; }();
   return EXA_CG_X_NT != 0 && (double)n * 8.0 * 6.0 >= min_mb * 1048576.0 * 0.25;
}
// (node_of: the weighted sums take a byNODES vector of at most three components)
static inline void check_3nn(int64_t n, int64_t nn) { if (n > 3 * nn) throw std::invalid_argument("weighted vector kernels: n must not exceed 3 * nn (byNODES, three components)"); }
void vk_dot(int64_t n, int64_t nn, const double* w, const double* a, const double* b, const double* flag, double* partial, double* out, hipStream_t s) {
   check_3nn(n, nn);
   const unsigned nb = gblk(n);
   hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(RBLK), 0, s, n, nn, w, a, b, flag, partial);
   hipLaunchKernelGGL(k_reduce, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, flag, out);
}
void vk_min3(int64_t nn, const double* x, double* partial, double* out3, hipStream_t s) {
   const unsigned nb = gblk(nn) < (unsigned)(DOT_BLOCKS / 3) ? gblk(nn) : (unsigned)(DOT_BLOCKS / 3);
   hipLaunchKernelGGL(k_min3_partial, dim3(nb), dim3(RBLK), 0, s, nn, x, partial);
   hipLaunchKernelGGL(k_min3_final, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, out3);
}
void vk_vgrad_velocity(int64_t nn, const uint8_t* m, const double* x, const double* org, const double* L9, double* v, hipStream_t s) {
   double9 L; for (int i = 0; i < 9; i++) L.a[i] = L9[i];
   hipLaunchKernelGGL(k_vgrad_velocity, dim3(nblk(nn)), dim3(256), 0, s, nn, m, x, org, L, v);
}
#ifndef EXA_CG_RED_GRID
#define EXA_CG_RED_GRID 1
#endif
// partialN != nullptr: consumer-side reduction of the (r, z) partial sums k_cg_step1 left there (one rank; S[6] / S[5] / S[16..17] updated by the launch itself)
void vk_cg_step2z(int64_t n, double* S, double* z, const double* r, double* d, bool ident, hipStream_t s, const double* partialN, int max_iter) {
   const unsigned nb = gblk(n);
   if (partialN) {
      const unsigned g = nb * EXA_CG_RED_GRID;      // (every block pays the prologue: fewer, longer blocks than the plain kernel's 4 nb)
      if (ident) hipLaunchKernelGGL((k_cg_step2z<true, true>), dim3(g), dim3(RBLK), 0, s, n, S, z, r, d, (int)nb, partialN, (double)max_iter);
      else hipLaunchKernelGGL((k_cg_step2z<false, true>), dim3(g), dim3(RBLK), 0, s, n, S, z, r, d, (int)nb, partialN, (double)max_iter);
   } else {
      if (ident) hipLaunchKernelGGL((k_cg_step2z<true, false>), dim3(nb * 4), dim3(RBLK), 0, s, n, S, z, r, d, 0, (const double*)nullptr, 0.0);
      else hipLaunchKernelGGL((k_cg_step2z<false, false>), dim3(nb * 4), dim3(RBLK), 0, s, n, S, z, r, d, 0, (const double*)nullptr, 0.0);
   }
}
// out == nullptr and fuse_den_S == nullptr: the partial sums stay in `partial` for the consumer-side reduction of k_cg_step1<.., RED>
void vk_mask_dot(int64_t n, int64_t nn, const double* w, const uint8_t* m, const double* a, double* b, const double* flag, double* partial, double* out, hipStream_t s, double* fuse_den_S) {
   check_3nn(n, nn);
   const unsigned nb = gblk(n);
   hipLaunchKernelGGL(k_mask_dot_partial, dim3(nb), dim3(RBLK), 0, s, n, nn, w, m, a, b, flag, partial);
   if (fuse_den_S) hipLaunchKernelGGL(k_reduce_cg<2>, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, fuse_den_S, 0.0);
   else if (out) hipLaunchKernelGGL(k_reduce, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, flag, out);
}
void vk_dot_partial(int64_t n, int64_t nn, const double* w, const double* a, const double* b, const double* flag, double* partial, hipStream_t s) {
   check_3nn(n, nn);
   hipLaunchKernelGGL(k_dot_partial, dim3(gblk(n)), dim3(RBLK), 0, s, n, nn, w, a, b, flag, partial);
}
// failed local solves of the constitutive launch behind a residual: the count goes to count_out (as a double, next to the norm in the read-back)
// and a non-zero count makes the local sum +inf, so that Newton sees a non-finite residual on every rank after the all-reduce
__global__ void k_poison_if_failed(const int* __restrict__ fail_count, double* __restrict__ sum, double* __restrict__ count_out) {
   const int f = fail_count[0];
   count_out[0] = (double)f;
   if (f > 0) sum[0] = __longlong_as_double(0x7ff0000000000000ll);
}
void vk_poison_if_failed(const int* fail_count, double* sum, double* count_out, hipStream_t s) { hipLaunchKernelGGL(k_poison_if_failed, dim3(1), dim3(1), 0, s, fail_count, sum, count_out); }
void vk_cg_init(double* S, double rel, double abs_, hipStream_t s) { hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(1), 0, s, S, rel, abs_); }
void vk_cg_den(double* S, hipStream_t s) { hipLaunchKernelGGL(k_cg_den, dim3(1), dim3(1), 0, s, S); }
void vk_cg_beta(double* S, int max_iter, hipStream_t s) { hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(1), 0, s, S, (double)max_iter); }
// fuse_beta: one rank, the reduction also performs the beta update (no vk_cg_beta launch afterwards)
// partialD != nullptr: consumer-side reductions - alpha from the denominator's partial sums in partialD, and the (r, z) partial sums are left in `partial` for vk_cg_step2z
void vk_cg_step1(int64_t n, int64_t nn, double* S, const double* w, const double* dinv, const double* d, double* x, double* r, double* z, double* partial, bool ident,
                 bool fuse_beta, int max_iter, hipStream_t s, const double* partialD) {
   check_3nn(n, nn);
   const unsigned nb = gblk(n);
   const bool xnt = vk_x_nt(n);
#define STEP1(I, X, R) hipLaunchKernelGGL((k_cg_step1<I, X, R>), dim3(nb), dim3(RBLK), 0, s, n, nn, S, w, dinv, d, x, r, z, partial, (int)nb, partialD)
   if (partialD) { if (ident) { if (xnt) STEP1(true, true, true); else STEP1(true, false, true); } else { if (xnt) STEP1(false, true, true); else STEP1(false, false, true); } return; }
   if (ident) { if (xnt) STEP1(true, true, false); else STEP1(true, false, false); } else { if (xnt) STEP1(false, true, false); else STEP1(false, false, false); }
#undef STEP1
   if (fuse_beta) hipLaunchKernelGGL(k_reduce_cg<1>, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, S, (double)max_iter);
   else hipLaunchKernelGGL(k_reduce, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, S + 6, S + 8);
}
void vk_cg2_init(double* S, double rel, double abs_, hipStream_t s) { hipLaunchKernelGGL(k_cg2_init, dim3(1), dim3(1), 0, s, S, rel, abs_); }
void vk_cg2_scalars(double* S, int max_iter, hipStream_t s) { hipLaunchKernelGGL(k_cg2_scalars, dim3(1), dim3(1), 0, s, S, (double)max_iter); }
void vk_cg2_update(int64_t n, const double* S, const double* dinv, double* x, double* r, double* u, double* p, double* sv, double* q, bool ident, hipStream_t s) {
   const bool xnt = vk_x_nt(n);
   if (ident) { if (xnt) hipLaunchKernelGGL((k_cg2_update<true, true>), dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, S, dinv, x, r, u, p, sv, q); else hipLaunchKernelGGL((k_cg2_update<true, false>), dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, S, dinv, x, r, u, p, sv, q); }
   else { if (xnt) hipLaunchKernelGGL((k_cg2_update<false, true>), dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, S, dinv, x, r, u, p, sv, q); else hipLaunchKernelGGL((k_cg2_update<false, false>), dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, S, dinv, x, r, u, p, sv, q); }
}
// local parts of (r, u)_w and (s, u)_w -> out2[0..1] (device); masks the essential rows of s on the way
void vk_cg2_dots(int64_t n, int64_t nn, const double* w, const uint8_t* m, const double* r, const double* u, double* sv, const double* flag, double* partial, double* out2,
                 bool ident, hipStream_t s) {
   check_3nn(n, nn);
   const unsigned nb = gblk(n) < (unsigned)(DOT_BLOCKS / 2) ? gblk(n) : (unsigned)(DOT_BLOCKS / 2);
   if (ident) hipLaunchKernelGGL(k_cg2_dots<true>, dim3(nb), dim3(RBLK), 0, s, n, nn, w, m, r, u, sv, flag, partial);
   else hipLaunchKernelGGL(k_cg2_dots<false>, dim3(nb), dim3(RBLK), 0, s, n, nn, w, m, r, u, sv, flag, partial);
   hipLaunchKernelGGL(k_reduce2, dim3(1), dim3(RBLK), 0, s, (int)nb, partial, flag, out2);
}
void vk_cg_step2(int64_t n, const double* S, const double* z, double* d, hipStream_t s) { hipLaunchKernelGGL(k_cg_step2, dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, S, z, d); }
void vk_qf_eb64_to_aos(int W, int Q, int64_t E, const double* src, double* dst, hipStream_t s) {
   const int64_t n = (int64_t)W * Q * E;
   hipLaunchKernelGGL(k_qf_eb64_to_aos, dim3(gblk(n) * 8), dim3(RBLK), 0, s, W, Q, n, src, dst);
}
void vk_max_abs_diff(int64_t n, const double* a, const double* b, double* out3_dev, hipStream_t s, int W, int skip) {
   (void)hipMemsetAsync(out3_dev, 0, 3 * sizeof(double), s);
   hipLaunchKernelGGL(k_max_abs_diff, dim3(gblk(n) * 4), dim3(RBLK), 0, s, n, a, b, reinterpret_cast<unsigned long long*>(out3_dev), W, skip);
}
void vk_pack(int64_t n, const int32_t* idx, const double* y, double* buf, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_pack, dim3(nblk(n)), dim3(256), 0, s, n, idx, y, buf); }
void vk_unpack_add(int64_t n, const int32_t* idx, const double* buf, double* y, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(k_unpack_add, dim3(nblk(n)), dim3(256), 0, s, n, idx, buf, y); }

}  // namespace exa_host
