// Cartesian hex RVE mesh + block domain decomposition (host side).
//   mesh generation : what the reference gets from Mesh::MakeCartesian3D(nx,ny,nz,HEX,sx,sy,sz,sfc=false)
//                     (reference src/mechanics_driver.cpp:247-253): vertices and elements x-fastest, p = 1 native vertex order
//   boundary ids    : reference src/mechanics_driver.cpp:1207-1227 (1 z-min, 2 x-min, 3 y-min, 4 z-max, 5 x-max, 6 y-max)
//   decomposition   : replaces ParMesh/METIS (reference src/mechanics_driver.cpp:312) by a structured block split; interface
//                     nodes are duplicated on every rank that touches them and kept consistent by a neighbour halo-sum
//                     (equivalent to the reference's P^T followed by P, spec src/mechanics_operator_ext.cpp:149-157).
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace exa_host {

struct Neighbor { int rank; std::vector<int32_t> dofs; };   // local dof = node + NN * comp, identical order on both sides

struct Partition {
   int N[3] = { 1, 1, 1 }; double len[3] = { 1, 1, 1 };
   int pg[3] = { 1, 1, 1 }, rank = 0, nranks = 1, rc[3] = { 0, 0, 0 };
   int e0[3], ne[3], nn[3];
   int p = 1, n = 8;                 // H1 order, nodes per element (p+1)^3
   int E = 0, NN = 0;
   std::vector<int32_t> conn;        // (n, E), native node order
   std::vector<double> X;            // byNODES (NN, 3)
   std::vector<int64_t> elem_gid;    // global element index, x fastest
   std::vector<double> weight;       // 1 / (number of ranks holding the node)
   std::vector<Neighbor> nbrs;
   int64_t E_global() const { return (int64_t)N[0] * N[1] * N[2]; }

   static void split(int n, int p, int r, int& start, int& cnt) { const int b = n / p, rem = n % p; cnt = b + (r < rem ? 1 : 0); start = r * b + (r < rem ? r : rem); }

   static std::array<int, 3> grid_for(int nranks) {
      std::array<int, 3> g = { 1, 1, 1 };
      int n = nranks, d = 2;   // distribute prime factors round-robin starting with z (slowest index)
      std::vector<int> f;
      while (n > 1) { if (n % d == 0) { f.push_back(d); n /= d; } else d++; }
      int k = 2;
      for (int i = (int)f.size() - 1; i >= 0; i--) { g[k] *= f[i]; k = (k + 2) % 3; }
      return g;
   }

   // lexicographic (i,j,k) -> native index of the order-p hexahedron (vertices, edge, face, volume interiors)
   static std::vector<int> native_order(int p) {
      const int np = p + 1;
      std::vector<int> m((size_t)np * np * np, -1);
      auto Lx = [&](int i, int j, int k) { return i + np * (j + np * k); };
      int c = 0;
      static const int V[8][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 1, 1, 1 }, { 0, 1, 1 } };
      for (auto& v : V) m[Lx(v[0] * p, v[1] * p, v[2] * p)] = c++;
      static const int Ed[12][2] = { { 0, 1 }, { 1, 2 }, { 3, 2 }, { 0, 3 }, { 4, 5 }, { 5, 6 }, { 7, 6 }, { 4, 7 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
      for (auto& ed : Ed) for (int t = 1; t < p; t++) {
         int ijk[3]; for (int d = 0; d < 3; d++) ijk[d] = V[ed[0]][d] * p + (V[ed[1]][d] - V[ed[0]][d]) * t;
         m[Lx(ijk[0], ijk[1], ijk[2])] = c++;
      }
      for (int f = 0; f < 6; f++) for (int t2 = 1; t2 < p; t2++) for (int t1 = 1; t1 < p; t1++) {
         int i = t1, j = t2, k = 0;
         if (f == 1) { i = t1; j = 0; k = t2; } else if (f == 2) { i = p; j = t1; k = t2; } else if (f == 3) { i = t1; j = p; k = t2; }
         else if (f == 4) { i = 0; j = t1; k = t2; } else if (f == 5) { i = t1; j = t2; k = p; }
         m[Lx(i, j, k)] = c++;
      }
      for (int k = 1; k < p; k++) for (int j = 1; j < p; j++) for (int i = 1; i < p; i++) m[Lx(i, j, k)] = c++;
      return m;
   }

   void build(const int Nn[3], const double L[3], int rank_, int nranks_, int order = 1) {
      p = order; n = (p + 1) * (p + 1) * (p + 1);
      for (int d = 0; d < 3; d++) { N[d] = Nn[d]; len[d] = L[d]; }
      rank = rank_; nranks = nranks_;
      auto g = grid_for(nranks); for (int d = 0; d < 3; d++) pg[d] = g[d];
      rc[0] = rank % pg[0]; rc[1] = (rank / pg[0]) % pg[1]; rc[2] = rank / (pg[0] * pg[1]);
      for (int d = 0; d < 3; d++) { split(N[d], pg[d], rc[d], e0[d], ne[d]); nn[d] = ne[d] * p + 1; }
      E = ne[0] * ne[1] * ne[2]; NN = nn[0] * nn[1] * nn[2];
      conn.resize((size_t)n * E); X.resize((size_t)3 * NN); elem_gid.resize(E); weight.resize(NN);
      const std::vector<int> nat = native_order(p);
      const int np = p + 1;
      for (int k = 0; k < ne[2]; k++) for (int j = 0; j < ne[1]; j++) for (int i = 0; i < ne[0]; i++) {
         const int e = i + ne[0] * (j + ne[1] * k);
         elem_gid[e] = (int64_t)(e0[0] + i) + (int64_t)N[0] * ((e0[1] + j) + (int64_t)N[1] * (e0[2] + k));
         for (int c = 0; c < np; c++) for (int b = 0; b < np; b++) for (int a = 0; a < np; a++)
            conn[nat[a + np * (b + np * c)] + (size_t)n * e] = (i * p + a) + nn[0] * ((j * p + b) + nn[1] * (k * p + c));
      }
      for (int k = 0; k < nn[2]; k++) for (int j = 0; j < nn[1]; j++) for (int i = 0; i < nn[0]; i++) {
         const int g = i + nn[0] * (j + nn[1] * k);
         const int gi[3] = { e0[0] * p + i, e0[1] * p + j, e0[2] * p + k };   // equispaced nodes (p <= 2: identical to Gauss-Lobatto)
         for (int d = 0; d < 3; d++) X[g + (size_t)NN * d] = len[d] * gi[d] / (N[d] * p);
         int mult = 1;
         for (int d = 0; d < 3; d++) {
            const int li = (d == 0 ? i : (d == 1 ? j : k));
            const bool lo = (li == 0 && rc[d] > 0), hi = (li == nn[d] - 1 && rc[d] < pg[d] - 1);
            if (lo || hi) mult *= 2;
         }
         weight[g] = 1.0 / mult;
      }
      nbrs.clear();
      for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
         if (!dx && !dy && !dz) continue;
         const int o[3] = { dx, dy, dz }; int r2[3]; bool ok = true;
         for (int d = 0; d < 3; d++) { r2[d] = rc[d] + o[d]; if (r2[d] < 0 || r2[d] >= pg[d]) ok = false; }
         if (!ok) continue;
         Neighbor nb; nb.rank = r2[0] + pg[0] * (r2[1] + pg[1] * r2[2]);
         int lo[3], hi[3];
         for (int d = 0; d < 3; d++) { if (o[d] < 0) { lo[d] = 0; hi[d] = 1; } else if (o[d] > 0) { lo[d] = nn[d] - 1; hi[d] = nn[d]; } else { lo[d] = 0; hi[d] = nn[d]; } }
         for (int c = 0; c < 3; c++) for (int k = lo[2]; k < hi[2]; k++) for (int j = lo[1]; j < hi[1]; j++) for (int i = lo[0]; i < hi[0]; i++)
            nb.dofs.push_back(i + nn[0] * (j + nn[1] * k) + NN * c);
         nbrs.push_back(std::move(nb));
      }
   }

   // is local node g on global boundary face id?
   bool on_face(int g, int id) const {
      const int i = g % nn[0] + e0[0] * p, j = (g / nn[0]) % nn[1] + e0[1] * p, k = g / (nn[0] * nn[1]) + e0[2] * p;
      switch (id) { case 1: return k == 0; case 2: return i == 0; case 3: return j == 0; case 4: return k == N[2] * p; case 5: return i == N[0] * p; default: return j == N[1] * p; }
   }
};

}  // namespace exa_host
