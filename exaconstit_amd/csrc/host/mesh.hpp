// Cartesian hex RVE mesh + block domain decomposition (host side).
//   mesh generation : what the reference gets from Mesh::MakeCartesian3D(nx,ny,nz,HEX,sx,sy,sz,sfc=false)
//                     (reference src/mechanics_driver.cpp:247-253): vertices and elements x-fastest, p = 1 native vertex order
//   boundary ids    : reference src/mechanics_driver.cpp:1207-1227 (1 z-min, 2 x-min, 3 y-min, 4 z-max, 5 x-max, 6 y-max)
//   decomposition   : replaces ParMesh/METIS (reference src/mechanics_driver.cpp:312) by a structured block split (generated
//                     meshes) or a recursive coordinate bisection of the element centroids (file meshes); interface
//                     nodes are duplicated on every rank that touches them and kept consistent by a neighbour halo-sum
//                     (equivalent to the reference's P^T followed by P, spec src/mechanics_operator_ext.cpp:149-157).
#pragma once
#include <array>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include <map>

void exa_gll_nodes_01(int np, std::vector<double>& x);   // host_tables.cpp

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
   // meshes read from a file (Mesh.type = "other"): grain id = element attribute, boundary ids = boundary-element attributes
   bool from_file = false; std::vector<int> elem_attr; std::vector<std::vector<uint8_t>> bdr_nodes;   // [attribute - 1][node]
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
      std::vector<double> gll; exa_gll_nodes_01(np, gll);
      for (int k = 0; k < ne[2]; k++) for (int j = 0; j < ne[1]; j++) for (int i = 0; i < ne[0]; i++) {
         const int e = i + ne[0] * (j + ne[1] * k);
         elem_gid[e] = (int64_t)(e0[0] + i) + (int64_t)N[0] * ((e0[1] + j) + (int64_t)N[1] * (e0[2] + k));
         for (int c = 0; c < np; c++) for (int b = 0; b < np; b++) for (int a = 0; a < np; a++)
            conn[nat[a + np * (b + np * c)] + (size_t)n * e] = (i * p + a) + nn[0] * ((j * p + b) + nn[1] * (k * p + c));
      }
      for (int k = 0; k < nn[2]; k++) for (int j = 0; j < nn[1]; j++) for (int i = 0; i < nn[0]; i++) {
         const int g = i + nn[0] * (j + nn[1] * k);
         const int gi[3] = { e0[0] * p + i, e0[1] * p + j, e0[2] * p + k };   // nodes of an element at the Gauss-Lobatto points of the H1 basis
         for (int d = 0; d < 3; d++) {
            const int ge = std::min(gi[d] / p, N[d] - 1), a = gi[d] - ge * p;
            X[g + (size_t)NN * d] = len[d] * (ge + gll[a]) / N[d];
         }
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

   // Elements that touch a node shared with another rank first (stable within the two groups).  The operator action then runs those
   // blocks, hands the shared dofs to the halo exchange, and computes the interior elements while the exchange is on the wire
   // ("easily overlapped with interior elements"; reference coupling sites src/mechanics_operator_ext.cpp:149-157).
   // Everything per element (connectivity, global id, attribute) moves together; nothing else is indexed by local element number.
   int E_bdr = 0;   // number of leading elements that touch shared nodes (0 until order_boundary_first ran, and on one rank)
   void order_boundary_first() {
      E_bdr = 0;
      if (nbrs.empty() || E == 0) return;
      std::vector<uint8_t> shared((size_t)NN, 0);
      for (const Neighbor& nb : nbrs) for (int32_t d : nb.dofs) shared[(size_t)(d % NN)] = 1;
      std::vector<int> perm; perm.reserve(E);
      std::vector<uint8_t> isb((size_t)E, 0);
      for (int e = 0; e < E; e++) { for (int a = 0; a < n; a++) if (shared[(size_t)conn[a + (size_t)n * e]]) { isb[e] = 1; break; } }
      for (int e = 0; e < E; e++) if (isb[e]) perm.push_back(e);
      E_bdr = (int)perm.size();
      for (int e = 0; e < E; e++) if (!isb[e]) perm.push_back(e);
      std::vector<int32_t> c2(conn.size()); std::vector<int64_t> g2(elem_gid.size()); std::vector<int> a2(elem_attr.size());
      for (int e = 0; e < E; e++) {
         for (int a = 0; a < n; a++) c2[a + (size_t)n * e] = conn[a + (size_t)n * perm[e]];
         if (!elem_gid.empty()) g2[e] = elem_gid[perm[e]];
         if (!elem_attr.empty()) a2[e] = elem_attr[perm[e]];
      }
      conn.swap(c2); elem_gid.swap(g2); elem_attr.swap(a2);
   }

   // MFEM mesh v1.0 reader for trilinear hexahedra (what the reference gets from `Mesh(mesh_file, 1, 1, true)`, src/mechanics_driver.cpp:239-241;
   // format of workflows/Stage3/main_simulations/simulation.mesh): sections `dimension`, `elements` (attr geom=5 v0..v7, MFEM vertex order =
   // this repo's native order), `boundary` (attr geom=3 v0..v3), `vertices` with inline coordinates or a `nodes` grid function (H1 order 1).
   // Every rank reads the whole file; with more than one rank the elements are split by recursive coordinate bisection of their
   // centroids (the reference uses METIS through ParMesh, src/mechanics_driver.cpp:312; any partition gives the same operator) and the
   // rank keeps its elements, the nodes they touch, and one Neighbor per rank it shares nodes with (dofs ordered by global node id on
   // both sides).
   void build_from_mfem_mesh(const std::string& path, int rank_, int nranks_, int order = 1) {
      std::ifstream f(path);
      if (!f) throw std::runtime_error("Cannot open mesh file: " + path);
      auto next_token_line = [&](std::string& line) {   // next non-empty, non-comment line
         while (std::getline(f, line)) { size_t a = line.find_first_not_of(" \t\r"); if (a == std::string::npos || line[a] == '#') continue; line = line.substr(a); while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back(); return true; }
         return false;
      };
      std::string line;
      if (!next_token_line(line) || line.rfind("MFEM mesh v1.0", 0) != 0) throw std::runtime_error("Not an MFEM mesh v1.0 file: " + path);
      p = 1; n = 8; rank = rank_; nranks = nranks_; from_file = true;
      for (int d = 0; d < 3; d++) { N[d] = 0; pg[d] = 1; rc[d] = 0; e0[d] = 0; ne[d] = 0; nn[d] = 0; }
      int nv = -1; std::vector<std::array<int, 5>> bdr;
      while (next_token_line(line)) {
         if (line == "dimension") { next_token_line(line); if (std::stoi(line) != 3) throw std::runtime_error("mesh: dimension must be 3"); }
         else if (line == "elements") {
            next_token_line(line); E = std::stoi(line);
            conn.resize((size_t)8 * E); elem_attr.resize(E); elem_gid.resize(E);
            for (int e = 0; e < E; e++) {
               next_token_line(line); std::istringstream is(line); int attr, geom; is >> attr >> geom;
               if (geom != 5) throw std::runtime_error("mesh: only hexahedra (geometry 5) are supported");
               elem_attr[e] = attr; elem_gid[e] = e;
               for (int a = 0; a < 8; a++) { int v; if (!(is >> v)) throw std::runtime_error("mesh: short element line"); conn[a + (size_t)8 * e] = v; }
            }
         } else if (line == "boundary") {
            next_token_line(line); const int nbe = std::stoi(line);
            for (int b = 0; b < nbe; b++) {
               next_token_line(line); std::istringstream is(line); int attr, geom; is >> attr >> geom;
               if (geom != 3) throw std::runtime_error("mesh: only quadrilateral boundary elements (geometry 3) are supported");
               std::array<int, 5> q; q[0] = attr; for (int a = 0; a < 4; a++) is >> q[1 + a];
               bdr.push_back(q);
            }
         } else if (line == "vertices") {
            next_token_line(line); nv = std::stoi(line); NN = nv; X.assign((size_t)3 * NN, 0.0);
            next_token_line(line);
            bool by_nodes_order = false;
            if (line == "nodes") {   // grid-function form: header lines up to the first coordinate line
               while (next_token_line(line)) {
                  if (line.rfind("Ordering:", 0) == 0) { by_nodes_order = (std::stoi(line.substr(9)) == 0); break; }
                  if (line.rfind("FiniteElementCollection:", 0) == 0 && line.find("H1_3D_P1") == std::string::npos) throw std::runtime_error("mesh: only H1_3D_P1 nodes are supported");
               }
            } else if (std::stoi(line) != 3) throw std::runtime_error("mesh: vertex dimension must be 3");
            if (by_nodes_order) { for (int d = 0; d < 3; d++) for (int g = 0; g < NN; g++) { next_token_line(line); X[g + (size_t)NN * d] = std::stod(line); } }
            else for (int g = 0; g < NN; g++) { next_token_line(line); std::istringstream is(line); for (int d = 0; d < 3; d++) is >> X[g + (size_t)NN * d]; }
         }
      }
      if (E <= 0 || nv <= 0) throw std::runtime_error("mesh: missing elements or vertices section");
      for (int32_t v : conn) if (v < 0 || v >= NN) throw std::runtime_error("mesh: vertex index out of range");
      int maxattr = 0; for (auto& q : bdr) maxattr = std::max(maxattr, q[0]);
      bdr_nodes.assign(maxattr, std::vector<uint8_t>(NN, 0));
      for (auto& q : bdr) for (int a = 0; a < 4; a++) bdr_nodes[q[0] - 1][q[1 + a]] = 1;
      if (order == 2) elevate_to_p2(bdr);
      else if (order >= 3 && order <= 6) elevate_to_order(order, bdr);
      else if (order != 1) throw std::runtime_error("mesh: file meshes run at p_refinement = 1 ... 6");
      weight.assign(NN, 1.0); nbrs.clear();
      if (nranks > 1) localize(rcb_owner(nranks));
   }

   // p_refinement = 2 on a file mesh (the reference raises the order of the nodal space of any mesh, src/mechanics_driver.cpp:300-306): one new
   // node per edge, per face and per element of the trilinear mesh, at the mean of the vertices it belongs to (which is where the trilinear map
   // puts the triquadratic nodes); local numbering = native_order(2) (vertices, edges, faces in MFEM's order, centre).  One node per entity
   // means no orientation bookkeeping - the reason orders above 2 are left to generated meshes.
   void elevate_to_p2(const std::vector<std::array<int, 5>>& bdr) {
      static const int Ed[12][2] = { { 0, 1 }, { 1, 2 }, { 3, 2 }, { 0, 3 }, { 4, 5 }, { 5, 6 }, { 7, 6 }, { 4, 7 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
      static const int Fc[6][4] = { { 0, 1, 2, 3 }, { 0, 1, 5, 4 }, { 1, 2, 6, 5 }, { 3, 2, 6, 7 }, { 0, 3, 7, 4 }, { 4, 5, 6, 7 } };
      const int nv = NN;
      std::map<std::array<int, 2>, int> edge_node; std::map<std::array<int, 4>, int> face_node;
      std::vector<std::array<double, 3>> xnew;
      auto coord = [&](int g, int d) { return g < nv ? X[g + (size_t)nv * d] : xnew[g - nv][d]; };
      auto add_node = [&](const int* vs, int k) { std::array<double, 3> c{ 0, 0, 0 }; for (int i = 0; i < k; i++) for (int d = 0; d < 3; d++) c[d] += X[vs[i] + (size_t)nv * d]; for (int d = 0; d < 3; d++) c[d] /= k; xnew.push_back(c); return nv + (int)xnew.size() - 1; };
      auto edge_of = [&](int a, int b) { std::array<int, 2> key{ std::min(a, b), std::max(a, b) }; auto it = edge_node.find(key); if (it != edge_node.end()) return it->second; const int vs[2] = { a, b }; const int g = add_node(vs, 2); edge_node.emplace(key, g); return g; };
      auto face_of = [&](const int* v4) { std::array<int, 4> key{ v4[0], v4[1], v4[2], v4[3] }; std::sort(key.begin(), key.end()); auto it = face_node.find(key); if (it != face_node.end()) return it->second; const int g = add_node(v4, 4); face_node.emplace(key, g); return g; };
      std::vector<int32_t> c2((size_t)27 * E);
      for (int e = 0; e < E; e++) {
         const int32_t* v = &conn[(size_t)8 * e];
         int32_t* w = &c2[(size_t)27 * e];
         for (int a = 0; a < 8; a++) w[a] = v[a];
         for (int k = 0; k < 12; k++) w[8 + k] = edge_of(v[Ed[k][0]], v[Ed[k][1]]);
         for (int k = 0; k < 6; k++) { const int v4[4] = { v[Fc[k][0]], v[Fc[k][1]], v[Fc[k][2]], v[Fc[k][3]] }; w[20 + k] = face_of(v4); }
         const int v8[8] = { v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7] };
         w[26] = add_node(v8, 8);
      }
      const int NN2 = nv + (int)xnew.size();
      std::vector<double> X2((size_t)3 * NN2);
      for (int g = 0; g < NN2; g++) for (int d = 0; d < 3; d++) X2[g + (size_t)NN2 * d] = coord(g, d);
      for (auto& b : bdr_nodes) b.resize(NN2, 0);
      for (auto& q : bdr) {      // a boundary quadrilateral constrains its edge nodes and its face node too
         for (int a = 0; a < 4; a++) { auto it = edge_node.find({ std::min(q[1 + a], q[1 + (a + 1) % 4]), std::max(q[1 + a], q[1 + (a + 1) % 4]) }); if (it != edge_node.end()) bdr_nodes[q[0] - 1][it->second] = 1; }
         std::array<int, 4> key{ q[1], q[2], q[3], q[4] }; std::sort(key.begin(), key.end());
         auto it = face_node.find(key); if (it != face_node.end()) bdr_nodes[q[0] - 1][it->second] = 1;
      }
      conn.swap(c2); X.swap(X2); NN = NN2; p = 2; n = 27;
   }

   // p_refinement = 3 ... 6 on a file mesh: (p - 1) nodes per edge, (p - 1)^2 per face and (p - 1)^3 per element of the trilinear mesh, at the images
   // of the Gauss-Lobatto points under the element's trilinear map (src/mechanics_driver.cpp:300-306 raises the order of the nodal space the same
   // way; straight-sided hexahedra).  Two elements that share an edge or a face traverse it in their own local directions, so a shared node is
   // identified by a key that does not depend on the element: an edge node by (smaller vertex id, larger vertex id, index counted from the smaller
   // one), a face node by the four sorted vertex ids and its (u, w) index in the frame whose origin is the face's smallest vertex id and whose
   // first axis points to the smaller of that corner's two neighbours on the face.  The Gauss-Lobatto points are symmetric, so both elements
   // compute the same position.  Local numbering = native_order(p).
   void elevate_to_order(const int order, const std::vector<std::array<int, 5>>& bdr) {
      const int pp = order, np = pp + 1, n2 = np * np * np, nv = NN;
      const std::vector<int> nat = native_order(pp);
      std::vector<double> gll; exa_gll_nodes_01(np, gll);
      static const int V[8][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 1, 1, 1 }, { 0, 1, 1 } };
      int corner_of[2][2][2];      // lexicographic corner (a, b, c) -> native vertex index
      for (int v = 0; v < 8; v++) corner_of[V[v][0]][V[v][1]][V[v][2]] = v;
      std::map<std::array<int64_t, 7>, int> shared_node;      // (kind, ids..., indices) -> global node
      std::vector<std::array<double, 3>> xnew;
      std::vector<int32_t> c2((size_t)n2 * E);
      auto edge_key = [&](int a, int b, int t) { return a < b ? std::array<int64_t, 7>{ 1, a, b, t, 0, 0, 0 } : std::array<int64_t, 7>{ 1, b, a, pp - t, 0, 0, 0 }; };
      // face with corner ids c[al][be] (al along the first free axis s, be along the second t), node at (s, t)
      auto face_key = [&](const int c[2][2], int s_, int t_) {
         int a0 = 0, b0 = 0;
         for (int al = 0; al < 2; al++) for (int be = 0; be < 2; be++) if (c[al][be] < c[a0][b0]) { a0 = al; b0 = be; }
         const int sp = a0 ? pp - s_ : s_, tp = b0 ? pp - t_ : t_;
         const bool s_first = c[1 - a0][b0] < c[a0][1 - b0];
         std::array<int64_t, 4> ids{ c[0][0], c[0][1], c[1][0], c[1][1] }; std::sort(ids.begin(), ids.end());
         return std::array<int64_t, 7>{ 2, ids[0], ids[1], ids[2], ids[3], s_first ? sp : tp, s_first ? tp : sp };
      };
      for (int e = 0; e < E; e++) {
         const int32_t* v = &conn[(size_t)8 * e];
         auto position = [&](int i, int j, int k) {
            std::array<double, 3> x{ 0, 0, 0 };
            const double w[3][2] = { { 1.0 - gll[i], gll[i] }, { 1.0 - gll[j], gll[j] }, { 1.0 - gll[k], gll[k] } };
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) {
               const int g = v[corner_of[a][b][c]]; const double ww = w[0][a] * w[1][b] * w[2][c];
               for (int d = 0; d < 3; d++) x[d] += ww * X[g + (size_t)nv * d];
            }
            return x;
         };
         for (int k = 0; k < np; k++) for (int j = 0; j < np; j++) for (int i = 0; i < np; i++) {
            const int idx[3] = { i, j, k };
            int fixed[3], nfix = 0, freeax[3], nfree = 0;
            for (int d = 0; d < 3; d++) { if (idx[d] == 0 || idx[d] == pp) fixed[nfix++] = d; else freeax[nfree++] = d; }
            int g;
            if (nfix == 3) g = v[corner_of[i / pp][j / pp][k / pp]];
            else if (nfix == 0) { xnew.push_back(position(i, j, k)); g = nv + (int)xnew.size() - 1; }
            else {
               std::array<int64_t, 7> key;
               if (nfix == 2) {      // edge along freeax[0]
                  int ca[3] = { i / pp, j / pp, k / pp }, cb[3] = { i / pp, j / pp, k / pp };
                  ca[freeax[0]] = 0; cb[freeax[0]] = 1;
                  key = edge_key(v[corner_of[ca[0]][ca[1]][ca[2]]], v[corner_of[cb[0]][cb[1]][cb[2]]], idx[freeax[0]]);
               } else {              // face: fixed[0] constant, free axes freeax[0] (s) and freeax[1] (t)
                  int c[2][2];
                  for (int al = 0; al < 2; al++) for (int be = 0; be < 2; be++) {
                     int cc[3]; cc[fixed[0]] = idx[fixed[0]] / pp; cc[freeax[0]] = al; cc[freeax[1]] = be;
                     c[al][be] = v[corner_of[cc[0]][cc[1]][cc[2]]];
                  }
                  key = face_key(c, idx[freeax[0]], idx[freeax[1]]);
               }
               auto it = shared_node.find(key);
               if (it != shared_node.end()) g = it->second;
               else { xnew.push_back(position(i, j, k)); g = nv + (int)xnew.size() - 1; shared_node.emplace(key, g); }
            }
            c2[nat[i + np * (j + np * k)] + (size_t)n2 * e] = g;
         }
      }
      const int NN2 = nv + (int)xnew.size();
      std::vector<double> X2((size_t)3 * NN2);
      for (int g = 0; g < NN2; g++) for (int d = 0; d < 3; d++) X2[g + (size_t)NN2 * d] = g < nv ? X[g + (size_t)nv * d] : xnew[g - nv][d];
      for (auto& b : bdr_nodes) b.resize(NN2, 0);
      for (auto& q : bdr) {      // a boundary quadrilateral constrains the nodes on its edges and in its interior too
         for (int a = 0; a < 4; a++) for (int t = 1; t < pp; t++) { auto it = shared_node.find(edge_key(q[1 + a], q[1 + (a + 1) % 4], t)); if (it != shared_node.end()) bdr_nodes[q[0] - 1][it->second] = 1; }
         const int c[2][2] = { { q[1], q[4] }, { q[2], q[3] } };      // corners in cyclic order q1 q2 q3 q4: s from q1 to q2, t from q1 to q4
         for (int s_ = 1; s_ < pp; s_++) for (int t_ = 1; t_ < pp; t_++) { auto it = shared_node.find(face_key(c, s_, t_)); if (it != shared_node.end()) bdr_nodes[q[0] - 1][it->second] = 1; }
      }
      conn.swap(c2); X.swap(X2); NN = NN2; p = pp; n = n2;
   }

   // element -> rank by recursive coordinate bisection: split the longest extent of the centroid cloud at the element count that
   // matches the share of ranks on each side; deterministic (ties broken by element index), so every rank computes the same map
   std::vector<int> rcb_owner(int nr) const {
      std::vector<std::array<double, 3>> c(E);
      for (int e = 0; e < E; e++) for (int d = 0; d < 3; d++) { double v = 0; for (int a = 0; a < n; a++) v += X[conn[a + (size_t)n * e] + (size_t)NN * d]; c[e][d] = v / n; }
      std::vector<int> owner(E, 0), ids(E);
      for (int e = 0; e < E; e++) ids[e] = e;
      struct Job { int lo, hi, r0, nr; };
      std::vector<Job> st; st.push_back({ 0, E, 0, nr });
      while (!st.empty()) {
         const Job j = st.back(); st.pop_back();
         if (j.nr == 1) { for (int i = j.lo; i < j.hi; i++) owner[ids[i]] = j.r0; continue; }
         double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
         for (int i = j.lo; i < j.hi; i++) for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], c[ids[i]][d]); hi[d] = std::max(hi[d], c[ids[i]][d]); }
         int ax = 0; for (int d = 1; d < 3; d++) if ((hi[d] - lo[d]) > (hi[ax] - lo[ax]) * (1.0 + 1e-12)) ax = d;
         const int nl = j.nr / 2; const int cut = j.lo + (int)(((int64_t)(j.hi - j.lo) * nl) / j.nr);
         std::sort(ids.begin() + j.lo, ids.begin() + j.hi, [&](int a, int b) { return c[a][ax] != c[b][ax] ? c[a][ax] < c[b][ax] : a < b; });
         st.push_back({ j.lo, cut, j.r0, nl }); st.push_back({ cut, j.hi, j.r0 + nl, j.nr - nl });
      }
      return owner;
   }

   // keep this rank's elements (original order) and the nodes they touch; build weights and neighbour lists from the global picture
   void localize(const std::vector<int>& owner) {
      const int NNg = NN, Eg = E;
      std::vector<std::vector<int>> sharers(NNg);   // ranks touching each global node (sorted, unique)
      for (int e = 0; e < Eg; e++) for (int a = 0; a < n; a++) { auto& v = sharers[conn[a + (size_t)n * e]]; if (std::find(v.begin(), v.end(), owner[e]) == v.end()) v.push_back(owner[e]); }
      for (auto& v : sharers) std::sort(v.begin(), v.end());
      std::vector<int> l2g, g2l(NNg, -1);
      for (int g = 0; g < NNg; g++) if (std::binary_search(sharers[g].begin(), sharers[g].end(), rank)) { g2l[g] = (int)l2g.size(); l2g.push_back(g); }
      const int NNl = (int)l2g.size();
      std::vector<int32_t> lconn; std::vector<int> lattr; std::vector<int64_t> lgid;
      for (int e = 0; e < Eg; e++) if (owner[e] == rank) {
         for (int a = 0; a < n; a++) lconn.push_back(g2l[conn[a + (size_t)n * e]]);
         lattr.push_back(elem_attr[e]); lgid.push_back(e);
      }
      std::vector<double> lX((size_t)3 * NNl);
      for (int l = 0; l < NNl; l++) for (int d = 0; d < 3; d++) lX[l + (size_t)NNl * d] = X[l2g[l] + (size_t)NNg * d];
      std::vector<std::vector<uint8_t>> lb(bdr_nodes.size(), std::vector<uint8_t>(NNl, 0));
      for (size_t k = 0; k < bdr_nodes.size(); k++) for (int l = 0; l < NNl; l++) lb[k][l] = bdr_nodes[k][l2g[l]];
      weight.assign(NNl, 1.0);
      std::vector<std::vector<int>> shared_with(nranks);   // local nodes shared with each other rank, ascending global id (l2g is ascending)
      for (int l = 0; l < NNl; l++) {
         const auto& v = sharers[l2g[l]];
         weight[l] = 1.0 / (double)v.size();
         for (int r : v) if (r != rank) shared_with[r].push_back(l);
      }
      nbrs.clear();
      for (int r = 0; r < nranks; r++) if (!shared_with[r].empty()) {
         Neighbor nb; nb.rank = r;
         for (int c = 0; c < 3; c++) for (int l : shared_with[r]) nb.dofs.push_back(l + NNl * c);
         nbrs.push_back(std::move(nb));
      }
      conn.swap(lconn); elem_attr.swap(lattr); elem_gid.swap(lgid); X.swap(lX); bdr_nodes.swap(lb);
      E = (int)elem_gid.size(); NN = NNl;
      if (E == 0) throw std::runtime_error("mesh: a rank received no elements (more ranks than the partitioner can serve)");
   }

   // number of boundary attributes (generated meshes: the six faces, reference src/mechanics_driver.cpp:1207-1227)
   int num_bdr_attr() const { return from_file ? (int)bdr_nodes.size() : 6; }
   // is local node g on global boundary face id?
   bool on_face(int g, int id) const {
      if (from_file) return id >= 1 && id <= (int)bdr_nodes.size() && bdr_nodes[id - 1][g] != 0;
      const int i = g % nn[0] + e0[0] * p, j = (g / nn[0]) % nn[1] + e0[1] * p, k = g / (nn[0] * nn[1]) + e0[2] * p;
      switch (id) { case 1: return k == 0; case 2: return i == 0; case 3: return j == 0; case 4: return k == N[2] * p; case 5: return i == N[0] * p; case 6: return j == N[1] * p; default: return false; }
   }
};

}  // namespace exa_host
