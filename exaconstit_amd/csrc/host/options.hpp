// options.toml reader for the stand-alone driver: the reference's option SCHEMA (reference src/options.toml,
// src/option_parser.cpp:26-932, src/option_types.hpp) for the subset of its hot path — auto-generated hex mesh,
// ExaCMech models, PA/EA assembly, NR/NRLS + PCG — parsed with a small TOML-subset reader (the reference vendors toml11;
// out of scope here).  Unknown keys are ignored like the reference's `toml::find_or` defaults; unsupported values abort
// with the reference's wording where one exists.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace exa_host {

struct TomlValue {
   enum Kind { NONE, NUM, STR, BOOL, ARR } kind = NONE;
   double num = 0; std::string str; bool b = false; std::vector<TomlValue> arr;
};

class TomlDoc {
 public:
   std::map<std::string, TomlValue> kv;   // "Table.Sub.key" -> value
   static TomlDoc parse_file(const std::string& path) {
      std::ifstream f(path);
      if (!f) throw std::runtime_error("Cannot open options file: " + path);
      std::stringstream ss; ss << f.rdbuf();
      return parse(ss.str());
   }
   static TomlDoc parse(const std::string& text) {
      TomlDoc d; size_t pos = 0; std::string table;
      const size_t n = text.size();
      auto skip_ws = [&](bool newlines) { while (pos < n) { char c = text[pos]; if (c == '#') { while (pos < n && text[pos] != '\n') pos++; } else if (c == ' ' || c == '\t' || c == '\r' || (newlines && c == '\n')) pos++; else break; } };
      while (true) {
         skip_ws(true);
         if (pos >= n) break;
         if (text[pos] == '[') {
            size_t e = text.find(']', pos); if (e == std::string::npos) throw std::runtime_error("toml: unterminated table header");
            table = trim(text.substr(pos + 1, e - pos - 1)); pos = e + 1; continue;
         }
         size_t eq = text.find('=', pos); if (eq == std::string::npos) throw std::runtime_error("toml: expected key = value");
         std::string key = trim(text.substr(pos, eq - pos)); pos = eq + 1;
         skip_ws(false);
         TomlValue v = parse_value(text, pos);
         d.kv[table.empty() ? key : table + "." + key] = v;
      }
      return d;
   }
   bool has(const std::string& k) const { return kv.count(k) > 0; }
   bool has_table(const std::string& t) const { for (auto& p : kv) if (p.first.compare(0, t.size() + 1, t + ".") == 0) return true; return false; }
   double num(const std::string& k, double def) const { auto it = kv.find(k); return (it != kv.end() && it->second.kind == TomlValue::NUM) ? it->second.num : def; }
   std::string str(const std::string& k, const std::string& def) const { auto it = kv.find(k); return (it != kv.end() && it->second.kind == TomlValue::STR) ? it->second.str : def; }
   bool boolean(const std::string& k, bool def) const { auto it = kv.find(k); return (it != kv.end() && it->second.kind == TomlValue::BOOL) ? it->second.b : def; }
   const TomlValue* get(const std::string& k) const { auto it = kv.find(k); return it == kv.end() ? nullptr : &it->second; }

 private:
   static std::string trim(const std::string& s) { size_t a = 0, b = s.size(); while (a < b && std::isspace((unsigned char)s[a])) a++; while (b > a && std::isspace((unsigned char)s[b - 1])) b--; return s.substr(a, b - a); }
   static TomlValue parse_value(const std::string& t, size_t& pos) {
      TomlValue v; const size_t n = t.size();
      auto skip = [&]() { while (pos < n) { char c = t[pos]; if (c == '#') { while (pos < n && t[pos] != '\n') pos++; } else if (std::isspace((unsigned char)c)) pos++; else break; } };
      if (t[pos] == '"' || t[pos] == '\'') {
         const char q = t[pos]; size_t e = t.find(q, pos + 1); if (e == std::string::npos) throw std::runtime_error("toml: unterminated string");
         v.kind = TomlValue::STR; v.str = t.substr(pos + 1, e - pos - 1); pos = e + 1;
      } else if (t[pos] == '[') {
         v.kind = TomlValue::ARR; pos++;
         while (true) { skip(); if (pos >= n) throw std::runtime_error("toml: unterminated array"); if (t[pos] == ']') { pos++; break; } if (t[pos] == ',') { pos++; continue; } v.arr.push_back(parse_value(t, pos)); }
      } else if (t.compare(pos, 4, "true") == 0) { v.kind = TomlValue::BOOL; v.b = true; pos += 4; }
      else if (t.compare(pos, 5, "false") == 0) { v.kind = TomlValue::BOOL; v.b = false; pos += 5; }
      else {
         size_t e = pos; while (e < n && (std::isalnum((unsigned char)t[e]) || t[e] == '+' || t[e] == '-' || t[e] == '.' || t[e] == '_')) e++;
         std::string tok = t.substr(pos, e - pos); std::string clean; for (char c : tok) if (c != '_') clean += c;
         char* endp = nullptr; v.num = std::strtod(clean.c_str(), &endp);
         if (endp == clean.c_str()) throw std::runtime_error("toml: cannot parse value near '" + tok + "'");
         v.kind = TomlValue::NUM; pos = e;
      }
      return v;
   }
};

enum class Assembly { PA, EA };            // reference src/option_types.hpp (FULL is mapped to EA: same operator, no sparse matrix/AMG)
enum class NLSolver { NR, NRLS };
enum class XtalType { FCC, BCC };
enum class SlipType { POWERVOCE, POWERVOCENL, MTSDD };

// comps < 0: velocity-gradient condition on components |comp| (reference src/option_parser.cpp:178-195)
struct BCEntry { int step; std::vector<int> ids, comps; std::vector<double> vals; double vgrad[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }; };

struct ExaOptions {
   std::string basedir;
   double temp_k = 298.0;
   std::string props_file; int nprops = 0;
   std::string ori_file, grain_file; int num_grains = 0; std::string ori_type = "quat";
   std::vector<BCEntry> bcs; bool vgrad_origin_flag = false; double vgrad_origin[3] = { 0, 0, 0 };
   XtalType xtal = XtalType::FCC; SlipType slip = SlipType::POWERVOCE;
   bool dt_cust = false, dt_auto = false; std::vector<double> cust_dt; double dt = 1.0, t_final = 1.0;
   double dt_min = 1.0, dt_scale = 0.25; int nsteps = 1; std::string auto_dt_fname = "auto_dt_out.txt";
   std::string avg_stress_fname = "avg_stress.txt", avg_def_grad_fname = "avg_def_grad.txt", avg_pl_work_fname = "avg_pl_work.txt", avg_dp_tensor_fname = "avg_dp_tensor.txt";
   bool additional_avgs = false;
   Assembly assembly = Assembly::EA; NLSolver nl_solver = NLSolver::NR; std::string integ_model = "FULL";
   int newton_iter = 25; double newton_rel = 1e-5, newton_abs = 1e-10;
   int krylov_iter = 200; double krylov_rel = 1e-10, krylov_abs = 1e-30; std::string krylov_solver = "PCG";
   int ref_ser = 0, order = 1; int ncuts[3] = { 1, 1, 1 }; double length[3] = { 1, 1, 1 }; std::string mesh_type = "auto", mesh_file;

   static std::vector<double> load_numbers(const std::string& path) {
      std::ifstream f(path); if (!f) throw std::runtime_error("Cannot open data file: " + path);
      std::vector<double> v; double x; while (f >> x) v.push_back(x); return v;
   }
   std::string resolve(const std::string& f) const { return (f.empty() || f[0] == '/') ? f : basedir + "/" + f; }

   static std::string lower(std::string s) { for (auto& c : s) c = (char)std::tolower((unsigned char)c); return s; }

   void parse_options(const std::string& path) {
      const size_t sl = path.find_last_of('/'); basedir = sl == std::string::npos ? "." : path.substr(0, sl);
      TomlDoc d = TomlDoc::parse_file(path);
      temp_k = d.num("Properties.temperature", 298.0);
      props_file = d.str("Properties.Matl_Props.floc", "props.txt"); nprops = (int)d.num("Properties.Matl_Props.num_props", 1);
      ori_file = d.str("Properties.Grain.ori_floc", "ori.txt"); grain_file = d.str("Properties.Grain.grain_floc", "grain_map.txt");
      num_grains = (int)d.num("Properties.Grain.num_grains", 0); ori_type = lower(d.str("Properties.Grain.ori_type", "euler"));
      if (ori_type != "quat" && ori_type != "quaternion") throw std::runtime_error("Only quaternion orientations (ori_type = \"quat\") are supported by this driver");
      // BCs (reference src/option_parser.cpp get_bcs): either flat arrays or arrays-of-arrays keyed by update_steps
      const bool changing = d.boolean("BCs.changing_ess_bcs", false);
      const TomlValue* ids = d.get("BCs.essential_ids"); const TomlValue* comps = d.get("BCs.essential_comps"); const TomlValue* vals = d.get("BCs.essential_vals");
      const TomlValue* vgr = d.get("BCs.essential_vel_grad");
      if (!ids || !comps) throw std::runtime_error("BCs.essential_ids / essential_comps are required");
      if (const TomlValue* vo = d.get("BCs.vgrad_origin")) {
         if (!vo->arr.empty()) { if (vo->arr.size() != 3) throw std::runtime_error("BCs.vgrad_origin when provided must contain 3 components."); vgrad_origin_flag = true; for (int k = 0; k < 3; k++) vgrad_origin[k] = vo->arr[k].num; }
      }
      auto flat = [](const TomlValue& a) { std::vector<double> o; for (auto& e : a.arr) o.push_back(e.num); return o; };
      auto fill_vgrad = [](const TomlValue* m, BCEntry& e) {   // [[a,b,c],[d,e,f],[g,h,i]] flattened row by row
         if (!m) return; int k = 0;
         for (auto& row : m->arr) for (auto& x : row.arr) { if (k < 9) e.vgrad[k] = x.num; k++; }
         if (k != 0 && k != 9) throw std::runtime_error("BCs.essential_vel_grad must be a 3 x 3 array");
      };
      if (changing) {
         const TomlValue* us = d.get("BCs.update_steps"); if (!us) throw std::runtime_error("BCs.update_steps was not provided any values.");
         bool has1 = false; for (auto& u : us->arr) has1 = has1 || (int)u.num == 1;
         if (!has1) throw std::runtime_error("BCs.update_steps must contain 1 in the array");
         if (ids->arr.size() != us->arr.size()) throw std::runtime_error("BCs.essential_ids did not contain the same number of arrays as number of update steps");
         if (comps->arr.size() != us->arr.size()) throw std::runtime_error("BCs.essential_comps did not contain the same number of arrays as number of update steps");
         for (size_t b = 0; b < us->arr.size(); b++) {
            BCEntry e; e.step = (int)us->arr[b].num;
            for (double v : flat(ids->arr[b])) e.ids.push_back((int)v);
            for (double v : flat(comps->arr[b])) e.comps.push_back((int)v);
            if (vals && b < vals->arr.size()) e.vals = flat(vals->arr[b]);
            if (vgr && b < vgr->arr.size()) fill_vgrad(&vgr->arr[b], e);
            bcs.push_back(e);
         }
      } else {
         BCEntry e; e.step = 1;
         for (double v : flat(*ids)) e.ids.push_back((int)v);
         for (double v : flat(*comps)) e.comps.push_back((int)v);
         if (vals) e.vals = flat(*vals);
         fill_vgrad(vgr, e); bcs.push_back(e);
      }
      for (auto& e : bcs) {
         if (e.comps.size() != e.ids.size()) throw std::runtime_error("BCs: essential_comps must hold one entry per essential id");
         bool need_vel = false, need_vg = false; for (int c : e.comps) { if (c > 0) need_vel = true; if (c < 0) need_vg = true; }
         if (e.vals.empty() && need_vel) throw std::runtime_error("BCs.essential_vals was not provided any values  but a boundary requires this.");
         if (!vgr && need_vg) throw std::runtime_error("BCs.essential_vel_grad was not provided any values but a boundary requires this.");
         if (e.vals.empty()) e.vals.assign(3 * e.ids.size(), 0.0);
         if (e.vals.size() != 3 * e.ids.size()) throw std::runtime_error("BCs: essential_vals must hold 3 values per essential id");
      }
      if (lower(d.str("Model.mech_type", "")) != "exacmech") throw std::runtime_error("Only mech_type = \"exacmech\" is supported (UMAT is CPU-only in the reference)");
      const std::string xt = lower(d.str("Model.ExaCMech.xtal_type", "")), st = lower(d.str("Model.ExaCMech.slip_type", ""));
      if (xt == "fcc") xtal = XtalType::FCC; else if (xt == "bcc") xtal = XtalType::BCC; else throw std::runtime_error("Unknown xtal_type: " + xt);
      if (st == "powervoce") slip = SlipType::POWERVOCE; else if (st == "powervocenl") slip = SlipType::POWERVOCENL; else if (st == "mtsdd") slip = SlipType::MTSDD; else throw std::runtime_error("Unknown slip_type: " + st);
      // time: Custom > Auto > Fixed (reference src/mechanics_driver.cpp:842-851)
      if (d.has_table("Time.Custom")) {
         dt_cust = true; nsteps = (int)d.num("Time.Custom.nsteps", 1);
         cust_dt = load_numbers(resolve(d.str("Time.Custom.floc", "custom_dt.txt")));
         if ((int)cust_dt.size() < nsteps) throw std::runtime_error("Custom dt file has fewer entries than nsteps");
      } else if (d.has_table("Time.Auto")) {
         dt_auto = true; dt = d.num("Time.Auto.dt_start", 1.0); dt_min = d.num("Time.Auto.dt_min", 1.0); dt_scale = d.num("Time.Auto.dt_scale", 0.25); t_final = d.num("Time.Auto.t_final", 1.0);
         auto_dt_fname = d.str("Time.Auto.auto_dt_file", "auto_dt_out.txt");
         if (changing) throw std::runtime_error("Automatic time stepping is currently not compatible with changing boundary conditions");   // src/option_parser.cpp:509-511
         if (dt_scale < 0.0 || dt_scale > 1.0) throw std::runtime_error("dt_scale for auto time stepping needs to be between 0 and 1.");
         nsteps = (int)std::ceil(t_final / dt_min);   // reference src/mechanics_driver.cpp:212
      } else { dt = d.num("Time.Fixed.dt", 1.0); t_final = d.num("Time.Fixed.t_final", 1.0); nsteps = (int)std::ceil(t_final / dt - 1e-9); }
      avg_stress_fname = d.str("Visualizations.avg_stress_fname", "avg_stress.txt");
      additional_avgs = d.boolean("Visualizations.additional_avgs", false);
      avg_def_grad_fname = d.str("Visualizations.avg_def_grad_fname", "avg_def_grad.txt");
      avg_pl_work_fname = d.str("Visualizations.avg_pl_work_fname", "avg_pl_work.txt");
      avg_dp_tensor_fname = d.str("Visualizations.avg_dp_tensor_fname", "avg_dp_tensor.txt");
      const std::string as = lower(d.str("Solvers.assembly", "FULL"));
      if (as == "pa") assembly = Assembly::PA; else if (as == "ea" || as == "full") assembly = Assembly::EA; else throw std::runtime_error("Unknown assembly: " + as);
      integ_model = d.str("Solvers.integ_model", "FULL");
      if (lower(integ_model) != "full" && lower(integ_model) != "bbar") throw std::runtime_error("Solvers.integ_model was not provided a valid type.");
      if (lower(integ_model) == "bbar" && assembly == Assembly::PA) throw std::runtime_error("integ_model = \"BBAR\" has no partial-assembly gradient (use EA or FULL), as in the reference");
      newton_iter = (int)d.num("Solvers.NR.iter", 25); newton_rel = d.num("Solvers.NR.rel_tol", 1e-5); newton_abs = d.num("Solvers.NR.abs_tol", 1e-10);
      { const std::string nl = lower(d.str("Solvers.NR.nl_solver", "NR"));   // reference src/option_parser.cpp:616-627 aborts on anything else
        if (nl == "nr") nl_solver = NLSolver::NR; else if (nl == "nrls") nl_solver = NLSolver::NRLS;
        else throw std::runtime_error("Solvers.NR.nl_solver was not provided a valid type."); }
      krylov_iter = (int)d.num("Solvers.Krylov.iter", 200); krylov_rel = d.num("Solvers.Krylov.rel_tol", 1e-10); krylov_abs = d.num("Solvers.Krylov.abs_tol", 1e-30);
      // reference src/option_parser.cpp:647-662: GMRES (its default), PCG or MINRES, anything else aborts.  Only PCG is built here (the
      // north-star path; the ExaCMech tangents the driver sees are symmetric to round-off, exa_grad_tangent_defect).  A file that asks
      // for - or defaults to - one of the other two is refused instead of silently running CG.
      krylov_solver = lower(d.str("Solvers.Krylov.solver", "GMRES"));
      if (krylov_solver == "gmres" || krylov_solver == "minres")
         throw std::runtime_error("Solvers.Krylov.solver = \"" + krylov_solver + "\" (the reference's default is GMRES) is not built in this driver: set Solvers.Krylov.solver = \"PCG\"");
      if (krylov_solver != "pcg") throw std::runtime_error("Solvers.Krylov.solver was not provided a valid type.");
      ref_ser = (int)d.num("Mesh.ref_ser", 0); order = (int)d.num("Mesh.p_refinement", 1);   // tests write "prefinement": ignored like the reference (src/option_parser.cpp:677)
      mesh_type = lower(d.str("Mesh.type", "other"));
      mesh_file = d.str("Mesh.floc", "");
      if (mesh_type == "other" || mesh_type == "cubit") {   // file mesh (reference src/mechanics_driver.cpp:239-241); MFEM mesh v1.0 hexahedra only
         if (mesh_file.empty()) throw std::runtime_error("Mesh.floc is required for Mesh.type = \"other\"");
         if (ref_ser != 0) throw std::runtime_error("Mesh.ref_ser > 0 is only built for auto-generated meshes");
         if (order < 1 || order > 6) throw std::runtime_error("File meshes run at p_refinement = 1 ... 6");
      } else if (mesh_type == "auto") {
         const TomlValue* nc = d.get("Mesh.Auto.ncuts"); const TomlValue* ln = d.get("Mesh.Auto.length");
         if (!nc || !ln || nc->arr.size() != 3 || ln->arr.size() != 3) throw std::runtime_error("Must input mesh geometry/discretization for hex_mesh_gen");
         for (int i = 0; i < 3; i++) { ncuts[i] = (int)nc->arr[i].num; length[i] = ln->arr[i].num; }
      } else throw std::runtime_error("Mesh.type must be \"auto\", \"other\" or \"cubit\"");
      if (order < 1 || order > 6) throw std::runtime_error("p_refinement must be between 1 and 6");
   }
};

}  // namespace exa_host
