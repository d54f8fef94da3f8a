// ani_model.cpp — flat model-file parser (layout: lammps-ani_amd/model_file.py).
#include "ani_model.h"

#include <cstdio>
#include <cstring>

namespace ani {

namespace {
struct Reader {
  FILE* f;
  bool ok = true;
  template <typename T>
  void get(T* p, size_t n) {
    if (ok && fread(p, sizeof(T), n, f) != n) ok = false;
  }
};
}  // namespace

std::string load_model(const std::string& path, int use_num_models, HostModel& m) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "cannot open model file '" + path + "'";
  Reader r{f};
  char magic[8];
  uint32_t h[6];
  double c[6];
  r.get(magic, 8);
  r.get(h, 6);
  r.get(c, 6);
  if (!r.ok || memcmp(magic, "ANIHIP01", 8) != 0) {
    fclose(f);
    return "'" + path + "' is not an ANIHIP01 model file (TorchScript archives of the reference are not readable here; "
           "convert with lammps-ani_amd/model_file.py)";
  }
  m.S = h[0]; m.M_file = h[1]; m.L = h[2]; m.nR = h[3]; m.nA = h[4]; m.nZ = h[5];
  if (m.S < 1 || m.S > kMaxSpecies || m.L < 2 || m.L > kMaxLayers || m.nR > kMaxShfR || m.nA > kMaxShfA ||
      m.nZ > kMaxShfZ || m.M_file < 1) {
    fclose(f);
    return "model file header out of supported range";
  }
  m.Rcr = c[0]; m.Rca = c[1]; m.EtaR = c[2]; m.EtaA = c[3]; m.Zeta = c[4]; m.alpha = c[5];
  m.ShfR.resize(m.nR); m.ShfA.resize(m.nA); m.ShfZ.resize(m.nZ);
  r.get(m.ShfR.data(), m.nR);
  r.get(m.ShfA.data(), m.nA);
  r.get(m.ShfZ.data(), m.nZ);
  m.symbols.resize(m.S); m.sae.resize(m.S); m.dims.assign(m.S, std::vector<int>(m.L + 1));
  for (int s = 0; s < m.S; s++) {
    char sym[5] = {0};
    uint32_t d[kMaxLayers + 1];
    r.get(sym, 4);
    r.get(&m.sae[s], 1);
    r.get(d, m.L + 1);
    m.symbols[s] = sym;
    for (int l = 0; l <= m.L; l++) m.dims[s][l] = (int)d[l];
  }
  if (!r.ok) { fclose(f); return "model file truncated (header)"; }
  m.radial_len = m.S * m.nR;
  m.angular_len = m.S * (m.S + 1) / 2 * m.nA * m.nZ;
  m.aev_len = m.radial_len + m.angular_len;
  for (int s = 0; s < m.S; s++)
    if (m.dims[s][0] != m.aev_len || m.dims[s][m.L] != 1) { fclose(f); return "network input/output width inconsistent with AEV length"; }
  if (use_num_models < 0) use_num_models = m.M_file;
  if (use_num_models < 1 || use_num_models > m.M_file) {
    fclose(f);
    return "use_num_models=" + std::to_string(use_num_models) + " outside 1.." + std::to_string(m.M_file);
  }
  m.M = use_num_models;
  m.W.assign(m.M, {});
  m.b.assign(m.M, {});
  std::vector<float> skip;
  for (int a = 0; a < m.M_file; a++) {
    if (a < m.M) { m.W[a].assign(m.S, {}); m.b[a].assign(m.S, {}); }
    for (int s = 0; s < m.S; s++) {
      if (a < m.M) { m.W[a][s].resize(m.L); m.b[a][s].resize(m.L); }
      for (int l = 0; l < m.L; l++) {
        size_t o = m.dims[s][l + 1], i = m.dims[s][l];
        if (a < m.M) {
          m.W[a][s][l].resize(o * i);
          m.b[a][s][l].resize(o);
          r.get(m.W[a][s][l].data(), o * i);
          r.get(m.b[a][s][l].data(), o);
        } else {
          skip.resize(o * i + o);
          r.get(skip.data(), o * i + o);
        }
      }
    }
  }
  if (!r.ok) { fclose(f); return "model file truncated (weights)"; }
  char tag[8];
  const size_t ntag = fread(tag, 1, 8, f);
  if (ntag != 0 && ntag != 8) { fclose(f); return "model file has trailing bytes"; }
  if (ntag == 8) {   // optional trailing block: pairwise repulsion tables
    if (memcmp(tag, "REPULXTB", 8) != 0) { fclose(f); return "model file has trailing bytes"; }
    m.rep_tables.resize((size_t)3 * m.S * m.S);
    r.get(&m.rep_cut, 1);
    r.get(m.rep_tables.data(), m.rep_tables.size());
    char extra;
    if (!r.ok || fread(&extra, 1, 1, f) == 1) { fclose(f); return "model file: malformed repulsion block"; }
    m.has_rep = true;
  }
  fclose(f);
  return "";
}

}  // namespace ani
