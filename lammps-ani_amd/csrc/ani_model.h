// ani_model.h — host-side model description parsed from the flat model file (lammps-ani_amd/model_file.py),
// and the POD parameter blocks handed to the HIP kernels.  Replaces what torch::jit::load + module attributes
// carried in the reference (src/ani_csrc/ani.cpp:46-90).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ani {

constexpr int kMaxSpecies = 16;
constexpr int kMaxLayers = 8;
constexpr int kMaxShfR = 32;
constexpr int kMaxShfA = 16;
constexpr int kMaxShfZ = 16;

struct HostModel {
  int S = 0, M_file = 0, M = 0, L = 0, nR = 0, nA = 0, nZ = 0;
  double Rcr = 0, Rca = 0, EtaR = 0, EtaA = 0, Zeta = 0, alpha = 0.1;
  std::vector<double> ShfR, ShfA, ShfZ;
  std::vector<std::string> symbols;
  std::vector<double> sae;              // [S] Hartree
  std::vector<std::vector<int>> dims;   // [S][L+1]
  // W[m][s][l] row-major [out][in]; b[m][s][l] [out]
  std::vector<std::vector<std::vector<std::vector<float>>>> W, b;
  int radial_len = 0, angular_len = 0, aev_len = 0;
  // optional pairwise repulsion block of the model file: [S*S] tables in atomic units, cutoff in Angstrom
  bool has_rep = false;
  double rep_cut = 0;
  std::vector<double> rep_tables;   // y_ab | sqrt_alpha_ab | k_rep_ab
};

// returns empty string on success, else the error text.  use_num_models < 0 = all (first-n semantics,
// models/lammps_ani.py:342).
std::string load_model(const std::string& path, int use_num_models, HostModel& out);

// Passed by value to the AEV kernels.
struct AevParams {
  int S, nR, nA, nZ, nAZ, radial_len, aev_len, aev_stride;
  int compat;  // 1: no radial screening ("pyaev"), 0: r <= Rcr ("cuaev")
  int full_cap;  // 1: size the radial LDS list for the longest neighbour list even when screened (ani_set_option)
  float Rcr, Rca, EtaR, EtaA, Zeta, pi_over_Rcr, pi_over_Rca;
  // equidistant shift grids (every ANI model): ShfR[k] = ShfR0 + k dShfR, ShfA[k] = ShfA0 + k dShfA.  The fast kernels use
  // these four numbers instead of the tables (24 scalar registers less in kernels that run out of them); equi = 0 sends
  // a model with irregular grids to the generic kernels
  float ShfR0, dShfR, ShfA0, dShfA;
  int equi;
  float ShfR[kMaxShfR], ShfA[kMaxShfA], cosZ[kMaxShfZ], sinZ[kMaxShfZ];
};

}  // namespace ani
