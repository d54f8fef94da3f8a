/* -*- c++ -*- ----------------------------------------------------------
   pair_style ani for LAMMPS on AMD MI355X — host adapter over the C ABI of libani_hip.so (include/ani_hip.h).

   Same input-script surface as the reference's PairANI (src/pair_ani.h:22-55, src/pair_ani.cpp):
       pair_style ani <cutoff> <model_file> <device> [num_models=-1] [cuaev|pyaev] [full|half] [single|double] [hostlist|devlist]
       pair_coeff * *
   (the eighth word is ours: `devlist` builds the full neighbour list on the GPU at re-neighbouring steps instead of
   taking LAMMPS' host list -- LAMMPS is then asked for an occasional list only, which it never builds)
   but no libtorch, no CUDA headers: the model runtime is the HIP library behind ani_hip.h.
   Compiled only when LAMMPS headers are available (LAMMPS_HEADER_DIR, like the reference's CMakeLists.txt:28-36);
   tests/mock_lammps provides a minimal stand-in for those headers so this file is exercised without LAMMPS.
------------------------------------------------------------------------- */

#ifdef PAIR_CLASS
// clang-format off
PairStyle(ani,PairANI);
// clang-format on
#else

#ifndef LMP_PAIR_ANI_HIP_H
#define LMP_PAIR_ANI_HIP_H

#include <cstdint>
#include <string>
#include <vector>

#include "ani_comm.h"
#include "ani_hip.h"
#include "pair.h"

namespace LAMMPS_NS {

class PairANI : public Pair {
 public:
  PairANI(class LAMMPS*);
  ~PairANI() override;
  void compute(int, int) override;
  void settings(int, char**) override;
  void coeff(int, char**) override;
  void init_style() override;
  double init_one(int, int) override;
  void* extract(const char*, int&) override;
  void write_restart(FILE*) override;
  void read_restart(FILE*) override;
  int pack_reverse_comm(int, int, double*) override;
  void unpack_reverse_comm(int, int*, double*) override;
  int pack_forward_comm(int, int*, double*, int, int*) override;
  void unpack_forward_comm(int, int, double*) override;

 protected:
  double cutoff = 0.0;
  ani_handle* ani = nullptr;
  std::string model_file, device_str;
  int use_num_models = -1;
  bool use_cuaev = true, use_fullnbr = true, use_single = true;
  bool use_devlist = false;  // not part of the restart record (kept byte-compatible with the reference): restarts come back as hostlist
  // `rcclcomm`: ghost forces go home on the device over RCCL (include/ani_comm.h) instead of through comm->reverse_comm(this)
  // on the host; not part of the restart record
  bool use_rccl = false;
  bool use_self_fold = false;   // one rank: a local communicator (no RCCL) sums the image rows on the device
  ani_comm* acomm = nullptr;
  std::vector<double> owner_info;   // [ntotal][2] {owning rank, index on that rank}, filled for ghosts by forward_comm(this)
  void build_rccl_maps(int nlocal, int nghost);
  bool profiling = false;  // LAMMPS_ANI_PROFILING: passed to the library as option "profiling" (stream sync before returning)

  // list epoch (rebuilt when neighbor->ago == 0), grown 1.5x like the reference (src/pair_ani.cpp:119-127)
  std::vector<int64_t> species;
  std::vector<int> flat_ilist, flat_numneigh, flat_jlist;
  std::vector<int64_t> atom_index12;
  int64_t npairs = 0;

  // ghost forces are reverse-communicated from here, not from atom->f (src/pair_ani.cpp:192-201)
  // a plain buffer the adapter owns, page-locked (ani_host_register) so that the per-step copy of the forces is a DMA transfer
  double* out_force = nullptr;
  size_t out_force_cap = 0;     // doubles
  bool out_force_pinned = false;
  bool direct_add = false;      // the library adds into atom->f itself (option out_force_accumulate)
  void grow_out_force(size_t doubles);
  // the atom->x block of the current neighbour-list epoch, page-locked at ago == 0 and released at the next ago == 0 before
  // anything else is touched (LAMMPS reallocates atom arrays only while re-neighbouring); LAMMPS_ANI_NO_PIN=1 turns both off
  const double* x_registered = nullptr;
  int x_registered_nmax = 0;
  bool pin_host = true;
  void release_pins();
  std::vector<double> out_eatom;

  void allocate();
  void create_model();
  int node_local_rank();
};

}  // namespace LAMMPS_NS

#endif
#endif
