// ani_plugin.cpp — LAMMPS plugin entry point registering `pair_style ani` (replaces src/ani_plugin.cpp:21-40 of the
// reference; the Kokkos style `ani/kk` is not registered: the device-resident path of this build is
// ani_compute_full_device()).
#include "lammpsplugin.h"
#include "pair_ani.h"
#include "version.h"

using namespace LAMMPS_NS;

static Pair* ani_hip_creator(LAMMPS* lmp) { return new PairANI(lmp); }

extern "C" void lammpsplugin_init(void* lmp, void* handle, void* regfunc) {
  lammpsplugin_t plugin;
  lammpsplugin_regfunc register_plugin = (lammpsplugin_regfunc)regfunc;
  plugin.version = LAMMPS_VERSION;
  plugin.style = "pair";
  plugin.name = "ani";
  plugin.info = "ANI pair style, MI355X/HIP build v0.1";
  plugin.author = "lammps-ani_amd";
  plugin.creator.v1 = (lammpsplugin_factory1*)&ani_hip_creator;
  plugin.handle = handle;
  (*register_plugin)(&plugin, lmp);
}
