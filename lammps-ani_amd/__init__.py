"""lammps-ani_amd — MI355X-native implementation of the `pair_style ani` hot path.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it with
``importlib.import_module("lammps-ani_amd")``; :func:`load` in the repository-root ``_pkg.py`` does this and
aliases the package as ``lammps_ani_amd``.

Sub-modules
  model_file   flat model-file format + seeded synthetic ANI-2x / ANI-1x shaped generators
  ani_hip      ctypes binding of the C ABI in include/ani_hip.h (libani_hip.so, HIP/gfx950)
  harness      LAMMPS stand-in: bricks, ghosts, neighbour lists, synthetic water boxes
  comm         ghost-force reverse / ghost-position forward exchange over torch.distributed (RCCL or gloo)
"""
