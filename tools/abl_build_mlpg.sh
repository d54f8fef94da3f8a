#!/bin/bash
# variants of the 16-row fused MLP kernel: extra -D flags on ani_kernels_mlpg.hip -> tools/abl/libani_<NAME>.so
# usage: tools/abl_build_mlpg.sh NAME "-DX -DY" [NAME2 "flags2" ...]
set -e
cd "$(dirname "$0")/../lammps-ani_amd/csrc"
OBJS="ani_hip.o ani_model.o ani_kernels_aev.o ani_kernels_mlp.o ani_kernels_mlpf.o ani_kernels_misc.o ani_kernels_f64.o ani_kernels_nbr.o ani_kernels_rep.o ani_kernels_md.o ani_comm.o"
mkdir -p ../../tools/abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I/opt/rocm/include $flags -c ani_kernels_mlpg.hip -o ../../tools/abl/mlpg_$name.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/abl/libani_$name.so $OBJS ../../tools/abl/mlpg_$name.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib -ldl && echo built $name ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.5; done
done
wait
