#!/bin/bash
# phase times of the default bench workload for every tools/abl/libani_<NAME>.so given (BASE = the shipped library)
# usage: tools/abl_run2.sh [--atoms N] NAME...   (env passthrough: ANI_AEV_WAVES_PER_CU)
ATOMS=100002
if [ "$1" = "--atoms" ]; then ATOMS=$2; shift 2; fi
for v in "$@"; do
  envs=""
  case $v in *@*) envs=${v#*@}; v=${v%%@*};; esac
  if [ "$v" = BASE ]; then unset ANI_HIP_LIB; else export ANI_HIP_LIB=$PWD/tools/abl/libani_$v.so; fi
  env $envs timeout -k 10 200 python bench.py --atoms $ATOMS --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $envs', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms'].items()}, flush=True)"
done
