#!/bin/bash
# ablation timing of the AEV backward kernel: bench phase times with variant libraries (tools/abl/, built by hand)
for v in "" XNOSTORE XNOALOAD XNOBOTHMEM; do
  if [ -z "$v" ]; then unset ANI_HIP_LIB; else export ANI_HIP_LIB=$PWD/tools/abl/libani_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-BASE}', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms'].items()})"
done
