#!/bin/bash
# A/B of library variants on the GPU box: tools/gpu/ab.sh WORKLOAD lib1.so lib2.so ...   (each twice, interleaved; "-" = the in-tree library)
W=$1; shift
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MDRP_LIB; else export MDRP_LIB=$PWD/$lib; fi
  python bench.py --workload $W --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 0 --host-steps 0 --inflight 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done; done
