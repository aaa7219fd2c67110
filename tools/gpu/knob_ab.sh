#!/bin/bash
# A/B of environment knobs on the GPU box: tools/gpu/knob_ab.sh WORKLOAD "A=1" "MDRP_CHUNKS=16,112 MDRP_LO_MERGE=1" ...   (each twice, interleaved)
W=$1; shift
for rep in 1 2; do for e in "$@"; do
  env $e python bench.py --workload $W --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 0 --host-steps 0 --inflight 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$e', round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done; done
