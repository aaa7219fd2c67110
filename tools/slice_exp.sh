#!/bin/bash
# experiment: sliced LO with different slice lengths (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for its in 25 8 4; do
  python3 -c "
from mdrp_amd import build
build.build(force=True, defines=('MDRP_LO_SLICE_ITS=$its',), out='gpurun_out/libmdrp_slice$its.so')" > /dev/null 2>&1
  MDRP_LIB=$R/gpurun_out/libmdrp_slice$its.so MDRP_LO_SLICE=1 timeout 300 python3 bench.py --batch 1024 --steps 5 --warmup 1 --cpu-pairs 0 --host-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('slice its $its:', d['value'], d['ms_per_step'])"
done
