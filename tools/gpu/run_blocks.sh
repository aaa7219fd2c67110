#!/bin/bash
# segment engine: blocks per workgroup of the dense sweeps, varying focal 1024 x 5000 (cost blocks, normal-equation blocks)
mkdir -p gpurun_out
out=gpurun_out/blocks.txt
: > $out
timeout 900 python -m pytest tests -q -m gpu -x -k "engines or schedule or headline" 2>&1 | grep -E "passed|failed|^FAILED|identical" >> $out
for v in "1 1" "2 2" "0 0" "3 3" "8 8" "5 2" "2 5" "5 3" "3 5"; do
  set -- $v
  echo "== cost_blocks $1 acc_blocks $2" >> $out
  MDRP_LME_COST_BLOCKS=$1 MDRP_LME_ACC_BLOCKS=$2 timeout 300 python bench.py --workload varying_n5000_i10k --steps 5 --warmup 2 --cpu-pairs 0 --host-steps 0 --inflight 0 --extra-configs 0 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])" >> $out
done
cat $out
