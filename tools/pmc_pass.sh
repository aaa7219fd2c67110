#!/bin/bash
# One rocprofv3 --pmc pass over bench.py (run ON the GPU box through gpurun); per-kernel sums via tools/pmc_kernels.py.
# usage: tools/pmc_pass.sh NAME "COUNTER1 COUNTER2 ..." [VAR=val ...]      -> gpurun_out/pmc/NAME.txt (+ NAME.csv)
# (--pmc only: never combined with tracing options)
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; counters=$2; shift 2
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
# (counter collection serialises dispatches: the fused tail cannot overlap, profile the unfused order - see tools/profile_round.sh)
MDRP_FUSE_TAIL=${MDRP_FUSE_TAIL:-0} rocprofv3 --pmc $counters --output-format csv -d /tmp/pmc_$name -o c -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --batch ${BATCH:-1024} --workload ${WORKLOAD:-calib_p3p_n2000_i10k} > $R/gpurun_out/pmc/$name.log 2>&1
F=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
if [ -z "$F" ]; then echo "$name: no counter file"; tail -5 $R/gpurun_out/pmc/$name.log; exit 1; fi
grep -E "Counter_Name|mdrp::" "$F" > $R/gpurun_out/pmc/$name.csv
python3 $R/tools/pmc_kernels.py $R/gpurun_out/pmc/$name.csv > $R/gpurun_out/pmc/$name.txt
echo "== $name ($counters)"; cat $R/gpurun_out/pmc/$name.txt
