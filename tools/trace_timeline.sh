#!/bin/bash
# Kernel timeline of one bench step under environment knobs (run ON the GPU box through gpurun).
# usage: tools/trace_timeline.sh NAME [VAR=val ...]   -> gpurun_out/trace/NAME_{timeline,stats}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 --c5-share 0 --latency 0 --batch ${BATCH:-1024} --workload ${WORKLOAD:-calib_p3p_n2000_i10k} > /dev/null 2>&1
DB=$(find /tmp/prof_$name -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" > $R/gpurun_out/trace/${name}_stats.txt
python3 $R/tools/rocpd_timeline.py "$DB" ${ROWS:-40} > $R/gpurun_out/trace/${name}_timeline.txt
echo "== $name ($*)"; cat $R/gpurun_out/trace/${name}_timeline.txt
