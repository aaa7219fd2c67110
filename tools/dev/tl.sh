# kernel timeline + stats of one workload:  bash tools/dev/tl.sh WORKLOAD  -> gpurun_out/q5/tl_WORKLOAD.txt
R=$GRAFT_REPO_ROOT; W=$1
mkdir -p $R/gpurun_out/q5
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk
rocprofv3 --kernel-trace --stats -d /tmp/pk -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 --c5-share 0 --latency 0 --workload $W > /tmp/bp.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
(python3 $R/tools/rocpd_timeline.py "$DB" 60 | cut -c1-110; python3 $R/tools/rocpd_summary.py "$DB" | head -14 | cut -c1-130) > $R/gpurun_out/q5/tl_$W.txt
cat $R/gpurun_out/q5/tl_$W.txt
