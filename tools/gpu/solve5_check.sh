# The 5-point solver after a change: its GPU tests, a kernel-trace summary of the workload, the bench line, and the bit comparison of the inlined
# elimination against the three-kernel one (tools/ubench/solve5_split_bits, built beforehand).  Run through gpurun: bash tools/gpu/solve5_check.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q5
timeout 900 python -m pytest tests/test_gpu_classic.py tests/test_gpu_headline.py -q -k "not 12500 and not large_batches" 2>&1 | tail -3 > gpurun_out/q5/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pk -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 --c5-share 0 --latency 0 --workload relpose_5pt_n2000_i10k > /tmp/bp.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$DB" | head -16 > $GRAFT_REPO_ROOT/gpurun_out/q5/stats.txt
cd $GRAFT_REPO_ROOT && python3 bench.py --workload relpose_5pt_n2000_i10k --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 2>/dev/null | tail -1 | cut -c1-200 > gpurun_out/q5/bench.txt
timeout 120 tools/ubench/solve5_split_bits 262144 > gpurun_out/q5/bits.txt 2>&1; cat gpurun_out/q5/pytest.txt gpurun_out/q5/stats.txt gpurun_out/q5/bench.txt gpurun_out/q5/bits.txt
