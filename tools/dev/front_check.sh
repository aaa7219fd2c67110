# after a change of the schedule / front: bench lines of the affected workloads (twice), single-pair latency, the dynamic-stopping rates, the GPU suite.
# Run through gpurun:  mkdir -p gpurun_out/q5; bash tools/dev/q6.sh > gpurun_out/q5/q6.txt 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q5
bash tools/dev/bench_set.sh "calib_p3p_n2000_i10k shared_n2000_i10k calib_shift_n2000_i10k varying_n5000_i10k calib_p3p_n2000_i10k_clean relpose_5pt_n2000_i10k calib_p3p_n2000_i10k_o75 calib_p3p_n2000_i10k_o85" 2
python tools/latency_trace.py 2000 10000 20; python tools/latency_trace.py 1000 1000 20
python tools/dynamic_bench.py 2>&1 | tail -4
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED"
