# round-6 measurement suite: run ON THE GPU BOX through gpurun from the repo root:  bash tools/gpu/run_r06.sh [workloads...]
# (every step under its own `timeout`)
set -x
mkdir -p gpurun_out
WL=${@:-"calib_p3p_n2000_i10k shared_n2000_i10k varying_n5000_i10k calib_shift_n2000_i10k calib_p3p_n2000_i10k_clean relpose_5pt_n2000_i10k fundamental_7pt_n2000_i10k"}
for w in $WL; do
  timeout 600 bash tools/profile_round.sh r06 $w > gpurun_out/pr_r06_$w.log 2>&1
done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_default_bench_line.json 2> gpurun_out/r06_default.err
timeout 300 python bench.py --batch 12500 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --extra-configs 0 --c5-share 0 --latency 0 > gpurun_out/r06_c5_bench_12500_pairs_1gpu.json 2> gpurun_out/r06_c5.err
timeout 300 python bench.py --total-pairs 12500 --gpus 1 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 > gpurun_out/r06_c5_total_pairs_1gpu.json 2>> gpurun_out/r06_c5.err
timeout 300 python bench.py --workload shared_6pt_n2000_i10k --batch 256 --steps 2 --warmup 1 --inflight 1 --host-steps 0 --cpu-pairs 8 > gpurun_out/r06_shared_6pt_n2000_i10k_bench.json 2> gpurun_out/r06_6pt.err
# beyond BASELINE.json: the headline shape at 75 % / 85 % outliers (the first chunk follows the previous call's inlier ratio), and the same under the fixed schedule of rounds 1-5
for w in calib_p3p_n2000_i10k_o75 calib_p3p_n2000_i10k_o85; do
  timeout 300 python bench.py --workload $w --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 > gpurun_out/r06_${w}_bench.json 2>> gpurun_out/r06_outl.err
  MDRP_CHUNKS=128 timeout 300 python bench.py --workload $w --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 > gpurun_out/r06_${w}_chunks128_bench.json 2>> gpurun_out/r06_outl.err
done
# the N > 1 launch on a one-GPU box: refused cleanly, rc 3
(python bench.py --gpus 2 --steps 1 --warmup 0; echo "rc=$?") > gpurun_out/r06_gpus2_on_one_gpu.txt 2>&1
# single-pair / small-batch latency through the drop-in entry points (the bench line's `latency` block has the same numbers)
(timeout 200 python tools/latency_trace.py 2000 10000 30; timeout 200 python tools/latency_trace.py 1000 1000 30; timeout 300 python tools/latency.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_latency.txt
# host-buffer path: per-call times and the copy / kernel timeline of one call
(timeout 200 python tools/host_path_trace.py 1024 10; timeout 200 python tools/host_path_trace.py 2048 6; timeout 200 python tools/host_path_trace.py 4096 4) 2>&1 | grep "pairs/s" > gpurun_out/r06_host_path.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/hp -o hp -- python3 $GRAFT_REPO_ROOT/tools/host_path_trace.py 1024 3 > /dev/null 2>&1; DB=$(find /tmp/hp -name "*.db" | head -1); python3 $GRAFT_REPO_ROOT/tools/host_path_timeline.py $DB 70) > gpurun_out/r06_host_path_timeline.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt && timeout 300 rocprofv3 --kernel-trace -d /tmp/lt -o lt -- python3 $GRAFT_REPO_ROOT/tools/latency_trace.py 2000 10000 4 > /dev/null 2>&1; DB=$(find /tmp/lt -name "*.db" | head -1); python3 $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 32) > gpurun_out/r06_latency_timeline.txt 2>&1
timeout 600 python tests/tools/stress_parity.py 96 > gpurun_out/r06_stress_parity.txt 2>&1
MDRP_STRESS_KINDS=3,5 timeout 900 python tests/tools/stress_parity_classic.py 384 > gpurun_out/r06_stress_parity_classic.txt 2>&1
MDRP_STRESS_KINDS=4 timeout 600 python tests/tools/stress_parity_classic.py 32 > gpurun_out/r06_stress_parity_sixpt.txt 2>&1
(timeout 120 python tools/entry_rate.py 2048 3; timeout 120 python tools/entry_rate.py 4096 3; timeout 200 python tools/entry_rate.py 8192 3; timeout 300 python tools/entry_rate.py 16384 2) 2>&1 | grep "^B " > gpurun_out/r06_python_entry_points.txt
timeout 600 python tests/tools/stress_options.py 512 777 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_stress_options.txt
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|^FAILED|pairs identical|pairs/s|REFERENCE|reference-NaN" > gpurun_out/r06_pytest_gpu.txt
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_* gpurun_out/profiles_r06/ 2>/dev/null
for f in r06_calib_p3p_n2000_i10k_o75_bench.json r06_calib_p3p_n2000_i10k_o75_chunks128_bench.json r06_calib_p3p_n2000_i10k_o85_bench.json r06_calib_p3p_n2000_i10k_o85_chunks128_bench.json r06_default_bench_line.json r06_c5_bench_12500_pairs_1gpu.json r06_c5_total_pairs_1gpu.json r06_shared_6pt_n2000_i10k_bench.json r06_gpus2_on_one_gpu.txt r06_latency.txt r06_latency_timeline.txt r06_host_path.txt r06_host_path_timeline.txt r06_stress_parity.txt r06_stress_parity_classic.txt r06_stress_parity_sixpt.txt r06_python_entry_points.txt r06_stress_options.txt r06_pytest_gpu.txt; do cp gpurun_out/$f gpurun_out/profiles_r06/ 2>/dev/null; done
ls gpurun_out/profiles_r06 | wc -l
