# round-5 measurement suite: run ON THE GPU BOX through gpurun from the repo root:  bash tools/gpu/run_r05.sh [workloads...]
# (every step under its own `timeout`: a profile pass of this round once waited 56 minutes for forked children under rocprofv3 --pmc)
set -x
mkdir -p gpurun_out
WL=${@:-"calib_p3p_n2000_i10k shared_n2000_i10k varying_n5000_i10k calib_shift_n2000_i10k calib_p3p_n2000_i10k_clean relpose_5pt_n2000_i10k fundamental_7pt_n2000_i10k"}
for w in $WL; do
  timeout 600 bash tools/profile_round.sh r05 $w > gpurun_out/pr_r05_$w.log 2>&1
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default_bench_line.json 2> gpurun_out/r05_default.err
timeout 300 python bench.py --batch 12500 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --extra-configs 0 > gpurun_out/r05_c5_bench_12500_pairs_1gpu.json 2> gpurun_out/r05_c5.err
timeout 300 python bench.py --total-pairs 12500 --gpus 1 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 > gpurun_out/r05_c5_total_pairs_1gpu.json 2>> gpurun_out/r05_c5.err
timeout 300 python bench.py --workload shared_6pt_n2000_i10k --batch 256 --steps 2 --warmup 1 --inflight 1 --host-steps 0 --cpu-pairs 8 > gpurun_out/r05_shared_6pt_n2000_i10k_bench.json 2> gpurun_out/r05_6pt.err
timeout 600 python tests/tools/stress_parity.py 96 > gpurun_out/r05_stress_parity.txt 2>&1
MDRP_STRESS_KINDS=3,5 timeout 900 python tests/tools/stress_parity_classic.py 384 > gpurun_out/r05_stress_parity_classic.txt 2>&1
MDRP_STRESS_KINDS=4 timeout 600 python tests/tools/stress_parity_classic.py 32 > gpurun_out/r05_stress_parity_sixpt.txt 2>&1
MDRP_FUSE_TAIL=0 timeout 300 python tools/lo_trace.py > gpurun_out/r05_lo_trace.txt 2>&1
MDRP_FUSE_TAIL=0 timeout 300 python tools/lo_trace.py varying_n5000_i10k > gpurun_out/r05_lo_trace_varying.txt 2>&1
(MDRP_PIPELINE_MIN=1000000 timeout 120 python tools/entry_rate.py 2048 3; MDRP_PIPELINE_MIN=1000000 timeout 120 python tools/entry_rate.py 4096 3; timeout 200 python tools/entry_rate.py 8192 3; timeout 300 python tools/entry_rate.py 16384 2) 2>&1 | grep "^B " > gpurun_out/r05_python_entry_points.txt
timeout 600 python tests/tools/stress_options.py 512 777 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_stress_options.txt
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|^FAILED|pairs identical|pairs/s|REFERENCE" > gpurun_out/r05_pytest_gpu.txt
mkdir -p gpurun_out/profiles_r05; cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null
ls gpurun_out/profiles_r05 | wc -l
