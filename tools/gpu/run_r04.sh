# round-4 measurement suite: run ON THE GPU BOX through gpurun from the repo root
set -x
mkdir -p gpurun_out
for w in calib_p3p_n2000_i10k shared_n2000_i10k varying_n5000_i10k calib_shift_n2000_i10k calib_p3p_n2000_i10k_clean relpose_5pt_n2000_i10k fundamental_7pt_n2000_i10k; do
  bash tools/profile_round.sh r04 $w > gpurun_out/pr_r04_$w.log 2>&1
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_default_bench_line.json 2> gpurun_out/r04_default.err
python bench.py --batch 12500 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --extra-configs 0 > gpurun_out/r04_c5_bench_12500_pairs_1gpu.json 2> gpurun_out/r04_c5.err
python bench.py --total-pairs 12500 --gpus 1 --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 > gpurun_out/r04_c5_total_pairs_1gpu.json 2>> gpurun_out/r04_c5.err
python bench.py --workload shared_6pt_n2000_i10k --batch 256 --steps 2 --warmup 1 --inflight 1 --host-steps 0 --cpu-pairs 8 > gpurun_out/r04_shared_6pt_n2000_i10k_bench.json 2> gpurun_out/r04_6pt.err
python tests/tools/stress_parity.py 96 > gpurun_out/r04_stress_parity.txt 2>&1
MDRP_STRESS_KINDS=3,5 python tests/tools/stress_parity_classic.py 384 > gpurun_out/r04_stress_parity_classic.txt 2>&1
MDRP_STRESS_KINDS=4 python tests/tools/stress_parity_classic.py 32 > gpurun_out/r04_stress_parity_sixpt.txt 2>&1
python tools/lo_trace.py > gpurun_out/r04_lo_trace.txt 2>&1
cp profiles/r04_* gpurun_out/profiles_r04/ 2>/dev/null
ls gpurun_out/profiles_r04 | wc -l
