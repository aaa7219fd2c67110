set -x
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "engines_agree or schedule_does" 2>&1 | tail -15 > gpurun_out/seg_tests.log
for e in 1 2; do MDRP_LM_ENGINE=$e python bench.py --workload varying_n5000_i10k --steps 5 --warmup 1 --cpu-pairs 0 --inflight 1 --host-steps 0 > gpurun_out/seg_var_e$e.json 2> gpurun_out/seg_var_e$e.err; done
for e in 0 2; do MDRP_LM_ENGINE=$e python bench.py --workload calib_p3p_n2000_i10k --steps 10 --warmup 2 --cpu-pairs 0 --inflight 1 --host-steps 0 --extra-configs 0 > gpurun_out/seg_cal_e$e.json 2> gpurun_out/seg_cal_e$e.err; done
tail -5 gpurun_out/seg_tests.log
