#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/exp.txt; : > $out
timeout 900 python -m pytest tests -q -m gpu -x -k "headline or schedule or engines" 2>&1 | grep -E "passed|failed|^FAILED|identical" >> $out
run() { wl=$1; shift; echo "== $wl $*" >> $out; env "$@" timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-pairs 0 --host-steps 0 --inflight 0 --extra-configs 0 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})" >> $out; }
run calib_p3p_n2000_i10k MDRP_LIB=$PWD/tools/gpu/libmdrp_exp.so
run calib_p3p_n2000_i10k A=1
run shared_n2000_i10k MDRP_LIB=$PWD/tools/gpu/libmdrp_exp.so
run shared_n2000_i10k A=1
run varying_n5000_i10k MDRP_LM_ENGINE=0 MDRP_LO_THREADS=256 MDRP_LIB=$PWD/tools/gpu/libmdrp_exp.so
run varying_n5000_i10k MDRP_LM_ENGINE=0 MDRP_LO_THREADS=256
run calib_shift_n2000_i10k A=1
run calib_p3p_n2000_i10k_clean A=1
echo "== trace headline" >> $out
timeout 300 python tools/lo_trace.py calib_p3p_n2000_i10k 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $out
cat $out
