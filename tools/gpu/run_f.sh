for v in "MDRP_FUSE_TAIL=0" "MDRP_FUSE_TAIL=1"; do
  echo "== shift $v" >> gpurun_out/f.log
  env $v python bench.py --workload calib_shift_n2000_i10k --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 0 --inflight 1 --host-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/f.log
done
