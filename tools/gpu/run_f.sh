python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -q -m gpu -x -k "bench_batch or schedule_does or statistical or stress_grid" 2>&1 | grep -E "passed|failed|^FAILED" >> gpurun_out/f.log
for v in 1 2 3 4; do
  echo "== MDRP_SOLVE_PARTS=$v" >> gpurun_out/f.log
  MDRP_SOLVE_PARTS=$v python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 3 --inflight 1 --host-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})
for c in d.get('configs',[]): print('  ', c['workload'], round(c['value']), round(c['ms_per_step'],2))" >> gpurun_out/f.log
done
