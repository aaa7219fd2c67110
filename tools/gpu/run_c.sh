for c in "128" "16,112" "32,96" "32,160" "48" "192"; do
  echo "== MDRP_CHUNKS=$c" >> gpurun_out/c_chunks.log
  MDRP_CHUNKS=$c python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 0 --inflight 1 --host-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/c_chunks.log
done
python -m pytest tests/test_gpu_boundary.py -q -m gpu -k "local_shard" 2>&1 | grep -E "passed|failed|Error" >> gpurun_out/c_chunks.log
