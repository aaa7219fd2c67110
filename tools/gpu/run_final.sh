python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" > gpurun_out/final_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -n 2 gpurun_out/final_smoke.log; tail -n 2 gpurun_out/final_tests.log
