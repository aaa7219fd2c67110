python -m pytest tests/test_gpu_headline.py tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_evalio.py tests/test_gpu_boundary.py tests/test_gpu_classic.py -q -m gpu 2>&1 | tail -8 > gpurun_out/a_tests.log
python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --extra-configs 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
tail -3 gpurun_out/a_tests.log
