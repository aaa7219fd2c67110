python tests/tools/stress_parity.py > gpurun_out/r04_stress_parity.txt 2>&1
MDRP_STRESS_KINDS=3,5 python tests/tools/stress_parity_classic.py 64 > gpurun_out/r04_stress_parity_classic.txt 2>&1
MDRP_STRESS_KINDS=4 python tests/tools/stress_parity_classic.py 16 > gpurun_out/r04_stress_parity_sixpt.txt 2>&1
tail -2 gpurun_out/r04_stress_parity.txt gpurun_out/r04_stress_parity_classic.txt gpurun_out/r04_stress_parity_sixpt.txt
