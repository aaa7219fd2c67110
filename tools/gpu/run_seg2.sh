python tools/gpu/engine_diff.py 0 1 CAUCHY > gpurun_out/ediff_01.log 2>&1
python tools/gpu/engine_diff.py 2 0 TRUNCATED_CAUCHY > gpurun_out/ediff_2.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for e in 1 2; do
rm -rf /tmp/prof_kt
MDRP_LM_ENGINE=$e rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --workload varying_n5000_i10k > $R/gpurun_out/seg_prof_e$e.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" > $R/gpurun_out/seg_kstats_e$e.txt
python3 $R/tools/rocpd_timeline.py "$DB" 400 > $R/gpurun_out/seg_timeline_e$e.txt
done
