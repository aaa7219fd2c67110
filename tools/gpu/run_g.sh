R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "8 1" "4 0"; do set -- $v
rm -rf /tmp/prof_kt
MDRP_LO_OVERLAP_WAVES=$1 MDRP_AUX2_PRIO=$2 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 > $R/gpurun_out/g_prof.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py "$DB" 45 > $R/gpurun_out/g_timeline_$1_$2.txt
done
