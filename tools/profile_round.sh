#!/bin/bash
# The round's measurement recipe for ONE workload, run ON THE GPU BOX through gpurun:  tools/profile_round.sh TAG [WORKLOAD] [BATCH]
# Writes gpurun_out/TAG_<workload>/ and copies the summaries the judge reads into profiles/:
#   profiles/TAG_<workload>_kernel_stats.txt   rocprofv3 --kernel-trace --stats (per-kernel calls / total / avg / min / max)
#   profiles/TAG_<workload>_timeline.txt       start / end of every launch of the last profiled step
#   profiles/TAG_pmc_<workload>.json           four --pmc passes merged (HBM bytes, VALU-issue and MFMA-busy fractions)
#   profiles/TAG_<workload>_bench.json         the bench line of the same command, un-profiled
TAG=${1:-round}; W=${2:-calib_p3p_n2000_i10k}; B=${3:-1024}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_$W
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 --c5-share 0 --latency 0 --workload $W --batch $B > $O/bench_prof.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" > $O/kernel_stats.txt
python3 $R/tools/rocpd_timeline.py "$DB" 40 > $O/timeline.txt
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  # counter collection serialises kernel dispatches: the fused tail (last LO launch and final refinements side by side) cannot
  # overlap then and would only show its bounded waits - the PMC passes profile the same kernels in the unfused order
  MDRP_FUSE_TAIL=0 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$i -o c -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --host-steps 0 --inflight 1 --extra-configs 0 --c5-share 0 --latency 0 --workload $W --batch $B > $O/pmc_$i.log 2>&1
  F=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then grep -E "Counter_Name|mdrp::" "$F" > $O/pmc_$i.csv; else echo "pass $i: no counters"; tail -3 $O/pmc_$i.log; fi
done
python3 $R/tools/pmc_json.py $W $B 2 $O/pmc_1.csv $O/pmc_2.csv $O/pmc_3.csv $O/pmc_4.csv > $O/pmc.json
cp $O/pmc.json $R/profiles/${TAG}_pmc_$W.json   # so that the bench line below cites this round's PMC passes
cd $R && python3 bench.py --workload $W --batch $B --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs ${CPU_PAIRS:-0} > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
cp $O/kernel_stats.txt $R/profiles/${TAG}_${W}_kernel_stats.txt; cp $O/timeline.txt $R/profiles/${TAG}_${W}_timeline.txt
tail -1 $O/bench.json > $R/profiles/${TAG}_${W}_bench.json
mkdir -p $R/gpurun_out/profiles_$TAG; cp $R/profiles/${TAG}_*${W}* $R/gpurun_out/profiles_$TAG/
head -14 $O/kernel_stats.txt
