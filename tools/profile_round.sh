#!/bin/bash
# The round's measurement recipe, run ON THE GPU BOX through gpurun:  tools/profile_round.sh TAG
# Writes gpurun_out/TAG/: pytest log, smoke log, kernel-trace stats (+ timeline), the two PMC passes (HBM bytes), the
# default bench.py JSON line.  Copy the summaries into profiles/ afterwards (see DESIGN.md "Measurement").
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 > $O/bench_prof.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py "$DB" > $O/kernel_stats.txt
python3 $R/tools/rocpd_timeline.py "$DB" 24 > $O/timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -o c -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then grep -E "Counter_Name|mdrp::" "$F" > $O/pmc_$C.csv; fi
done
python3 $R/tools/pmc_hbm_json.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv calib_p3p_n2000_i10k 1024 2 > $O/pmc_hbm.json
cp $O/pmc_hbm.json $R/profiles/${TAG}_pmc_hbm.json   # so that the bench line below cites this round's PMC pass
cd $R && python3 bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-220
head -12 $O/kernel_stats.txt
