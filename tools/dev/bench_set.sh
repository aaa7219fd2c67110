# bench.py lines of a list of workloads:  bash tools/dev/bench_set.sh "w1 w2 ..." [repeats]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${2:-1}); do for W in $1; do
  echo -n "$W  "
  python3 bench.py --workload $W --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items() if 'solve' in k})"
done; done
