# bench.py under a list of MDRP_CHUNKS schedules:  bash tools/dev/sweep_chunks.sh WORKLOAD "128 128,1024 256" [extra bench args]
cd $GRAFT_REPO_ROOT
W=$1; shift; L=$1; shift
for C in $L; do
  echo -n "$W MDRP_CHUNKS=$C  "
  MDRP_CHUNKS=$C python3 bench.py --workload $W --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
done
