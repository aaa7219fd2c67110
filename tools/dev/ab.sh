# A/B of builds in ONE box, alternating:  bash tools/dev/ab.sh "LIB_B LIB_C ..." "workloads" repeats     (experiment builds: mdrp_amd.build.build(defines=..., out=...))
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${3:-2}); do for W in $2; do
  for L in "" $1; do
  echo -n "$W  lib=${L:-default}  "
  MDRP_LIB=${L:+$GRAFT_REPO_ROOT/$L} python3 bench.py --workload $W --extra-configs 0 --c5-share 0 --latency 0 --cpu-pairs 0 --host-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items() if 'lo' in k or 'final' in k})"
  done
done; done
