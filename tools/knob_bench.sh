#!/bin/bash
# Time bench.py under different environment knobs on the GPU box (run through gpurun).
# usage: tools/knob_bench.sh "NAME1:VAR=val VAR2=val" "NAME2:..."   (env: BATCH, STEPS, WORKLOAD)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/knobs
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 600 python bench.py --batch ${BATCH:-1024} --steps ${STEPS:-5} --warmup 1 --cpu-pairs 0 --workload ${WORKLOAD:-calib_p3p_n2000_i10k} > gpurun_out/knobs/$name.json 2> gpurun_out/knobs/$name.err || { echo "$name: RUN FAILED"; tail -3 gpurun_out/knobs/$name.err; continue; }
  grep "super-chunk" gpurun_out/knobs/$name.err | tail -1
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
l = [x for x in open(f"gpurun_out/knobs/{name}.json") if x.startswith("{")][-1]
d = json.loads(l)
print(f"{name:28s} pairs/s {d['value']:10.1f}  ms/step {d['ms_per_step']:8.2f}  inl {d['quality']['mean_inlier_ratio']:.5f} Rerr {d['quality']['median_rotation_error_deg_first64']:.5f}")
PY
done
