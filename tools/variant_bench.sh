#!/bin/bash
# Build library variants with different -D knobs and time each on the GPU box (run through gpurun).
# usage: tools/variant_bench.sh "NAME1:-DFOO=1" "NAME2:-DFOO=2 -DBAR" ...   (env: BATCH, STEPS, WORKLOAD)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/variants
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=fast -no-hip-rt $flags mdrp_amd/csrc/mdrp_capi.hip -o mdrp_amd/libmdrp_hip.so 2> gpurun_out/variants/$name.build.log || { echo "$name: BUILD FAILED"; tail -5 gpurun_out/variants/$name.build.log; continue; }
  timeout 600 python bench.py --batch ${BATCH:-512} --steps ${STEPS:-3} --warmup 1 --cpu-pairs 0 --workload ${WORKLOAD:-calib_p3p_n2000_i10k} > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err || { echo "$name: RUN FAILED"; tail -3 gpurun_out/variants/$name.err; continue; }
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
l = [x for x in open(f"gpurun_out/variants/{name}.json") if x.startswith("{")][-1]
d = json.loads(l)
print(f"{name:28s} pairs/s {d['value']:10.1f}  ms/step {d['ms_per_step']:8.2f}  sweep avg ms {d['roofline']['avg_launch_ms']:8.3f}  share {d['roofline']['sweep_share_of_step']:.3f}  inl {d['quality']['mean_inlier_ratio']:.5f} Rerr {d['quality']['median_rotation_error_deg_first64']:.5f}")
PY
done
