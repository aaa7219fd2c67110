#!/bin/bash
# Build the reference shim (this container only; needs /root/reference). See oracle/refshim/refshim.cpp.
# The reference's PoseLib exists only as a binary inside a cp312 wheel: it is extracted to /tmp (never into
# the repo) and dlopen()ed by the shim.  Outputs go to oracle/_ref/ only.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
WHL=/root/reference/demo/poselib-2.0.5-cp312-cp312-linux_x86_64.whl
if [ ! -f "$WHL" ]; then echo "build_ref: reference wheel not present; skipping (GPU box)"; exit 0; fi
mkdir -p /tmp/mdrp_ref_whl "$HERE/_ref"
if [ ! -f /tmp/mdrp_ref_whl/poselib/_core.cpython-312-x86_64-linux-gnu.so ]; then
  python3 -m zipfile -e "$WHL" /tmp/mdrp_ref_whl
fi
g++ -O2 -std=c++17 -mavx -fPIC -shared "$HERE/refshim/refshim.cpp" -o "$HERE/_ref/librefshim.so" \
    -Wl,--no-as-needed -lpython3.10 -ldl
echo "built $HERE/_ref/librefshim.so"
