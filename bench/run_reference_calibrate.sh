#!/bin/sh
# the real Julia calibrate!() as a baseline, only where the toolchain already exists.
if ! command -v julia >/dev/null 2>&1 || ! julia -e 'using PhyloGaussianBeliefProp' >/dev/null 2>&1; then
  echo '{"available": false, "reason": "no julia with PhyloGaussianBeliefProp on this machine (cannot be installed: no network)"}'
  exit 0
fi
here=$(dirname "$0")
python3 "$here/dump_workload.py" /tmp/pgbp_workload.bin "$@" && julia "$here/reference_calibrate.jl" /tmp/pgbp_workload.bin 5
