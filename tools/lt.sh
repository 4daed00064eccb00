#!/bin/bash
# per-launch times of one cfg3 calibrate under a given environment: bash tools/lt.sh <tag> [VAR=value ...]
tag=$1; shift
mode=${LT_MODE:-run}
out=$PWD/gpurun_out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/lt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$tag -- python3 tools/level_times.py $mode > $out/lt_${tag}_run.txt 2>&1 || { echo "run failed"; tail -5 $out/lt_${tag}_run.txt; exit 1; }
python3 tools/level_times.py parse /tmp/lt_$tag $out/lt_${tag}.json > $out/lt_${tag}.txt 2>&1
rm -rf /tmp/lt_$tag
python3 - $out/lt_${tag}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(d["summary"]))
for i, l in enumerate(d["launches"]):
    k = l["kernel"].replace("bp_", "")
    print(i, k[:34], l["workgroups"], l["us"])
PY
