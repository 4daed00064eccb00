#!/bin/bash
# A/B of environment settings on one box: bash tools/sweep_env.sh <tag> "<bench args>" "VAR=a,VAR2=b" "VAR=c" ...
# (each setting: comma-separated VAR=value pairs, "-" for none); prints ms_per_step of each
tag=$1; args=$2; shift 2
out=$PWD/gpurun_out
mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i+1))
  (
    if [ "$setting" != "-" ]; then IFS=','; for kv in $setting; do export "$kv"; done; unset IFS; fi
    python3 bench.py $args > $out/${tag}_$i.json 2> $out/${tag}_$i.err
  ) || { echo "FAILED $setting"; tail -5 $out/${tag}_$i.err; exit 1; }
  python3 - $out/${tag}_$i.json "$setting" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
extra = {k: (round(v["ms_per_step"], 4) if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in d.items() if isinstance(v, dict) and "ms_per_step" in v}
print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 4), extra, flush=True)
PY
done
