#!/bin/bash
# scratch: chunk width sweep on cfg5
for mt in 384 1024 2048 4096; do
  for g in joingraph bethe; do
    PGBP_CHUNK_MAX_TASKS=$mt timeout -k 10 200 python bench.py --workload network --graph $g --steps 50 --warmup 5 > gpurun_out/sw_${g}_${mt}.json 2>/dev/null || exit 1
    python -c "
import json,sys
d=json.loads(open('gpurun_out/sw_${g}_${mt}.json').read().strip().splitlines()[-1]); print('$g', $mt, round(d['ms_per_step'],4))"
  done
done
