set -e
for rep in 1 2; do for v in old v6; do
  PGBP_LIB=build/libpgbp_$v.so timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sites-block --no-network-block --ll-batch 1 > gpurun_out/ab_$v.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "cfg3 ms", round(d["ms_per_step"],4), "cfg2 ms", round(d.get("cfg2_bethe_10k_tips_8_traits",{}).get("ms_per_step",0),4), "50k-clique ms", round(d.get("alt_reading_50k_cliques",{}).get("ms_per_step",0),4), "ll/s", round(d.get("ll_evals_per_s",0)))
PY
done; done
