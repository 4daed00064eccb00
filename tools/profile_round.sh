#!/bin/bash
# End-of-round profile collection on ONE MI355X (run from the repo root through gpurun); summaries land in gpurun_out/prof_<tag>/
# and are copied into profiles/ by hand.  usage: bash tools/profile_round.sh <tag>
set -o pipefail
tag=${1:-rXX}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
step() { echo "== $1" | tee -a $out/log.txt; }
keep_stats() {  # $1 = rocprof output dir, $2 = name
  f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/$2
}
cd $R
step "cfg3 kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-reading --ll-batch 1 > $out/bench_cfg3_under_profiler.json 2>$out/e1.txt || exit 1
keep_stats /tmp/p1 cfg3_kernel_stats.csv; rm -rf /tmp/p1
step "cfg3 level times"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python3 tools/level_times.py run > $out/lt_run.txt 2>&1 || exit 1
python3 tools/level_times.py parse /tmp/p2 $out/cfg3_level_times.json > $out/cfg3_level_times.txt 2>&1; rm -rf /tmp/p2
step "cfg3 pmc fetch"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -- python3 tools/level_times.py run > $out/pmc_f.txt 2>&1 || exit 1
step "cfg3 pmc write"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -- python3 tools/level_times.py run > $out/pmc_w.txt 2>&1 || exit 1
F=$(find /tmp/p3 -name "*counter_collection.csv" | head -1); W=$(find /tmp/p4 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W 0 $out/cfg3_pmc_traffic.json none bp_fast16+bp_loop16 8 > $out/cfg3_pmc_traffic.txt 2>&1; rm -rf /tmp/p3 /tmp/p4
step "cfg3 log-likelihood evaluation: pmc fetch / write (ll_eval.roofline.traffic)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3l -- python3 tools/level_times.py run-ll > $out/pmc_ll_f.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4l -- python3 tools/level_times.py run-ll > $out/pmc_ll_w.txt 2>&1 || exit 1
F=$(find /tmp/p3l -name "*counter_collection.csv" | head -1); W=$(find /tmp/p4l -name "*counter_collection.csv" | head -1)
# (the last 8 evaluations: 16 launches each -- fill, flag reset, 7 level + 6 loop launches of the postorder, root integrate)
python3 tools/pmc_traffic.py $F $W 0 $out/cfg3_ll_eval_pmc_traffic.json none bm_tree_fill_fast+reset_flags_kernel+bp_fast16+bp_loop16+integrate_kernel 8 128 > $out/cfg3_ll_eval_pmc_traffic.txt 2>&1; rm -rf /tmp/p3l /tmp/p4l
step "cfg3 instruction issue (pmc)"
bash tools/pmc_issue.sh ${tag}_cfg3 > /dev/null 2>&1; cp $PWD/gpurun_out/${tag}_cfg3_pmc_issue.txt $out/cfg3_pmc_issue.txt 2>/dev/null
step "cfg2 kernel stats + level times"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2b -- python3 tools/level_times.py run-bethe > $out/lt2_run.txt 2>&1 || exit 1
keep_stats /tmp/p2b cfg2_kernel_stats.csv
python3 tools/level_times.py parse /tmp/p2b $out/cfg2_level_times.json > $out/cfg2_level_times.txt 2>&1; rm -rf /tmp/p2b
step "cfg4 kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 bench.py --workload sites --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_cfg4_under_profiler.json 2>$out/e5.txt || exit 1
keep_stats /tmp/p5 cfg4_kernel_stats.csv; rm -rf /tmp/p5
step "cfg4 pmc fetch"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p6 -- python3 bench.py --workload sites --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc4_f.txt 2>&1 || exit 1
step "cfg4 pmc write"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p7 -- python3 bench.py --workload sites --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc4_w.txt 2>&1 || exit 1
F=$(find /tmp/p6 -name "*counter_collection.csv" | head -1); W=$(find /tmp/p7 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W 0 $out/cfg4_pmc_traffic.json none bp_level_uni1+bp_chunk_uni1 > $out/cfg4_pmc_traffic.txt 2>&1; rm -rf /tmp/p6 /tmp/p7
step "cfg5 kernel stats (join graph)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -- python3 bench.py --workload network --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_cfg5_joingraph_under_profiler.json 2>$out/e8.txt || exit 1
keep_stats /tmp/p8 cfg5_joingraph_kernel_stats.csv; rm -rf /tmp/p8
step "cfg5 kernel stats (Bethe)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p9 -- python3 bench.py --workload network --graph bethe --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_cfg5_bethe_under_profiler.json 2>$out/e9.txt || exit 1
keep_stats /tmp/p9 cfg5_bethe_kernel_stats.csv; rm -rf /tmp/p9
step "cfg5 pmc traffic"
bash tools/pmc_network.sh $tag joingraph > /dev/null 2>&1; cp $PWD/gpurun_out/${tag}_pmc_traffic_network_joingraph.json $out/ 2>/dev/null
bash tools/pmc_network.sh $tag bethe > /dev/null 2>&1; cp $PWD/gpurun_out/${tag}_pmc_traffic_network_bethe.json $out/ 2>/dev/null
step "cfg5 level times"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p9b -- python3 tools/level_times.py run-network joingraph > $out/lt5_run.txt 2>&1 || exit 1
python3 tools/level_times.py parse /tmp/p9b $out/cfg5_joingraph_level_times.json > $out/cfg5_joingraph_level_times.txt 2>&1; rm -rf /tmp/p9b
step "cfg5: the two wavefronts of a task of bp_chunk_pair (phase stamps; instrumented build build/libpgbp_pstamp.so)"
[ -f build/libpgbp_pstamp.so ] && PGBP_LIB=$PWD/build/libpgbp_pstamp.so timeout -k 10 300 python3 tools/stamp_pair.py joingraph > $out/cfg5_joingraph_pair_phase_stamps.txt 2>$out/e_stamp.txt
step "cfg4 at a rank's share of 8 GPUs (1 000 problems)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5b -- python3 bench.py --workload sites --sites 125 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_cfg4_125_sites_under_profiler.json 2>$out/e5b.txt || exit 1
keep_stats /tmp/p5b cfg4_125_sites_kernel_stats.csv; rm -rf /tmp/p5b
for s in 1000 500 250 125; do python3 bench.py --workload sites --sites $s --no-cpu-baseline 2>>$out/e5c.txt; done > $out/bench_cfg4_strong_scaling_projection.jsonl || exit 1
step "seam microbenchmark"
[ -x build/exp/grid_barrier ] && { timeout -k 10 120 build/exp/grid_barrier 0 200; timeout -k 10 120 build/exp/grid_barrier 128 200; } > $out/grid_barrier_microbench.txt 2>&1
step "Mueller clique tree: large beliefs, KL residuals, free energy"
python3 tools/time_muller.py 2 3 4 6 > $out/muller_cliquetree_times.jsonl 2>$out/e_muller.txt
step "plain bench lines"
python3 bench.py > $out/bench_default.json 2>$out/e10.txt || exit 1
python3 bench.py --workload sites > $out/bench_cfg4.json 2>$out/e11.txt || exit 1
python3 bench.py --workload network > $out/bench_cfg5_joingraph.json 2>$out/e12.txt || exit 1
python3 bench.py --workload network --graph bethe > $out/bench_cfg5_bethe.json 2>$out/e13.txt || exit 1
step "done"
