#!/bin/bash
# HBM traffic of the cfg5 network workload (join graph / Bethe): two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs)
# of `tools/level_times.py run-network <graph>` (8 calibrate iterations and nothing else on these kernels), reduced by
# tools/pmc_traffic.py.  usage (through gpurun, repo root): bash tools/pmc_network.sh <tag> [joingraph|bethe]
tag=${1:-rXX}; graph=${2:-joingraph}
out=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pn_f -- python3 tools/level_times.py run-network $graph > $out/${tag}_pmcnet_f.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pn_w -- python3 tools/level_times.py run-network $graph > $out/${tag}_pmcnet_w.txt 2>&1 || exit 1
F=$(find /tmp/pn_f -name "*counter_collection.csv" | head -1); W=$(find /tmp/pn_w -name "*counter_collection.csv" | head -1)
# launches of one iteration (2 spanning trees), counted from a kernel trace of the same program (tools/level_times.py parse:
# 75 on the join graph since bp_level_small4 took over the mixed levels, 79 before)
rocprofv3 --kernel-trace --output-format csv -d /tmp/pn_t -- python3 tools/level_times.py run-network $graph > $out/${tag}_pmcnet_t.txt 2>&1 || exit 1
python3 tools/level_times.py parse /tmp/pn_t $out/${tag}_cfg5_${graph}_level_times.json > $out/${tag}_cfg5_${graph}_level_times.txt 2>&1
per_iter=$(python3 -c "import json; print(json.load(open('$out/${tag}_cfg5_${graph}_level_times.json'))['summary']['launches'])")
python3 tools/pmc_traffic.py $F $W 0 $out/${tag}_pmc_traffic_network_$graph.json none bp_level_generic+bp_level_small4+bp_chunk_generic+bp_chunk_pair+bp_fast16 8 $((8 * per_iter)) > $out/${tag}_pmc_traffic_network_$graph.txt 2>&1
rm -rf /tmp/pn_f /tmp/pn_w /tmp/pn_t
tail -12 $out/${tag}_pmc_traffic_network_$graph.txt
