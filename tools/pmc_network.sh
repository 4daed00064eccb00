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
# launches of one iteration (2 spanning trees): 79 on the join graph, 129 on the Bethe graph (tools/level_times.py parse)
per_iter=79; [ "$graph" = "bethe" ] && per_iter=129
python3 tools/pmc_traffic.py $F $W 0 $out/${tag}_pmc_traffic_network_$graph.json none bp_level_generic+bp_chunk_generic+bp_fast16 8 $((8 * per_iter)) > $out/${tag}_pmc_traffic_network_$graph.txt 2>&1
rm -rf /tmp/pn_f /tmp/pn_w
tail -12 $out/${tag}_pmc_traffic_network_$graph.txt
