#!/bin/bash
# SQ counters of the message kernel on cfg3 (three separate --pmc passes of `tools/level_times.py run`); CSVs under gpurun_out/sq/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/sq
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/sq$i -- python3 tools/level_times.py run > gpurun_out/sq/run$i.txt 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/sq/run$i.txt; continue; }
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" gpurun_out/sq/pass$i.json <<'PY'
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per dispatch id: kernel, grid, counters
d = collections.OrderedDict()
for r in rows:
    if "bp_fast16" not in r["Kernel_Name"]:
        continue
    k = r["Dispatch_Id"]
    e = d.setdefault(k, {"kernel": r["Kernel_Name"].split("(")[0][-28:], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
L = list(d.values())
per = len(L) // 8
json.dump(L[-per:], open(sys.argv[2], "w"), indent=0)
print(len(L), "dispatches,", per, "per calibrate")
PY
  rm -rf /tmp/sq$i
done
