#!/bin/bash
# Instruction issue of the message kernel's launches (one cfg3 calibrate, `tools/level_times.py run`): two --pmc passes
# (instruction counts; busy cycles), reduced to per-launch VALU / SALU / LDS instructions per wavefront and VALU-busy share.
# usage (through gpurun, repo root): bash tools/pmc_issue.sh <tag> [VAR=value ...]
tag=${1:-rXX}; shift
for kv in "$@"; do export "$kv"; done
mode=${LT_MODE:-run}
out=$PWD/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/pi_a /tmp/pi_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pi_a -- python3 tools/level_times.py $mode > $out/${tag}_pmci_a.txt 2>&1 || { tail -5 $out/${tag}_pmci_a.txt; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pi_b -- python3 tools/level_times.py $mode > $out/${tag}_pmci_b.txt 2>&1 || { tail -5 $out/${tag}_pmci_b.txt; exit 1; }
A=$(find /tmp/pi_a -name "*counter_collection.csv" | head -1); B=$(find /tmp/pi_b -name "*counter_collection.csv" | head -1)
python3 - $A $B > $out/${tag}_pmc_issue.txt <<'PY'
import csv, sys, collections
def load(p):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return d
A, B = load(sys.argv[1]), load(sys.argv[2])
ka = [k for k in A if "bp_" in A[k]["kernel"]]
kb = [k for k in B if "bp_" in B[k]["kernel"]]
n = 0
for x, y in zip(ka, kb):
    a, b = A[x], B[y]
    w = max(a.get("SQ_WAVES", 0.0), 1.0)
    name = a["kernel"].split("(")[0].replace("void pgbp::", "")[:40]
    print(f"{n:3d} {name:40s} wgs {a['grid'] // a['wg']:6d} waves {w:8.0f} valu/wave {a.get('SQ_INSTS_VALU', 0) / w:7.0f} "
          f"salu/wave {a.get('SQ_INSTS_SALU', 0) / w:7.0f} lds/wave {a.get('SQ_INSTS_LDS', 0) / w:6.0f} | "
          f"active_valu {b.get('SQ_ACTIVE_INST_VALU', 0):12.0f} busy {b.get('SQ_BUSY_CYCLES', 0):12.0f} "
          f"wave_cycles {b.get('SQ_WAVE_CYCLES', 0):12.0f} gui {b.get('GRBM_GUI_ACTIVE', 0):10.0f}")
    n += 1
    if n >= 80:
        break
PY
rm -rf /tmp/pi_a /tmp/pi_b
head -45 $out/${tag}_pmc_issue.txt
