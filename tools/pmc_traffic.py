#!/usr/bin/env python3
"""
Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes) of `bench.py` into per-launch HBM
traffic of the message kernel.

gfx950 corrections (same guide): counters are in KiB; FETCH_SIZE reports exactly 1/2 of the bytes of a
16-B-per-lane coalesced read -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  The correction is
re-validated inside the same run on the reset-from-factors copy kernel, whose byte count is known exactly
(the cluster records, read once and written once: `copy_strided_kernel` in the plain layout,
`copy_records_kernel` -- only the part of each slot in use -- in the packed layout).

usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv copy_bytes out.json [copy_kernel [kernel [calibrates]]]
  copy_kernel: calibration kernel with a known byte count each way (default copy_strided_kernel; a name without "pgbp::" is
               looked up as given, e.g. __amd_rocclr_copyBuffer for the site-minor reset of the sites workload)
  kernel:      the message kernel to reduce (default bp_fast16: every launch mode of it; bp_level_uni1 for the sites workload;
               several kernels joined by '+': bp_level_generic+bp_chunk_generic+bp_fast16 for the network workload)
  calibrates:  how many calibrate!() iterations the profiled program ran and nothing else on that kernel (e.g.
               `tools/level_times.py run` = 8): adds launches_per_calibrate and hbm_bytes_per_calibrate
  last_n:      count only the last last_n launches of these kernels (dispatch order): the network workload's one-off
               regularisation walk sends its messages through the same kernels before the timed iterations
usage: ... [copy_kernel [kernel [calibrates [last_n]]]]
"""
import csv
import json
import sys


LAST = [0]   # > 0: of the kernels asked for, only the last LAST[0] launches in dispatch order count (what ran before
             # them on the same kernels -- the one-off regularisation walk of the network workload -- is not the path)
KEEP = [None]


def per_kernel(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            name = name.split("<")[0]
            k = int(r["Dispatch_Id"])
            e = rows.setdefault(k, [name, 0.0])
            e[1] += float(r["Counter_Value"])
    order = sorted(rows)
    if LAST[0] > 0 and KEEP[0]:
        mine = [k for k in order if rows[k][0] in KEEP[0]]
        drop = set(mine[:-LAST[0]]) if len(mine) > LAST[0] else set()
        order = [k for k in order if k not in drop]
    d = {}
    for k in order:
        d.setdefault(rows[k][0], []).append(rows[k][1])
    return d


def main():
    f_csv, w_csv, copy_bytes, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    ck = sys.argv[5] if len(sys.argv) > 5 else "copy_strided_kernel"
    if not ck.startswith("__"):
        ck = "pgbp::" + ck
    kname = sys.argv[6] if len(sys.argv) > 6 else "bp_fast16"
    ncal = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    LAST[0] = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    KEEP[0] = {"pgbp::" + x for x in kname.split("+")}
    F = per_kernel(f_csv, "FETCH_SIZE")
    W = per_kernel(w_csv, "WRITE_SIZE")
    have_cal = copy_bytes > 0 and ck in F and ck in W   # copy_bytes 0: calibrated elsewhere (tools/copy8_microbench.hip)
    cal_f = 2.0 * 1024 * sum(F[ck]) / len(F[ck]) if have_cal else 0.0
    cal_w = 1024 * sum(W[ck]) / len(W[ck]) if have_cal else 0.0
    # kernel: one name or several joined by '+' (the network workload runs bp_level_generic + bp_chunk_generic + bp_fast16)
    ks = ["pgbp::" + x for x in kname.split("+")]
    missing = [k for k in ks if k not in F or k not in W]
    ks = [k for k in ks if k not in missing]
    n = sum(len(F[k]) for k in ks)
    nw = sum(len(W[k]) for k in ks)
    fetch = 2.0 * 1024 * sum(sum(F[k]) for k in ks)
    write = 1024 * sum(sum(W[k]) for k in ks)
    res = {
        "kernel": kname, "launches": n,
        "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / nw,
        "hbm_bytes_per_launch": fetch / n + write / nw,
        "corrections": "KiB units; FETCH_SIZE x2 (16-B/lane reads on gfx950); WRITE_SIZE exact",
        "calibration_copy_kernel": ({"kernel": ck, "known_bytes_each_way": copy_bytes, "fetch_corrected": cal_f, "write": cal_w,
                                     "fetch_ratio": cal_f / copy_bytes, "write_ratio": cal_w / copy_bytes} if have_cal else
                                    "tools/copy8_microbench.hip under the same two passes: FETCH_SIZE x 2 = bytes read and "
                                    "WRITE_SIZE = bytes written, to three digits, for 8-B-per-lane and 16-B-per-lane accesses"),
        "note": "average over every %s launch of one bench.py run (postorder and preorder levels)" % kname,
    }
    # stamp: which kernels these counters belong to (bench.py prints roofline.traffic_stale when the stamp is not the running
    # tree's); the git head is informative only (the profile is usually taken on an uncommitted tree)
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    res["csrc_sha16"] = bench.csrc_sha16()
    try:
        res["git_head"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                         timeout=20).stdout.strip() or None
    except Exception:
        res["git_head"] = None
    if ncal > 0:
        res["calibrates"] = ncal
        res["launches_per_calibrate"] = n / ncal
        res["hbm_bytes_per_calibrate"] = (fetch + write) / ncal
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
