#!/usr/bin/env python3
"""Every launch of ONE step of a bench run, from the kernel trace of the stats pass of profiles/tools/collect.sh:

    python3 profiles/tools/timeline.py gpurun_out/default_r05/stats HET_rgat_aggregate_runs_packed 0.62 [occurrence] > default_timeline.txt

The step is the window between two consecutive launches of the anchor kernel (one per step), shifted back by `lead_ms` so that it
starts where the step does (the anchor is not the step's first launch).  Columns: start (ms from the window's first launch of a HET_
kernel), duration, queue (HIP stream), kernel, grid."""
import csv
import glob
import os
import sys


def main():
    d, anchor, lead = sys.argv[1], sys.argv[2], float(sys.argv[3])
    occ = int(sys.argv[4]) if len(sys.argv) > 4 else -4
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"],
                         r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    rows.sort()
    hits = [i for i, r in enumerate(rows) if anchor in r[3]]
    if len(hits) < 3:
        print(f"anchor {anchor!r}: {len(hits)} launches in {d}")
        return 1
    a, b = rows[hits[occ]][0], rows[hits[occ + 1]][0]
    t0, t1 = a - lead * 1e6, b - lead * 1e6
    win = [r for r in rows if t0 <= r[0] < t1]
    first = next((r[0] for r in win if "HET_" in r[3]), win[0][0])
    queues = {}
    print(f"One step of `{os.path.basename(os.path.dirname(os.path.normpath(d)))}` (bench.py under rocprofv3 --kernel-trace, profiles/tools/collect.sh; launch "
          f"{occ} of {anchor} .. the next one, window moved {lead} ms back); step length {(b - a) / 1e6:.3f} ms")
    print("start ms   dur ms  queue  kernel")
    for s, e, q, name, grid in win:
        qn = queues.setdefault(q, len(queues) + 1)
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{(s - first) / 1e6:8.3f} {(e - s) / 1e6:8.4f}  q={qn}   {name[:60]} grid={grid}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
