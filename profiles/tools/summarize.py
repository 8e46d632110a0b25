"""Turn the rocprofv3 output of profiles/tools/collect.sh into the two files committed per round:
   <dir>/kernel_stats.csv (copy of the --stats table) and <dir>/pmc.json (per-kernel HBM bytes and MFMA busy).
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are KiB and on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 B/lane) read stream (MI355X_MICROARCH.md, section HBM).  Calibration of that rule on
this repo's own kernels is printed at the end (HET_rowdot_fwd reads E*H*D*4 bytes, HET_rowdot_bwd_dx writes them)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

d = sys.argv[1]


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n.strip() if n.startswith("HET_") else n[:60]


def counters(sub):
    f = sorted(glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True))
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return acc
    for r in csv.DictReader(open(f[-1])):
        # launches of one kernel on differently sized inputs are kept apart by their grid size
        acc[short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


st = sorted(glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True))
if st:
    shutil.copy(st[-1], os.path.join(d, "kernel_stats.csv"))
    rows = list(csv.DictReader(open(st[-1])))
    print("== kernel stats (16 steps: 3 warm-up + 10 timed + 3 for the per-op breakdown; one-time grouping and layout builds included) ==")
    for r in rows[:22]:
        print(f"{short(r['Name']):45s} calls={int(r['Calls']):5d} avg_ms={float(r['AverageNs'])/1e6:8.3f} "
              f"total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")

def kernel_source_sha16():
    """Digest of the kernel sources the profiled library was built from (het_amd/csrc/*): bench.py quotes a counter of this
    file only while the tree's digest is the same, i.e. while no kernel changed since the counters were collected."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "het_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


fe, wr, mf = counters("fetch"), counters("write"), counters("mfma")
out = {"kernel_source_sha16": kernel_source_sha16(),
       "_how": "profiles/tools/collect.sh; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (mean over launches); "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): GRBM_GUI_ACTIVE is summed "
               "over the 8 XCDs (it equals 8 * kernel duration * ~2.0 GHz here) and one f32 32x32x2 MFMA holds its SIMD's "
               "matrix pipe for 64 cycles (measured: busy cycles / MFMAs issued = 64.0)",
       "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    if not k.startswith("HET_"):
        continue
    F = fe.get(k, {}).get("FETCH_SIZE", [])
    W = wr.get(k, {}).get("WRITE_SIZE", [])
    e = {"launches": len(F) or len(W)}
    if F:
        e["FETCH_SIZE_KiB"] = round(sum(F) / len(F), 1)
    if W:
        e["WRITE_SIZE_KiB"] = round(sum(W) / len(W), 1)
    if F and W:
        e["hbm_bytes_per_launch"] = int((2 * sum(F) / len(F) + sum(W) / len(W)) * 1024)
    m = mf.get(k, {})
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
        busy, act = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(m["GRBM_GUI_ACTIVE"])
        e["mfma_busy_cycles"] = round(busy / len(m["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
        e["gui_active_cycles"] = round(act / len(m["GRBM_GUI_ACTIVE"]), 1)
        e["mfma_busy_frac"] = round(busy / (act / 8 * 256 * 4), 4) if act else None
    out["kernels"][k] = e
json.dump(out, open(os.path.join(d, "pmc.json"), "w"), indent=1)
print("== PMC ==")
for k, e in out["kernels"].items():
    print(f"{k:52s} n={e['launches']:3d} hbm_GB={e.get('hbm_bytes_per_launch', 0)/1e9:7.3f} "
          f"fetch_GiBx2={2*e.get('FETCH_SIZE_KiB', 0)/1048576:7.3f} write_GiB={e.get('WRITE_SIZE_KiB', 0)/1048576:7.3f} "
          f"mfma_busy={e.get('mfma_busy_frac')}")
