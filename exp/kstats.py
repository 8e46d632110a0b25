import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f, "total ms/step", tot / 1e6 / steps)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:100]:100s} n/step={float(r['Calls'])/steps:5.1f} avg_ms={float(r['AverageNs'])/1e6:8.3f} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.2f}")
