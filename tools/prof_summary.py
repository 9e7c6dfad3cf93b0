"""print a compact per-kernel summary from a rocprofv3 kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel ms: {tot/1e6:.2f}  (/{steps:g} steps = {tot/1e6/steps:.2f} ms/step)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:100]:100s} n={int(r['Calls']):5d} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
