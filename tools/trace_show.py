"""Print a rocprofv3 kernel_stats.csv: python tools/trace_show.py <csv> [steps] [substring filters...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
keys = sys.argv[3:]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows:
    n = r['Name']
    if keys and not any(k in n for k in keys):
        continue
    print("%5s %9.1f %8.2f %5.2f%%  %s" % (r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, float(r['Percentage']), n[:130]))
