"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output directory as argv[1])."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%6s %9.1f %8.2f %5.2f %s" % (r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                       100 * float(r["TotalDurationNs"]) / tot, r["Name"][:120]))
print("total ms", tot / 1e6)
