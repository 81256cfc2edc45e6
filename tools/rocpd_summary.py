"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as text:
per kernel calls / total / average / share.  Usage: python tools/rocpd_summary.py <db> [out.txt]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "rocBLAS/Tensile " + name[:14] + " MT" + (m.group(1) if m else "")
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    db = sys.argv[1]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % db,
             "# total kernel time %.3f ms over %d kernels" % (tot / 1e6 if tot > 1e7 else tot / 1e3, len(rows)),
             "%8s %12s %10s %7s  %s" % ("calls", "total_us", "avg_us", "share%", "kernel")]
    unit = 1e3 if tot > 1e7 else 1.0          # ns -> us when the db stores ns
    for name, calls, total, avg, pct in rows:
        lines.append("%8d %12.1f %10.2f %7.2f  %s" % (calls, total / unit, avg / unit, pct, short(name)))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
