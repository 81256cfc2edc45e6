"""Post-process tools/profile_roofline.sh output: per-kernel time (kernel-trace stats) and HBM traffic
(FETCH_SIZE / WRITE_SIZE counter passes).  Units and the gfx950 correction follow
MI355X_MICROARCH.md "HBM": the counters are in KiB-like units of 1024 B... rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in kilobytes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) streaming
reads, so read bytes = 2 x FETCH_SIZE x 1024 for our kernels (all stream with 16-B loads); WRITE_SIZE is
used as reported (uncalibrated, said so in the output).

usage: python tools/pmc_summary.py <prof_dir> <summary.txt> <traffic.json>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?\(", name)
    if name.startswith("Cijk_"):
        return "rocBLAS " + name[:20]
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


def key(row):
    """kernel template + number of workgroups: one template serves several layer shapes."""
    try:
        if "Grid_Size" in row:
            g, w = int(row["Grid_Size"]), int(row["Workgroup_Size"])
        else:
            g = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])
            w = int(row["Workgroup_Size_X"]) * int(row["Workgroup_Size_Y"]) * int(row["Workgroup_Size_Z"])
        wg = str(g // max(w, 1))
    except (KeyError, ValueError):
        wg = "?"
    return short(row["Kernel_Name"]) + " |wg=" + wg


def read_counter(prof_dir, sub, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(prof_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    per[key(row)].append(float(row["Counter_Value"]))
    return per


def main():
    prof, out_txt, out_json = sys.argv[1:4]
    times = defaultdict(list)
    for f in glob.glob(os.path.join(prof, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                times[key(row)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    fetch = read_counter(prof, "fetch", "FETCH_SIZE")
    write = read_counter(prof, "write", "WRITE_SIZE")
    total = sum(sum(v) for v in times.values())
    rows, traffic = [], {}
    for k, v in sorted(times.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        fk = (sum(fetch[k]) / len(fetch[k])) if k in fetch else None
        wk = (sum(write[k]) / len(write[k])) if k in write else None
        rd = 2.0 * fk * 1024 if fk is not None else None            # gfx950: FETCH_SIZE = 1/2 of wide reads
        wr = wk * 1024 if wk is not None else None
        rows.append((k, len(v), sum(v), avg, 100 * sum(v) / total, rd, wr))
        if rd is not None or wr is not None:
            traffic[k] = {"avg_us": avg, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                          "hbm_bytes_per_launch": (rd or 0) + (wr or 0), "calls": len(v)}
    with open(out_txt, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace + --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: bench.py --steps 4 --warmup 2\n")
        fh.write("# read MB = 2 x FETCH_SIZE KB (gfx950 wide-load correction), write MB = WRITE_SIZE KB (uncalibrated)\n")
        fh.write("%7s %11s %10s %7s %10s %10s  %s\n" % ("calls", "total_us", "avg_us", "share%", "read_MB", "write_MB", "kernel"))
        for k, n, tot, avg, pct, rd, wr in rows[:90]:
            fh.write("%7d %11.1f %10.2f %7.2f %10s %10s  %s\n" % (
                n, tot, avg, pct, "%.2f" % (rd / 1e6) if rd is not None else "-",
                "%.2f" % (wr / 1e6) if wr is not None else "-", k[:110]))
        # per device function (all launch geometries together): what bench.py's `roofline.kernel` refers to
        sym = defaultdict(lambda: [0, 0.0])
        for k, n, tot, avg, pct, rd, wr in rows:
            name = k.split(" |wg=")[0]
            sym[name][0] += n
            sym[name][1] += tot
        fh.write("\n# per device function, all launch geometries together\n")
        fh.write("%7s %11s %10s %7s  %s\n" % ("calls", "total_us", "avg_us", "share%", "kernel"))
        for name, (n, tot) in sorted(sym.items(), key=lambda kv: -kv[1][1])[:25]:
            fh.write("%7d %11.1f %10.2f %7.2f  %s\n" % (n, tot, tot / n, 100 * tot / total, name[:120]))
    json.dump(traffic, open(out_json, "w"), indent=1)
    print(open(out_txt).read()[:6000])


if __name__ == "__main__":
    main()
