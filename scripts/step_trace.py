"""One step's launch sequence from a rocprofv3 kernel_trace.csv: every launch between the last two launches of a marker kernel
(default rmsprop_kernel), in start order, with its duration and the gap to its predecessor.
usage: step_trace.py <kernel_trace.csv> <out.txt> [marker]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[3] if len(sys.argv) > 3 else "rmsprop_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)\(", n)
    return (m.group(1) if m else n)[:70]


idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
with open(sys.argv[2], "w") as f:
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    tot = 0
    f.write("# launches %d, wall %.1f us\n" % (b - a, (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3))
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += e - s
        f.write("%9.1f  dur %7.1f  gap %6.1f  grid %8s wg %4s lds %6s  %s\n" % (
            (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")),
            r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"), short(r["Kernel_Name"])))
        prev_end = max(prev_end, e)
    f.write("# sum of durations %.1f us\n" % (tot / 1e3))
