"""Kernel mix of a rocprofv3 kernel trace between the last two occurrences of a marker kernel (or of the whole trace).
usage: trace_mix.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.OrderedDict()
for r in rows:
    c = agg.setdefault(r["Kernel_Name"][:110], [0, 0.0])
    c[0] += 1
    c[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
wall = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("kernels %d, kernel time %.1f ms, first start -> last end %.1f ms" % (len(rows), tot / 1e3, wall / 1e3))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-112s %6d %10.0f us %5.1f %%" % (n, c, d, 100 * d / tot))
