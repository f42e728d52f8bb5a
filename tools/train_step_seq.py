#!/usr/bin/env python3
"""Every kernel of the LAST (steady-state) training step of a rocprofv3 kernel trace of `python3 bench.py --workload train ...`,
in launch order: start offset, duration, grid, name.  usage: train_step_seq.py <kernel_trace.csv> [out.txt]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adam_step_kernel" in r["Kernel_Name"]]
groups = []
for i in ends:
    if groups and i - groups[-1][-1] <= 3: groups[-1].append(i)
    else: groups.append([i])
a, b = groups[-2][-1] + 1, groups[-1][-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%9.1f us  %8.1f us  grid %-9s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:110]))
text = "\n".join(out) + "\n"
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
