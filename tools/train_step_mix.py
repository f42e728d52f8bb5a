#!/usr/bin/env python3
"""Kernel mix of the LAST (steady-state) training step from a rocprofv3 kernel trace of
`python3 bench.py --workload train ...` (the first steps contain MIOpen's one-off autotune for the pieces it still runs).
usage: train_step_mix.py <kernel_trace.csv> [out.md]"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "multi_tensor_apply" in n or "adam_step_kernel" in n]
groups = []
for i in ends:
    if groups and i - groups[-1][-1] <= 3: groups[-1].append(i)
    else: groups.append([i])
a, b = groups[-2][-1] + 1, groups[-1][-1] + 1
step = rows[a:b]
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
agg = collections.OrderedDict()
for r in step:
    c = agg.setdefault(r["Kernel_Name"][:100], [0, 0.0])
    c[0] += 1; c[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
out = ["# steady-state training step (last of the traced run): %.2f ms wall, %.2f ms of kernel time, %d kernels\n\n" % (wall, tot / 1e3, len(step)),
       "| kernel | launches | us | share |\n|---|---|---|---|\n"]
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    out.append("| `%s` | %d | %.0f | %.1f %% |\n" % (n.replace("|", "/"), c, d, 100 * d / tot))
text = "".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
