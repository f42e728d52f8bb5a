#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into profiles/:
   <tag>_<run>_kernel_stats.csv  top rows of rocprofv3 --kernel-trace --stats
   <tag>_summary.md              the fused kernel's average duration, PMC traffic, roofline fractions
   pmc_traffic.json              HBM bytes per launch (FETCH_SIZE doubled on gfx950 + WRITE_SIZE, KiB -> bytes)"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "dasv::dmha_fwd_vec"


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # newest run wins
    return g[-1] if g else None


def stats(run):
    path = one("%s_trace/*/*kernel_stats.csv" % run)
    rows = list(csv.DictReader(open(path)))
    keep = [r for r in rows if KERNEL in r["Name"]] + [r for r in rows if KERNEL not in r["Name"]][:12]
    with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, run)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    # per (kernel, grid size) from the trace itself: bench.py launches the config-2 kernel at B = 64 (the quoted figure) and, as
    # context, at B = 512 — same kernel name, eight times the grid
    tr = one("%s_trace/*/*kernel_trace.csv" % run)
    groups = {}
    for r in csv.DictReader(open(tr)):
        if KERNEL in r["Kernel_Name"]:
            groups.setdefault((r["Kernel_Name"], int(r["Grid_Size_X"])), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: (len(v), sum(v) / len(v), min(v)) for k, v in groups.items()}


def pmc(run, counter):
    path = one("%s_%s/*/*counter_collection.csv" % (run, "fetch" if counter == "FETCH_SIZE" else "write"))
    vals = {}
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / max(1, len(v[2:])) if len(v) > 2 else sum(v) / len(v) for k, v in vals.items()}


def bench_line(run):
    for line in open(os.path.join(src, "%s_trace.log" % run)):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


import subprocess
try:
    head = subprocess.check_output(["git", "-C", ROOT, "log", "-1", "--format=%h %s"], text=True).strip()
except Exception:
    head = "unknown"
traffic, md = {}, ["# rocprofv3 summary, round tag `%s`\n\n" % tag,
                   "Collected with the sources of commit `%s` (the last commit touching a kernel precedes it).\n\n" % head,
                   "Commands: `tools/collect_profiles.sh %s` (kernel-trace + stats pass, then separate `--pmc FETCH_SIZE` and "
                   "`--pmc WRITE_SIZE` passes) around `python3 bench.py --workload pool ...` and `python3 bench.py ...`.\n\n" % tag,
                   "FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md "
                   "section HBM); both counters are KiB.\n"]
alg = {"pool": 39954688, "embed": 24965760}   # embed: the in-model [64,19,5120] launch (alignments not materialised)
for run in ("pool", "embed"):
    st = stats(run)
    fe, wr = pmc(run, "FETCH_SIZE"), pmc(run, "WRITE_SIZE")
    bl = bench_line(run)
    md.append("\n## `%s` workload\n\n" % run)
    gmin = {}
    for (name, grid) in st:
        gmin[name] = min(grid, gmin.get(name, grid))
    for (name, grid), (calls, avg, mn) in sorted(st.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        f = fe.get((name, grid), float("nan")) * 1024 * 2
        w = wr.get((name, grid), float("nan")) * 1024
        hbm = f + w
        if grid != gmin[name]:      # the B = 512 context launches of bench.py (`large_batch_context`): eight times config 2
            a8 = 8 * alg["pool"] - 7 * 4 * (64 * 8 + 64)
            md.append("* kernel `%s`, grid %d (bench.py's B = 512 context launches): %d calls, average %.2f us; FETCH_SIZE x2 + WRITE_SIZE = "
                      "%.1f MB vs %.1f MB algorithmic (x%.3f) -> %.0f GB/s = %.1f %% of 8 TB/s\n"
                      % (name, grid, calls, avg, hbm / 1e6, a8 / 1e6, hbm / a8, a8 / avg / 1e3, a8 / avg / 1e3 / 80))
            continue
        # inside the embed run the <16, 1, 4> variant is bench.py's config-2 measurement (`roofline_pooling`), the PERM variant the
        # launch inside the model
        c2 = run == "embed" and "16, 1, 4" in name
        ab = alg["pool"] if c2 else alg[run]
        if not c2:
            traffic["%s_fp32" % run] = {"kernel": name, "fetch_bytes_x2": f, "write_bytes": w, "hbm_bytes_per_launch": int(hbm),
                                        "avg_kernel_us": avg}
        md.append("* kernel `%s`, grid %d%s: %d calls, **average %.2f us** (min %.2f us)\n"
                  % (name, grid, " (BASELINE config 2, bench.py's `roofline_pooling`)" if c2 else "", calls, avg, mn))
        md.append("* PMC per launch: FETCH_SIZE x2 = %.2f MB, WRITE_SIZE = %.2f MB -> HBM traffic %.2f MB vs algorithmic %.2f MB (x%.3f)\n"
                  % (f / 1e6, w / 1e6, hbm / 1e6, ab / 1e6, hbm / ab))
        md.append("* algorithmic bytes / kernel time = %.0f GB/s = %.1f %% of 8 TB/s (spec), %.1f %% of 6.29 TB/s (measured copy peak)\n"
                  % (ab / avg / 1e3, ab / avg / 1e3 / 80, ab / avg / 1e3 / 62.9))
    if run == "embed":
        rows = list(csv.DictReader(open(one("embed_trace/*/*kernel_stats.csv"))))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        # The run holds the headline passes (conv operand mode split8: conv3x3_window_kernel<3,..>) and bench.py's companion
        # passes in the other operand modes (`conv_operand_modes`: <0,..> = split-bf16, <2,..> = fp16); the other kernels are shared.
        names = {"3": "split8 (fp16 products + block-scaled e4m3 cross terms; the headline)", "0": "split-bf16 (companion)",
                 "2": "fp16 operands (companion)"}
        pipe = {"3": 2, "0": 3, "2": 1}      # matrix-pipe time per algorithmic FLOP in units of a plain bf16 MFMA
        md.append("* kernel mix of the whole run (all GPU kernels, %.2f ms of kernel time):\n" % (tot / 1e6))
        for r in rows[:10]:
            md.append("  * `%s` — %s calls, avg %.1f us, %.1f %% of kernel time\n"
                      % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
        flops = 64 * (39.13e9 - 0.055e9)   # seven wide layers of VGG4L/1024 at T=300 (SURVEY appendix D.1 minus conv11)
        for code in ("3", "0", "2"):
            ks = [r for r in rows if "conv3x3_window_kernel<%s," % code in r["Name"]]
            if not ks:
                continue
            calls = sum(int(r["Calls"]) for r in ks)
            per_step = sum(float(c["TotalDurationNs"]) for c in ks) / (calls / 7.0) / 1e3
            md.append("* `conv3x3_window_kernel<%s,..>` %s: %d launches, %.0f us per step over the 7 layers (average launch %.1f us) -> %.0f "
                      "TFLOP/s algorithmic = %.1f %% of the 2.5 PFLOP/s dense bf16 peak; matrix-pipe time x%d -> %.0f %% of the pipe\n"
                      % (code, names[code], calls, per_step, per_step / 7, flops / per_step / 1e6, 100 * flops / per_step / 1e6 / 2500,
                         pipe[code], 100 * pipe[code] * flops / per_step / 1e6 / 2500))
        # HBM traffic of the conv launches per operand mode (separate PMC passes; same gfx950 doubling of FETCH_SIZE)
        def conv_pmc(which, counter, code):
            path = one("embed_%s/*/*counter_collection.csv" % which)
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                 if "conv3x3_window_kernel<%s," % code in r["Kernel_Name"] and r["Counter_Name"] == counter]
            return sum(v) / len(v) if v else float("nan")
        layers = [(128, 128, 300, 80), (128, 256, 150, 40), (256, 256, 150, 40), (256, 512, 75, 20), (512, 512, 75, 20),
                  (512, 1024, 38, 10), (1024, 1024, 38, 10)]
        # algorithmic bytes of an average launch: read x, write y (fp32 NHWC), read both 16-bit weight planes once
        calg = sum(64 * h * w * (ci + co) * 4 + 2 * 2 * 9 * ci * co for ci, co, h, w in layers) / len(layers)
        for code, key in (("3", "split8"), ("0", "split")):
            cf, cw = conv_pmc("fetch", "FETCH_SIZE", code) * 1024 * 2, conv_pmc("write", "WRITE_SIZE", code) * 1024
            if cf == cf and cw == cw:
                traffic["embed_conv_%s" % key] = {"kernel": "dasv::conv3x3_window_kernel<%s,..>" % code, "fetch_bytes_x2": cf, "write_bytes": cw,
                                                  "hbm_bytes_per_launch": int(cf + cw), "algorithmic_bytes_per_launch": int(calg)}
                md.append("* conv PMC per launch, mode %s (average over the seven layers): FETCH_SIZE x2 = %.1f MB, WRITE_SIZE = %.1f MB -> HBM "
                          "traffic %.1f MB vs %.1f MB algorithmic (x + y + both weight planes once): x%.2f.  Writes are exact.  The counters "
                          "sit on the L2's fabric side and include Infinity-Cache hits (MI355X_MICROARCH.md): the excess reads are the "
                          "weight planes (up to 37.7 MB for conv42, resident in the 256 MiB Infinity Cache) streamed once per XCD and "
                          "per round of workgroups, plus the window halo.  The kernel stays MFMA-bound\n"
                          % (key, cf / 1e6, cw / 1e6, (cf + cw) / 1e6, calg / 1e6, (cf + cw) / calg))
    if bl:
        pr = bl.get("roofline_pooling") or bl["roofline"]
        md.append("* bench.py under the profiler: value %.1f %s; pooling-kernel avg_launch_us %.2f by events (includes the event/launch gap)\n"
                  % (bl["value"], bl["unit"], pr["avg_launch_us"]))
# MFMA utilisation from the PMC passes (derived counter MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs))
util = {}
for run in ("embed", "train"):
    path = one("%s_mfma/*/*counter_collection.csv" % run)
    if not path:
        continue
    agg = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "MfmaUtil" and ("conv3x3_window_kernel" in r["Kernel_Name"] or "conv3x3_wgrad_kernel" in r["Kernel_Name"]):
            agg.setdefault(r["Kernel_Name"].replace("void ", "")[:70], []).append(float(r["Counter_Value"]))
    if agg:
        md.append("\n## MFMA utilisation (`rocprofv3 --pmc MfmaUtil --kernel-trace`, `%s` workload)\n\n" % run)
        md.append("MfmaUtil = matrix-core busy cycles / (GPU active cycles x SIMDs), per dispatch, in percent of the matrix pipe's "
                  "time (the pipe's peak is the 2.5 PFLOP/s dense bf16 figure).\n\n| kernel | dispatches | mean | max |\n|---|---|---|---|\n")
        for k, v in sorted(agg.items()):
            md.append("| `%s` | %d | %.1f %% | %.1f %% |\n" % (k, len(v), sum(v) / len(v), max(v)))
            util["%s:%s" % (run, k)] = round(sum(v) / len(v), 2)
traffic["mfma_util_percent"] = util
# the round's diagnostic logs, verbatim (tools/collect_profiles.sh): the steady-state training step's kernel mix, the power test
# (random vs all-zero data), the two weight-gradient forms, and the split8 tile stamps with the in-kernel clock
for name, title in (("train_step_mix.md", None), ("conv_power_test.log", "power test: the same split8 launches on random and on all-zero data (tools/_diag/conv_power_test.py)"),
                    ("wgrad_x8_ab.log", "weight gradient: split-bf16 vs fp16 + e4m3 form (tools/_diag/wgrad_x8_ab.py)"),
                    ("conv_stamps.log", "split8 tile stamps and in-kernel clock (tools/conv_stamps.py)")):
    path = os.path.join(src, name)
    if not os.path.exists(path):
        continue
    text = "".join(l for l in open(path) if "amdgpu.ids" not in l)
    if title is None:
        open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write(text)
        md.append("\n## steady-state training step\n\n" + text)
        seq = os.path.join(src, "train_step_sequence.txt")
        if os.path.exists(seq):      # every kernel of that step in launch order (tools/train_step_seq.py)
            open(os.path.join(dst, "%s_train_step_sequence.txt" % tag), "w").write(open(seq).read())
    else:
        md.append("\n## %s\n\n```\n%s```\n" % (title, text))
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("".join(md))
print("".join(md))
