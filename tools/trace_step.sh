#!/bin/bash
# usage: tools/trace_step.sh [bench args...]  -- every device activity (kernels, copies) of the LAST bench step, in order,
# with the idle gaps between them: where a step's time goes when no kernel runs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_step
timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_step -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_step.log 2>&1
grep -E "^\{" gpurun_out/prof_step.log | cut -c1-220
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/prof_step/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tgx::", "").replace("(anonymous namespace)::", "")[:44]))
for f in glob.glob("gpurun_out/prof_step/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))[:30]))
ev.sort()
# one whole step: a step has one partition_init per key column (two in the bench's suite): from the 4th-last to the
# 2nd-last of them
idx = [i for i, e in enumerate(ev) if "partition_init" in e[2]]
if len(idx) >= 4:
    a, b = idx[-4], idx[-2]
else:
    a, b = max(0, len(ev) - 60), len(ev)
t0 = ev[a][0]
busy = 0
for i in range(a, b):
    s, e, n = ev[i]
    gap = (s - ev[i - 1][1]) / 1e3 if i > a else 0.0
    busy += e - s
    print("  +%8.1f us  %-46s %8.1f us   (idle before: %6.1f us)" % ((s - t0) / 1e3, n, (e - s) / 1e3, gap))
print("  step: %.1f us from first to last activity, %.1f us busy" % ((ev[b - 1][1] - t0) / 1e3, busy / 1e3))
PY
rm -rf gpurun_out/prof_step
