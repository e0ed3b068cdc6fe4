#!/bin/bash
# usage: tools/trace_step_api.sh [bench args...]  -- the HIP calls of the host between the end of the LAST step's scan and
# the next step's first kernel, with the device activity beside them: which host round trips make up a step's tail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_api
timeout -k 5 300 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_api -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_api.log 2>&1
grep -E "^\{" gpurun_out/prof_api.log | cut -c1-200
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/prof_api/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  " + r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tgx::", "").replace("(anonymous namespace)::", "")[:44]))
for f in glob.glob("gpurun_out/prof_api/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  copy " + r.get("Direction", "")[:30]))
for f in glob.glob("gpurun_out/prof_api/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "HOST " + r["Function"]))
ev.sort()
scans = [i for i, e in enumerate(ev) if "GPU  scan_kernel" in e[2]]
a = scans[-2]
t0 = ev[a][1]
inits = [i for i, e in enumerate(ev) if "partition_init" in e[2] and i > a]
b = inits[0] if inits else len(ev)
for s, e, n in ev[a:b]:
    if s < t0 - 5000: continue
    print("  +%8.1f us  %-60s %8.1f us" % ((s - t0) / 1e3, n, (e - s) / 1e3))
PY
rm -rf gpurun_out/prof_api
