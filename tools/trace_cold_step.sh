#!/bin/bash
# usage: tools/trace_cold_step.sh [cold_step.py args]  -- the HIP calls and device activity of the LAST cold state
# (create -> update -> finalize -> destroy) of tools/cold_step.py, one line per call
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cold
timeout -k 5 300 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_cold -- python3 tools/cold_step.py "$@" > gpurun_out/prof_cold.log 2>&1
grep -E "warm|cold" gpurun_out/prof_cold.log
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/prof_cold/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  " + r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tgx::", "")[:44]))
for f in glob.glob("gpurun_out/prof_cold/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  copy " + r.get("Direction", "")[:30]))
for f in glob.glob("gpurun_out/prof_cold/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "HOST " + r["Function"]))
ev.sort()
# the last cold state: from the last hipDeviceSynchronize that follows a long quiet (torch.cuda.synchronize) ... simpler:
# the events after the second to last `scan_kernel`'s end up to the end of the trace
scans = [i for i, e in enumerate(ev) if "GPU  scan_kernel" in e[2]]
a = scans[-2]
t0 = ev[a][1]
for s, e, n in ev[a:]:
    if s < t0: continue
    print("  +%8.1f us  %-60s %8.1f us" % ((s - t0) / 1e3, n, (e - s) / 1e3))
PY
rm -rf gpurun_out/prof_cold
