#!/bin/bash
# usage: tools/trace_feed.sh <binary> [args...]  -- HIP API time by call (count, total ms) of a C feeder run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_feed2
timeout -k 5 300 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_feed2 -- "$@" > gpurun_out/prof_feed2.log 2>&1
grep -E "^\{" gpurun_out/prof_feed2.log | cut -c1-260
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob("gpurun_out/prof_feed2/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        t = tot[r["Function"]]
        t[0] += 1; t[1] += d; t[2] = max(t[2], d)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-36s calls %7d  total %9.2f ms  max %8.3f ms" % (k, v[0], v[1], v[2]))
kt = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/prof_feed2/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tgx::", "")[:40]
        kt[n][0] += 1; kt[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  GPU %-40s calls %6d  total %8.2f ms" % (k, v[0], v[1]))
mc = [0, 0.0]
for f in glob.glob("gpurun_out/prof_feed2/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        mc[0] += 1; mc[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("  GPU copies: %d, total %.2f ms" % tuple(mc))
PY
rm -rf gpurun_out/prof_feed2
