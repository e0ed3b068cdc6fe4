#!/bin/bash
# usage: tools/exp_sort_trace_all.sh <rows> <from_ms> <to_ms>  -- every kernel of the last Spearman step between two times
rows=$1; from=$2; to=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_trace
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python3 tools/bench_spearman.py --rows $rows --steps 1 > gpurun_out/prof_trace.log 2>&1
f=$(find gpurun_out/prof_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" $from $to <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at its convert / compact kernel, or -- a lent batch -- at the first ranking's sample
conv = [i for i, r in enumerate(rows) if "spearman_co" in r["Kernel_Name"]]
big = [i for i, r in enumerate(rows) if "sr_sample_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
last = max(conv) if conv and (not big or max(conv) > big[-2]) else big[-2]
t0 = int(rows[last]["Start_Timestamp"])
lo, hi = float(sys.argv[2]), float(sys.argv[3])
prev_end = None
for r in rows[last:]:
    st = (int(r["Start_Timestamp"]) - t0) / 1e6
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if lo <= st <= hi:
        name = r["Kernel_Name"].replace("tgx::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3 if prev_end else 0
        print("   +%8.3f ms  %-60s %8.1f us  (gap %6.1f us)" % (st, name, d * 1e3, gap))
    prev_end = int(r["End_Timestamp"])
PY
rm -rf gpurun_out/prof_trace
