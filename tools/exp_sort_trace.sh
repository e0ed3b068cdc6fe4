#!/bin/bash
# usage: tools/exp_sort_trace.sh <rows> "<ENV=..>"   -- every launch of the sorter's big kernels in the LAST step, in order
rows=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -rf gpurun_out/prof_trace
  ( export $cfg; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python3 tools/bench_spearman.py --rows $rows --steps 1 > gpurun_out/prof_trace.log 2>&1 )
  echo "=== $cfg"; grep -E "^\{" gpurun_out/prof_trace.log | cut -c1-160
  f=$(find gpurun_out/prof_trace -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tgx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last spearman_convert / compact kernel
# the last step starts at its convert / compact kernel, or -- a lent batch -- at the first ranking's sample
conv = [i for i, r in enumerate(rows) if "spearman_co" in r["Kernel_Name"]]
big = [i for i, r in enumerate(rows) if "sr_sample_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
last = max(conv) if conv and (not big or max(conv) > big[-2]) else big[-2]
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.3:
        name = r["Kernel_Name"].split("tgx::")[1].replace("(anonymous namespace)::", "").split("(")[0]
        print("   +%8.3f ms  %-34s %8.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, name, d))
print("   end at +%.3f ms" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6))
PY
  rm -rf gpurun_out/prof_trace
done
