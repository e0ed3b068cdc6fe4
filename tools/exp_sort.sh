#!/bin/bash
# usage: tools/exp_sort.sh <rows> <steps> "<ENV=.. ENV=..>" ...   -- kernel totals per Spearman step for each setting of
# the sorter's knobs (kernels/sortrank.hip); one rocprofv3 --kernel-trace --stats run per setting
rows=$1; shift; steps=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1)); tag=sortexp$i
  rm -rf gpurun_out/prof_$tag
  ( export $cfg; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 tools/bench_spearman.py --rows $rows --steps $steps > gpurun_out/prof_$tag.log 2>&1 )
  echo "=== $cfg"
  grep -E "^\{" gpurun_out/prof_$tag.log | cut -c1-200
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $((steps+1)) <<'PY'
import csv, sys
steps = int(sys.argv[2])
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "tgx::" in n:
        print("   %-46s calls/step %6.1f  ms/step %8.3f  max %8.3f" % (n.split("tgx::")[1].replace("(anonymous namespace)::", "")[:46],
              int(row["Calls"]) / steps, float(row["TotalDurationNs"]) / 1e6 / steps, float(row["MaxNs"]) / 1e6))
PY
  rm -rf gpurun_out/prof_$tag
done
