#!/bin/bash
# usage: tools/pmc_any.sh <tag> <counter> <script.py> [args...]  -- one rocprofv3 --pmc pass of any tool script
# (counters in their own run: never combined with --stats / trace domains other than --kernel-trace)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$tag
timeout -k 5 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
    if "tgx::" not in r["Kernel_Name"]: continue
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-62s %-12s dispatches %4d  mean %14.1f KiB" % (k[0], k[1], n, v / n))
PY
rm -rf gpurun_out/pmc_$tag
