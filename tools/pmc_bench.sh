#!/bin/bash
# usage: tools/pmc_bench.sh <tag> <counter> [bench args...]  -- one rocprofv3 --pmc pass (counters in their own run)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$tag
timeout -k 5 420 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py "$@" > gpurun_out/pmc_$tag.log 2>&1
tail -1 gpurun_out/pmc_$tag.log | cut -c1-200
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
head -1 $f
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][:90], r["Counter_Name"])
    if "tgx::" not in k[0]: continue
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(k[0], k[1], "dispatches", n, "mean", v / n)
PY
mkdir -p gpurun_out/pmc_keep_$tag && cp $f gpurun_out/pmc_keep_$tag/counter_collection.csv
rm -rf gpurun_out/pmc_$tag && mv gpurun_out/pmc_keep_$tag gpurun_out/pmc_$tag
rm -rf gpurun_out/pmc_$tag
