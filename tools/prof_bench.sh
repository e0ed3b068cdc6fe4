#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   -- rocprofv3 kernel stats of bench.py, tgx kernels only
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 5 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
grep "^{" gpurun_out/prof_$tag.log | tail -1   # (the bench line of the profiled run: not the profiler's last log line)
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
head -1 $f; grep "tgx::" $f
# keep the summary (what profiles/ commits), drop the raw traces: gpurun_out/ travels back only while it stays small
(head -1 $f; grep "tgx::" $f) > gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
