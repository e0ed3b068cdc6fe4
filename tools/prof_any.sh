#!/bin/bash
# usage: tools/prof_any.sh <tag> <script.py> [args...]  -- rocprofv3 kernel stats (tgx kernels) of any tool script
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
grep -E "^\{|^rows|^col" gpurun_out/prof_$tag.log | cut -c1-300
python3 tools/kstats.py gpurun_out/prof_$tag
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
(head -1 $f; grep "tgx::" $f) > gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
