#!/bin/bash
# per-kernel times of tools/bench_strings.py (optionally with another build of the library: TGX_LIB)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=${1:-bs}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/bench_strings.py --steps 5 > gpurun_out/$tag.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "fp_" in r["Name"]: print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
