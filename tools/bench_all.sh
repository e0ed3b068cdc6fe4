#!/bin/bash
# Reproduces every number quoted in BASELINE.md / DESIGN.md on ONE MI355X (about 6 minutes):
#   gpurun --timeout 1200 -- 'bash tools/bench_all.sh > gpurun_out/bench_all.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo "### $*"; timeout -k 5 400 "$@" 2>/dev/null | grep -E '^\{|^rows |ms/step|G rows/s' | cut -c1-400; }
run python bench.py                                   # headline: 1 G rows x 16 columns, + cpu_baseline
run python bench.py --force-distributed --rows 125000000 --steps 10 --warmup 3 --no-cpu-baseline   # one rank of the 8-way shard
run python tools/bench_configs.py --steps 3           # C2, C4, C5
run python tools/bench_regex.py                       # C3 (LargeUtf8)
run python tools/bench_regex.py --view                # C3 held as Utf8View
run python tools/bench_strings.py                     # string DISTINCT / LENGTH
run python tools/bench_tuples.py                      # multi-column uniqueness
run python tools/bench_distinct.py --rows 100000000 --steps 3 --sparse-rows 100000000   # uniqueness passes alone; sparse keys
run python tools/bench_kll.py
run python tools/bench_numeric32.py                   # Int32 / Float32 columns next to Int64 / Float64
run python tools/bench_spearman.py --ranks 8       # 100 M pairs + the distributed ranking over 8 threaded ranks on this one GPU
run python tools/bench_spearman.py --rows 1000000000 --steps 3   # 1 G pairs
run build/feed_batches                                # 8192-row batches through the C ABI from plain C (make -C tools): the coalescing rates
run python tools/bench_batches.py                     # the same through the Python binding (a ctypes call costs 2-4 us)
run python tools/bench_host_batches.py                # PCIe-inclusive rate
run build/feed_strings                                # a HOST string column as Utf8 / Utf8View / Dictionary in 8192-row batches, plain C
run python tools/bench_host_strings.py                # the same through the Python binding
