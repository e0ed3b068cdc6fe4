#!/bin/bash
# usage: tools/copy_round.sh rNN   -- what tools/collect_round.sh left in gpurun_out/ under the names profiles/ uses
t=$1
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/${t}_bench_1gpu.json $p/${t}_bench_1gpu.json
cp $g/${t}_bench_all.txt $p/${t}_bench_all.txt
(cat $g/${t}_gpu_tests.txt; echo; echo "# tools/sim_bench_ranks.py"; cat $g/${t}_sim_ranks.txt) > $p/${t}_gpu_tests_and_sim_ranks.txt
cp $g/${t}_kernel_stats.csv $p/${t}_kernel_stats_1Brows_16cols.csv
grep "^{" $g/prof_${t}.log | tail -1 > $p/${t}_kernel_stats_1Brows_16cols.benchline.json
cp $g/${t}_C2_kernel_stats.csv $p/${t}_kernel_stats_C2_100Mrows_8cols.csv
cp $g/${t}_C4_kernel_stats.csv $p/${t}_kernel_stats_C4_1Brows_16cols.csv
cp $g/${t}_C5_kernel_stats.csv $p/${t}_kernel_stats_C5_250Mrows_64cols.csv
cp $g/${t}_distinct_kernel_stats.csv $p/${t}_kernel_stats_distinct_1Brows.csv
cp $g/${t}_ordered_kernel_stats.csv $p/${t}_kernel_stats_ordered_keys_1Brows.csv
cp $g/${t}_spearman_kernel_stats.csv $p/${t}_kernel_stats_spearman_1Bpairs.csv
cp $g/${t}_spearman_timeline.txt $p/${t}_spearman_timeline.txt
cp $g/${t}_shard_step_tail.txt $p/${t}_shard_step_tail.txt
cp $g/${t}_pmc_1Brows_16cols.json $p/${t}_pmc_1Brows_16cols.json 2>/dev/null
(echo "# FETCH_SIZE pass"; cat $g/${t}_pmc_fetch.txt; echo; echo "# WRITE_SIZE pass"; cat $g/${t}_pmc_write.txt) > $p/${t}_pmc_passes.txt
cp $g/${t}_regex_kernel_stats.csv $p/${t}_kernel_stats_regex.csv 2>/dev/null
cp $g/${t}_regex_view_kernel_stats.csv $p/${t}_kernel_stats_regex_utf8view.csv 2>/dev/null
(echo "# tools/bench_regex.py under rocprofv3 --pmc (own runs, --kernel-trace only): FETCH_SIZE (KiB per dispatch; gfx950: x2 for 16 B / lane streams)"; cat $g/${t}_pmc_regex_fetch.txt; echo; echo "# SQ counters"; cat $g/${t}_pmc_regex_sq.txt) > $p/${t}_pmc_regex.md 2>/dev/null
(echo "# tools/bench_distinct.py (1 G rows; sparse keys through the key lists) and tools/bench_strings.py (100 M x 28 B) under rocprofv3 --pmc, own runs: KiB per dispatch"; for f in lists_fetch lists_write strings_fetch strings_write; do echo; echo "## $f"; cat $g/${t}_pmc_$f.txt; done) > $p/${t}_pmc_lists.md 2>/dev/null
cp $g/${t}_cold_step.txt $p/${t}_cold_step.txt 2>/dev/null
cp $g/${t}_cold_step_trace.txt $p/${t}_cold_step_trace.txt 2>/dev/null
cp $g/${t}_exp_chunked_distinct.txt $p/${t}_exp_chunked_distinct.txt 2>/dev/null
ls -la $p/${t}_*
