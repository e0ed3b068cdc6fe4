#!/bin/bash
# usage: tools/collect_round.sh rNN   -- everything profiles/ holds for a round, from the tree as it is (one MI355X, ~25 min):
#   gpurun --timeout 2700 -- 'bash tools/collect_round.sh r04'
tag=$1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${tag}_gpu_tests.txt
bash tools/bench_all.sh > gpurun_out/${tag}_bench_all.txt 2>&1
# (the headline's region alone: --no-secondary keeps the other configs' scan launches out of the scan kernel's average)
bash tools/prof_bench.sh ${tag} --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_prof_bench.txt 2>&1
bash tools/pmc_bench.sh fetch FETCH_SIZE --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_pmc_fetch.txt 2>&1
bash tools/pmc_bench.sh write WRITE_SIZE --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_pmc_write.txt 2>&1
python tools/make_pmc_json.py ${tag} 1000000000 > gpurun_out/${tag}_make_pmc.txt 2>&1
cp profiles/${tag}_pmc_1Brows_16cols.json gpurun_out/ 2>/dev/null
for c in C2 C4 C5; do
  bash tools/prof_any.sh ${tag}_${c} tools/bench_configs.py --only $c --steps 3 > gpurun_out/${tag}_prof_${c}.txt 2>&1
done
bash tools/prof_any.sh ${tag}_ordered tools/bench_distinct.py --rows 1000000000 --steps 3 --ordered-only > gpurun_out/${tag}_prof_ordered.txt 2>&1
bash tools/prof_any.sh ${tag}_distinct tools/bench_distinct.py --rows 1000000000 --steps 3 --sparse-rows 1000000000 > gpurun_out/${tag}_prof_distinct.txt 2>&1
bash tools/prof_any.sh ${tag}_spearman tools/bench_spearman.py --rows 1000000000 --steps 3 > gpurun_out/${tag}_prof_spearman.txt 2>&1
# the pattern kernels (C3): kernel stats, bytes fetched, SQ counters -- one and three patterns, Utf8 and Utf8View
bash tools/prof_any.sh ${tag}_regex tools/bench_regex.py --steps 5 > gpurun_out/${tag}_prof_regex.txt 2>&1
bash tools/prof_any.sh ${tag}_regex_view tools/bench_regex.py --steps 5 --view > gpurun_out/${tag}_prof_regex_view.txt 2>&1
bash tools/pmc_any.sh ${tag}_rxf FETCH_SIZE tools/bench_regex.py --steps 3 > gpurun_out/${tag}_pmc_regex_fetch.txt 2>&1
bash tools/pmc_any.sh ${tag}_rxs "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES" tools/bench_regex.py --steps 3 > gpurun_out/${tag}_pmc_regex_sq.txt 2>&1
# the list paths (sparse keys, strings): bytes fetched and written per pass
bash tools/pmc_any.sh ${tag}_lf FETCH_SIZE tools/bench_distinct.py --rows 1000000000 --steps 2 --sparse-rows 1000000000 > gpurun_out/${tag}_pmc_lists_fetch.txt 2>&1
bash tools/pmc_any.sh ${tag}_lw WRITE_SIZE tools/bench_distinct.py --rows 1000000000 --steps 2 --sparse-rows 1000000000 > gpurun_out/${tag}_pmc_lists_write.txt 2>&1
bash tools/pmc_any.sh ${tag}_sf FETCH_SIZE tools/bench_strings.py > gpurun_out/${tag}_pmc_strings_fetch.txt 2>&1
bash tools/pmc_any.sh ${tag}_sw WRITE_SIZE tools/bench_strings.py > gpurun_out/${tag}_pmc_strings_write.txt 2>&1
# a state per table: create -> update -> finalize -> destroy beside the warm step, and the last cold state's timeline
python tools/cold_step.py > gpurun_out/${tag}_cold_step.txt 2>&1
python tools/cold_step.py --rows 1000000 >> gpurun_out/${tag}_cold_step.txt 2>&1
bash tools/trace_cold_step.sh > gpurun_out/${tag}_cold_step_trace.txt 2>&1
# the experiments of the round whose outcome was "dropped": kept as evidence
python tools/exp_chunked_distinct.py > gpurun_out/${tag}_exp_chunked_distinct.txt 2>&1
bash tools/exp_sort_trace.sh 1000000000 TGX_SORT_DEBUG=1 > gpurun_out/${tag}_spearman_timeline.txt 2>&1
bash tools/trace_step_api.sh --force-distributed --rows 125000000 --steps 6 --warmup 3 2>&1 | grep -v hipEventQuery > gpurun_out/${tag}_shard_step_tail.txt
python tools/sim_bench_ranks.py > gpurun_out/${tag}_sim_ranks.txt 2>&1
python bench.py > gpurun_out/${tag}_bench_1gpu.json 2> gpurun_out/${tag}_bench_1gpu.err
tail -3 gpurun_out/${tag}_gpu_tests.txt; tail -c 600 gpurun_out/${tag}_bench_1gpu.json
