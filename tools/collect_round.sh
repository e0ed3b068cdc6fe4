#!/bin/bash
# usage: tools/collect_round.sh rNN   -- everything profiles/ holds for a round, from the tree as it is (one MI355X, ~25 min):
#   gpurun --timeout 2700 -- 'bash tools/collect_round.sh r04'
tag=$1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${tag}_gpu_tests.txt
bash tools/bench_all.sh > gpurun_out/${tag}_bench_all.txt 2>&1
bash tools/prof_bench.sh ${tag} --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_prof_bench.txt 2>&1
bash tools/pmc_bench.sh fetch FETCH_SIZE --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.txt 2>&1
bash tools/pmc_bench.sh write WRITE_SIZE --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.txt 2>&1
python tools/make_pmc_json.py ${tag} 1000000000 > gpurun_out/${tag}_make_pmc.txt 2>&1
cp profiles/${tag}_pmc_1Brows_16cols.json gpurun_out/ 2>/dev/null
for c in C2 C4 C5; do
  bash tools/prof_any.sh ${tag}_${c} tools/bench_configs.py --only $c --steps 3 > gpurun_out/${tag}_prof_${c}.txt 2>&1
done
bash tools/prof_any.sh ${tag}_ordered tools/bench_distinct.py --rows 1000000000 --steps 3 --ordered-only > gpurun_out/${tag}_prof_ordered.txt 2>&1
bash tools/prof_any.sh ${tag}_distinct tools/bench_distinct.py --rows 1000000000 --steps 3 --sparse-rows 1000000000 > gpurun_out/${tag}_prof_distinct.txt 2>&1
bash tools/prof_any.sh ${tag}_spearman tools/bench_spearman.py --rows 1000000000 --steps 3 > gpurun_out/${tag}_prof_spearman.txt 2>&1
bash tools/exp_sort_trace.sh 1000000000 TGX_SORT_DEBUG=1 > gpurun_out/${tag}_spearman_timeline.txt 2>&1
bash tools/trace_step_api.sh --force-distributed --rows 125000000 --steps 6 --warmup 3 2>&1 | grep -v hipEventQuery > gpurun_out/${tag}_shard_step_tail.txt
python tools/sim_bench_ranks.py > gpurun_out/${tag}_sim_ranks.txt 2>&1
python bench.py > gpurun_out/${tag}_bench_1gpu.json 2> gpurun_out/${tag}_bench_1gpu.err
tail -3 gpurun_out/${tag}_gpu_tests.txt; tail -c 600 gpurun_out/${tag}_bench_1gpu.json
