cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c5
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5 -- python3 tools/bench_configs.py --only C5 --steps 2 > gpurun_out/c5.log 2>&1
grep -E "^\{|mismatch" gpurun_out/c5.log | cut -c1-400
python3 tools/kstats.py gpurun_out/c5
