cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pd_$1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd_$1 -- python3 tools/bench_distinct.py --steps 3 > gpurun_out/pd_$1.log 2>&1
grep "^col" gpurun_out/pd_$1.log
python3 tools/kstats.py gpurun_out/pd_$1
