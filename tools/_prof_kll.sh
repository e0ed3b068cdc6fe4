cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pk_$1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk_$1 -- python3 tools/bench_kll.py --steps 2 > gpurun_out/pk_$1.log 2>&1
grep "^rows" gpurun_out/pk_$1.log
python3 tools/kstats.py gpurun_out/pk_$1
