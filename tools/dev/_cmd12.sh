cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/dev/steplaunches.py > gpurun_out/steplaunches_final.txt 2>&1
tail -2 gpurun_out/steplaunches_final.txt
bash tools/dev/run_prof.sh r04 > gpurun_out/run_prof.log 2>&1
tail -3 gpurun_out/run_prof.log | cut -c1-400
