cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/dev/steplaunches.py > gpurun_out/steplaunches.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/refstyle_prof -o rs -- python tools/dev/refstyle_prof.py > gpurun_out/refstyle_prof.log 2>&1
f=$(ls gpurun_out/refstyle_prof/*kernel_stats.csv gpurun_out/refstyle_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
head -40 $f > gpurun_out/refstyle_kernel_stats.csv
find gpurun_out/refstyle_prof -name "*.db" -delete
tail -3 gpurun_out/steplaunches.txt
