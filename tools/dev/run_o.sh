cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3o; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_channel_mix.py tests/test_hip_blocks.py tests/test_hip_bf16_block.py tests/test_hip_mixed.py tests/test_hip_c5.py tests/test_hip_zz_dist.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
python tools/dev/steplaunches.py > $o/launches.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-extras > $o/bench.log 2>&1
tail -3 $o/tests.log; grep '^{' $o/bench.log | head -c 250; echo; sed -n 28,36p $o/launches.txt; tail -1 $o/launches.txt
