cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_layout.py tests/test_hip_channel_mix.py tests/test_hip_redzone.py tests/test_hip_blocks.py tests/test_hip_headline_parity.py -q > gpurun_out/t3.txt 2>&1
tail -15 gpurun_out/t3.txt
timeout 300 python tools/dev/steplaunches.py > gpurun_out/steplaunches_k9s.txt 2>&1
UNO_CW_SPLIT_OFF=1 timeout 300 python tools/dev/steplaunches.py > gpurun_out/steplaunches_k9s_off.txt 2>&1
grep -h 'wgrad\|^sum' gpurun_out/steplaunches_k9s.txt gpurun_out/steplaunches_k9s_off.txt
