cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_channel_mix.py tests/test_hip_bf16_block.py tests/test_hip_mixed.py tests/test_hip_c5.py tests/test_hip_redzone.py tests/test_hip_headline_parity.py -q > gpurun_out/t4.txt 2>&1
tail -8 gpurun_out/t4.txt
timeout 300 python tools/dev/steplaunches.py > gpurun_out/steplaunches_k9s2.txt 2>&1
grep -h 'wgrad\|^sum' gpurun_out/steplaunches_k9s2.txt
timeout 300 python tools/dev/c5_ab.py 2>&1 | grep C5
UNO_CW_SPLIT_OFF=1 timeout 300 python tools/dev/c5_ab.py 2>&1 | grep C5
