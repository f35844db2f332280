cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_layout.py tests/test_hip_redzone.py tests/test_hip_blocks.py -x -q > gpurun_out/t_layout.txt 2>&1
tail -5 gpurun_out/t_layout.txt
timeout 300 python tools/dev/refstyle_time.py prof > gpurun_out/refstyle_time.txt 2>&1
grep -v '^  \|^--' gpurun_out/refstyle_time.txt | head -20
