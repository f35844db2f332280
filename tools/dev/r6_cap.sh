cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_capture.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_hip_mixed.py tests/test_harness_ns.py tests/test_hip_spectral3d.py -x -q 2>&1 | tail -3
