cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/dev/fusetime.py 20 2>&1 | grep -E "round|fused|paired"
