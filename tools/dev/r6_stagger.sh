cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 0 1 2 3 4 6 8 12; do UNO_K3A_STAGGER=$s python tools/dev/k3a_time.py 2>&1 | grep stagger; done
