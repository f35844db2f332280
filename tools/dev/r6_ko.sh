cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 0 1 2 3 4 8 16 24 25 27 31 7 ; do UNO_K3A_STAGGER=$s python tools/dev/k3a_time.py 2>&1 | grep "446"; done
