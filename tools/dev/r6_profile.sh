cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06 > gpurun_out/profile_r06.log 2>&1
python tools/dev/steplaunches.py > profiles/r06_step_launches.txt 2>&1
python tools/dev/fusetime.py 20 > profiles/r06_fuse_ab.txt 2>&1
mkdir -p gpurun_out/r06prof; cp profiles/r06_* profiles/block_rocprof.json profiles/block_traffic.json profiles/hbm_traffic.json gpurun_out/r06prof/ 2>/dev/null
tail -3 profiles/r06_step_launches.txt; grep -E "fused|K3" profiles/r06_fuse_ab.txt | tail -6; ls gpurun_out/r06prof | head -40
