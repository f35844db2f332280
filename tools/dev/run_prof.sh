# everything profiles/ holds for a round, on one box: blocks + step profiles (tools/profile_round.sh), NS-2D / NS-3D kernel
# statistics, the per-launch table of one step, the same-process A/B of the round's switches, the bench line with the CPU baseline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r04}
bash tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1
bash tools/prof_ns_kernels.sh 2d > gpurun_out/ns2d_stats.txt 2>&1
cp $(ls gpurun_out/ns2d_prof/*/*kernel_stats.csv | head -1) profiles/${tag}_ns2d_kernel_stats.csv
bash tools/prof_ns_kernels.sh 3d 32 > gpurun_out/ns3d_stats.txt 2>&1
cp $(ls gpurun_out/ns3d_prof/*/*kernel_stats.csv | head -1) profiles/${tag}_ns3d_w32_kernel_stats.csv
rm -rf gpurun_out/ns2d_prof gpurun_out/ns3d_prof
python tools/dev/steplaunches.py > profiles/${tag}_step_launches.txt 2>&1
python tools/dev/fusetime.py 20 > profiles/${tag}_fuse_ab.txt 2>&1
python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
grep '^{' gpurun_out/bench_${tag}.json | tail -1 > profiles/${tag}_bench_n1.json
mkdir -p gpurun_out/profiles_$tag; cp profiles/${tag}_* profiles/block_rocprof.json profiles/block_traffic.json profiles/hbm_traffic.json gpurun_out/profiles_$tag/ 2>/dev/null
find gpurun_out/prof_$tag -name "*.db" -delete; find gpurun_out/prof_$tag -name "*_agent_info.csv" -delete
du -sh gpurun_out/prof_$tag; ls -la profiles/ | grep $tag; head -c 400 profiles/${tag}_bench_n1.json
