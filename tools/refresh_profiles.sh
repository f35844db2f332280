cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh gpurun_out/pmc_traffic
python tools/traffic_json.py gpurun_out/pmc_traffic gpurun_out/hbm_traffic.json gpurun_out/pmc_traffic.txt
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -1 gpurun_out/bench_n1.json | cut -c1-300
rm -rf gpurun_out/bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof -- python bench.py --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
ls gpurun_out/bench_prof/*/ | head
