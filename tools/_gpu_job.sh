cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shard or config5" 2>&1 | tail -5 > gpurun_out/r5d/pytest_shards.txt
timeout 1500 python -m pytest tests/test_gpu_host.py -x -q -m gpu -k "shard or rccl" 2>&1 | tail -5 >> gpurun_out/r5d/pytest_shards.txt
timeout 900 bash tools/shards_ab.sh 600 300 2 > gpurun_out/r5d/shards_ab.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu > gpurun_out/r5d/bench_n1.json 2>gpurun_out/r5d/bench_n1.err
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/trace -o tr -- python3 $GRAFT_REPO_ROOT/tools/shards_trace.py root > $GRAFT_REPO_ROOT/gpurun_out/r5d/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/shards_trace.py --summarise gpurun_out/r5d/trace > gpurun_out/r5d/timeline.txt 2>&1
rm -rf gpurun_out/r5d/trace
cat gpurun_out/r5d/pytest_shards.txt gpurun_out/r5d/shards_ab.txt gpurun_out/r5d/timeline.txt; cut -c1-300 gpurun_out/r5d/bench_n1.json
