set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench_20.json 2> gpurun_out/r2b/bench_20.err
python tools/tick_sequence.py > gpurun_out/r2b/tick_sequence.json 2> gpurun_out/r2b/tick_sequence.err
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r2b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
