cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s2
timeout -k 10 900 python3 -m pytest tests/test_split_gpu.py tests/test_northstar_gpu.py tests/test_models_gpu.py -q -x > gpurun_out/s2/tests.txt 2>&1; tail -5 gpurun_out/s2/tests.txt | cut -c1-250
cd /tmp
for v in 0 1; do
GMD_SPLIT_LC=$v timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-drift --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/s2/bench_$v.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/s2/bench_$v.json')); print('GMD_SPLIT_LC=$v', d['value'], d['tolerance_path']['images_per_s'], d['tolerance_path']['ms_each'])"
done
