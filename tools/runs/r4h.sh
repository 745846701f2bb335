cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4h
timeout -k 10 900 python3 -m pytest tests/test_split_gpu.py tests/test_distributed_gpu.py tests/test_northstar_gpu.py tests/test_pipeline_gpu.py -q -x -k "range or f32_mode or sd15_width or side_stream or northstar or 16bit or decode_tail" > gpurun_out/r4h/tests.txt 2>&1; tail -12 gpurun_out/r4h/tests.txt
cd /tmp && timeout -k 10 600 python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r4h/bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4h/bench_line.err; echo rc=$?; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r4h/bench_line.err
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4h/bench_line.json"))
print(d["value"], d["ms_per_step"], d["tolerance_path"]["images_per_s"] if d.get("tolerance_path") else None, d["latent_rms_vs_f32"])
print({k:(v.get("tflops") or v.get("gbps"), v["frac"], v["share"]) for k,v in d["kernels"].items()})
print(d["roofline"]["kind"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"))
PY
