cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O
for v in 1 i 0; do
  GMD_PP=$v timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-drift --no-tolerance-path > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("GMD_PP="+sys.argv[2], d["value"], d["ms_per_step"], {k:(round(v.get("tflops") or v.get("gbps")), v["frac"], v["share"], v["avg_us"]) for k,v in d["kernels"].items() if k in ("gemm_nt","conv3x3","attention")})
PY
done
