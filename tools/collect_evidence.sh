set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; mkdir -p $O
# 1. the default bench command, un-profiled, then under the kernel trace
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
rm -rf /tmp/prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/prof.err
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $S $O/bench_kernel_stats.csv
python3 $R/tools/rocprof_kinds.py $T > $O/bench_kinds.txt
python3 $R/tools/rocprof_shapes.py $T 2 > $O/bench_shapes.txt
# 2. PMC traffic passes (one counter per pass) on single-op drivers
pmc() { # name counter cmd...
  n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" "$c" <<'PY' >> $O/pmc.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if ("gemm" in r["Kernel_Name"] or "attn" in r["Kernel_Name"] or "ff_fused" in r["Kernel_Name"]) and "splitk" not in r["Kernel_Name"] and "split_weights" not in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"][:80], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn)
PY
}
: > $O/pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  pmc "gemm_bf16_M2048_N1280_K1280_res" $c python3 $R/tools/one_gemm.py 2048 1280 1280 res
  GMD_ONE_DTYPE=f32 pmc "gemm_f32split_M32768_N320_K1280_res" $c python3 $R/tools/one_gemm.py 32768 320 1280 res
  GMD_ONE_DTYPE=f32 pmc "conv_f32split_8x64x64_640_320" $c python3 $R/tools/one_conv.py 8 64 64 640 320
  pmc "gemm_bf16_M32768_N2560_K320_geglu" $c python3 $R/tools/one_gemm.py 32768 2560 320 geglu
done
GMD_ONE_DTYPE=f32 pmc "conv_f32split_8x64x64_640_320" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python3 $R/tools/one_conv.py 8 64 64 640 320
pmc "conv_bf16_8x64x64_640_320" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python3 $R/tools/one_conv.py 8 64 64 640 320
GMD_ONE_DTYPE=f32 pmc "conv_f32split_8x64x64_640_320" "GRBM_GUI_ACTIVE" python3 $R/tools/one_conv.py 8 64 64 640 320
# 3. timings of the same single ops (un-profiled)
python3 $R/tools/one_gemm.py 2048 1280 1280 res > $O/one_ops.txt 2>&1
GMD_ONE_DTYPE=f32 python3 $R/tools/one_gemm.py 32768 320 1280 res >> $O/one_ops.txt 2>&1
# 4. 50-step drift
timeout -k 10 400 python3 $R/tools/drift_fullsize.py 50 > $O/drift50.txt 2>&1
cat $O/pmc.txt; tail -12 $O/drift50.txt; cat $O/bench_kinds.txt | head -12
