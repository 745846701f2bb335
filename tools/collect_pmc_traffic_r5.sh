# Round 5: launch-weighted HBM-side traffic of the gemm_nt kind under the CO-RUNNING plan family (GMD_ONE_FAMILY=1: what the shipped
# two-stream pipeline launches -- 256-row tiles, K slices reduced inside the kernel): FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes
# (--kernel-trace only beside --pmc) for the gemm_nt shapes that carry the most time of one loop iteration (tools/gemm_shape_census.py),
# the level-0 fused feed-forward included, plus three convolution shapes.  Run through gpurun; tools/pmc_traffic_json.py folds the result.
cd /tmp && export TMPDIR=/tmp
export GMD_ONE_FAMILY=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc; mkdir -p $O; : > $O/pmc_raw.txt
pmc() { # tag counter cmd...
  n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY' >> $O/pmc_raw.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("gemm_", "conv_patch", "ff_fused")) and "splitk" not in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:70], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn, sep="\t")
PY
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc "ff_geglu_fused 32768" $c python3 $R/tools/one_ff.py 32768
  pmc "gemm 8192 5120 640 geglu" $c python3 $R/tools/one_gemm.py 8192 5120 640 geglu
  pmc "gemm 8192 640 640 res" $c python3 $R/tools/one_gemm.py 8192 640 640 res
  pmc "gemm 2048 10240 1280 geglu" $c python3 $R/tools/one_gemm.py 2048 10240 1280 geglu
  pmc "gemm 32768 320 320 res" $c python3 $R/tools/one_gemm.py 32768 320 320 res
  pmc "gemm 2048 1280 1280 res" $c python3 $R/tools/one_gemm.py 2048 1280 1280 res
  pmc "gemm 16384 320 320 res" $c python3 $R/tools/one_gemm.py 16384 320 320 res
  pmc "gemm 1024 1280 1280 res" $c python3 $R/tools/one_gemm.py 1024 1280 1280 res
  pmc "gemm 4096 640 640 res" $c python3 $R/tools/one_gemm.py 4096 640 640 res
  pmc "gemm 2048 1280 5120 res" $c python3 $R/tools/one_gemm.py 2048 1280 5120 res
  pmc "gemm 8192 640 2560 res" $c python3 $R/tools/one_gemm.py 8192 640 2560 res
  pmc "gemm 1024 10240 1280 geglu" $c python3 $R/tools/one_gemm.py 1024 10240 1280 geglu
  pmc "ff_geglu_fused 16384" $c python3 $R/tools/one_ff.py 16384
  pmc "gemm 4096 5120 640 geglu" $c python3 $R/tools/one_gemm.py 4096 5120 640 geglu
  pmc "conv 8 64 64 640 320" $c python3 $R/tools/one_conv.py 8 64 64 640 320
  pmc "conv 8 64 64 320 320" $c python3 $R/tools/one_conv.py 8 64 64 320 320
  pmc "conv 8 32 32 640 640" $c python3 $R/tools/one_conv.py 8 32 32 640 640
done
pmc "conv 8 64 64 640 320" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python3 $R/tools/one_conv.py 8 64 64 640 320
pmc "conv 8 32 32 640 640" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python3 $R/tools/one_conv.py 8 32 32 640 640
pmc "gemm 8192 640 640 res" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" python3 $R/tools/one_gemm.py 8192 640 640 res
cat $O/pmc_raw.txt
