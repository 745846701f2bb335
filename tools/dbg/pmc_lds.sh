cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; : > $O/lds.txt
pmc() { n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY' >> $O/lds.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("gemm_", "conv_patch")) and "splitk" not in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:50], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn, sep="\t")
PY
}
for mode in 2 1 0; do
  export GMD_TUNING=1 GMD_CONV_PATCH=$mode
  pmc "conv 8 64 64 320 320 patch=$mode" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" python3 $R/tools/one_conv.py 8 64 64 320 320
  pmc "conv 8 32 32 640 640 patch=$mode" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" python3 $R/tools/one_conv.py 8 32 32 640 640
done
unset GMD_CONV_PATCH
pmc "gemm 8192 640 2560" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" python3 $R/tools/one_gemm.py 8192 640 2560 res
cat $O/lds.txt
