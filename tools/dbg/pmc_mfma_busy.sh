# MFMA-busy / VALU / wait counters of the contraction kernels on four pipeline shapes, co-running plan family -> gpurun_out/r5pmc5/busy.txt
cd /tmp && export TMPDIR=/tmp
export GMD_ONE_FAMILY=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc5; mkdir -p $O; : > $O/busy.txt
pmc() { n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY' >> $O/busy.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("gemm_", "conv_patch", "ff_fused")) and "splitk" not in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:44], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn, sep="\t")
PY
}
C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
C2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
for c in "$C1" "$C2"; do
  pmc "gemm 32768 320 320 res" "$c" python3 $R/tools/one_gemm.py 32768 320 320 res
  pmc "gemm 8192 640 640 res" "$c" python3 $R/tools/one_gemm.py 8192 640 640 res
  pmc "gemm 8192 5120 640 geglu" "$c" python3 $R/tools/one_gemm.py 8192 5120 640 geglu
  pmc "conv 8 64 64 320 320" "$c" python3 $R/tools/one_conv.py 8 64 64 320 320
done
cat $O/busy.txt
