cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4x
timeout -k 10 600 python3 -m pytest tests/test_pp_gpu.py tests/test_split_gpu.py -q -x > gpurun_out/r4x/tests.txt 2>&1; tail -5 gpurun_out/r4x/tests.txt | cut -c1-250
cd /tmp; export TMPDIR=/tmp
for sh in "8192 5120 640 geglu" "2048 10240 1280 geglu"; do
  python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $sh
  rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $sh > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$sh" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_" in r["Kernel_Name"]]
print("FETCH_SIZE KB", sys.argv[2], rows[-1]["Counter_Value"], rows[-1]["Kernel_Name"][:60])
PY
done
