# FETCH_SIZE of the W-heavy GEGLU projections against the tile group (GMD_TILE_GROUP was a temporary measurement knob in pick_tile_group, csrc/gemm.hip: re-add it to re-run), co-running plan family -> profiles/r05_pmc_tile_group.txt
cd /tmp && export TMPDIR=/tmp
export GMD_ONE_FAMILY=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc3; mkdir -p $O; : > $O/pmc_raw.txt
pmc() { n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY' >> $O/pmc_raw.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_" in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:40], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn, sep="\t")
PY
}
for g in 1 2 4 8 16; do
  export GMD_TILE_GROUP=$g
  pmc "g=$g gemm 8192 5120 640 geglu" FETCH_SIZE python3 $R/tools/one_gemm.py 8192 5120 640 geglu
  pmc "g=$g gemm 4096 5120 640 geglu" FETCH_SIZE python3 $R/tools/one_gemm.py 4096 5120 640 geglu
  pmc "g=$g gemm 2048 10240 1280 geglu" FETCH_SIZE python3 $R/tools/one_gemm.py 2048 10240 1280 geglu
  pmc "g=$g gemm 1024 10240 1280 geglu" FETCH_SIZE python3 $R/tools/one_gemm.py 1024 10240 1280 geglu
done
cat $O/pmc_raw.txt
