# PMC counters of attn40_kernel on the level-0 self-attention (B=8 N=4096 d=40) and the 77-key cross-attention -> gpurun_out/r5pmc4/attn.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc4; mkdir -p $O; : > $O/attn.txt
pmc() { n=$1; c=$2; shift 2; rm -rf /tmp/pmc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY' >> $O/attn.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "attn" in r["Kernel_Name"]]
last = {}
for r in rows:
    last[r["Counter_Name"]] = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:40], float(r["Counter_Value"]))
for k, (kn, v) in last.items():
    print(sys.argv[2], k, v, kn, sep="\t")
PY
}
pmc "self B=8 N=4096 d=40" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python3 $R/tools/one_attn.py 8 4096 4096 8 40
pmc "self B=8 N=4096 d=40" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" python3 $R/tools/one_attn.py 8 4096 4096 8 40
pmc "self B=8 N=4096 d=40" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" python3 $R/tools/one_attn.py 8 4096 4096 8 40
cat $O/attn.txt
