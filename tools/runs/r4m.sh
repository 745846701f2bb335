cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
for c in 0 64 128 320; do
  echo "== GMD_CONV_CBLK=$c" >> $O/cblk.txt
  GMD_TUNING=1 GMD_CONV_CBLK=$c timeout -k 10 300 python3 $R/tools/bench_gemm.py --conv-only >> $O/cblk.txt 2>&1
done
cat $O/cblk.txt
