cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
for c in 0 1; do
  echo "== GMD_CONV_PATCH=$c" >> $O/conv.txt
  GMD_CONV_PATCH=$c timeout -k 10 300 python3 $R/tools/bench_gemm.py --conv-only >> $O/conv.txt 2>&1
done
cat $O/conv.txt
timeout -k 10 900 python3 $R/tools/ab_bench.py --rounds 2 --steps 4 GMD_PP=0 GMD_CONV_PATCH=0 GMD_CONV_PATCH=1 > $O/ab.txt 2>&1
tail -3 $O/ab.txt
