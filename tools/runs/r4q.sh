cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4q
GMD_CONV_PATCH=2 timeout -k 10 600 python3 -m pytest tests/test_pp_gpu.py -x -q -k "conv_patch" > gpurun_out/r4q/test_cont.txt 2>&1; tail -5 gpurun_out/r4q/test_cont.txt
cd /tmp
for c in 0 1 2; do
  echo "== GMD_CONV_PATCH=$c" >> $GRAFT_REPO_ROOT/gpurun_out/r4q/conv.txt
  GMD_CONV_PATCH=$c timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py --conv-only 2>&1 | head -14 >> $GRAFT_REPO_ROOT/gpurun_out/r4q/conv.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r4q/conv.txt
