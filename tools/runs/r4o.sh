cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4o
timeout -k 10 600 python3 -m pytest tests/test_pp_gpu.py -x -q > gpurun_out/r4o/test_pp.txt 2>&1; tail -25 gpurun_out/r4o/test_pp.txt
