cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4g && timeout -k 10 1100 python3 -m pytest tests -q -m gpu > gpurun_out/r4g/gpu_tests.txt 2>&1; tail -15 gpurun_out/r4g/gpu_tests.txt
