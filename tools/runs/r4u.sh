cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4u && timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -k "conv3x3_groupnorm" > gpurun_out/r4u/gpu_tests2.txt 2>&1; tail -4 gpurun_out/r4u/gpu_tests2.txt | cut -c1-300
bash tools/collect_evidence_r4.sh > gpurun_out/r4u/evidence.log 2>&1; tail -60 gpurun_out/r4u/evidence.log
