set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
python3 $R/tools/vs_library_gemm.py > $O/vs_library.txt 2>&1
rm -rf /tmp/prof; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 $R/tools/vs_library_gemm.py > $O/vs_library_profiled.txt 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
head -1 $T > $O/trace_header.txt
python3 $R/tools/trace_kernels.py $T Cijk,gemm_,conv,igemm,Conv,naive > $O/lib_kernels.txt 2>&1
cat $O/vs_library.txt
