# Round-5 evidence (run on the GPU box through gpurun): the default bench line, its rocprofv3 kernel trace folded per kind / shape,
# the library yardstick, the shape census of one loop iteration under the co-running plan family, the two-stream timeline.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${EVID_DIR:-r5z}; mkdir -p $O
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
rm -rf /tmp/prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/prof.err
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $S $O/bench_kernel_stats.csv
python3 $R/tools/rocprof_kinds.py $T > $O/bench_kinds.txt
python3 $R/tools/rocprof_shapes.py $T 2 > $O/bench_shapes.txt
python3 $R/tools/gemm_shape_census.py > $O/gemm_shape_census.jsonl 2> $O/census.err
python3 $R/tools/vs_library_gemm.py > $O/vs_library.txt 2>&1
python3 $R/tools/timeline.py > $O/timeline.txt 2>&1
head -12 $O/bench_kinds.txt; cat $O/vs_library.txt; tail -16 $O/timeline.txt; cat $O/bench_line.json
