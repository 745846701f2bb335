cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O
timeout -k 10 300 python3 $R/tools/gemm_shape_census.py > $O/census.jsonl 2> $O/census.err; tail -3 $O/census.err; head -24 $O/census.jsonl; tail -1 $O/census.jsonl
