cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
timeout -k 10 900 python3 $R/tools/ab_bench.py --rounds 2 --steps 4 --args=--no-overlap GMD_PP=0 GMD_PP=1 > $O/ab_nooverlap.txt 2>&1
tail -2 $O/ab_nooverlap.txt
