cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O
timeout -k 10 1100 python3 $R/tools/ab_bench.py --rounds 2 --steps 4 GMD_PP=b GMD_PP=b1 GMD_PP=b2 GMD_PP=b3 > $O/ab2.txt 2>&1
tail -4 $O/ab2.txt
