cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O
timeout -k 10 1100 python3 $R/tools/ab_bench.py --rounds 3 --steps 4 GMD_PP=0 GMD_PP=i GMD_CONV_PATCH=0 GMD_CONV_PATCH=1 GMD_CONV_PATCH=2 > $O/ab.txt 2>&1
tail -5 $O/ab.txt
