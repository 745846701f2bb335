cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
timeout -k 10 600 python3 $R/tools/sweep_pp.py --round3 > $O/sweep_vs_round3.txt 2>&1
cat $O/sweep_vs_round3.txt
timeout -k 10 900 python3 $R/tools/ab_bench.py --rounds 3 --steps 5 GMD_PP=0 GMD_PP=1 > $O/ab.txt 2>&1
tail -3 $O/ab.txt
