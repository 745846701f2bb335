cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_pp_gpu.py -x -q > $O/test_pp.txt 2>&1
tail -15 $O/test_pp.txt
cd /tmp && timeout -k 10 900 python3 $R/tools/ab_bench.py --rounds 3 --steps 5 GMD_PP=0 GMD_PP=1 > $O/ab_pp.txt 2>&1
cat $O/ab_pp.txt
