cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O
timeout -k 10 800 python3 $R/tools/sweep_lc.py > $O/sweep_lc.txt 2>&1
cat $O/sweep_lc.txt
