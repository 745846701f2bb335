cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
timeout -k 10 900 python3 $R/tools/sweep_pp.py > $O/sweep_pp.txt 2>&1
cat $O/sweep_pp.txt
