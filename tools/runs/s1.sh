cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s1; mkdir -p $O
timeout -k 10 400 python3 $R/tools/ab_split_lc.py > $O/ab_split_lc.txt 2>&1; cat $O/ab_split_lc.txt
