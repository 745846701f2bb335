cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
timeout -k 10 300 python3 $R/tools/pp_diag.py 256,160,283,1 > $O/pp_diag.txt 2>&1
cat $O/pp_diag.txt
