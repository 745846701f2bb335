cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
timeout -k 10 600 python3 $R/tools/timeline.py > $O/timeline.txt 2>&1
cat $O/timeline.txt
