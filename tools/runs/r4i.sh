cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
timeout -k 10 600 python3 $R/tools/check_ring.py 128,160,244,1 64,160,244,1 128,128,244,1 64,128,244,1 > $O/check_lc.txt 2>&1
cat $O/check_lc.txt
