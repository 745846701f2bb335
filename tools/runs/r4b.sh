set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O
timeout -k 10 400 python3 $R/tools/check_ring.py 256,160,283,1 256,128,283,1 > $O/check_pp.txt 2>&1
cat $O/check_pp.txt
cd $R && timeout -k 10 900 python3 -m pytest tests/test_northstar_gpu.py -x -q -s > $O/northstar.txt 2>&1
tail -30 $O/northstar.txt
