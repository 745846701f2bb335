cd /tmp
for dt in f32 bf16 f16; do DTYPE=$dt NREP=400 INNER=12 timeout -k 10 500 python3 $GRAFT_REPO_ROOT/tools/stress_colstats.py 2>&1 | grep -v Warn | tail -2 | cut -c1-300 || exit 1; done
