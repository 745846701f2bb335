cd /tmp
for dt in f32 bf16 f16; do DTYPE=$dt REPS=25 STEPS=4 timeout -k 10 500 python3 $GRAFT_REPO_ROOT/tools/stress_pipeline_determinism.py 2>&1 | grep -v Warn | tail -3 | cut -c1-300; done
