# Side lines of bench.py at the current build (one box): value, ms per step, outputs finite -> gpurun_out/$1/side_lines.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-side}; mkdir -p $O; : > $O/side_lines.txt
run() { python3 $R/bench.py --no-cpu-baseline --no-drift --no-tolerance-path --no-kernel-timing "$@" 2>/dev/null | python3 -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
print(' '.join(sys.argv[1:]) or '(default)', d.get('value'), d.get('ms_per_step'), d.get('outputs_finite'))" "$@" >> $O/side_lines.txt; }
run
run --batch 8
run --dtype f16
run --scheduler ddpm
run --scheduler dpm++ --inference-steps 75
run --res 1024 --batch 8 --steps 1 --warmup 1
run --no-overlap
cat $O/side_lines.txt
