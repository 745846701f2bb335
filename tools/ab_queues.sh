#!/bin/bash
# HIP hardware-queue count on the headline (hip_ops.side_stream): the default against GPU_MAX_HW_QUEUES=2 / 8, twice, on one box.
# Round-3 result (bf16, ms per batch): default 850.9 / 851.7, 2 queues 853.6 / 852.3, 8 queues 1158.5 / 1154.3; a high-priority GM
# stream (torch.cuda.Stream(priority=-1), since removed) 1393.1 / 1393.5.
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-tolerance-path --no-drift"
for r in 1 2; do
for v in "base" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8"; do
  if [ "$v" = base ]; then $B 2>/dev/null > gpurun_out/ab_q.json; else env $v $B 2>/dev/null > gpurun_out/ab_q.json; fi
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_q.json)"
done; done
