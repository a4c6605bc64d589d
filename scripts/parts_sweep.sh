#!/bin/bash
# Sweep the number of book parts (streams) of the split pipeline, with the HIP runtime's default 4 hardware queues and with 8.
mkdir -p gpurun_out
for q in 4 8; do
  for p in 2 3 4 6 8; do
    echo "== GPU_MAX_HW_QUEUES=$q parts=$p"
    GPU_MAX_HW_QUEUES=$q BOURSE_AMD_SPLIT_PARTS=$p timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['ms_per_step'], d['roofline']['kernels']['k_agents_fsm']['avg_launch_ms'], d['roofline']['kernels']['k_step_batch']['avg_launch_ms'])"
  done
done
