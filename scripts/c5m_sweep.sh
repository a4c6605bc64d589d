# C5 as written (256 momentum + 256 noise agents) vs. batch size and pipeline (GPU box): where the lane-per-book
# members' update overtakes the fused kernel (bk_run's auto threshold).
for b in 512 1024 2048 3072 4096 6144 8192 32768 65536; do
  for p in fused split; do
    python bench.py --workload C5M --books $b --pipeline $p --steps 30 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('B=$b $p', round(d['value']/1e6,2),'M', {n:round(v['avg_launch_ms'],3) for n,v in k.items()})"
  done
done
