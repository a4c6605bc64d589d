for b in 2048 8192 32768 65536; do python bench.py --workload C5M --books $b --pipeline split --steps 30 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('B=$b', round(d['value']/1e6,2),'M', {n:round(v['avg_launch_ms'],3) for n,v in k.items()})"; done
