R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for P in 3 4 5 6 8; do echo -n "C3 parts $P: "; BOURSE_AMD_SPLIT_PARTS=$P BOURSE_AMD_MIN_PART=2048 run; done
for S in 0 15 30 45 60 90; do echo -n "C3 4 parts stagger $S us: "; BOURSE_AMD_STAGGER_US=$S run; done
for S in 15 30 60; do echo -n "C3 4 parts stagger $S us, 20/5: "; BOURSE_AMD_STAGGER_US=$S run --steps 20 --warmup 5; done
for B in 20000 24576 28000 32768; do for PIPE in split wave_split; do echo -n "books $B $PIPE: "; run --books $B --pipeline $PIPE; done; done
