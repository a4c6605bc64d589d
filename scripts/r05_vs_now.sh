# same-box comparison of round 5's tree (build_variants/r05tree: `git worktree add build_variants/r05tree e552d10`, library built
# in the container) with this tree on the lines both can print.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%7.1f M  (median %s)  %s' % (d['value']/1e6, round(d.get('runs',{}).get('median',0)/1e6,1), {k: round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()}))"; }
for rep in 1 2; do
for T in $R/build_variants/r05tree $R; do
  cd $T; echo "== $T"
  for A in "--workload INGRESS --steps 24 --warmup 6" "--workload INGRESS --books 65536 --steps 24 --warmup 6" "--steps 200 --warmup 50 --repeats 2" "--books 8192 --steps 200 --warmup 50 --repeats 2" "--workload C5 --steps 100 --warmup 30 --repeats 2" "--workload C5M --steps 100 --warmup 30 --repeats 2" "--workload C2 --steps 100 --warmup 30 --repeats 2"; do
    printf "  %-62s" "$A"; python3 bench.py $A --no-cpu-baseline 2>/dev/null | line
  done
done
done
