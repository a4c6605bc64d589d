# the driver's 20-step region against the parts' start offset (GPU box): bash scripts/exp_stagger.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 4 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%8.1f M (median %8.1f)  runs %s' % (d['value']/1e6, d['runs']['median']/1e6, [round(x/1e6,1) for x in d['runs']['values']]))"; }
for s in default 0 10 20 30 45; do
  if [ $s = default ]; then unset BOURSE_AMD_STAGGER_US; else export BOURSE_AMD_STAGGER_US=$s; fi
  echo -n "stagger $s: "; run
  echo -n "stagger $s: "; run
done
