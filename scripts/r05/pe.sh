R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do for pe in 8 0 16 32; do echo -n "--profile-every $pe: "; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --profile-every $pe 2>/dev/null | python -c "
import json,sys,numpy as np
d=json.loads(sys.stdin.readline()); v=d['runs']['values']
print('%.1f first, repeats %s, first/median(repeats) %.3f, sampled launches %d' % (v[0]/1e6, [round(x/1e6,1) for x in v[1:]], v[0]/np.median(v[1:]), d['roofline']['launches']))"; done; done
