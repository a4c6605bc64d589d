# k_step_batch books-per-wave (prefetch) x the lane kernel's VGPR claim, at C3 and 32 768 books.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for LIB in "" lib_fsm167.so lib_fsm127.so lib_fsm63.so; do
  for BPW in 1 2 3 4; do
    echo -n "C3 fsm-claim ${LIB:-231(in-tree)} bpw $BPW: "; BOURSE_AMD_STEP_BPW=$BPW BOURSE_AMD_LIBRARY=${LIB:+$R/build_variants/$LIB} run
  done
done
for BPW in 1 2; do echo -n "32768 in-tree bpw $BPW: "; BOURSE_AMD_STEP_BPW=$BPW run --books 32768; done
for BPW in 1 2; do echo -n "8192 in-tree bpw $BPW: "; BOURSE_AMD_STEP_BPW=$BPW run --books 8192; done
