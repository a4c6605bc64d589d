# Do the HIP-event kernel durations of bench.py agree with a kernel trace of the same run?  GPU box.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kt_check; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for B in 65536 8192; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$B -o kt -f csv -- python3 $R/bench.py --books $B --steps 200 --warmup 50 --no-cpu-baseline --preheat-steps 0 --repeats 0 > $OUT/bench_$B.json 2> $OUT/kt_$B.err
  head -3 $OUT/kt_$B/kt_kernel_stats.csv | cut -c1-120
  python3 -c "
import json
d=json.loads(open('$OUT/bench_$B.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench.py HIP events:', round(d['value']/1e6,1), 'M', {k:(round(v['avg_launch_ms']*1e3,1), v['launches']) for k,v in r['kernels'].items()}, 'standalone', r['standalone']['avg_launch_ms'], r['standalone']['launches'])
"
done
