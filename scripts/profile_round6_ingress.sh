# the external-agents part of scripts/profile_round6.sh alone (after a change to k_step_events only): same outputs, same places
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
MIX="--modify-frac 0.05 --market-frac 0.02"
python3 $R/bench.py --workload INGRESS --steps 24 --warmup 6 > $OUT/bench_INGRESS.json 2> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --steps 24 --warmup 6 $MIX > $OUT/bench_INGRESS_mixed.json 2>> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --books 65536 --steps 24 --warmup 6 --no-cpu-baseline > $OUT/bench_INGRESS_65536.json 2>> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --books 65536 --steps 24 --warmup 6 --no-cpu-baseline $MIX > $OUT/bench_INGRESS_mixed_65536.json 2>> $OUT/bench_INGRESS.err
kt() { d=$1; shift; rocprofv3 --kernel-trace --stats -d $OUT/$d -o kt -f csv -- python3 $R/bench.py "$@" --no-cpu-baseline --preheat-steps 0 > $OUT/bench_${d#kt_}_under_rocprof.json 2> $OUT/$d.err
       f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$d.csv; rm -rf $OUT/$d; }
kt kt_INGRESS --workload INGRESS --steps 24 --warmup 6
kt kt_INGRESS_mixed --workload INGRESS --steps 24 --warmup 6 $MIX
python3 $R/scripts/device_ingress_rate.py 8192 > $OUT/device_ingress_rate.txt 2>&1
python3 $R/scripts/device_ingress_rate.py 65536 >> $OUT/device_ingress_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 8192 > $OUT/host_driven_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 65536 >> $OUT/host_driven_rate.txt 2>&1
bash $R/scripts/pmc_all.sh ${TAG}i INGRESS:8192 INGRESS:65536 INGRESSMIX:8192 INGRESSMIX:65536 > $OUT/pmc_ingress.log 2>&1
bash $R/scripts/pmc_step_events.sh 8192 > $OUT/pmc_step_events_8192.log 2>&1
for f in $OUT/bench_INGRESS*.json; do python3 -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); r=d['roofline']
print('%-40s %8.1f M  keyed %.4f  frac %.3f  launch %.1f us traffic %s' % ('$(basename $f)', d['value']/1e6, d['keyed_frac'], r['frac'], r['avg_launch_ms']*1e3, r['traffic']))
"; done
