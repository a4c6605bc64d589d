# Round-6 profile (GPU box): bench lines of every configuration, rocprofv3 kernel stats of the headline, of the C4 shard, of C5 and of
# the external-agents stream, the PMC record of every configuration (scripts/pmc_all.sh), the ingress rates.
#   bash scripts/profile_round6.sh [tag]   -> gpurun_out/prof_<tag>/ + gpurun_out/pmc_<tag>/   (scripts/collect_round4.py <tag> <round> folds them into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MIX="--modify-frac 0.05 --market-frac 0.02"
python3 $R/bench.py --steps 200 --warmup 50 > $OUT/bench_C3.json 2> $OUT/bench_C3.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C3_driver_args.json 2>> $OUT/bench_C3.err
python3 $R/bench.py --books 8192 --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_C4_shard_8192.json 2> $OUT/bench_C4.err
for N in 32768 16384; do python3 $R/bench.py --books $N --steps 100 --warmup 50 --no-cpu-baseline > $OUT/bench_C4_shard_$N.json 2>> $OUT/bench_C4.err; done
for W in C2 C5 C5M; do python3 $R/bench.py --workload $W --steps 100 --warmup 30 > $OUT/bench_$W.json 2> $OUT/bench_$W.err; done
# external agents: the clean stream (round 5's) and the mixed one (5 % modifications, 2 % market orders), both sizes
python3 $R/bench.py --workload INGRESS --steps 24 --warmup 6 > $OUT/bench_INGRESS.json 2> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --steps 24 --warmup 6 $MIX > $OUT/bench_INGRESS_mixed.json 2>> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --books 65536 --steps 24 --warmup 6 --no-cpu-baseline > $OUT/bench_INGRESS_65536.json 2>> $OUT/bench_INGRESS.err
python3 $R/bench.py --workload INGRESS --books 65536 --steps 24 --warmup 6 --no-cpu-baseline $MIX > $OUT/bench_INGRESS_mixed_65536.json 2>> $OUT/bench_INGRESS.err
# the multi-rank code path on the one GPU there is (collectives over gloo; NOT an 8-GPU measurement)
python3 $R/bench.py --gpus 8 --dry-ranks --steps 20 --warmup 5 --no-cpu-baseline --repeats 1 > $OUT/bench_dry_ranks_8.json 2> $OUT/bench_dry_ranks_8.err
# kernel stats (no pre-heat: its launches would be averaged into the same kernel names)
kt() { d=$1; shift; rocprofv3 --kernel-trace --stats -d $OUT/$d -o kt -f csv -- python3 $R/bench.py "$@" --no-cpu-baseline --preheat-steps 0 > $OUT/bench_${d#kt_}_under_rocprof.json 2> $OUT/$d.err
       f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$d.csv; rm -rf $OUT/$d; }
kt kt_C3 --steps 200 --warmup 50
kt kt_C4 --books 8192 --steps 200 --warmup 50
kt kt_C5 --workload C5 --steps 60 --warmup 20
kt kt_C5M --workload C5M --steps 60 --warmup 20
kt kt_C2 --workload C2 --steps 100 --warmup 30
kt kt_INGRESS --workload INGRESS --steps 24 --warmup 6
kt kt_INGRESS_mixed --workload INGRESS --steps 24 --warmup 6 $MIX
python3 $R/scripts/device_ingress_rate.py 8192 > $OUT/device_ingress_rate.txt 2>&1
python3 $R/scripts/device_ingress_rate.py 65536 >> $OUT/device_ingress_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 8192 > $OUT/host_driven_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 65536 >> $OUT/host_driven_rate.txt 2>&1
# three more driver-argument runs (fresh processes): the first region against the median of its five
for i in 2 3 4; do python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C3_driver_args_$i.json 2>> $OUT/bench_C3.err; done
bash $R/scripts/pmc_all.sh $TAG C3:65536 C3:32768 C3:16384 C3:8192 C2:4096 C5:8192 C5M:8192 INGRESS:8192 INGRESS:65536 INGRESSMIX:8192 INGRESSMIX:65536 > $OUT/pmc_all.log 2>&1
tail -n 3 $OUT/*.err | tail -n 40; cat $OUT/device_ingress_rate.txt $OUT/host_driven_rate.txt | grep -v amdgpu.ids
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); r=d['roofline']
    print('%-38s %8.1f M  frac %.3f node %.3f  traffic %s' % ('$(basename $f)', d['value']/1e6, r['frac'], r.get('frac_node',0), r['traffic']))
except Exception as e: print('$f', e)
"; done
