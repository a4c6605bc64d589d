# rocprofv3 kernel stats of the INGRESS bench line (the kernel-trace durations beside the line's HIP-event ones)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_INGRESS -o kt -f csv -- python3 $R/bench.py --workload INGRESS --steps 24 --warmup 6 --no-cpu-baseline --preheat-steps 0 > $OUT/bench_INGRESS_under_rocprof.json 2> $OUT/kt_INGRESS.err
f=$(find $OUT/kt_INGRESS -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_kt_INGRESS.csv; head -8 $OUT/kernel_stats_kt_INGRESS.csv | cut -c1-200
find $OUT/kt_INGRESS -name "*.csv" -delete; find $OUT/kt_INGRESS -name "*.db" -delete
tail -c 600 $OUT/bench_INGRESS_under_rocprof.json | cut -c1-100
