# rocprofv3 kernel stats for the other workloads (GPU box): bash scripts/profile_extra.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/c5m -o k -f csv -- python3 $R/bench.py --workload C5M --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_C5M_under_rocprof.json 2> $OUT/c5m.err
rocprofv3 --kernel-trace --stats -d $OUT/mkt -o k -f csv -- python3 $R/scripts/market_rate.py > $OUT/market_rate.txt 2> $OUT/mkt.err
rocprofv3 --kernel-trace --stats -d $OUT/c2 -o k -f csv -- python3 $R/bench.py --workload C2 --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_C2_under_rocprof.json 2> $OUT/c2.err
head -5 $OUT/c5m/k_kernel_stats.csv; head -4 $OUT/mkt/k_kernel_stats.csv; head -3 $OUT/c2/k_kernel_stats.csv; cat $OUT/market_rate.txt
