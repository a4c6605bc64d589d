# Round profile: rocprofv3 kernel stats of the default bench command + HBM traffic counters (separate passes).
# Usage on the GPU box: bash scripts/profile_round.sh r01
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 50 > $OUT/bench_full_C3.json 2> $OUT/bench_full_C3.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -f csv -- python3 $R/bench.py --steps 50 --warmup 50 --no-cpu-baseline --profile-every 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -f csv -- python3 $R/bench.py --steps 50 --warmup 50 --no-cpu-baseline --profile-every 0 > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
cat $OUT/bench_full_C3.json
