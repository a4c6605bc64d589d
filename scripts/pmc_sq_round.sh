# SQ instruction counters of the split kernels at C3 (one part: a launch covers all 65 536 books).  GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export BOURSE_AMD_SPLIT_PARTS=1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_WAVES -d $R/gpurun_out/pmc_sq -o p -f csv -- python3 $R/bench.py --steps 20 --warmup 20 --steps-per-launch 20 --no-cpu-baseline --profile-every 0 > /dev/null 2> $R/gpurun_out/pmc_sq.err
ls $R/gpurun_out/pmc_sq
