cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcM/p$i -o out -f csv -- python3 $R/bench.py --workload C5M --steps 20 --warmup 10 --steps-per-launch 10 --no-cpu-baseline --profile-every 0 > /dev/null 2>&1
done
python3 $R/bench.py --workload C5M --steps 50 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
