R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/prof_r05; mkdir -p $O
python3 $R/scripts/device_ingress_rate.py 8192 > $O/device_ingress_rate.txt 2>&1
python3 $R/scripts/device_ingress_rate.py 65536 >> $O/device_ingress_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 8192 > $O/host_driven_rate.txt 2>&1
python3 $R/scripts/host_driven_rate.py 65536 >> $O/host_driven_rate.txt 2>&1
bash scripts/pmc_events.sh 8192 > /dev/null 2>&1; cat gpurun_out/pmc_events/summary_8192.txt
grep -v amdgpu.ids $O/device_ingress_rate.txt $O/host_driven_rate.txt | cut -c1-260
