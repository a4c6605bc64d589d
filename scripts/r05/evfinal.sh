R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
FUZZ_LO=3000 FUZZ_HI=4500 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/fuzz_keyed_events.txt
BOURSE_AMD_EV_WAVE_SHUFFLE_MIN=2 FUZZ_LO=4500 FUZZ_HI=5500 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/fuzz_keyed_events.txt
FUZZ_LO=57000 FUZZ_HI=57400 python3 scripts/fuzz_host.py 2>&1 | tail -4
FUZZ_LO=6000 FUZZ_HI=6400 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -2
for b in 8192 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids; done | tee $O/device_ingress_rate.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
