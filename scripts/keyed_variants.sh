#!/bin/bash
# The parity tests on library variants of the keyed event loop (built into build_var/ by bourse_amd/_build.py
# build(out=..., defines=...)): BOURSE_AMD_KEY_SEQ_BITS=8 / 24 narrow the key's arrival window (8) or its price window (24)
# so that ordinary configurations leave it and steps alternate between the keyed loop and the fallback;
# BOURSE_AMD_KEYED_EVENTS=0 is the fallback alone; BOURSE_AMD_ASM_EVENTS=0 the C++ keyed loop instead of the assembly.
# GPU box.  Every leg runs under `timeout`.
for f in ${@:-build_var/lib_*.so}; do
  echo "== $f"
  BOURSE_AMD_LIBRARY=$PWD/$f timeout 400 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "random or fuzz or keyed or full or c2 or c3 or c5 or wave or split" 2>&1 | tail -4
  BOURSE_AMD_LIBRARY=$PWD/$f FUZZ_LO=${FUZZ_LO:-20000} FUZZ_HI=${FUZZ_HI:-20600} timeout 400 python scripts/fuzz_random.py 2>&1 | tail -3
  # AgentSets with Noise / Momentum members: market orders through the keyed loop of the larger pools
  BOURSE_AMD_LIBRARY=$PWD/$f timeout 400 python scripts/fuzz_wave_members.py ${MEMBERS_LO:-740000} ${MEMBERS_N:-300} 2>&1 | tail -2
done
