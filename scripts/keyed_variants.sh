#!/bin/bash
# The RandomAgents parity tests on library variants of the keyed event loop (built into build_var/ by
# bourse_amd/_build.py build(out=..., defines=...)): BOURSE_AMD_KEY_SEQ_BITS=8 / 24 move the key's field split so that
# ordinary configurations leave the arrival window (8) or the price window (24) and steps alternate between the keyed
# loop and the fallback; BOURSE_AMD_KEYED_EVENTS=0 is the fallback alone.  GPU box.  Every leg runs under `timeout`.
for f in ${@:-build_var/lib_*.so}; do
  echo "== $f"
  BOURSE_AMD_LIBRARY=$PWD/$f timeout 400 python -m pytest tests/test_gpu_parity.py -q -x --timeout 120 -k "random or fuzz or keyed or full or c2 or c3 or c5 or wave or split" 2>&1 | tail -4
  BOURSE_AMD_LIBRARY=$PWD/$f FUZZ_LO=${FUZZ_LO:-20000} FUZZ_HI=${FUZZ_HI:-20600} timeout 400 python scripts/fuzz_random.py 2>&1 | tail -3
done
