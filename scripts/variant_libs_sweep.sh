#!/bin/bash
# A/B of library variants built into build_var/ (bourse_amd/_build.py build(out=..., defines=...)) at the batch sizes given
# usage: variant_libs_sweep.sh books...   (default: 65536 32768 8192)
sizes=${@:-65536 32768 8192}
for f in build_var/lib_*.so; do
  export BOURSE_AMD_LIBRARY=$PWD/$f
  line="$f:"
  for b in $sizes; do
    v=$(python bench.py --no-cpu-baseline --books $b --repeats 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f (median %.1f)' % (d['value']/1e6, d['runs']['median']/1e6))")
    line="$line  $b: $v |"
  done
  echo "$line"
done
