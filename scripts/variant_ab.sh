for f in build_var/lib_*.so default; do
  if [ $f = default ]; then unset BOURSE_AMD_LIBRARY; else export BOURSE_AMD_LIBRARY=$PWD/$f; fi
  for b in 65536 8192; do
  python bench.py --no-cpu-baseline --books $b --repeats 2 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', $b, round(j['value']/1e6,1), round(j['runs']['median']/1e6,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in j['roofline']['kernels'].items()})"
  done
done
