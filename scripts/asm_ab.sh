# A/B of the event loop: the parity suite and the bench on the build with the COMPILED C++ event loop
# (-DBOURSE_AMD_ASM_EVENTS=0) next to the shipped one (hand-written gfx950 assembly, event_asm.hpp).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
python -c "from bourse_amd import _build; print(_build.build(out='/tmp/libbourse_amd_cxx.so', defines=['BOURSE_AMD_ASM_EVENTS=0']))"
echo "== parity suite, compiled C++ event loop"; BOURSE_AMD_LIBRARY=/tmp/libbourse_amd_cxx.so python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
echo "== parity suite, assembly event loop";     python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
for V in cxx asm; do
  L=""; [ $V = cxx ] && L=/tmp/libbourse_amd_cxx.so
  for B in 65536 8192; do
    BOURSE_AMD_LIBRARY=$L python bench.py --books $B --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print('$V', $B, 'books:', round(d['value'] / 1e6, 1), 'M book-steps/s (median', round(d['runs']['median'] / 1e6, 1), ');', {n: round(v['avg_launch_ms'] * 1e3, 1) for n, v in k.items()}, 'us per launch')"
  done
done
