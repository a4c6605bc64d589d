# Everything a round's hand-in is judged on, in one go on a GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/round_check.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
echo "== parity suite (through the C ABI, bit-exact vs the oracle)"; python -m pytest tests -m gpu -q 2>&1 | tail -2
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (default)"; python bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(f\"{d['value']/1e6:.1f} M {d['unit']} | {r['kernel']} {r['avg_launch_ms']*1e3:.1f} us/launch, {r['achieved']:.0f} GB/s = {r['frac']*100:.1f} % of HBM, traffic {r['traffic']/1e6 if r['traffic'] else None} MB/launch | scalar port {r['issue']['scalar_port_frac']*100:.0f} % vector port {r['issue']['vector_port_frac']*100:.0f} % | cpu {d['cpu_baseline']['value']/1e6:.2f} M on {d['cpu_baseline']['cores']} threads\")"
echo "== long-run and full-size parity"; python scripts/soak.py 2000 | tail -1; python scripts/soak_agents.py | tail -2; python scripts/c5m_fullsize_parity.py | tail -1; python scripts/c3_fullsize_parity.py 200 | tail -1
