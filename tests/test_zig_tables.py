"""The regenerated 256-layer ziggurat tables of the standard normal (bourse_amd/csrc/zig_norm_tables.inc =
oracle/zig_norm_tables.inc, tools/gen_zig_tables.py) against INDEPENDENT facts - rand_distr 0.4.3's literal tables are
not in the reference tree, so this is what can be pinned here: the defining equal-area property of every
layer, the base layer + tail area, F = exp(-X^2/2), and the leading constants of the classic Marsaglia-Tsang /
rand_distr table (ZIG_NORM_R = 3.654152885361008796, X[0] = 3.910757959537090045, X[2] = 3.449278298560964462)."""
import math
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 3.654152885361008796
V = 0.00492867323399


def _tables():
    txt = open(os.path.join(ROOT, "bourse_amd", "csrc", "zig_norm_tables.inc")).read()
    out = {}
    for name in ("ZIG_NORM_X", "ZIG_NORM_F"):
        body = re.search(r"ZIG_TABLE_BEGIN\(%s\)(.*?)ZIG_TABLE_END" % name, txt, re.S).group(1)
        out[name] = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+p[-+]?\d+", body)]
    return out["ZIG_NORM_X"], out["ZIG_NORM_F"]


def test_ziggurat_tables_satisfy_their_defining_properties():
    x, f = _tables()
    assert len(x) == 257 and len(f) == 257
    assert x[1] == R and x[256] == 0.0 and f[256] == 1.0
    assert all(a > b for a, b in zip(x, x[1:]))                        # strictly decreasing layer edges
    for i in range(257):
        assert abs(f[i] - math.exp(-x[i] * x[i] / 2.0)) <= 2 * math.ulp(f[i])
    # every rectangle layer i = 1..255 has area V: x[i] * (f(x[i+1]) - f(x[i]))
    for i in range(1, 256):
        area = x[i] * (f[i + 1] - f[i])
        assert abs(area - V) < 1e-11, (i, area)  # V is a 12-digit literal in the recipe; the last layer absorbs its rounding
    # the base layer: the rectangle R * f(R) plus the tail beyond R; x[0] = V / f(R) is its virtual edge
    tail = math.sqrt(math.pi / 2.0) * math.erfc(R / math.sqrt(2.0))
    assert abs(R * f[1] + tail - V) < 1e-13
    assert abs(x[0] - V / f[1]) < 1e-12
    # leading constants of the published table
    assert "%.18f" % x[0] == "3.910757959537090045" and "%.18f" % x[2] == "3.449278298560964462"
    assert abs(x[3] - 3.320244733839166074) < 1e-15 and abs(x[4] - 3.224575052047029100) < 1e-15
