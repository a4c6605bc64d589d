"""Static instruction counts of every kernel in the SHIPPED library, and of its loops - a baseline a rebuild is diffed against.

The step kernels are tuned at the ISA level (hand-written / generated assembly statements, scheduler flags, a register
claim, branch-free searches: DESIGN.md 2.3-2.5, 7) and bound by instruction issue, so "the compiler now emits 8 % more
instructions in the decode's window block" IS "the shard is 5 % slower" - but until round 5 only a slower BENCH would have
said so (VERDICT r4 Weak #10).  This tool disassembles the gfx950 code objects inside bourse_amd/csrc/libbourse_amd.so
(llvm-objcopy -> clang-offload-bundler -> llvm-objdump, a second: no recompile) and reports per kernel

  * instructions by issue class: salu, branch, valu, lds, vmem, smem, waitcnt / nop, total;
  * every LOOP, as the span of a backward branch (target .. branch, inclusive), sizes sorted - the hot loops of this code
    (the keyed event loop's match blocks, the decode's window / walk / chase loops, k_agents_fsm's draw loop) are the
    largest backward spans of their kernels;

and compares them with profiles/kernel_isa_baseline.json (tests/test_kernel_isa_baseline.py: totals and loop spans of
every kernel within 3 %).  After an INTENDED kernel change:  python tools/kernel_isa_counts.py --update

usage: kernel_isa_counts.py [--update | --check] [--lib path] [name-filter ...]
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = os.path.join(ROOT, "profiles", "kernel_isa_baseline.json")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TOL = 0.03


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def code_objects(lib, tmp):
    """The gfx950 code objects of a HIP shared library (one bundle per translation unit in its .hip_fatbin section)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([_tool("llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True,
                   capture_output=True)
    d = open(fat, "rb").read()
    pos, i = [], d.find(MAGIC)
    while i >= 0:
        pos.append(i)
        i = d.find(MAGIC, i + 1)
    out = []
    for k, p in enumerate(pos):
        b, o = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"co{k}.o")
        open(b, "wb").write(d[p:pos[k + 1] if k + 1 < len(pos) else len(d)])
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--input={b}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={o}"], check=True, capture_output=True)
        out.append(o)
    return out


def classify(op):
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
        return "branch"
    if op.startswith("s_waitcnt") or op in ("s_nop", "s_sleep", "s_endpgm", "s_barrier", "s_setprio", "s_sethalt", "s_trap",
                                            "s_icache_inv", "s_set_gpr_idx_off", "s_set_gpr_idx_on", "s_set_gpr_idx_mode"):
        return "other"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache", "s_atc")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


INS = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")
SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")


def kernels_of(obj):
    """{mangled kernel name: {"counts": {...}, "loops": [sizes, descending]}} of one code object."""
    txt = subprocess.run([_tool("llvm-objdump"), "-d", obj], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = SYM.match(line)
        if m:
            if m.group(2).startswith("_Z"):  # (the labels of the assembly statements are symbols too: they stay inside)
                cur = funcs.setdefault(m.group(2), [])
            continue
        m = INS.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2) or ""))
    out = {}
    for name, ins in funcs.items():
        counts = {k: 0 for k in ("salu", "branch", "valu", "lds", "vmem", "smem", "other")}
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, op, args) in enumerate(ins):
            op = re.sub(r"_e(32|64)$|_dpp$|_sdwa$|_e64_dpp$", "", op)
            counts[classify(op)] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                mm = re.match(r"(\d+)", args)
                if mm:
                    imm = int(mm.group(1))
                    imm = imm - 65536 if imm >= 32768 else imm
                    tgt = a + 4 + 4 * imm
                    if tgt <= a and tgt in addr_index:
                        loops.append(i - addr_index[tgt] + 1)
        counts["total"] = len(ins)
        out[name] = {"counts": counts, "loops": sorted(loops, reverse=True)[:24]}
    return out


def demangle(names):
    f = shutil.which("c++filt") or _tool("llvm-cxxfilt")
    if not f:
        return {n: n for n in names}
    res = subprocess.run([f], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {n: re.sub(r"^void bkd::", "", d.split("(")[0]) for n, d in zip(names, res)}


def measure(lib):
    with tempfile.TemporaryDirectory(prefix="bourse_isa_") as tmp:
        ks = {}
        for o in code_objects(lib, tmp):
            ks.update(kernels_of(o))
    names = demangle(sorted(ks))
    return {names[n]: v for n, v in sorted(ks.items()) if names[n].startswith("k_")}


def toolchain():
    try:
        v = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
        return " | ".join(l.strip() for l in v.splitlines() if "HIP version" in l or "clang version" in l)
    except Exception:  # noqa: BLE001
        return "unknown"


def compare(base, now, tol=TOL):
    """List of human-readable differences beyond `tol` (relative; +-2 instructions always pass)."""
    bad = []

    def off(a, b):
        return abs(a - b) > max(2, tol * max(a, b))

    for k in sorted(set(base) | set(now)):
        if k not in now:
            bad.append(f"{k}: kernel gone")
            continue
        if k not in base:
            bad.append(f"{k}: new kernel (not in the baseline)")
            continue
        b, n = base[k], now[k]
        for c in ("total", "salu", "branch", "valu", "lds", "vmem"):
            if off(b["counts"][c], n["counts"][c]):
                bad.append(f"{k}: {c} instructions {b['counts'][c]} -> {n['counts'][c]}")
        if len(b["loops"]) != len(n["loops"]):
            bad.append(f"{k}: {len(b['loops'])} loops -> {len(n['loops'])}")
        else:
            for i, (x, y) in enumerate(zip(b["loops"], n["loops"])):
                if off(x, y):
                    bad.append(f"{k}: loop #{i} (by size) {x} -> {y} instructions")
    return bad


def main(argv):
    lib = os.path.join(ROOT, "bourse_amd", "csrc", "libbourse_amd.so")
    if "--lib" in argv:
        lib = argv[argv.index("--lib") + 1]
        argv = [a for a in argv if a not in ("--lib", lib)]
    filt = [a for a in argv if not a.startswith("-")]
    now = measure(lib)
    if "--update" in argv:
        json.dump({"_toolchain": toolchain(), "_tolerance": TOL,
                   "_how": "python tools/kernel_isa_counts.py --update (after an intended kernel change); checked by tests/test_kernel_isa_baseline.py",
                   "kernels": now}, open(BASELINE, "w"), indent=1, sort_keys=True)
        print(f"{len(now)} kernels -> {os.path.relpath(BASELINE, ROOT)}")
        return 0
    if "--check" in argv:
        base = json.load(open(BASELINE))
        bad = compare(base["kernels"], now, base.get("_tolerance", TOL))
        for b in bad:
            print(b)
        print(f"{len(now)} kernels, {len(bad)} differences beyond {base.get('_tolerance', TOL) * 100:.0f} % "
              f"(baseline: {base.get('_toolchain')}; now: {toolchain()})")
        return 1 if bad else 0
    for k, v in now.items():
        if filt and not any(f in k for f in filt):
            continue
        c = v["counts"]
        print(f"{k[:58]:58s} total {c['total']:6d}  salu {c['salu']:5d} br {c['branch']:4d} valu {c['valu']:5d} lds {c['lds']:4d} "
              f"vmem {c['vmem']:4d} smem {c['smem']:3d}  loops {v['loops'][:10]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
