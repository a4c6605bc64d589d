"""Keep INTEGRATION.md section 1's Rust listing IDENTICAL to integration/rust/gpu_env.rs (the hand-written snippet of
rounds 1-3 had drifted from the ABI: a `reserved` field where the header has `assets`).  The listing between the
markers `<!-- BEGIN gpu_env.rs -->` / `<!-- END gpu_env.rs -->` is replaced by the file's text.
usage: sync_integration_md.py [--check]   (--check: exit 1 if INTEGRATION.md would change)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN gpu_env.rs -->", "<!-- END gpu_env.rs -->"


def render():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rs = open(os.path.join(ROOT, "integration", "rust", "gpu_env.rs")).read().rstrip("\n")
    a, b = md.index(BEGIN) + len(BEGIN), md.index(END)
    return md[:a] + "\n```rust\n" + rs + "\n```\n" + md[b:]


def check_against_sys():
    """Every `sys::NAME` gpu_env.rs uses exists in the generated FFI, and its BkConfig literal names exactly the struct's
    fields (no `..Default::default()` that would hide a renamed field)."""
    rs = open(os.path.join(ROOT, "integration", "rust", "gpu_env.rs")).read()
    sysrs = open(os.path.join(ROOT, "integration", "rust", "bourse_amd_sys.rs")).read()
    declared = set(re.findall(r"pub (?:fn|struct|const) (\w+)", sysrs))
    used = set(re.findall(r"sys::(\w+)", rs))
    missing = sorted(used - declared)
    assert not missing, f"gpu_env.rs uses names bourse_amd_sys.rs does not declare: {missing}"
    fields = re.findall(r"pub (?:r#)?(\w+):", re.search(r"pub struct BkConfig \{(.*?)\}", sysrs, re.S).group(1))
    lit = re.search(r"sys::BkConfig \{(.*?)\};", rs, re.S).group(1)
    names = [re.match(r"\s*(\w+)", part).group(1) for part in lit.split(",") if part.strip()]
    assert names == fields, f"BkConfig literal {names} != struct fields {fields}"


if __name__ == "__main__":
    check_against_sys()
    new = render()
    path = os.path.join(ROOT, "INTEGRATION.md")
    if "--check" in sys.argv:
        sys.exit(0 if new == open(path).read() else 1)
    open(path, "w").write(new)
    print("INTEGRATION.md section 1 listing = integration/rust/gpu_env.rs")
