"""Undefined-name lint for the repo's Python (no pyflakes in the image).

Round 4 shipped `bench.py --profile-every 0` with a NameError (`KINDS`, never defined) that killed every PMC pass AFTER
its timed region.  A name a function loads as a GLOBAL must be bound somewhere at module level (assignment, import, def,
class, `global` store in a function), or be a builtin.  This catches exactly that class of bug from the compiled code
objects - no execution, no imports.

usage: python tools/lint_names.py [files...]     (default: every tracked .py of the product, bench, scripts and tools)
"""
import builtins
import dis
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["bench.py", "__graft_entry__.py", "bourse_amd/**/*.py", "scripts/*.py", "tools/*.py", "oracle/*.py", "tests/*.py"]
MODULE_DUNDERS = {"__name__", "__file__", "__doc__", "__builtins__", "__spec__", "__package__", "__loader__", "__path__",
                  "__debug__", "__annotations__", "__class__", "__qualname__", "__module__"}


def code_objects(co):
    yield co
    for c in co.co_consts:
        if hasattr(c, "co_code"):
            yield from code_objects(c)


def undefined_names(path):
    src = open(path).read()
    top = compile(src, path, "exec")
    bound, star = set(MODULE_DUNDERS), False
    for co in code_objects(top):
        for ins in dis.get_instructions(co):
            if ins.opname in ("STORE_GLOBAL", "DELETE_GLOBAL") or (co is top and ins.opname in ("STORE_NAME", "IMPORT_NAME")):
                bound.add(ins.argval.split(".")[0] if ins.opname == "IMPORT_NAME" else ins.argval)
            if ins.opname == "IMPORT_STAR":
                star = True
    # class bodies bind with STORE_NAME in their own code object: names loaded there may be class-local
    out = []
    for co in code_objects(top):
        local_names = set()
        if co is not top:
            local_names = {i.argval for i in dis.get_instructions(co) if i.opname == "STORE_NAME"}
        line = co.co_firstlineno
        for ins in dis.get_instructions(co):
            if ins.starts_line:  # only the first instruction of a source line carries it
                line = ins.starts_line
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
                n = ins.argval
                if n in bound or n in local_names or hasattr(builtins, n) or star:
                    continue
                out.append((line, n, co.co_name))
    return sorted(set(out))


def main(argv):
    files = argv or sorted({f for pat in DEFAULT for f in glob.glob(os.path.join(ROOT, pat), recursive=True)})
    bad = 0
    for f in files:
        for line, name, where in undefined_names(f):
            print(f"{os.path.relpath(f, ROOT)}:{line}: undefined name '{name}' in {where}")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
