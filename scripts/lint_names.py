"""Undefined-name check for the Python sources (no pyflakes in this image).

GPU-marked tests never execute on the CPU box, so a typo such as a fixture that is used but not
requested only shows up on the MI355X.  This walks every function with `symtable` and reports names
that are read as globals but defined neither at module level nor as builtins.
usage: python scripts/lint_names.py [paths...]   (exit code 1 if anything is found)
"""
import ast
import builtins
import os
import symtable
import sys


def module_names(tree):
    names = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                names.add((a.asname or a.name).split(".")[0])
    return names


def check(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    top = symtable.symtable(src, path, "exec")
    defined = module_names(tree) | {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    # names bound by `global x` inside functions
    bad = []

    def walk(tab):
        for s in tab.get_symbols():
            if s.is_global() and s.is_assigned():
                defined.add(s.get_name())
        for c in tab.get_children():
            walk(c)

    walk(top)

    def visit(tab):
        if tab.get_type() in ("function", "class"):
            for s in tab.get_symbols():
                if s.is_referenced() and s.is_global() and s.get_name() not in defined:
                    bad.append((tab.get_lineno(), tab.get_name(), s.get_name()))
        for c in tab.get_children():
            visit(c)

    visit(top)
    return bad


def main(argv):
    roots = argv or ["tests", "bench.py", "__graft_entry__.py", "instant-distance_amd", "oracle", "scripts"]
    files = []
    for r in roots:
        if os.path.isdir(r):
            for d, _, fs in os.walk(r):
                files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
        elif r.endswith(".py"):
            files.append(r)
    n = 0
    for f in sorted(files):
        for line, fn, name in check(f):
            print(f"{f}:{line}: undefined name '{name}' in {fn}()")
            n += 1
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
