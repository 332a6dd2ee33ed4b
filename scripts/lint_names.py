#!/usr/bin/env python
"""Poor man's pyflakes (none is installed here, and a NameError on the GPU box costs minutes of budget): report names
that a function loads but that are bound nowhere -- not in the function, an enclosing function, the module, builtins.
    python scripts/lint_names.py tumblr_emotions_amd tests bench.py"""
import ast
import builtins
import os
import sys


def bound_names(node):
    names = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        elif isinstance(n, ast.arg):
            names.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                names.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
    return names


def check(path):
    tree = ast.parse(open(path).read(), path)
    module = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__"}
    bad = []

    def visit(fn, outer):
        scope = outer | bound_names(fn)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
                bad.append((path, n.lineno, n.id))
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
            visit(n, module)
    return bad


def main(args):
    files = []
    for a in args:
        if os.path.isdir(a):
            for d, _, fs in os.walk(a):
                files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
        else:
            files.append(a)
    bad = [b for f in sorted(files) for b in check(f)]
    for b in sorted(set(bad)):
        print("%s:%d: undefined name %r" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["tumblr_emotions_amd", "tests", "bench.py", "scripts", "__graft_entry__.py"]))
