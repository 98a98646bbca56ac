"""Every name a function of the package, of bench.py or of a tool reads as a global must exist at module level or among
the builtins: a wrapper edited for one call and broken for another (a parameter that only the other one has) fails here,
on CPU, not as a NameError in the first GPU test that reaches the line."""
import builtins
import glob
import os
import symtable

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(REPO, "fastq_utils_amd", "*.py")) + glob.glob(os.path.join(REPO, "tools", "*.py")) +
               glob.glob(os.path.join(REPO, "oracle", "*.py")) + glob.glob(os.path.join(REPO, "tests", "*.py")) +
               [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")])


def undefined_globals(path):
    with open(path) as f:
        src = f.read()
    top = symtable.symtable(src, path, "exec")
    known = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__", "__loader__"}
    for s in top.get_symbols():
        if s.is_assigned() or s.is_imported() or s.is_namespace():
            known.add(s.get_name())
    # names a function declares `global` and assigns are module names too
    def declared(t):
        for s in t.get_symbols():
            if s.is_declared_global() and s.is_assigned():
                known.add(s.get_name())
        for c in t.get_children():
            declared(c)
    declared(top)
    bad = []

    def walk(t):
        for s in t.get_symbols():
            if t.get_type() != "module" and s.is_global() and s.is_referenced() and s.get_name() not in known:
                bad.append((t.get_name(), t.get_lineno(), s.get_name()))
        for c in t.get_children():
            walk(c)

    walk(top)
    return bad


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, REPO) for p in FILES])
def test_no_function_reads_a_global_that_does_not_exist(path):
    assert undefined_globals(path) == []
