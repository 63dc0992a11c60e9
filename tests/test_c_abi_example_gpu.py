"""The drop-in boundary is a C-ABI: examples/c_abi_fit.c -- a complete fit (anchors, locality, features, sampling step, device-fitted
models, selection, refinement, bound update, graph) written in plain C against include/annchor_hip.h -- is compiled with gcc,
linked against the in-tree library and run; it checks its own graph against annchor_brute_force."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    import __graft_entry__ as g

    g.build()   # (the in-tree library the example links against)
    exe = str(tmp_path / "c_abi_fit")
    lib = os.path.join(ROOT, "annchor_amd")
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_fit.c"),
                        "-L", lib, "-lannchor_hip", "-Wl,-rpath," + lib, "-lm", "-o", exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    """(no GPU needed: the header is C, every entry point the example binds exists in the library)"""
    _compile(tmp_path)


@pytest.mark.gpu
def test_c_example_fits_and_checks_its_graph(tmp_path):
    exe = _compile(tmp_path)
    r = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "errors of" in r.stdout
    print(r.stdout)
