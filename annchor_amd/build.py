"""Builds libannchor_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libannchor_hip.so")
SOURCES = ["ctx", "lev", "euclid", "emd", "picker", "scan", "locality", "features", "select", "refine", "state", "brute", "hostrng", "hostols", "streamed", "knnbf", "knnh", "knnbk", "sharded", "model", "enemies", "comm", "repair"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-ffp-contract=off"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "annchor_hip.h"))

    sources = [os.path.join(CSRC, n + ".hip") for n in SOURCES]
    if not force and not _stale(LIB, sources + headers):
        return LIB   # (the GPU box receives the library without the object files: .gpurunignore)

    def compile_one(name):
        src, obj = os.path.join(CSRC, name + ".hip"), os.path.join(OBJ, name + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr))
            if verbose and r.stderr.strip():
                print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
