"""Builds the engine's shared library in-tree with hipcc for gfx950.

    python semantic-search_amd/build.py [--force]

Output: semantic-search_amd/lib/libssmap.so (git-ignored; it travels with the tree to the
GPU box).  geometry.hip, label.hip, ego.hip and engine.cpp are compiled with -ffp-contract=off: their
float64 geometry has to round exactly like the CPU statement of the same formulas.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libssmap.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = [
    ("geometry.hip", ["-ffp-contract=off"]),
    ("label.hip", ["-ffp-contract=off"]),
    ("ego.hip", ["-ffp-contract=off"]),
    ("fuse.hip", []),
    ("engine.cpp", ["-ffp-contract=off"]),
]
COMMON = ["-x", "hip", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "ssmap_internal.h"), os.path.join(CSRC, "rows.h"),
               os.path.join(ROOT, "include", "ssmap.h")]
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, procs = [], []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if not force and _newer(obj, [src] + headers):
            continue
        cmd = [hipcc] + COMMON + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {name}")
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
