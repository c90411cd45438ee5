#!/usr/bin/env python
"""In-tree native build: ``python build.py [--force] [--jobs N]``.

* ``csrc/kernels/*.hip``  → ``distributeddeeplearning_amd/_ddl_kernels.so``
  (``hipcc --offload-arch=gfx950``; the .hip sources are authored directly, nothing is
  hipified, and no torch header is included so each TU compiles in seconds)
* ``csrc/engine/*.cpp``   → ``distributeddeeplearning_amd/_ddl_engine.so``  (host C++17)

Objects are cached under ``build/`` and rebuilt when their source or a header changed.  The
.so files are git-ignored but travel to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "distributeddeeplearning_amd")
BUILD = os.path.join(ROOT, "build")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def build_kernels(force=False, jobs=8, verbose=True):
    srcs = sorted(glob.glob(os.path.join(ROOT, "csrc", "kernels", "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(ROOT, "csrc", "kernels", "*.h")))
    out = os.path.join(PKG, "_ddl_kernels.so")
    if not srcs:
        return None
    os.makedirs(os.path.join(BUILD, "kernels"), exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-math-errno", "-ffp-contract=fast",
             "-fno-gpu-sanitize", "-Wno-unused-result", "-I", os.path.join(ROOT, "csrc", "kernels")]
    flags += os.environ.get("DDL_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DDDL_CONV_PROBES (timing probes)
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(BUILD, "kernels", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            todo.append((s, o))

    def compile_one(so):
        s, o = so
        _run([hipcc] + flags + ["-c", s, "-o", o])
        return s

    with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        for s in ex.map(compile_one, todo):
            if verbose:
                print("[build] compiled", os.path.relpath(s, ROOT), flush=True)
    if todo or force or _newer(out, objs):
        _run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", out] + objs)
        if verbose:
            print("[build] linked", os.path.relpath(out, ROOT), flush=True)
    return out


def build_engine(force=False, jobs=8, verbose=True):
    srcs = sorted(glob.glob(os.path.join(ROOT, "csrc", "engine", "*.cpp")))
    hdrs = sorted(glob.glob(os.path.join(ROOT, "csrc", "engine", "*.h")))
    out = os.path.join(PKG, "_ddl_engine.so")
    if not srcs:
        return None
    os.makedirs(os.path.join(BUILD, "engine"), exist_ok=True)
    cxx = shutil.which("g++") or shutil.which("c++")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    flags = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__",
             "-I", os.path.join(ROOT, "csrc", "engine"), "-I", os.path.join(rocm, "include")]
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(BUILD, "engine", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            todo.append((s, o))

    def compile_one(so):
        s, o = so
        _run([cxx] + flags + ["-c", s, "-o", o])
        return s

    with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        for s in ex.map(compile_one, todo):
            if verbose:
                print("[build] compiled", os.path.relpath(s, ROOT), flush=True)
    if todo or force or _newer(out, objs):
        _run([cxx, "-shared", "-fPIC", "-pthread", "-o", out] + objs +
             ["-L" + os.path.join(rocm, "lib"), "-Wl,-rpath," + os.path.join(rocm, "lib"),
              "-lamdhip64", "-lrccl"])
        if verbose:
            print("[build] linked", os.path.relpath(out, ROOT), flush=True)
    return out


def build_probes(force=False, verbose=True):
    """Standalone HIP measurement programs under benchmarks/native (one executable per .hip)."""
    outs = []
    hipcc = _hipcc()
    for s in sorted(glob.glob(os.path.join(ROOT, "benchmarks", "native", "*.hip"))):
        out = s[:-4]
        if force or _newer(out, [s]):
            _run([hipcc, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fno-gpu-sanitize",
                  "-Wno-unused-result", "-o", out, s])
            if verbose:
                print("[build] built", os.path.relpath(out, ROOT), flush=True)
        outs.append(out)
    return outs


def build_engine_stress(tsan=True, force=False, verbose=True):
    """Native stress driver of the engine's TCP data plane (csrc/engine/tests/engine_stress.cpp),
    by default under ThreadSanitizer.  Uses the ROCm LLVM clang++: g++ 11's libtsan does not
    intercept pthread_cond_clockwait (std::condition_variable::wait_for) and reports phantom
    double locks.  Host code only: no HIP/RCCL objects are linked."""
    srcs = [os.path.join(ROOT, "csrc", "engine", f) for f in ("engine.cpp", "store.cpp", "tests/engine_stress.cpp")]
    hdrs = sorted(glob.glob(os.path.join(ROOT, "csrc", "engine", "*.h")))
    out = os.path.join(BUILD, "engine_tsan" if tsan else "engine_plain", "engine_stress")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    clang = os.path.join(rocm, "lib", "llvm", "bin", "clang++")
    cxx = clang if os.path.exists(clang) else (shutil.which("clang++") or shutil.which("g++"))
    if tsan and "clang" not in os.path.basename(cxx):
        raise RuntimeError("a clang++ with the ThreadSanitizer runtime is needed for --tsan")
    if force or _newer(out, srcs + hdrs):
        _run([cxx, "-O1", "-g", "-std=c++17", "-pthread", "-Wall"] + (["-fsanitize=thread"] if tsan else []) +
             ["-I", os.path.join(ROOT, "csrc", "engine")] + srcs + ["-o", out])
        if verbose:
            print("[build] built", os.path.relpath(out, ROOT), flush=True)
    return out


def build_all(force=False, jobs=8, verbose=True):
    libs = build_kernels(force, jobs, verbose), build_engine(force, jobs, verbose)
    build_probes(force, verbose)
    return libs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--tsan", action="store_true", help="also build the engine stress driver under ThreadSanitizer")
    a = ap.parse_args()
    if a.tsan:
        print(build_engine_stress(True, a.force))
    print(build_all(a.force, a.jobs))
