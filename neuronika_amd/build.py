"""Build libneuronika_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m neuronika_amd.build          # incremental (mtime based)
    python -m neuronika_amd.build --force

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot."""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libneuronika_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "neuronika_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lrccl",
               "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[neuronika_amd.build] {LIB} ({'rebuilt' if rebuilt else 'up to date'}; {len(objs)} objects)")
    return LIB


def build_tape(force: bool = False, verbose: bool = True) -> str:
    """Build the C++ host tape mirror (host/*.cpp) + its pybind11 glue as neuronika_amd/_tape*.so,
    linked against libneuronika_hip.so (rpath = $ORIGIN/lib)."""
    import sysconfig

    import pybind11

    host = os.path.join(ROOT, "host")
    srcs = [os.path.join(host, "neuronika.cpp"), os.path.join(host, "data.cpp"), os.path.join(host, "pymodule.cpp")]
    deps = srcs + [os.path.join(host, "neuronika.hpp"), os.path.join(host, "data.hpp"),
                   os.path.join(ROOT, "include", "neuronika_hip.h")]
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    out = os.path.join(HERE, "_tape" + ext)
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(d) for d in deps):
        if verbose:
            print(f"[neuronika_amd.build] {out} (up to date)")
        return out
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + host, "-I" + pybind11.get_include(),
           "-I" + sysconfig.get_paths()["include"], *srcs, "-o", out,
           "-L" + LIBDIR, "-lneuronika_hip", "-Wl,-rpath,$ORIGIN/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host tape build failed:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    # a shared object links fine with undefined symbols: load it once so that a declared-but-undefined function fails
    # the build here, not the first test on the GPU box
    chk = subprocess.run([sys.executable, "-c", "import importlib.util,sys; s=importlib.util.spec_from_file_location('neuronika_amd._tape', sys.argv[1]); m=importlib.util.module_from_spec(s); s.loader.exec_module(m)", out],
                         capture_output=True, text=True)
    if chk.returncode != 0:
        os.remove(out)
        raise RuntimeError(f"host tape module does not load:\n{chk.stderr}")
    if verbose:
        print(f"[neuronika_amd.build] {out} (rebuilt)")
    return out


def build_all(force: bool = False, verbose: bool = True):
    build(force, verbose)
    build_tape(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
