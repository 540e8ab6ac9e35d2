"""The Rust side of the drop-in boundary (integration/neuronika-variable/) cannot be compiled in this image (no rustc /
cargo), so it is checked structurally: `ffi.rs` is generated from include/neuronika_hip.h and must be up to date, carry
every prototype of the header exactly once with the header's arity, and every `ffi::nk_*` call in the hand-written
modules must name a header function and pass the number of arguments its prototype takes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIP = os.path.join(ROOT, "integration", "neuronika-variable", "src", "hip")


def _top_level_args(argstr):
    depth, n, seen = 0, 0, False
    for ch in argstr:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
        if not ch.isspace():
            seen = True
    return n + 1 if seen else 0


def test_ffi_is_generated_from_the_header_and_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import gen_rust_ffi as g
    protos = g.prototypes()
    assert len(protos) > 100
    src = open(os.path.join(HIP, "ffi.rs")).read()
    decls = re.findall(r"pub fn (nk_[a-z0-9_]+)\((.*?)\)(?: -> [^;]+)?;", src)
    assert [d[0] for d in decls] == [p[1] for p in protos]                      # every symbol, once, header order
    for (name, args), (_, pname, params) in zip(decls, protos):
        assert _top_level_args(args) == len(params), name                       # same arity
    # the ctypes table used by the GPU tests binds the same symbols: three views of one ABI
    from neuronika_amd import capi
    assert sorted(capi.EXPORTED) == sorted(p[1] for p in protos)


def test_type_mapping():
    import gen_rust_ffi as g
    for c, rs in (("nk_device*", "*mut nk_device"), ("const nk_comm*", "*const nk_comm"), ("nk_event**", "*mut *mut nk_event"),
                  ("const float*", "*const c_float"), ("float**", "*mut *mut c_float"), ("float* const*", "*const *mut c_float"),
                  ("const size_t*", "*const usize"), ("uint64_t", "u64"), ("long long", "c_longlong"), ("const char*", "*const c_char"),
                  ("void**", "*mut *mut c_void"), ("double*", "*mut c_double"), ("int", "c_int")):
        assert g.rust_type(c) == rs, (c, g.rust_type(c))


def test_handwritten_modules_call_the_abi_consistently():
    import gen_rust_ffi as g
    arity = {p[1]: len(p[2]) for p in g.prototypes()}
    calls = 0
    for dp, _, fs in os.walk(HIP):
        for f in fs:
            if not f.endswith(".rs") or f == "ffi.rs":
                continue
            src = open(os.path.join(dp, f)).read()
            for m in re.finditer(r"ffi::(nk_[a-z0-9_]+)\(", src):
                name = m.group(1)
                assert name in arity, (f, name)
                i, depth = m.end(), 1
                while depth:                                   # matching parenthesis of the call
                    depth += {"(": 1, ")": -1}.get(src[i], 0)
                    i += 1
                assert _top_level_args(src[m.end():i - 1]) == arity[name], (f, name)
                calls += 1
    assert calls >= 20
    # the module list names files that exist
    for d in (HIP, os.path.join(HIP, "node")):
        mods = re.findall(r"^mod (\w+);", open(os.path.join(d, "mod.rs")).read(), re.M)
        assert mods
        for mname in mods:
            assert os.path.exists(os.path.join(d, mname + ".rs")) or os.path.exists(os.path.join(d, mname, "mod.rs")), mname
