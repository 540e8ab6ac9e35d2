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


# every §8(a) entry point of INTEGRATION.md section 3 and the HipVar / HipVarDiff method that must reach it
REQUIRED = {
    "mm": ["nk_mm_fwd", "nk_mm_bwd_left", "nk_mm_bwd_right"],
    "mm_t": ["nk_mm_t_fwd", "nk_mm_t_bwd_left", "nk_mm_t_bwd_right"],
    "convolution": ["nk_conv_fwd", "nk_conv_bwd_input", "nk_conv_bwd_kernel"],
    "binary": ["nk_binary_fwd", "nk_binary_bwd_left", "nk_binary_bwd_right"],
    "sum": ["nk_sum_fwd", "nk_sum_bwd"], "mean": ["nk_mean_fwd", "nk_mean_bwd"],
    "softmax": ["nk_softmax_fwd", "nk_softmax_bwd"], "log_softmax": ["nk_log_softmax_fwd", "nk_log_softmax_bwd"],
    "dropout": ["nk_dropout_fwd", "nk_dropout_bwd"], "relu": ["nk_relu_fwd", "nk_relu_bwd"],
    "mse": ["nk_mse_fwd", "nk_mse_bwd"], "pad_with": ["nk_pad_const_fwd", "nk_pad_reflective_fwd", "nk_pad_replicative_fwd", "nk_pad_bwd"],
    "chunks": ["nk_chunk_fwd", "nk_chunk_bwd"], "cat": ["nk_concat_fwd_part", "nk_concat_bwd_part"],
    "t": ["nk_transpose_fwd", "nk_transpose_bwd"], "heads_attention": ["nk_attention_fwd", "nk_attention_bwd"],
}


def _rust_params(argstr):
    """Top-level parameter count of a Rust parameter list (commas inside <...> generics do not count)."""
    depth, n, seen = 0, 0, False
    for ch in argstr:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
        if not ch.isspace():
            seen = True
    return n + 1 if seen else 0


def _rust_nodes():
    """struct name -> (number of `new` parameters or None, set of ffi functions its Forward / Backward bodies call)."""
    nodes = {}
    node_dir = os.path.join(HIP, "node")
    for f in sorted(os.listdir(node_dir)):
        if not f.endswith(".rs") or f == "mod.rs":
            continue
        src = open(os.path.join(node_dir, f)).read()
        for m in re.finditer(r"pub\(crate\) struct (\w+)", src):
            nodes.setdefault(m.group(1), [None, set()])
        for m in re.finditer(r"impl(?:<[^{]*?>)? (\w+)(?:<[^{]*?>)?\s*(?:where[^{]*)?\{\s*(?:///[^\n]*\n\s*)*(?:#\[[^\]]*\]\s*)*pub\(crate\) fn new\(", src):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            nodes[m.group(1)][0] = _rust_params(src[m.end():i - 1])
        for m in re.finditer(r"impl(?:<[^{]*?>)? (?:Forward|Backward) for (\w+)", src):
            body_start = src.index("{", m.end())
            i, depth = body_start + 1, 1
            while depth:
                depth += {"{": 1, "}": -1}.get(src[i], 0)
                i += 1
            nodes[m.group(1)][1] |= set(re.findall(r"ffi::(nk_[a-z0-9_]+)\(", src[body_start:i]))
    return nodes


def test_every_hot_path_entry_point_is_reachable_from_a_variable_method():
    """VERDICT round 2, missing #3: node structs that nothing constructs are not a binding.  Every §8(a) op has a
    `HipVar` / `HipVarDiff` method in hipvar.rs; the nodes that method constructs (`Node::new(...)`, with the arity of the
    node's `new`) call, between them, every entry point INTEGRATION.md section 3 lists for the op."""
    src = open(os.path.join(HIP, "hipvar.rs")).read()
    nodes = _rust_nodes()
    # split hipvar.rs into method bodies
    methods = {}
    for m in re.finditer(r"(?:pub(?:\(crate\))? )?fn (\w+)(?:<[^(]*>)?\(", src):
        b = src.index("{", m.end())
        i, depth = b + 1, 1
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        methods.setdefault(m.group(1), "")
        methods[m.group(1)] += src[b:i]
    constructed = set()
    for name, body in methods.items():
        for m in re.finditer(r"\b([A-Z]\w+)::new\(", body):
            if m.group(1) in ("Rc", "RefCell", "Cell", "HashMap"):
                continue
            assert m.group(1) in nodes, (name, m.group(1))
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(body[i], 0)
                i += 1
            assert nodes[m.group(1)][0] == _top_level_args(body[m.end():i - 1]), (name, m.group(1), nodes[m.group(1)][0])
            constructed.add(m.group(1))
    def reach(method, seen=None):
        seen = seen or set()
        if method in seen or method not in methods:
            return set()
        seen.add(method)
        body = methods[method]
        out = set()
        for m in re.finditer(r"\b([A-Z]\w+)::new\(", body):
            out |= nodes.get(m.group(1), [None, set()])[1]
        for callee in re.findall(r"\.(\w+)\(", body):           # var-level twin (self.var.relu()), pad_zero -> pad_with, ...
            if callee in methods and callee != method:
                out |= reach(callee, seen)
        return out
    for method, wanted in REQUIRED.items():
        assert method in methods, method
        got = reach(method)
        assert set(wanted) <= got, (method, sorted(set(wanted) - got))
    # every node struct with a Forward / Backward body is constructed by some method (no dead node files)
    dead = [n for n, (arity, calls) in nodes.items() if calls and n not in constructed]
    assert dead == [], dead
    # the dropout nodes advance the Philox offset by the calls one forward consumes: ceil(n / 8)
    assert "+ 7) / 8" in open(os.path.join(HIP, "node", "pointwise.rs")).read()
    assert "+ 7) / 8" in open(os.path.join(HIP, "node", "attention.rs")).read()
