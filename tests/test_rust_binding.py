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
    "mm": ["nk_mm_fwd", "nk_mm_bwd_left", "nk_mm_bwd_right", "nk_mm_bwd"],
    "mm_t": ["nk_mm_t_fwd", "nk_mm_t_bwd_left", "nk_mm_t_bwd_right", "nk_mm_t_bwd"],
    "convolution": ["nk_conv_fwd", "nk_conv_bwd_input", "nk_conv_bwd_kernel"],
    "binary": ["nk_binary_fwd", "nk_binary_bwd_left", "nk_binary_bwd_right"],
    "sum": ["nk_sum_fwd", "nk_sum_bwd"], "mean": ["nk_mean_fwd", "nk_mean_bwd"],
    "softmax": ["nk_softmax_fwd", "nk_softmax_bwd"], "log_softmax": ["nk_log_softmax_fwd", "nk_log_softmax_bwd"],
    "dropout": ["nk_dropout_fwd", "nk_dropout_bwd"], "relu": ["nk_relu_fwd", "nk_relu_bwd"],
    "mse": ["nk_mse_fwd", "nk_mse_bwd"], "pad_with": ["nk_pad_const_fwd", "nk_pad_reflective_fwd", "nk_pad_replicative_fwd", "nk_pad_bwd"],
    "chunks": ["nk_chunk_fwd", "nk_chunk_bwd"], "cat": ["nk_concat_fwd_part", "nk_concat_bwd_part"],
    "t": ["nk_transpose_fwd", "nk_transpose_bwd"], "heads_attention": ["nk_attention_fwd", "nk_attention_bwd"],
    # the fused entries the driver's numbers are measured on (VERDICT round 4, next #2b): what `neuronika_nn::hip`'s layers call
    "linear": ["nk_linear_fwd", "nk_linear_relu_fwd", "nk_linear_bwd_input_relu", "nk_mm_t_bwd_left", "nk_mm_t_bwd_right", "nk_unbroadcast_add",
               "nk_relu_bwd_assign"],
    "linear_diff": ["nk_linear_fwd", "nk_linear_relu_fwd", "nk_mm_t_bwd_right", "nk_unbroadcast_add"],
    "convolution_bias": ["nk_conv_bias_fwd", "nk_conv_bwd_input", "nk_conv_bwd_kernel_bias"],
    "convolution_bias_padded": ["nk_conv_bias_fwd_padded", "nk_conv_bwd_input_padded", "nk_conv_bwd_kernel_bias_padded"],
    "packed_heads_attention": ["nk_attention_qkv_fwd", "nk_attention_qkv_bwd"],
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
        for m in re.finditer(r"pub(?:\(crate\))? struct (\w+)", src):
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


# ---- static checks a never-compiled crate needs ---------------------------------------------------
INTEGRATION = os.path.join(ROOT, "integration")
NN = os.path.join(INTEGRATION, "neuronika-nn", "src", "hip.rs")


def _rust_files():
    out = []
    for dp, _, fs in os.walk(INTEGRATION):
        out += [os.path.join(dp, f) for f in fs if f.endswith(".rs")]
    return sorted(out)


def _strip(src):
    """Source without comments, string literals and char literals (lifetimes survive: they contain no delimiters)."""
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src, flags=re.S)
    src = re.sub(r"'(?:\\.|[^\\'])'", "' '", src)
    return src


def test_rust_sources_have_balanced_delimiters():
    files = _rust_files()
    assert len(files) >= 18
    pairs = {")": "(", "]": "[", "}": "{"}
    for path in files:
        stack = []
        for n, line in enumerate(_strip(open(path).read()).split("\n"), 1):
            for ch in line:
                if ch in "([{":
                    stack.append((ch, n))
                elif ch in ")]}":
                    assert stack and stack[-1][0] == pairs[ch], (path, n, ch, stack[-1:] if stack else None)
                    stack.pop()
        assert not stack, (path, stack[-1])


def _items(path):
    """Names a Rust file defines or re-exports at any visibility: struct / enum / trait / fn / type / mod / macro + `pub use` leaves."""
    src = _strip(open(path).read())
    names = set(re.findall(r"\b(?:struct|enum|trait|fn|type|mod|union)\s+(\w+)", src))
    names |= set(re.findall(r"macro_rules!\s*(\w+)", src))
    for m in re.finditer(r"pub(?:\([^)]*\))?\s+use\s+([^;]+);", src, re.S):
        names |= set(re.findall(r"(\w+)\s*(?:,|\}|$)", m.group(1)))
        g = re.match(r"\s*(\w+)::\*\s*$", m.group(1))            # glob re-export of a sibling module: its items too
        if g:
            sub = _module_file(os.path.dirname(path), g.group(1))
            if sub:
                names |= _items(sub)
    return names


def _module_file(base_dir, name):
    for cand in (os.path.join(base_dir, name + ".rs"), os.path.join(base_dir, name, "mod.rs")):
        if os.path.exists(cand):
            return cand
    return None


def _use_leaves(tree):
    """`a::b::{c, d::{e, f}, g as h}` -> [[a, b, c], [a, b, d, e], [a, b, d, f], [a, b, g]]."""
    tree = tree.strip()
    m = re.match(r"^([\w:]*?)(?:::)?\{(.*)\}$", tree, re.S)
    if not m:
        return [[p for p in re.sub(r"\s+as\s+\w+", "", tree).split("::") if p]]
    prefix = [p for p in m.group(1).split("::") if p]
    parts, depth, cur = [], 0, ""
    for ch in m.group(2):
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    out = []
    for part in parts:
        for leaf in _use_leaves(part):
            out.append(prefix + leaf)
    return out


def test_rust_use_paths_resolve_inside_the_integration_tree():
    """Every `use` of an item that lives in THIS tree (`super::`, `crate::hip::`, `neuronika_variable::hip::`) names a module
    file that exists and an item that file defines or re-exports.  Paths into the reference crate (`crate::autograd`,
    `crate::gradient`, `crate::utils`, ...) and into external crates are checked against a list of the names they are known
    to export (read off /root/reference when this test was written)."""
    hip_mod = os.path.join(HIP, "mod.rs")
    reference = {"autograd": {"Backward", "Forward"}, "gradient": {"Gradient", "NoGrad"}, "history": {"History"},
                 "utils": {"check_conv_args", "check_groups_args", "cobroadcast", "conv_out_shape", "Broadcast", "Shared"}}
    checked = 0
    for path in _rust_files():
        if os.path.basename(path) in ("autograd_hip_ext.rs", "gradient_hip_impls.rs", "build.rs"):
            continue
        here = os.path.dirname(path)
        src = _strip(open(path).read())
        for m in re.finditer(r"^\s*(?:pub(?:\([^)]*\))?\s+)?use\s+([^;]+);", src, re.M | re.S):
            for leaf in _use_leaves(re.sub(r"\s+", " ", m.group(1))):
                if leaf[0] == "neuronika_variable" and leaf[1] == "hip":
                    base, rest = HIP, leaf[2:]
                    assert rest and rest[0] in _items(hip_mod), (path, leaf)
                    checked += 1
                    continue
                if leaf[0] == "crate" and len(leaf) > 1 and leaf[1] == "hip":
                    base, rest = HIP, leaf[2:]
                elif leaf[0] == "super":
                    base = here if os.path.basename(path) != "mod.rs" else os.path.dirname(here)
                    base = os.path.dirname(path) if os.path.basename(path) != "mod.rs" else os.path.dirname(os.path.dirname(path))
                    rest = leaf[1:]
                    if os.path.basename(path) != "mod.rs":
                        # `super` of hip/node/x.rs is hip/node/mod.rs; of hip/x.rs it is hip/mod.rs
                        if len(rest) == 1:
                            assert rest[0] in _items(os.path.join(os.path.dirname(path), "mod.rs")), (path, leaf)
                            checked += 1
                            continue
                        base = os.path.dirname(path)
                elif leaf[0] == "crate" and len(leaf) > 2 and leaf[1] in reference:
                    assert leaf[2] in reference[leaf[1]], (path, leaf)
                    checked += 1
                    continue
                else:
                    continue                                    # std, ndarray, rand, crate::Reduction, ...
                if not rest:
                    continue
                # walk module components, the last one is the item
                cur = base
                for comp in rest[:-1]:
                    f = _module_file(cur, comp)
                    assert f, (path, leaf, comp)
                    cur = os.path.join(cur, comp) if os.path.isdir(os.path.join(cur, comp)) else cur
                    last_file = f
                if len(rest) == 1:
                    f = _module_file(cur, rest[0])
                    assert f or rest[0] in _items(os.path.join(cur, "mod.rs")), (path, leaf)
                else:
                    assert rest[-1] in _items(last_file) or rest[-1] == "*", (path, leaf)
                checked += 1
    assert checked >= 40, checked


def _pub_methods(path):
    """method name -> parameter count (without self) of every `pub fn` / `pub(crate) fn` in a file."""
    src = _strip(open(path).read())
    out = {}
    for m in re.finditer(r"pub(?:\(crate\))?\s+fn\s+(\w+)(?:<[^(]*>)?\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        params = src[m.end():i - 1]
        n = _rust_params(params)
        if re.match(r"\s*(?:mut\s+)?&?\s*(?:mut\s+)?self\b", params):
            n -= 1
        out.setdefault(m.group(1), set()).add(n)
    return out


def test_nn_layers_exist_and_call_variable_methods_that_exist():
    """`neuronika_nn::hip`: every layer `north_star` names is a struct with the reference's fields, a `new` in the reference's
    argument order (+ device) and a `forward`; every method the layers call on a device variable exists in hipvar.rs with
    that many arguments."""
    src = _strip(open(NN).read())
    for layer in ("Linear", "Dropout", "MultiheadAttention"):
        assert re.search(r"pub struct %s\b" % layer, src), layer
        body = src[src.index("impl %s" % layer):]
        assert "pub fn new(" in body and "pub fn forward" in body, layer
    conv = re.findall(r"conv_layer!\((\w+), (\w+),", src)
    assert conv == [("Conv1d", "GroupedConv1d"), ("Conv2d", "GroupedConv2d"), ("Conv3d", "GroupedConv3d")]
    macro = src[src.index("macro_rules! conv_layer"):src.index("conv_layer!(Conv1d")]
    for field in ("padding", "padding_mode", "stride", "dilation", "weight", "bias"):          # neuronika-nn/src/lib.rs:633-642
        assert re.search(r"pub %s:" % field, macro), field
    assert re.search(r"pub groups: usize", macro)
    order = re.search(r"pub fn new\(in_channels: usize, out_channels: usize, kernel_size: \$size, padding: \$size, padding_mode: PaddingMode,\s*"
                      r"stride: \$size, dilation: \$size, device: &Device\)", macro)
    assert order, "Conv*::new argument order (neuronika-nn/src/lib.rs:671-679)"
    assert re.search(r"pub weight: HipVarDiff<Ix2>,\s*pub bias: HipVarDiff<Ix1>", src)             # Linear fields, :406-409
    methods = _pub_methods(os.path.join(HIP, "hipvar.rs"))
    # the layers are built from the FUSED variable methods: Linear = one node (nk_linear_fwd / nk_linear_relu_fwd), Conv* = one node with
    # the bias (nk_conv_bias_fwd / nk_conv_bwd_kernel_bias), MultiheadAttention = packed projections + nk_attention_qkv_*
    used = {"linear": 3, "linear_diff": 3, "pad": 2, "convolution_bias": 5, "convolution_bias_padded": 6, "padding_folds": 5, "dropout": 2,
            "packed_heads_attention": 7, "shape": 0, "parameter": 2}
    for slow in ("mm_t", "mm_t_diff", "convolution", "heads_attention"):                       # ... and from nothing slower
        assert not re.search(r"\.%s\(" % slow, src), slow
    for name, nargs in used.items():
        assert re.search(r"[.:]%s\(" % name, src), name
        assert name in methods and nargs in methods[name], (name, methods.get(name))
    # call sites pass that many arguments
    for m in re.finditer(r"\.(linear|linear_diff|pad|convolution_bias|convolution_bias_padded|padding_folds|dropout|packed_heads_attention)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        assert _top_level_args(src[m.end():i - 1]) == used[m.group(1)], m.group(1)


def test_backward_sync_overlaps_the_exchange():
    """Every device backward node names the gradients it writes (`targets`), `backward_sync` derives the last writer from
    them and hands the gradient over (`grad_ready`) right after issuing that node - the overlapped exchange `north_star` asks
    for, as `VarDiff::run_backward` in host/neuronika.cpp."""
    node_dir = os.path.join(HIP, "node")
    n_impl = n_targets = 0
    for f in sorted(os.listdir(node_dir)):
        src = _strip(open(os.path.join(node_dir, f)).read())
        for m in re.finditer(r"impl(?:<[^{]*?>)? Backward for (\w+)", src):
            b = src.index("{", m.end())
            i, depth = b + 1, 1
            while depth:
                depth += {"{": 1, "}": -1}.get(src[i], 0)
                i += 1
            n_impl += 1
            assert "fn targets(&self) -> Vec<usize>" in src[b:i], (f, m.group(1))
            n_targets += 1
    assert n_impl == n_targets >= 20
    hv = _strip(open(os.path.join(HIP, "hipvar.rs")).read())
    body = hv[hv.index("pub fn backward_sync("):]
    body = body[:body.index("\n    }\n") + 7]
    assert "op.targets()" in body and "sync.grad_ready(" in body and "last_writer" in body and ".rev()" in body
    assert body.index("op.backward()") < body.index("sync.grad_ready(")           # issued first, handed over right after
    ext = open(os.path.join(INTEGRATION, "neuronika-variable", "src", "autograd_hip_ext.rs")).read()
    assert "fn targets(&self) -> Vec<usize>" in ext and "Vec::new()" in ext       # defaulted: CPU nodes compile unchanged
    dp = _strip(open(os.path.join(HIP, "dp.rs")).read())
    assert "pub(crate) fn bucket_of(" in dp and "pub struct SyncEntry" in dp


def test_optimizer_step_is_the_multi_tensor_launch():
    """VERDICT round 4, next #2b: the Rust optimizer updates every registered parameter with ONE launch (`nk_sgd_step_multi`, what
    the driver's C4 step is measured with), not one `nk_sgd_step` per parameter."""
    opt = _strip(open(os.path.join(HIP, "optimizer.rs")).read())
    step = opt[opt.index("pub fn step("):]
    step = step[:step.index("\n    }") + 6]
    assert "sgd_step_multi(" in step and "nk_sgd_step(" not in opt
    node = _strip(open(os.path.join(HIP, "node", "optim.rs")).read())
    body = node[node.index("fn sgd_step_multi"):]
    assert "ffi::nk_sgd_step_multi(" in body
    assert "seen.insert" in body                          # a parameter registered twice is updated once
    mod = open(os.path.join(HIP, "mod.rs")).read()
    assert "mod optimizer;" in mod and "optimizer::SGD" in mod
    for name in ("register", "step", "zero_grad", "with_momentum"):      # neuronika-optim/src/optimizer.rs:60-94, sgd/mod.rs:62-110
        assert re.search(r"pub fn %s[(<]" % name, opt), name
