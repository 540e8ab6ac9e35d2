#!/usr/bin/env python3
"""Transcribe the reference's own known-answer vectors into JSON fixtures.

The reference is Rust and cannot be compiled here, so its unit tests cannot be run.  Their
expected values are still the only ground truth for the hot path, so
this script PARSES the numeric literals (`vec![..]`, `array![..]`) out of the reference's
`#[test]` functions and writes them — together with the shapes/arguments each test uses and
the file:line they came from — to `tests/golden/reference_fixtures.json`.

Run in the authoring container (needs /root/reference; the GPU box does not have it):

    python tests/golden/extract_reference_fixtures.py

The per-test "spec" tables below only say WHICH literal of a test plays which role (input,
expected output, ...); every number in the JSON is read from the reference file, none is
typed in here.  Where a reference test builds its input procedurally (`(0..150).map(..)`,
`linspace`, `ones`) the fixture records the recipe (`{"arange": n}` etc.) instead.
"""
from __future__ import annotations

import json
import os
import re
import sys

REF = os.environ.get("NEURONIKA_REFERENCE", "/root/reference")
NODE = os.path.join(REF, "neuronika-variable", "src", "node")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json")


def _read(rel):
    with open(os.path.join(NODE, rel)) as f:
        return f.read()


def _fn_body(src, name, mod=None):
    """Return (body_text, first_line_no) of `fn name()`; if `mod` is given, search inside
    `mod <mod> { .. }` only."""
    start = 0
    if mod is not None:
        m = re.search(r"^mod\s+%s\s*\{" % re.escape(mod), src, re.M)
        start = m.end() if m else 0
    m = re.compile(r"fn\s+%s\s*\(" % re.escape(name)).search(src, start)
    if not m:
        raise KeyError(f"fn {name}")
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return src[i : j + 1], src.count("\n", 0, i) + 1


_NUM = re.compile(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?")


def _literals(body, line0):
    """All `vec![..]` / `array![..]` numeric literals of a function body, in source order:
    list of (line, flat_values)."""
    out = []
    for m in re.finditer(r"\b(vec|array)!\s*\[", body):
        i = m.end() - 1
        depth, j = 0, i
        while True:
            c = body[j]
            if c == "[":
                depth += 1
            elif c == "]":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        inner = body[i + 1 : j]
        inner = re.sub(r"//[^\n]*", "", inner)
        rep = re.fullmatch(r"\s*(-?\d+\.?\d*)\s*;\s*(\d+)\s*", inner)
        if rep:
            vals = [float(rep.group(1))] * int(rep.group(2))
        else:
            if re.search(r"[A-Za-z_]", inner):
                continue  # not a pure numeric literal (e.g. vec![a.view(), ..])
            vals = [float(t) for t in _NUM.findall(inner)]
        out.append((line0 + body.count("\n", 0, m.start()), vals))
    return out


def scalars(body):
    """`arr0(x)` scalar literals of a function body, in source order."""
    return [float(x) for x in re.findall(r"arr0\((-?\d+\.?\d*)\)", body)]


def from_elems(body):
    """`from_elem(shape, value)` / `ones(shape)` / `zeros(shape)` / `ndarray_zeros(shape)`
    constructors of a function body, in source order: [{"shape": [...], "value": v}]."""
    out = []
    for m in re.finditer(r"(from_elem|ones|ndarray_zeros|zeros)\(\s*(\([0-9, ]+\)|\d+)\s*(?:,\s*(-?\d+\.?\d*))?\)", body):
        kind, shape, val = m.group(1), _tuple(m.group(2)), m.group(3)
        v = float(val) if kind == "from_elem" else (1.0 if kind == "ones" else 0.0)
        out.append({"shape": shape, "value": v})
    return out


def lits(rel, fn, mod=None):
    body, l0 = _fn_body(_read(rel), fn, mod)
    return _literals(body, l0)


def cite(rel, line):
    return f"neuronika-variable/src/node/{rel}:{line}"


# ----------------------------------------------------------------------------------------
# convolution/test.rs — uniform structure, fully parsed (shapes, stride, dilation, groups)
# ----------------------------------------------------------------------------------------

CONV_TESTS = [
    "conv1d", "conv2d", "conv3d",
    "conv1d_strided", "conv2d_strided", "conv3d_strided",
    "conv1d_dilated", "conv2d_dilated", "conv3d_dilated",
    "grouped_conv1d", "grouped_conv2d", "grouped_conv3d",
]


def _tuple(s):
    return [int(t) for t in re.findall(r"\d+", s)]


def conv_cases():
    rel = "convolution/test.rs"
    src = _read(rel)
    cases = []
    for fn in CONV_TESTS:
        body, l0 = _fn_body(src, fn)
        in_shape = _tuple(re.search(r"into_shape\(\(([^)]*)\)\)", body).group(1))
        k_shape = _tuple(re.search(r"let kernel = Array::<f32, _>::ones\(\(([^)]*)\)\)", body).group(1))
        stride = _tuple(re.search(r"let stride = &\[([^\]]*)\]", body).group(1))
        dilation = _tuple(re.search(r"let dilation = &\[([^\]]*)\]", body).group(1))
        g = re.search(r"let groups = (\d+)", body)
        named = {}
        for m in re.finditer(r"let (true_\w+)(?:\s*:\s*[^=]+)?\s*=\s*", body):
            sub = body[m.end():]
            lit = _literals(sub, 0)
            # first literal after the `let` is the value
            line = l0 + body.count("\n", 0, m.start())
            named[m.group(1)] = (line, lit[0][1])
        n_in = 1
        for d in in_shape:
            n_in *= d
        assert re.search(r"\(0\.\.%d\)" % n_in, body.replace("_", "")), fn  # input = 0..n as f32
        cases.append({
            "name": fn,
            "cite": cite(rel, l0),
            "input": {"arange": n_in, "shape": in_shape},
            "kernel": {"ones": k_shape},
            "grad": "ones",
            "stride": stride, "dilation": dilation, "groups": int(g.group(1)) if g else 1,
            "output": named["true_output_elems"][1],
            "output_cite": cite(rel, named["true_output_elems"][0]),
            "input_grad": named["true_input_grad_elems"][1],
            "kernel_grad": named["true_kernel_grad_elems"][1],
            "exact": True,
        })
    # im2col layout fixture (convolution/test.rs:11-84)
    body, l0 = _fn_body(src, "im2col")
    l = _literals(body, l0)
    cases_im2col = {
        "cite": cite(rel, l0),
        "image": l[0][1], "image_shape": [3, 4, 4],      # one sample, 3 channels of 4x4
        "batch": 2, "kernel_shape": [1, 3, 3, 3], "stride": [1, 1], "dilation": [1, 1],
        # the literal is im2col^T per sample, shape (27, 4): rows = (ci,kh,kw)... listed as
        # 27 rows of 4; the test compares `im2col.t()` stacked twice against columns (N, L, K).
        "im2col_T": l[1][1], "im2col_T_shape": [27, 4],
    }
    return cases, cases_im2col


# ----------------------------------------------------------------------------------------
# everything else: (file, fn, mod) -> which literal is what
# ----------------------------------------------------------------------------------------


def L(rel, fn, mod, idx):
    ls = lits(rel, fn, mod)
    line, vals = ls[idx]
    return vals, cite(rel, line)


def other_cases():
    c = {}

    # ---- MatMul (enabled tests) matrix_matrix_mul/test.rs:66-136: linspace inputs
    v1, c1 = L("matrix_matrix_mul/test.rs", "left_base_case", "backward", 0)
    v2, _ = L("matrix_matrix_mul/test.rs", "left_base_case", "backward", 1)
    v3, c3 = L("matrix_matrix_mul/test.rs", "right_base_case", "backward", 0)
    v4, _ = L("matrix_matrix_mul/test.rs", "right_base_case", "backward", 1)
    c["mm_backward"] = {
        "cite": c1,
        "left_bwd": {"right": {"linspace": [10.0, 18.0, 9], "shape": [3, 3]}, "grad": "ones",
                     "left_grad_once": v1, "left_grad_twice": v2},
        "right_bwd": {"left": {"linspace": [1.0, 9.0, 9], "shape": [3, 3]}, "grad": "ones",
                      "right_grad_once": v3, "right_grad_twice": v4, "cite": c3},
        "tol": 4.88e-4,
    }

    # ---- MatMulT (module commented out; numbers valid) matrix_matrix_mul_t/test.rs
    ls = lits("matrix_matrix_mul_t/test.rs", "forward", "forward")
    c["mm_t_forward"] = {
        "cite": cite("matrix_matrix_mul_t/test.rs", ls[0][0]),
        "left": ls[0][1], "left_shape": [3, 3],
        "right": ls[1][1], "right_shape": [2, 3],
        "out": ls[2][1], "out_shape": [3, 2], "tol": 4.88e-4,
    }
    ls = lits("matrix_matrix_mul_t/test.rs", "backward", "backward")
    c["mm_t_backward"] = {"cite": cite("matrix_matrix_mul_t/test.rs", ls[0][0]),
                          "literals": [v for _, v in ls], "tol": 4.88e-4}

    # ---- Softmax / LogSoftmax (modules commented out; numbers valid)
    for op in ("softmax", "logsoftmax"):
        rel = f"{op}/test.rs"
        for fn, axis in (("forward_rows", 0), ("forward_columns", 1)):
            ls = lits(rel, fn, "forward")
            c[f"{op}_{fn}"] = {"cite": cite(rel, ls[0][0]), "axis": axis, "shape": [3, 3],
                               "input": ls[0][1], "out": ls[1][1], "tol": 4.88e-4}
        for fn, axis in (("backward_rows", 0), ("backward_columns", 1)):
            ls = lits(rel, fn, "backward")
            c[f"{op}_{fn}"] = {"cite": cite(rel, ls[0][0]), "axis": axis, "shape": [3, 3],
                               "literals": [v for _, v in ls], "tol": 4.88e-4}

    # ---- ReLU / Sum / Mean / SquaredError / Transpose / MultiConcatenate (commented out)
    for rel, fn, mod in (
        ("relu/test.rs", "forward", "forward"), ("relu/test.rs", "backward", "backward"),
        ("sum/test.rs", "forward", "forward"), ("sum/test.rs", "backward", "backward"),
        ("mean/test.rs", "forward", "forward"), ("mean/test.rs", "backward", "backward"),
        ("transpose/test.rs", "forward", "forward"), ("transpose/test.rs", "backward", "backward"),
        ("squared_error/test.rs", "mean", None), ("squared_error/test.rs", "sum", None),
    ):
        try:
            ls = lits(rel, fn, mod)
        except KeyError as e:
            print(f"  (skip {rel}::{fn}: {e})", file=sys.stderr)
            continue
        body, _l0 = _fn_body(_read(rel), fn, mod)
        c[f"{rel.split('/')[0]}_{fn}"] = {"cite": cite(rel, ls[0][0]) if ls else cite(rel, 1),
                                          "literals": [v for _, v in ls], "scalars": scalars(body),
                                          "tol": 4.88e-4}

    # ---- pointwise unary nodes ("next" row): forward = (input, expected), backward = (.., input,
    # grad, grad, once, twice, ..) in the old-API files; exp/logn are enabled new-API tests
    for rel_dir in ("negation", "sqrt", "sigmoid", "tanh", "softplus", "leaky_relu", "power"):
        rel = f"{rel_dir}/test.rs"
        for fn, mod in (("forward", "forward"), ("backward", "backward"), ("backward_negative_exp", "backward")):
            try:
                body, l0 = _fn_body(_read(rel), fn, mod)
            except KeyError:
                continue
            ls = _literals(body, l0)
            ints_ = [int(v) for v in re.findall(r"(?:Power|PowerBackward)::new\([^;]*?,\s*(-?\d+)\s*\)", body)]
            c[f"{rel_dir}_{fn}"] = {"cite": cite(rel, l0), "literals": [v for _, v in ls], "exp": ints_[:1], "tol": 4.88e-4}

    # ---- broadcast binaries (enabled tests)
    for op in ("addition", "subtraction", "multiplication", "division"):
        rel = f"{op}/test.rs"
        src = _read(rel)
        names = re.findall(r"fn\s+(\w+)\s*\(", src)
        for mod in ("forward", "backward"):
            try:
                m = re.search(r"^mod\s+%s\s*\{" % mod, src, re.M)
            except Exception:
                m = None
            if not m:
                continue
            for fn in dict.fromkeys(names):
                try:
                    body, l0 = _fn_body(src, fn, mod)
                except KeyError:
                    continue
                ls = _literals(body, l0)
                if mod == "forward" and fn not in ("base_case", "left_broadcast", "right_broadcast"):
                    continue
                if mod == "backward" and not fn.endswith(("_base_case", "_reduction")):
                    continue
                lin = [[float(a), float(b), int(n)] + _tuple(sh) for a, b, n, sh in re.findall(
                    r"linspace\((-?\d+\.?\d*),\s*(-?\d+\.?\d*),\s*(\d+)\)\.into_shape\(\(([0-9, ]+)\)\)", body)]
                c[f"{op}_{mod}_{fn}"] = {"cite": cite(rel, l0), "constructors": from_elems(body),
                                         "linspace_start_stop_n_shape": lin, "tol": 4.88e-4}

    # ---- pad (enabled): pad/zero/test.rs, pad/constant/test.rs
    for mode in ("zero", "constant"):
        rel = f"pad/{mode}/test.rs"
        src = _read(rel)
        for fn in dict.fromkeys(re.findall(r"fn\s+(\w+)\s*\(", src)):
            body, l0 = _fn_body(src, fn)
            ls = _literals(body, l0)
            c[f"pad_{mode}_{fn}"] = {"cite": cite(rel, l0), "literals": [v for _, v in ls],
                                     "constructors": from_elems(body),
                                     "padding": _tuple(re.search(r"&\[([0-9, ]+)\]|\(([0-9, ]+)\)\s*\.into_dimension", body).group(0)) if re.search(r"&\[([0-9, ]+)\]|\(([0-9, ]+)\)\s*\.into_dimension", body) else None}

    # ---- pad modes (enabled): pad/reflective/test.rs, pad/replicative/test.rs — base = range(0, n) reshaped,
    # padded shape from `zeros(..)`, padding from `[..].into_dimension()`, expectation = the array! literal
    for mode in ("reflective", "replicative"):
        rel = f"pad/{mode}/test.rs"
        src = _read(rel)
        for fn in ("test_1d", "test_2d", "test_3d"):
            body, l0 = _fn_body(src, fn)
            ls = _literals(body, l0)
            n = int(float(re.search(r"range\(0\.?0?,\s*(\d+)\.?0?,", body).group(1)))
            base_shape = re.search(r"into_shape\(\(([0-9, ]+)\)\)", body)
            padded_shape = re.search(r"zeros\(\(?([0-9, ]+)\)?\)", body)
            c[f"pad_{mode}_{fn}"] = {"cite": cite(rel, l0), "arange": n,
                                     "base_shape": _tuple(base_shape.group(1)) if base_shape else [n],
                                     "padded_shape": _tuple(padded_shape.group(1)),
                                     "padding": _tuple(re.search(r"\[([0-9, ]+)\]\s*\.into_dimension", body).group(1)),
                                     "expected": ls[-1][1]}

    # ---- losses: bce + absolute_error (enabled, forward/backward mods); nll, kldiv, bce_with_logits
    # (test modules commented out — stale API, numbers valid): literals / arr0 scalars / linspace recipes in
    # source order
    def _loss_case(rel, fn, mod):
        body, l0 = _fn_body(_read(rel), fn, mod)
        lin = [[float(a), float(b), int(n)] + _tuple(sh) for a, b, n, sh in re.findall(
            r"linspace\((-?\d+\.?\d*),\s*(-?\d+\.?\d*),\s*(\d+)\)\.into_shape\(\(([0-9, ]+)\)\)", body)]
        return {"cite": cite(rel, l0), "literals": [v for _, v in _literals(body, l0)], "scalars": scalars(body),
                "linspace_start_stop_n_shape": lin,
                "from_elem": [[float(v)] + _tuple(sh) for sh, v in re.findall(r"from_elem\(\(([0-9, ]+)\),\s*(-?\d+\.?\d*)\)", body)],
                "tol": 4.88e-4}
    for loss in ("bce", "absolute_error"):
        for mod in ("forward", "backward"):
            for fn in ("base_case_mean", "base_case_sum"):
                c[f"{loss}_{mod}_{fn}"] = _loss_case(f"{loss}/test.rs", fn, mod)
    for loss in ("nll", "kldiv", "bce_with_logits"):
        for fn in ("mean", "sum"):
            c[f"{loss}_{fn}"] = _loss_case(f"{loss}/test.rs", fn, None)

    # ---- matrix-vector / vector-matrix / vector-vector (test modules commented out — stale API, numbers
    # valid): all numeric literals + arr0 scalars of `forward` / `backward`, in source order
    for node in ("matrix_vector_mul", "vector_matrix_mul", "vector_vector_mul"):
        rel = f"{node}/test.rs"
        for mod in ("forward", "backward"):
            body, l0 = _fn_body(_read(rel), mod, mod)
            c[f"{node}_{mod}"] = {"cite": cite(rel, l0), "literals": [v for _, v in _literals(body, l0)],
                                  "scalars": scalars(body), "tol": 4.88e-4}

    # ---- chunk (enabled)
    rel = "chunk/test.rs"
    src = _read(rel)
    for mod in ("forward", "backward"):
        for fn in dict.fromkeys(re.findall(r"fn\s+(\w+)\s*\(", src)):
            try:
                body, l0 = _fn_body(src, fn, mod)
            except KeyError:
                continue
            ls = _literals(body, l0)
            if ls:
                c[f"chunk_{mod}_{fn}"] = {"cite": cite(rel, l0), "literals": [v for _, v in ls]}
    return c


def data_cases():
    """neuronika-data/src/test.rs: the CSV text of both modules and, per test, the `vec![..]` literals of its
    assertions in source order (records / labels of kfold folds, batches, splits)."""
    path = os.path.join(REF, "neuronika-data", "src", "test.rs")
    src = open(path).read()
    out = {}
    for mod in ("dataset", "labeled_dataset"):
        m = re.search(r"^mod\s+%s\s*\{" % mod, src, re.M)
        body_start = m.end()
        st = re.search(r'static DATASET: &str = "\\\n(.*?)";', src[body_start:], re.S)
        csv = "".join(line.strip().rstrip("\\") for line in st.group(1).splitlines()).replace("\\n", "\n")
        case = {"cite": f"neuronika-data/src/test.rs:{src.count(chr(10), 0, body_start) + 1}", "csv": csv, "tests": {}}
        lab = re.search(r"with_labels\(&\[([0-9, ]+)\]\)", src[body_start:])
        if mod == "labeled_dataset":
            case["label_columns"] = _tuple(lab.group(1))
        for fn in ("from_reader", "kfold", "batch", "split", "drop_last"):
            body, l0 = _fn_body(src, fn, mod)
            case["tests"][fn] = {"line": l0, "literals": [v for _, v in _literals(body, l0)],
                                 "shapes": [_tuple(t) for t in re.findall(r"from_shape_vec\(\s*\(([0-9, ]+)\)", body)]}
        out[mod] = case
    return out


def quickstart_weights():
    """The 3->5->5->1 MLP weights embedded as JSON in examples/quickstart.rs:53-169 (config C1)."""
    src = open(os.path.join(REF, "examples", "quickstart.rs")).read()
    m = re.search(r'r#"(.*?)"#', src, re.S)
    line = src.count("\n", 0, m.start()) + 1
    model = json.loads(m.group(1))
    out = {"cite": f"examples/quickstart.rs:{line}"}
    for name in ("lin1", "lin2", "lin3"):
        for p in ("weight", "bias"):
            t = model[name][p]
            out[f"{name}.{p}"] = {"dim": t["dim"], "data": t["data"]}
    return out


def main():
    if not os.path.isdir(NODE):
        sys.exit(f"reference not found at {REF}; fixtures can only be regenerated in the authoring container")
    conv, im2col = conv_cases()
    fixtures = {
        "_generated_by": "tests/golden/extract_reference_fixtures.py",
        "_reference": "neuronika/neuronika @ /root/reference (v0.2.0 workspace snapshot)",
        "convolution": conv,
        "im2col": im2col,
        "nodes": other_cases(),
        "quickstart_mlp": quickstart_weights(),
        "data": data_cases(),
    }
    with open(OUT, "w") as f:
        json.dump(fixtures, f, indent=1)
    print(f"wrote {OUT}: {len(conv)} conv cases, {len(fixtures['nodes'])} node cases")


if __name__ == "__main__":
    main()
