"""neuronika_amd — MI355X (gfx950) dense-tensor backend for neuronika's Var/VarDiff op graph.

Layers:
  * `csrc/` + `include/neuronika_hip.h` — hand-written HIP kernels behind a C ABI
    (`lib/libneuronika_hip.so`): the drop-in boundary;
  * `capi`  — ctypes binding of that C ABI (one function per entry point);
  * `_tape` — the C++ host mirror of neuronika's tape (`host/neuronika.hpp`): Var, VarDiff,
    nn.Linear / Conv2d / MultiheadAttention, optim.SGD, dp.GradientSync.

There is no CPU fallback: importing `capi` or `tape` without the built HIP library raises.
"""
__all__ = ["capi", "tape", "build"]
__version__ = "0.1.0"


def __getattr__(name):
    import importlib
    if name == "capi":
        return importlib.import_module("neuronika_amd.capi")
    if name == "tape":
        try:
            return importlib.import_module("neuronika_amd._tape")
        except ImportError as e:
            raise ImportError("neuronika_amd._tape is not built: run `python -m neuronika_amd.build` "
                              "(the HIP backend has no CPU fallback)") from e
    raise AttributeError(name)
