"""Build a VARIANT of libneuronika_hip.so with extra -D flags for same-box A/B measurements.

    python benchmarks/ab_build.py kc_linear -DNK_AB_KC_LINEAR      -> benchmarks/_ab/kc_linear.so
    NEURONIKA_HIP_LIB=benchmarks/_ab/kc_linear.so python benchmarks/ab_gemm.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neuronika_amd", "csrc")


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "benchmarks", "_ab")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-I" + CSRC, *flags, "-c", os.path.join(CSRC, src), "-o", obj], check=True, stderr=subprocess.DEVNULL)
        objs.append(obj)
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-L/opt/rocm/lib", "-lrccl",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
