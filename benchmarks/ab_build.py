"""Build a VARIANT of libneuronika_hip.so with extra -D flags for same-box A/B measurements.

    python benchmarks/ab_build.py kc_linear -DNK_AB_KC_LINEAR      -> benchmarks/_ab/kc_linear.so
    NEURONIKA_HIP_LIB=benchmarks/_ab/kc_linear.so python benchmarks/ab_gemm.py
    python benchmarks/ab_build.py nofold --sed 's/KFOLD_TILES = 32;/KFOLD_TILES = 1 << 20;/' nk_gemm.hip
        -> the same from a COPY of csrc/ with the sed expression applied to the named file (variants that are an edit, not a flag)
"""
import shutil
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
    csrc = CSRC
    if "--src" in flags:          # a prepared copy of csrc/ (variants that are more than a one-line edit)
        i = flags.index("--src")
        csrc = os.path.abspath(flags[i + 1])
        del flags[i:i + 2]
    while "--sed" in flags:
        i = flags.index("--sed")
        expr, fname = flags[i + 1], flags[i + 2]
        del flags[i:i + 3]
        if csrc == CSRC:
            csrc = os.path.join(out_dir, "src_" + name)
            shutil.rmtree(csrc, ignore_errors=True)
            shutil.copytree(CSRC, csrc)
        before = open(os.path.join(csrc, fname)).read()
        subprocess.run(["sed", "-i", expr, os.path.join(csrc, fname)], check=True)
        if open(os.path.join(csrc, fname)).read() == before:
            raise SystemExit(f"--sed {expr!r} changed nothing in {fname}")
    objs = []
    import concurrent.futures as cf
    def one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-I" + csrc, *flags, "-c", os.path.join(csrc, src), "-o", obj], check=True, stderr=subprocess.DEVNULL)
        return obj
    with cf.ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))))
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-L/opt/rocm/lib", "-lrccl",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
