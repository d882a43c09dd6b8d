"""Builds libznippy_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libznippy_hip.so")


def sources():
    # *.hip: kernels + the C ABI; host/*.cpp: host code that uses the HIP runtime; host/*.cc: plain C++ (no device pass)
    return (sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp"))) +
            sorted(glob.glob(os.path.join(CSRC, "host", "*.cc"))))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = (sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "host", "*.h")) +
            glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        if src.endswith(".cc"):
            cmd = [hipcc, "-x", "c++", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj, "-Wall"]
        else:
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o", obj,
                   "-Wall", "-Wno-unused-function"] + os.environ.get("ZN_CFLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
