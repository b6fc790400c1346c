"""In-tree build of the native pieces (explicit compiler invocations, no JIT cache).

  build_hip()     hipcc --offload-arch=gfx950 -> localexpstereo_amd/csrc/liblocalexp_hip.so   (the product)
  build_host()    g++ -> localexpstereo_amd/host/les_host_demo (C++ host adapter self-test, links the .so)
  build_oracle()  g++ -> oracle/libles_oracle.so        (CPU restatement: test infrastructure only)
  build_sim()     g++ -> tools/hipsim/liblocalexp_sim.so (CPU SIMT simulator build of the same sources:
                  test infrastructure only, never loaded by the package)
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
HOST = os.path.join(_HERE, "host")
HIP_SO = os.path.join(CSRC, "liblocalexp_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
HIPCC_LIBS = ["-ldl", "-lpthread"]      # dlopen of librccl at run time (les_hip_exchange_tiles): no link-time dependency on RCCL


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build liblocalexp_hip.so for gfx950)")


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]      # les_hip.hip (one translation unit) + its parts + the kernel headers
    srcs.append(os.path.join(ROOT, "include", "localexp_hip.h"))
    srcs += [os.path.join(HOST, f) for f in ("ResidualCut.h", "GridMaxFlow.h", "GridPushRelabel.h", "BandPool.h")]      # the host cores' finisher of the tiled max-flow
    if not force and _newer(HIP_SO, srcs):
        return HIP_SO
    cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "les_hip.hip"), "-o", HIP_SO] + HIPCC_LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return HIP_SO


PLAIN_SO = os.path.join(CSRC, "libles_plain.so")
PLAIN_FLAGS = ["-DLES_MARCH_SCAN_PLAIN", "-DLES_MARCH_STATS_PLAIN", "-DLES_SIMT_PLAIN"]


def build_hip_plain(force=False):
    """hipcc -> localexpstereo_amd/csrc/libles_plain.so: the SAME sources with every inline-assembly path of the march kernel replaced by
    plain C++ (the DPP scan, the tied-destination statistics loads with hand-kept vmcnt, the SDWA / cvt / med3 / mad64 primitives).  Test
    infrastructure: the GPU tests run the product and this build on the same inputs and require bit-identical outputs (the CPU simulator
    cannot see the assembly).  Never loaded by the package."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(ROOT, "include", "localexp_hip.h")]
    srcs += [os.path.join(HOST, f) for f in ("ResidualCut.h", "GridMaxFlow.h", "GridPushRelabel.h", "BandPool.h")]
    if not force and _newer(PLAIN_SO, srcs):
        return PLAIN_SO
    cmd = [_hipcc()] + HIPCC_FLAGS + PLAIN_FLAGS + [os.path.join(CSRC, "les_hip.hip"), "-o", PLAIN_SO] + HIPCC_LIBS
    subprocess.check_call(cmd, cwd=CSRC)
    return PLAIN_SO


def build_host(force=False):
    exe = os.path.join(HOST, "les_host_demo")
    srcs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".h", ".cpp"))] if os.path.isdir(HOST) else []
    main = os.path.join(HOST, "les_host_demo.cpp")
    if not os.path.exists(main):
        return None
    if not force and _newer(exe, srcs + [HIP_SO]):
        return exe
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), main, "-o", exe,
           "-L", CSRC, "-llocalexp_hip", "-Wl,-rpath," + CSRC, "-ldl"]
    subprocess.check_call(cmd, cwd=HOST)
    return exe


HOST_SO = os.path.join(HOST, "liblocalexp_host.so")


def build_host_lib(force=False):
    """g++ -> localexpstereo_amd/host/liblocalexp_host.so: C ABI of the host graph-cut fusion (include/localexp_host.h)."""
    src = os.path.join(HOST, "les_gc.cpp")
    deps = [src, os.path.join(ROOT, "include", "localexp_host.h")] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not force and _newer(HOST_SO, deps):
        return HOST_SO
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", HOST_SO]
    subprocess.check_call(cmd, cwd=HOST)
    return HOST_SO


def build_host_selfcheck(force=False):
    """tests/cpp/gc_selfcheck: graph-cut host logic on a CPU-only test energy (test infrastructure)."""
    src = os.path.join(ROOT, "tests", "cpp", "gc_selfcheck.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "gc_selfcheck")
    if not os.path.exists(src):
        return None
    deps = [src] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not force and _newer(exe, deps + [HIP_SO]):
        return exe
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", HOST, src, "-o", exe,
           "-L", CSRC, "-llocalexp_hip", "-Wl,-rpath," + CSRC]
    subprocess.check_call(cmd, cwd=ROOT)
    return exe


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    args = ["make", "-C", d] + (["-B"] if force else []) + ["libles_oracle.so"]
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return os.path.join(d, "libles_oracle.so")


def build_sim(force=False):
    d = os.path.join(ROOT, "tools", "hipsim")
    args = ["make", "-C", d] + (["-B"] if force else []) + ["liblocalexp_sim.so"]
    subprocess.check_call(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(d, "liblocalexp_sim.so")
