"""In-tree build of libkapre_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "kapre_hip.hip")
# kapre_hip.hip is one translation unit that includes every header of csrc/ (FFT building blocks,
# then one header per kernel family) and the public C ABI header
HDRS = sorted(os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".h")) + \
       [os.path.join(os.path.dirname(_HERE), "include", "kapre_hip.h")]
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libkapre_hip.so")
# torch-free consumer of the C ABI (tests/abi/test_abi.cc): built here so that it travels to the GPU box
ABI_SRC = os.path.join(os.path.dirname(_HERE), "tests", "abi", "test_abi.cc")
ABI_EXE = os.path.join(os.path.dirname(_HERE), "tests", "abi", "test_abi")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in [SRC] + HDRS)


def build_abi_test(force: bool = False, verbose: bool = False) -> str:
    """tests/abi/test_abi.cc -> tests/abi/test_abi (host program: HIP runtime + libkapre_hip.so through the
    public header only; rpath relative to the binary so the pair can be moved together)."""
    if not os.path.exists(ABI_SRC):
        return ""
    hdr = os.path.join(os.path.dirname(_HERE), "include", "kapre_hip.h")
    if not force and os.path.exists(ABI_EXE) and \
            os.path.getmtime(ABI_EXE) >= max(os.path.getmtime(ABI_SRC), os.path.getmtime(hdr), os.path.getmtime(LIB)):
        return ABI_EXE
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-x", "c++", ABI_SRC, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-o", ABI_EXE, "-L" + LIB_DIR, "-lkapre_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../../kapre_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return ABI_EXE


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile kapre_amd/csrc/kapre_hip.hip -> kapre_amd/lib/libkapre_hip.so (gfx950 only), then the
    torch-free ABI test program."""
    if not force and not _stale():
        build_abi_test(force, verbose)
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-pass-failed", "-DKPR_RING_DEPTH=3", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    build_abi_test(True, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
