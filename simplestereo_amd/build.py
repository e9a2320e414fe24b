"""Build libssamd.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libssamd.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
               "-Wno-int-to-pointer-cast"]      # (host pass of the LDS byte-offset casts in asw_wave_kernel.hip.h)


def _sources():
    return sorted(os.path.join(_SRC, f) for f in os.listdir(_SRC) if f.endswith((".hip", ".h", ".inc")))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _sources() + [os.path.join(os.path.dirname(_PKG), "include", "ssamd.h")]
    return any(os.path.getmtime(s) > t for s in deps if os.path.exists(s))


# translation units: (source, extra flags).  The LLVM scheduling strategy is a per-translation-unit option and the kernel families
# disagree on the best one (profiles/r05_llvm_sched_strategy_ab.txt), so two of them are compiled apart (round 5).
UNITS = [("ssamd_api.hip", []),
         ("asw_pipe_tu.hip", ["-mllvm", "--amdgpu-sched-strategy=max-memory-clause"]),
         ("asw_wave6_tu.hip", ["-mllvm", "--amdgpu-sched-strategy=max-ilp"])]


def build_native(force=False, verbose=False):
    """Compile simplestereo_amd/csrc/*.hip into simplestereo_amd/libssamd.so."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libssamd.so cannot be built (ROCm toolchain required)")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    objs, procs = [], []
    for src, extra in UNITS:
        obj = os.path.join(_SRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + flags + extra + ["-c", "-o", obj, os.path.join(_SRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    for o in objs:
        os.remove(o)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
