"""Build the native libraries in-tree (so they travel to the GPU box with gpurun).

    libjpeg_gpu_amd.so  the product: host entropy stage + layout (C, gcc),
                        HIP kernels for gfx950 + C-ABI wrappers + plugin vtable +
                        pipeline (hipcc).  Kernels are built -ffp-contract=off and
                        the emitted ISA is checked for fused multiply-adds
                        (FMA contraction changes results).
    libjga_synth.so     synthetic-JPEG writer used by tests and bench.py.
    jpeg_gpu_hip        headless harness with the reference program's options
                        (csrc/harness.c, plain C over the C-ABI).

Usage: python -m jpeg_gpu_amd.build [--force]
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"

C_SOURCES = ["layout.c", "entropy.c", "libjpeg_vtbl.c", "band.c"]
HIP_SOURCES = ["idct_kernels.hip", "huff_kernels.hip", "pack_kernels.hip", "unstuff_kernels.hip", "copy_kernel.hip"]   # device code: hipcc
CXX_SOURCES = ["device_api.cpp", "vtbl.cpp", "pipeline.cpp", "huff_prepare.cpp",
               "huff_api.cpp"]                           # host only: g++ + HIP API
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
LIB = os.path.join(HERE, "libjpeg_gpu_amd.so")
# The same objects with layout.c compiled -DJGA_TUNING: the A/B knobs of rounds 1-3 (JGA_* environment
# variables, csrc/jga_tune.h) exist in this one only.  For tests of alternate code paths and tools/.
TUNING_LIB = os.path.join(HERE, "libjpeg_gpu_amd_tuning.so")
TUNING_HARNESS = os.path.join(HERE, "jpeg_gpu_hip_tuning")
SYNTH_LIB = os.path.join(HERE, "libjga_synth.so")
HARNESS = os.path.join(HERE, "jpeg_gpu_hip")            # headless harness (csrc/harness.c)

FMA_RE = re.compile(r"^\s+(v_fma\w*|v_fmac\w*|v_mad_f32\w*|v_mac_f32\w*|v_pk_fma\w*|v_mad_legacy\w*)\b",
                    re.M)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build step failed: " + cmd[0])
    return r.stdout


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def check_no_fma(asm_path):
    """The IDCT must stay operation-for-operation IEEE binary32 (no contraction)."""
    text = open(asm_path).read()
    hits = FMA_RE.findall(text)
    if hits:
        raise RuntimeError("fused multiply-add found in %s: %s" %
                           (asm_path, sorted(set(hits))))
    return len(text)


def build(force=False, verbose=False):
    """Compile what is out of date.  Safe to call from several processes at once (one rank per
    GPU calls it): an exclusive file lock serialises them, the others find everything built."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    headers = [os.path.join(CSRC, h) for h in ("jga_internal.h", "kernel_params.h", "huff_common.h",
                                               "huff_kernels.h", "huff_prepare.h", "pack_params.h",
                                               "libjpeg8_abi.h", "unstuff_kernels.h", "jga_tune.h", "host_wait.h")]
    headers.append(os.path.join(HERE, "..", "include", "jpeg_gpu_amd.h"))
    all_src = [os.path.join(CSRC, s) for s in C_SOURCES + HIP_SOURCES + CXX_SOURCES] + headers

    synth_src = os.path.join(CSRC, "synth_encode.c")
    if force or not _newer(SYNTH_LIB, [synth_src]):
        _run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared",
              "-fvisibility=hidden", "-o", SYNTH_LIB, synth_src, "-lm"])

    if not force and _newer(LIB, all_src) and _newer(TUNING_LIB, all_src):
        _build_harness(force)
        return LIB
    hipcc = _hipcc()
    objs = []
    tuning_objs = []
    for s in C_SOURCES:
        o = os.path.join(OBJ, s + ".o")
        cc = ["gcc", "-std=gnu11", "-O3", "-Wall", "-Wextra", "-fPIC",
              "-fvisibility=hidden", "-c", os.path.join(CSRC, s)]
        _run(cc + ["-o", o])
        objs.append(o)
        if s == "layout.c":                      # (where jga_tune() lives)
            o = os.path.join(OBJ, s + ".tuning.o")
            _run(cc + ["-DJGA_TUNING", "-o", o])
        tuning_objs.append(o)
    for s in CXX_SOURCES:
        o = os.path.join(OBJ, s + ".o")
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-fPIC", "-fvisibility=hidden",
              "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
              "-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    for s in HIP_SOURCES:
        o = os.path.join(OBJ, s + ".o")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
               "-fvisibility=hidden", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall",
               "-c", os.path.join(CSRC, s), "-o", o]
        if s.endswith(".hip"):
            cmd.insert(1, "-save-temps=obj")
        cmd[1:1] = os.environ.get("JGA_EXTRA_HIPFLAGS", "").split()      # tuning experiments
        _run(cmd)
        objs.append(o)
        if s == "idct_kernels.hip":      # the float path; huff_kernels is integer/byte code
            stem = s[:-4]
            asm = os.path.join(OBJ, "%s-hip-amdgcn-amd-amdhsa-%s.s" % (stem, ARCH))
            if not os.path.exists(asm):
                # -save-temps naming differs between hipcc versions: find it
                cands = [f for f in os.listdir(OBJ) if f.endswith(ARCH + ".s")]
                if not cands:
                    raise RuntimeError("kernel ISA listing not produced")
                asm = os.path.join(OBJ, cands[0])
            n = check_no_fma(asm)
            if verbose:
                print("ISA check ok: %s (%d bytes, no fma)" % (os.path.basename(asm), n))
    _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs +
         ["-lpthread", "-ldl"])
    _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", TUNING_LIB] + tuning_objs +
         objs[len(tuning_objs):] + ["-lpthread", "-ldl"])
    _build_harness(True)
    return LIB


def _build_harness(force):
    src = os.path.join(CSRC, "harness.c")
    for exe, lib, name in ((HARNESS, LIB, "jpeg_gpu_amd"), (TUNING_HARNESS, TUNING_LIB, "jpeg_gpu_amd_tuning")):
        if force or not _newer(exe, [src, lib]):
            _run(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-o", exe, src,
                  "-L" + HERE, "-l" + name, "-Wl,-rpath,$ORIGIN",
                  "-Wl,-rpath," + os.path.join(ROCM, "lib")])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
