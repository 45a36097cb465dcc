"""Builds the gfx950 C-ABI library `locate_amd/csrc/liblocate_hip.so` in-tree with hipcc.

No libtorch linkage: the library only needs the HIP runtime (libamdhip64.so.7, resolved at load time to
the copy PyTorch already has in the process).  Run as `python -m locate_amd.build` or via
`__graft_entry__.build()`."""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liblocate_hip.so")
# debug variant: the same objects, except that conv.hip is compiled with -DLOCATE_DEBUG_KNOBS (the LOCATE_DISABLE kernel-flavour
# switch used by tests/test_gpu_ops.py::test_bf16x6_kernels_match_fp32_mfma_kernels and the A/B tools).  Loaded only when
# LOCATE_HIP_DEBUG_LIBRARY=1 is set before `import locate_amd`; the product library has no such switch compiled in.
LIB_DBG = os.path.join(CSRC, "liblocate_hip_dbg.so")
DBG_SOURCES = ["conv.hip", "convwin.hip"]
SOURCES = ["runtime.hip", "elementwise.hip", "norm.hip", "softmax.hip", "resample.hip", "spectral.hip", "conv.hip", "convwin.hip", "convfp8.hip",
           "grouped.hip", "nadam.hip", "loss.hip", "finalise.hip", "parallel.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    """every header under csrc/: an edit to any of them rebuilds every object (they are few and small)"""
    return [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]


def needs_build():
    if not (os.path.exists(LIB) and os.path.exists(LIB_DBG)):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_DBG))
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
             "-I", CSRC]
    procs = []
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    for src in _sources():
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            continue
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    dbg_objs = list(objs)
    for src in DBG_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", "_dbg.o"))
        dbg_objs[dbg_objs.index(os.path.join(CSRC, src.replace(".hip", ".o")))] = obj
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            continue
        cmd = [hipcc] + flags + ["-DLOCATE_DEBUG_KNOBS", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src + " (debug)", subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_DBG] + dbg_objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
