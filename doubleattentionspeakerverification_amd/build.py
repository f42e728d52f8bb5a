"""Build libdasv_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

No torch.utils.cpp_extension (it hipifies), no cmake: one hipcc command per
translation unit, linked into doubleattentionspeakerverification_amd/lib/.
hipcc cross-compiles gfx950 code objects without a GPU, so this also runs in
the CPU-only build container.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libdasv_hip.so"
# the same sources with -DDASV_DEBUG: the developer surface of include/dasv_hip.h (tuning knobs, read-floor probe) lives ONLY in
# this second library; the product library exports none of it
DEBUG_LIBNAME = "libdasv_hip_dbg.so"
DEBUG_SOURCES = ["dmha_fwd.hip", "conv3x3.hip"]        # the translation units that contain DASV_DEBUG code
SOURCES = ["dmha_fwd.hip", "dmha_bwd.hip", "conv3x3.hip", "conv3x3_wgrad.hip", "frontend_misc.hip", "amsoftmax.hip", "adam.hip"]
HEADERS = ["dasv_common.hpp", os.path.join("..", "..", "include", "dasv_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _stamp():
    h = hashlib.sha256()
    for rel in SOURCES + HEADERS:
        path = os.path.join(CSRC, rel)
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(rel.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def debug_lib_path():
    return os.path.join(LIBDIR, DEBUG_LIBNAME)


def is_current():
    stamp = os.path.join(LIBDIR, "build.stamp")
    return (os.path.exists(lib_path()) and os.path.exists(debug_lib_path()) and os.path.exists(stamp)
            and open(stamp).read().strip() == _stamp())


def build(force=False, verbose=True):
    """Compile every .hip source for gfx950 and link the shared library.  Returns its path."""
    if not force and is_current():
        return lib_path()
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, dbg_objs = [], []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
        if verbose:
            print("[dasv build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
        if src in DEBUG_SOURCES:
            dobj = os.path.join(LIBDIR, src.replace(".hip", "_dbg.o"))
            cmd = [hipcc] + FLAGS + ["-DDASV_DEBUG", "-c", path, "-o", dobj]
            if verbose:
                print("[dasv build]", " ".join(cmd), flush=True)
            procs.append((src + " (debug)", subprocess.Popen(cmd, cwd=CSRC)))
            dbg_objs.append(dobj)
        else:
            dbg_objs.append(obj)
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    for out, these in ((lib_path(), objs), (debug_lib_path(), dbg_objs)):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + these
        if verbose:
            print("[dasv build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    with open(os.path.join(LIBDIR, "build.stamp"), "w") as f:
        f.write(_stamp())
    return lib_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
