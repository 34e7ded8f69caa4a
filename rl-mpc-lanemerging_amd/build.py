"""Build the gfx950 shared library (``libstmpc.so``) in-tree with hipcc.

The library is the product's only compute path; importing the package does not build
anything, and every solver entry fails loudly if the library or a GPU is missing.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libstmpc.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    # bit-exactness with the reference: one IEEE op per source op, no reassociation
    "-ffp-contract=off", "-fno-fast-math",
    "-fPIC", "-shared",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def source_hash():
    """sha256[:16] over the library's sources (csrc/* and include/stmpc.h, by name).  Compiled into the library
    (``stmpc_backend_info`` ends with ``src=<hash>``) and stamped into profiles/*/measured.json, so that counters
    taken from another build of the kernels are recognisable as stale."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    files.append(os.path.join(REPO_ROOT, "include", "stmpc.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO_ROOT, "include", "stmpc.h")]
    return any(os.path.getmtime(s) > lib_m for s in srcs)


def build(force=False, verbose=False):
    """Compile csrc/stmpc.hip for gfx950 into ``libstmpc.so`` next to this file."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + ['-DSTMPC_SRC_HASH="%s"' % source_hash(), "-I" + os.path.join(REPO_ROOT, "include"),
                                         os.path.join(CSRC, "stmpc.hip"), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB_PATH
