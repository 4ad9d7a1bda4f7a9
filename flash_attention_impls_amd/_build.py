"""In-tree build of the HIP shared library (hipcc cross-compiles for gfx950 without a GPU).

The product is one C-ABI shared object, ``flash_attention_impls_amd/libfa_mi355.so``
(header: ``include/fa_mi355.h``).  It is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libfa_mi355.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)

SOURCES = [os.path.join(CSRC, "fa_capi.hip"), os.path.join(CSRC, "fa_bwd_capi.hip")]
DEPS = SOURCES + [os.path.join(CSRC, n) for n in ("fa_fwd_kernel.hpp", "fa_fwd_kernel16.hpp", "fa_bwd_kernel.hpp", "fa_bwd_dkdv_kernel.hpp",
                                                  "fa_capi_common.hpp")] + \
    [os.path.join(PKG_DIR, "..", "include", "fa_mi355.h")]

# -fno-slp-vectorize: SLP packs the softmax's scalar f32 adds into v_pk_add_f32 chains placed behind the
# MFMAs of a block (measured -2.6 % on the forward kernel); the kernel wants single-issue VALU fillers.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-slp-vectorize", "-Wno-unused-result"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", LIB_PATH + ".tmp", *SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


CLI_SRC = os.path.join(PKG_DIR, "cli", "fa_bench.cpp")
CLI_PATH = os.path.join(PKG_DIR, "cli", "fa_bench")


def build_cli(verbose: bool = False) -> str:
    """Compile the native bench/verify CLI (host-only C++ over the C ABI) next to its source."""
    build()
    cmd = [hipcc_path(), "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", CLI_SRC, "-o", CLI_PATH,
           "-L" + PKG_DIR, "-lfa_mi355", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (cli) failed:\n" + res.stdout + res.stderr)
    return CLI_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_cli(verbose=True))
