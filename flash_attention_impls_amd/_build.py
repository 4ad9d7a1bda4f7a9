"""In-tree build of the HIP shared library (hipcc cross-compiles for gfx950 without a GPU).

The product is one C-ABI shared object, ``flash_attention_impls_amd/libfa_mi355.so``
(header: ``include/fa_mi355.h``).  It is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libfa_mi355.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)

SOURCES = [os.path.join(CSRC, "fa_capi.hip"), os.path.join(CSRC, "fa_bwd_capi.hip"), os.path.join(CSRC, "fa_diag.hip")]
# every file under csrc/ is a dependency (kernel headers are included from the two sources above): a header added later is
# picked up without touching this list, and the digest below covers exactly what the compiler reads
DEPS = SOURCES + sorted(os.path.join(CSRC, n) for n in os.listdir(CSRC)
                        if n.endswith((".hpp", ".h", ".hip")) and os.path.join(CSRC, n) not in SOURCES) + \
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


DIGEST_PATH = LIB_PATH + ".digest"      # sha256 of the sources and flags the library was built from (travels with it)


def sources_digest() -> str:
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in DEPS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def unit_sources(unit: str) -> list:
    """The files under csrc/ that one translation unit's DEVICE code is made of: the unit and every csrc header it includes,
    transitively (`#include "x.hpp"`).  include/fa_mi355.h -- C declarations and the version number, host side only -- is not
    among them."""
    import re
    seen, todo = [], [os.path.join(CSRC, unit)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or os.path.dirname(f) != os.path.normpath(CSRC) or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f) as fh:
            todo += [os.path.join(os.path.dirname(f), m) for m in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M)]
    return sorted(seen)


def unit_digest(unit: str) -> str:
    """sha256 of the compiler flags and of `unit_sources(unit)`: what a kernel of that translation unit was built from.
    Counter measurements (profiles/hbm_traffic.json) are stamped with the digest of the unit their kernel lives in, so that an
    edit to the backward's sources does not invalidate what was measured on the forward kernels, and vice versa."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in sorted({f for u in unit.split(",") for f in unit_sources(u)}):       # "a.hip,b.hip": a measurement over kernels of both
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True if the library is missing or was built from other sources.  Decided by content (a digest written next to the
    library), not by modification times: a copy of the tree to another machine may reorder those, and N ranks
    starting together must not all decide to rebuild."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(DIGEST_PATH) as f:
            return f.read().strip() != sources_digest()
    except OSError:
        t = os.path.getmtime(LIB_PATH)       # no digest (library built by hand): fall back to modification times
        return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if missing or built from other sources; returns its path.  Safe to call from several
    processes at once: one builds (file lock, private temporary, atomic rename), the others wait and find it done."""
    if not force and not is_stale():
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():          # built by another process while this one waited
                return LIB_PATH
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", tmp, *SOURCES]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIB_PATH)
            with open(DIGEST_PATH + ".tmp", "w") as f:
                f.write(sources_digest() + "\n")
            os.replace(DIGEST_PATH + ".tmp", DIGEST_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
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
