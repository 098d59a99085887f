"""Build libsynthhip.so (hipcc, gfx950) and libsynthhost.so (g++, host-side table building) in-tree.

    python -m synthesizer_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with the
repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libsynthhip.so"
SOURCES = ["runtime.hip", "osc_bank.hip", "osc_render.hip", "osc_render_lean.hip", "osc_render_combined.hip", "osc_generate.hip", "osc_mixbus.hip", "osc_scan.hip", "pcm.hip", "pcm_ops.hip", "dist.hip"]
HEADERS = sorted(p.name for p in CSRC.glob("*.hpp")) + ["../../include/synthhip.h"]      # (every header of csrc/: a header left out of the hash is a stale library unnoticed)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",          # fused multiply-add only where fma() is written (bit-exact PCM paths)
    "-fno-fast-math",
    "-fgpu-rdc" if False else "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
    "-I/opt/rocm/include",
]


def extra_flags() -> list:
    """SYNTHHIP_BUILD_FLAGS: extra compiler flags of a diagnostic build (e.g. -DSH_DIAG: in-kernel timestamps; never the
    shipped library -- the flags are part of the embedded source hash)."""
    return os.environ.get("SYNTHHIP_BUILD_FLAGS", "").split()


def source_hash() -> str:
    """SHA-256 (16 hex digits) over the sources, headers and compiler flags the library is built from."""
    h = hashlib.sha256()
    for f in [CSRC / s for s in SOURCES] + [(CSRC / x).resolve() for x in HEADERS]:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(" ".join(FLAGS + extra_flags()).encode())
    return h.hexdigest()[:16]


def built_hash() -> str:
    """The hash embedded in libsynthhip.so (sh_version(): "... src:<hash>"), read from the file without loading it."""
    if not LIB.exists():
        return ""
    m = re.search(rb"synthhip [0-9.]+ \(gfx950\) src:([0-9a-f]{16})", LIB.read_bytes())
    return m.group(1).decode() if m else ""


def needs_build() -> bool:
    """Stale = the library's embedded source hash differs from the tree's (mtimes of a shipped .so mean nothing)."""
    return built_hash() != source_hash()


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    # one builder at a time: the ranks of a multi-GPU job all import the package at once, and a stale library must be compiled
    # by ONE of them (the others wait on the lock and then find it up to date); the result is moved into place atomically
    import fcntl
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or needs_build():
            tmp = LIB.with_suffix(".so.tmp%d" % os.getpid())
            objdir = HERE / "build" / ("obj%d" % os.getpid())
            objdir.mkdir(parents=True, exist_ok=True)
            defs = ['-DSH_SOURCE_HASH="%s"' % source_hash()] + extra_flags()
            # one hipcc per translation unit, all at once (no relocatable device code: every kernel lives in the unit that launches it)
            jobs = []
            for src in SOURCES:
                obj = objdir / (src + ".o")
                cmd = [HIPCC] + FLAGS + defs + ["-c", str(CSRC / src), "-o", str(obj)]
                if verbose:
                    print(" ".join(cmd), flush=True)
                jobs.append((src, obj, subprocess.Popen(cmd)))
            try:
                failed = [src for src, _obj, p in jobs if p.wait() != 0]
                if failed:
                    raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
                link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(obj) for _s, obj, _p in jobs] + ["-o", str(tmp), "-ldl"]
                if verbose:
                    print(" ".join(link), flush=True)
                subprocess.run(link, check=True)
                os.replace(tmp, LIB)
            finally:
                for _s, _obj, p in jobs:
                    if p.poll() is None:
                        p.kill()
                if tmp.exists():
                    tmp.unlink()
                import shutil
                shutil.rmtree(objdir, ignore_errors=True)
    return LIB


# ---- libsynthhost.so: host-side table building in native code (include/synthhost.h; plain C++, no HIP) ----------------------
HOST_LIB = HERE / "libsynthhost.so"
HOST_SOURCES = ["host_tables.cpp"]
HOST_HEADERS = ["../../include/synthhost.h"]
CXX = os.environ.get("CXX", "g++")
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall"]


def host_source_hash() -> str:
    h = hashlib.sha256()
    for f in [CSRC / s for s in HOST_SOURCES] + [(CSRC / x).resolve() for x in HOST_HEADERS]:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(" ".join(HOST_FLAGS).encode())
    return h.hexdigest()[:16]


def host_built_hash() -> str:
    if not HOST_LIB.exists():
        return ""
    m = re.search(rb"synthhost [0-9.]+ src:([0-9a-f]{16})", HOST_LIB.read_bytes())
    return m.group(1).decode() if m else ""


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """Compile libsynthhost.so when it is missing or its embedded source hash differs from the tree's (a second or two of g++)."""
    if not force and host_built_hash() == host_source_hash():
        return HOST_LIB
    import fcntl
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or host_built_hash() != host_source_hash():
            tmp = HOST_LIB.with_suffix(".so.tmp%d" % os.getpid())
            cmd = [CXX] + HOST_FLAGS + ['-DSHH_SOURCE_HASH="%s"' % host_source_hash()] + [str(CSRC / s) for s in HOST_SOURCES] + ["-o", str(tmp)]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, HOST_LIB)
            finally:
                if tmp.exists():
                    tmp.unlink()
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_host(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(HOST_LIB)
