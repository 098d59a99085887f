"""Build libsynthhip.so (hipcc, gfx950) in-tree.

    python -m synthesizer_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with the
repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libsynthhip.so"
SOURCES = ["runtime.hip", "osc.hip", "pcm.hip", "pcm_ops.hip", "dist.hip"]
HEADERS = ["common.hpp", "devmath.hpp", "../../include/synthhip.h"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # fused multiply-add only where fma() is written (bit-exact PCM paths)
    "-fno-fast-math",
    "-fgpu-rdc" if False else "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
    "-I/opt/rocm/include",
]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [str(CSRC / s) for s in SOURCES] + ["-o", str(LIB), "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
