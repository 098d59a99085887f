#!/usr/bin/env python
"""Build a VARIANT of libsynthhip.so beside the shipped one (A/B timings, diagnostic builds):

    python tools/build_variant.py NAME [extra hipcc flags ...]     ->  synthesizer_amd/build/libsynthhip_NAME.so

Load it with SYNTHHIP_LIB=<that path>.  The shipped library (synthesizer_amd/libsynthhip.so) is not touched.
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from synthesizer_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out = B.HERE / "build" / ("libsynthhip_%s.so" % name)
    objdir = B.HERE / "build" / ("obj_" + name)
    objdir.mkdir(parents=True, exist_ok=True)
    defs = ['-DSH_SOURCE_HASH="%s"' % B.source_hash()] + extra
    jobs = []
    for src in B.SOURCES:
        obj = objdir / (src + ".o")
        jobs.append((src, obj, subprocess.Popen([B.HIPCC] + B.FLAGS + defs + ["-c", str(B.CSRC / src), "-o", str(obj)])))
    bad = [s for s, _o, p in jobs if p.wait() != 0]
    if bad:
        sys.exit("failed: " + " ".join(bad))
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for _s, o, _p in jobs] + ["-o", str(out), "-ldl"], check=True)
    import shutil
    shutil.rmtree(objdir, ignore_errors=True)
    print(out)


if __name__ == "__main__":
    main()
