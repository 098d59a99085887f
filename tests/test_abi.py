"""The C-ABI library: it loads without a GPU, exports every symbol include/synthhip.h declares, its
struct layouts match the ctypes/numpy mirrors, and the product path fails LOUDLY when no GPU is there
(no CPU fallback, no route through oracle/)."""
import ctypes
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "synthhip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(sh_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_loads():
    from synthesizer_amd import build as B
    lib = B.build()
    assert lib.exists()
    from synthesizer_amd import _native as N
    assert N.lib().sh_version().startswith(b"synthhip")


def test_every_declared_symbol_is_exported_and_bound():
    from synthesizer_amd import _native as N
    names = declared_functions()
    assert len(names) >= 40
    handle = ctypes.CDLL(str(N.LIB_PATH))
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    unbound = [n for n in names if n not in N.exported_symbols()]
    assert not unbound, unbound
    extra = [n for n in N.exported_symbols() if n not in names]
    assert not extra, extra


def test_struct_layouts_match_the_header(tmp_path):
    from synthesizer_amd import _native as N
    src = tmp_path / "sz.c"
    src.write_text('#include "%s"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(sh_segment), sizeof(sh_partial),'
                   'sizeof(sh_envelope), sizeof(sh_voice), offsetof(sh_voice, env), offsetof(sh_voice, gain_l),'
                   'offsetof(sh_voice, frequency), sizeof(sh_devinfo), offsetof(sh_voice, noise_seed), offsetof(sh_voice, noise_hold));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", str(src), "-o", str(exe)], check=True)      # the header is plain C
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    V = N.VOICE_DTYPE
    want = [N.SEGMENT_DTYPE.itemsize, N.PARTIAL_DTYPE.itemsize, N.ENVELOPE_DTYPE.itemsize, V.itemsize,
            V.fields["env"][1], V.fields["gain_l"][1], V.fields["frequency"][1], ctypes.sizeof(N.DevInfo),
            V.fields["noise_seed"][1], V.fields["noise_hold"][1]]
    assert got == want


def test_no_gpu_means_loud_failure_not_fallback():
    from synthesizer_amd import _native as N
    L = N.lib()
    if L.sh_device_count() > 0:
        pytest.skip("a GPU is visible here")
    assert L.sh_init(0) == N.SH_ERR_NOTINIT
    assert b"no HIP device" in L.sh_last_error()
    assert L.sh_sync() == N.SH_ERR_NOTINIT
    from synthesizer_amd.oscillators import Sine
    with pytest.raises(N.SynthHipError):
        Sine(440).render(16)
    from synthesizer_amd.sample import Sample
    a = Sample.from_raw_frames(b"\1\0\2\0", 2, 8000, 1)
    with pytest.raises(N.SynthHipError):
        a.mix(a.copy())
    with pytest.raises(N.SynthHipError):
        a.resample(4000)
    # pure arithmetic entry points work without a device
    assert L.sh_resample_out_frames(1000, 96000, 44100) == (999 * 147) // 320 + 1
    assert L.sh_resample_out_frames(8, 96000, 44100) == 4 and L.sh_resample_out_frames(0, 3, 7) == 0


def test_product_has_no_ml_framework():
    """north_star: 'no PyTorch'.  Not a mention of it anywhere under synthesizer_amd/ (sources; bench.py owns the launcher plumbing)."""
    for f in (ROOT / "synthesizer_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".h"):
            assert "torch" not in f.read_text().lower(), f


def test_product_never_imports_the_oracle():
    for py in (ROOT / "synthesizer_amd").rglob("*.py"):
        text = py.read_text()
        assert "import oracle" not in text and "from oracle" not in text, py
    for src in (ROOT / "synthesizer_amd" / "csrc").iterdir():
        assert "oracle" not in src.read_text().replace("the oracle", ""), src


def test_host_library_exports_what_its_header_declares(tmp_path):
    """include/synthhost.h / libsynthhost.so (host-side table building, plain C++): every declared function is exported, the header
    is plain C, and shh_segment has sh_segment's layout -- the records go into a bank's table of pieces as they are."""
    from synthesizer_amd import build as B
    from synthesizer_amd import _native as N
    header = ROOT / "include" / "synthhost.h"
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    names = sorted(set(re.findall(r"\b(shh_[a-z0-9_]+)\s*\(", text)))
    assert names == ["shh_phase_table", "shh_version"]
    lib = ctypes.CDLL(str(B.build_host()))
    assert all(hasattr(lib, n) for n in names)
    lib.shh_version.restype = ctypes.c_char_p
    assert lib.shh_version().decode() == "synthhost 0.1 src:" + B.host_source_hash()
    src = tmp_path / "sz.c"
    src.write_text('#include "%s"\n#include "%s"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(shh_segment), sizeof(sh_segment), offsetof(shh_segment, t0),'
                   'offsetof(sh_segment, t0), offsetof(shh_segment, dt), offsetof(sh_segment, dt));return 0;}\n' % (header, HEADER))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [24, 24, 8, 8, 16, 16] and N.SEGMENT_DTYPE.itemsize == 24


def test_abi_guard_accepts_the_built_library():
    from synthesizer_amd import _native as N
    got = (ctypes.c_uint32 * 7)()
    assert N.lib().sh_abi(got, 7) == 7
    assert list(got) == N._abi_expected()
    assert got[0] == N.SH_ABI_VERSION and got[4] == N.VOICE_DTYPE.itemsize


def test_abi_guard_refuses_a_library_with_other_layouts(tmp_path):
    """a stale .so with the same symbols but another sh_voice must not be loaded silently."""
    from synthesizer_amd import _native as N
    want = N._abi_expected()
    for label, words in (("old_voice", [want[0], want[1], want[2], want[3], 240, want[5], want[6]]),
                         ("old_abi", [want[0] - 1] + want[1:])):
        src = tmp_path / (label + ".c")
        src.write_text("#include <stdint.h>\nint sh_abi(uint32_t* out, int n){const uint32_t v[7]={%s};for(int k=0;k<n&&k<7;++k)out[k]=v[k];return 7;}\n"
                       % ",".join(str(w) for w in words))
        so = tmp_path / (label + ".so")
        subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
        with pytest.raises(N.NativeLibraryStale):
            N._check_abi(ctypes.CDLL(str(so)), so)
    src = tmp_path / "none.c"
    src.write_text("int sh_other(void){return 0;}\n")
    so = tmp_path / "none.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    with pytest.raises(N.NativeLibraryStale):
        N._check_abi(ctypes.CDLL(str(so)), so)


def test_a_tree_that_does_not_compile_is_an_error_not_a_stale_load(monkeypatch):
    """A failing hipcc (the compiler IS there) must not fall back to the library of an older tree; a missing compiler may."""
    from synthesizer_amd import _native as N
    from synthesizer_amd import build as B
    N.lib()                                     # the real one is loaded and cached; work on a fresh cache below
    saved = N._lib
    try:
        monkeypatch.delenv("SYNTHHIP_LIB", raising=False)
        monkeypatch.delenv("SYNTHHIP_ALLOW_STALE", raising=False)

        def fails(**kw):
            raise subprocess.CalledProcessError(1, "hipcc -c osc_render.hip")
        monkeypatch.setattr(B, "build", fails)
        N._lib = None
        with pytest.raises(N.NativeLibraryStale):
            N.lib()
        monkeypatch.setenv("SYNTHHIP_ALLOW_STALE", "1")
        with pytest.warns(RuntimeWarning):
            assert N.lib() is not None
        # no compiler on the box: the shipped library is loaded with a warning
        monkeypatch.delenv("SYNTHHIP_ALLOW_STALE")

        def no_compiler(**kw):
            raise FileNotFoundError(2, "No such file or directory", "/nonexistent/hipcc")
        monkeypatch.setattr(B, "build", no_compiler)
        monkeypatch.setattr(B, "HIPCC", "/nonexistent/hipcc")
        N._lib = None
        with pytest.warns(RuntimeWarning):
            assert N.lib() is not None
    finally:
        N._lib = saved


def test_the_source_hash_covers_every_file_the_library_is_built_from():
    """The hash embedded in libsynthhip.so decides whether the tree's library is stale (build.needs_build, the binding's guard): every
    header a translation unit includes must be hashed (round 4: the render kernels' header was not, and an edit of it alone left the old
    library in place)."""
    import re
    from synthesizer_amd import build as B
    hashed = {Path(h).name for h in B.HEADERS} | set(B.SOURCES)
    for f in list(B.CSRC.glob("*.hip")) + list(B.CSRC.glob("*.hpp")) + list(B.CSRC.glob("*.h")):
        if f.suffix == ".hip":
            assert f.name in B.SOURCES, f.name
        for inc in re.findall(r'#include\s+"([^"]+)"', f.read_text()):
            assert Path(inc).name in hashed, (f.name, inc)
