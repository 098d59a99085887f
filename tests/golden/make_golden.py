"""Generate the committed golden vectors.  Run from the repo root:  python tests/golden/make_golden.py

Two families, with different standing:

* ``audioop_*.npz`` -- inputs and outputs of CPython 3.10.12's own ``audioop.add`` / ``audioop.ratecv``
  (the third-party dependency synthplayer's Sample.mix / Sample.resample / mixer delegate to),
  produced by calling the real module in this container.  These PIN the integer PCM rows.
  Header of every file: oracle = CPython 3.10.12 audioop; upstream delegation recalled, not citable
  (reference tree not mounted, /root/reference/README.md:1-2).
* ``audioop_ops.npz`` -- the same for the elementwise calls behind Sample.amplify / bias / reverse / mono / stereo /
  make_16bit / peak / rms (``mul``, ``bias``, ``reverse``, ``tomono``, ``tostereo``, ``lin2lin``, ``max``, ``rms``) and for
  the editing methods composed from them (echo, envelope, speed, modulate_amp; oracle/sample_oracle.py over the
  live module).
* ``osc_*.npy`` -- outputs of oracle/synth_oracle.py, i.e. of this repository's own restatement of the
  oscillator formulas.  They guard the oracle against accidental edits; they are NOT reference
  outputs (parity unpinned for these rows).
"""
import audioop
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import synth_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent
assert sys.version_info[:2] == (3, 10), "golden vectors are pinned to CPython 3.10 audioop"


def audioop_add_vectors():
    rng = np.random.default_rng(1234)
    out = {}
    for width, dt in ((1, np.int8), (2, np.int16), (4, np.int32)):
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max + 1, 2048, dtype=np.int64).astype(dt)
        b = rng.integers(info.min, info.max + 1, 2048, dtype=np.int64).astype(dt)
        # force the saturation corners
        a[:8] = [info.max, info.min, info.max, info.min, 0, -1, 1, info.max - 1]
        b[:8] = [info.max, info.min, 1, -1, 0, -1, info.max, 1]
        r = np.frombuffer(audioop.add(a.tobytes(), b.tobytes(), width), dtype=dt)
        out["a%d" % width], out["b%d" % width], out["sum%d" % width] = a, b, r
    # the mixer's chain of saturating adds, order-dependent
    chunks = rng.integers(-20000, 20001, (12, 1024), dtype=np.int64).astype(np.int16)
    mixed = chunks[0].tobytes()
    for c in chunks[1:]:
        mixed = audioop.add(mixed, c.tobytes(), 2)
    out["chain_in"] = chunks
    out["chain_out"] = np.frombuffer(mixed, dtype=np.int16)
    np.savez_compressed(OUT / "audioop_add.npz", **out)


def audioop_ratecv_vectors():
    rng = np.random.default_rng(4321)
    out = {}
    cases = []
    for (i, o) in ((96000, 44100), (48000, 44100), (44100, 48000), (44100, 22050), (8000, 44100), (3, 7)):
        for nch in (1, 2, 8):
            for width, dt in ((2, np.int16), (4, np.int32), (1, np.int8)):
                if width != 2 and nch == 8:
                    continue
                cases.append((i, o, nch, width, dt))
    for n, (i, o, nch, width, dt) in enumerate(cases):
        info = np.iinfo(dt)
        frames = 100
        x = rng.integers(info.min, info.max + 1, frames * nch, dtype=np.int64).astype(dt)
        y = np.frombuffer(audioop.ratecv(x.tobytes(), width, nch, i, o, None)[0], dtype=dt)
        out["case%d_meta" % n] = np.array([i, o, nch, width], dtype=np.int64)
        out["case%d_in" % n] = x
        out["case%d_out" % n] = y
    # SURVEY appendix A probe: ramp 0,1000,..7000 96k->44.1k gives 0, 2176, 4353, 6530
    ramp = (np.arange(8) * 1000).astype(np.int16)
    out["ramp_in"] = ramp
    out["ramp_out"] = np.frombuffer(audioop.ratecv(ramp.tobytes(), 2, 1, 96000, 44100, None)[0], dtype=np.int16)
    assert out["ramp_out"].tolist() == [0, 2176, 4353, 6530]
    np.savez_compressed(OUT / "audioop_ratecv.npz", **out)


def audioop_ops_vectors():
    from oracle.sample_oracle import RefSample
    rng = np.random.default_rng(777)
    out = {}
    for width, dt in ((1, np.int8), (2, np.int16), (4, np.int32)):
        info = np.iinfo(dt)
        x = rng.integers(info.min, info.max + 1, 1024, dtype=np.int64).astype(dt)
        x[:8] = [info.max, info.min, info.min + 1, info.max - 1, 0, -1, 3, -3]
        raw = x.tobytes()
        out["x%d" % width] = x
        out["mul%d_1p5" % width] = np.frombuffer(audioop.mul(raw, width, 1.5), dtype=dt)
        out["mul%d_m0p333" % width] = np.frombuffer(audioop.mul(raw, width, -0.333), dtype=dt)
        out["bias%d_1000" % width] = np.frombuffer(audioop.bias(raw, width, 1000), dtype=dt)
        out["reverse%d" % width] = np.frombuffer(audioop.reverse(raw, width), dtype=dt)
        out["tomono%d" % width] = np.frombuffer(audioop.tomono(raw, width, 0.75, 0.5), dtype=dt)
        out["tostereo%d" % width] = np.frombuffer(audioop.tostereo(raw, width, 0.3, 1.2), dtype=dt)
        out["max_rms%d" % width] = np.array([audioop.max(raw, width), audioop.rms(raw, width)], dtype=np.int64)
        for nw, ndt in ((1, np.int8), (2, np.int16), (4, np.int32)):
            if nw != width:
                out["lin2lin%d_%d" % (width, nw)] = np.frombuffer(audioop.lin2lin(raw, width, nw), dtype=ndt)
    # editing methods (upstream compositions over the live module), 16-bit stereo 0.25 s at 8 kHz
    y = (rng.integers(-32768, 32768, 4000, dtype=np.int64) * 0.4).astype(np.int16)
    out["edit_in"] = y
    out["edit_echo"] = np.frombuffer(RefSample(y.tobytes(), 2, 8000, 2).echo(0.1, 3, 0.05, 0.6).frames, dtype=np.int16)
    out["edit_envelope"] = np.frombuffer(RefSample(y.tobytes(), 2, 8000, 2).envelope(0.05, 0.05, 0.5, 0.08).frames, dtype=np.int16)
    out["edit_speed_1p26"] = np.frombuffer(RefSample(y.tobytes(), 2, 8000, 2).speed(1.26).frames, dtype=np.int16)
    m = rng.integers(-32768, 32768, 333, dtype=np.int64).astype(np.int16)
    out["edit_mod"] = m
    out["edit_modulate"] = np.frombuffer(RefSample(y.tobytes(), 2, 8000, 2).modulate_amp(RefSample(m.tobytes(), 2, 8000, 1)).frames,
                                         dtype=np.int16)
    np.savez_compressed(OUT / "audioop_ops.npz", **out)


def osc_vectors():
    sine = np.array(O.Sine(440, samplerate=44100).take(44100))
    np.save(OUT / "osc_sine440_44k1.npy", sine.astype(np.float64)[np.r_[0:4096, 40004:44100]].copy())
    out = {}
    sr = 48000
    out["saw"] = np.array(O.Sawtooth(1000, 0.8, phase=0.1, bias=0.05, samplerate=sr).take(2048))
    out["square"] = np.array(O.Square(1000, samplerate=sr).take(2048))
    out["pulse"] = np.array(O.Pulse(441, pulsewidth=0.25, samplerate=sr).take(2048))
    out["harm"] = np.array(O.Harmonics(220, [(k, 1.0 / k) for k in range(1, 17)], 0.5, samplerate=sr).take(2048))
    out["fm_sine"] = np.array(O.Sine(440, fm_lfo=O.Sine(5, 0.03, samplerate=sr), samplerate=sr).take(2048))
    out["adsr"] = np.array(O.EnvelopeFilter(O.Sine(440, samplerate=sr), 0.01, 0.01, 0.01, 0.6, 0.01).take(2048))
    out["adsr_cycle"] = np.array(O.EnvelopeFilter(O.Sine(440, samplerate=sr), 0.004, 0.003, 0.005, 0.6, 0.006, cycle=True).take(2048))
    out["quant"] = np.array(O.quantise(out["harm"] * 0.5), dtype=np.int16)
    np.savez_compressed(OUT / "osc_misc.npz", **out)


if __name__ == "__main__":
    audioop_add_vectors()
    audioop_ratecv_vectors()
    audioop_ops_vectors()
    osc_vectors()
    for p in sorted(OUT.glob("*.np*")):
        print(p.name, p.stat().st_size)
