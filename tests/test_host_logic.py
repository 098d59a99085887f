"""Host-side logic of the package that needs no GPU: envelope boundaries, voice packing, the harmonic
polynomial, sharding, Sample bookkeeping."""
import itertools
import math

import numpy as np
import pytest

from oracle import synth_oracle as O
from synthesizer_amd import _native as N
from synthesizer_amd import oscillators as G
from synthesizer_amd.dist import shard_range, shard_sizes
from synthesizer_amd.sample import Sample


def oracle_envelope_gains(a, d, s, sl, r, sr, n):
    """gain per sample of the reference EnvelopeFilter: feed it the constant 1."""
    class One(O.Oscillator):
        def blocks(self):
            while True:
                yield [1.0] * O.norm_osc_blocksize
    return O.EnvelopeFilter(One(sr), a, d, s, sl, r).take(n)


@pytest.mark.parametrize("adsr", [(0.01, 0.05, 0.5, 0.6, 0.2), (0.0, 0.05, 0.1, 0.6, 0.1), (0.02, 0.0, 0.0, 1.0, 0.05),
                                   (0.013, 0.007, 0.0, 0.3, 0.0), (0.0, 0.0, 0.01, 0.5, 0.0), (0.05, 0.05, 0.05, 0.0, 0.05),
                                   (0.001, 0.001, 0.001, 0.9, 0.001)])
@pytest.mark.parametrize("sr", [48000, 44100, 8000])
def test_envelope_spec_replays_the_reference_loops(adsr, sr):
    a, d, s, sl, r = adsr
    e = G.envelope_spec(a, d, s, sl, r, sr)
    n = e.n_release_end + 10
    want = np.array(oracle_envelope_gains(a, d, s, sl, r, sr, n))
    idx = np.arange(n, dtype=np.float64)
    got = np.zeros(n)
    A, D, S, R = e.n_attack_end, e.n_decay_end, e.n_sustain_end, e.n_release_end
    got[:A] = idx[:A] * e.attack_slope
    got[A:D] = 1.0 + (idx[A:D] - A) * e.decay_slope
    got[D:S] = e.sustain_level
    got[S:R] = e.sustain_level + (idx[S:R] - S) * e.release_slope
    if e.has_tail:
        got[R] = e.tail_amp
    assert A <= D <= S <= R
    assert np.max(np.abs(got - want)) < 1e-12
    # piece boundaries are exact: the first sample of every piece matches to rounding, no off-by-one
    for b in (A, D, S, R):
        if 0 < b < n:
            assert abs(got[b] - want[b]) < 1e-12 and abs(got[b - 1] - want[b - 1]) < 1e-12
    # stream length with stop_at_end
    o = O.EnvelopeFilter(O.Sine(1.0, samplerate=sr), a, d, s, sl, r, stop_at_end=True)
    assert sum(len(b) for b in o.blocks()) == e.length


def test_pack_voices_layout_and_sharing():
    sr = 48000
    lfo = G.Sine(5, 0.02, samplerate=sr)
    voices = [G.Sine(440, samplerate=sr), G.Sine(440, samplerate=sr), G.Square(100, phase=0.5, samplerate=sr),
              G.Harmonics(220, [(1, 1), (3, 0.5)], samplerate=sr, fm_lfo=lfo),
              G.Harmonics(330, [(1, 1), (3, 0.5)], samplerate=sr),
              G.Harmonics(110, [(k, 1.0 / k) for k in range(1, 30)], samplerate=sr),
              G.EnvelopeFilter(G.Pulse(50, pulsewidth=0.2, samplerate=sr), 0.1, 0.1, 0.1, 0.5, 0.1)]
    v, segs, coefs, partials = G.pack_voices([x.spec() for x in voices], [(0.25, 0.75)] * len(voices))
    assert v.dtype == N.VOICE_DTYPE and segs.dtype == N.SEGMENT_DTYPE
    assert (v["seg_offset"][0], v["seg_count"][0]) == (v["seg_offset"][1], v["seg_count"][1])      # shared table
    assert v["kind"].tolist() == [0, 0, 2, 4, 4, 4, 3]
    assert v["fm_mode"].tolist() == [0, 0, 0, 1, 0, 0, 0]
    assert v["harm_dense"][3] == 2 and v["harm_dense"][4] == 2 and v["harm_offset"][3] == v["harm_offset"][4]
    assert v["harm_dense"][5] == 1 and v["harm_count"][5] == 32
    assert v["env"]["enabled"].tolist() == [0, 0, 0, 0, 0, 0, 1]
    assert np.allclose(v["gain_l"], 0.25) and np.allclose(v["gain_r"], 0.75)
    for i in range(len(voices)):
        off = v["time_seg_offset"][i] if v["fm_mode"][i] else v["seg_offset"][i]
        assert segs["n0"][off] == 0
    # closed-form LFO constants
    d = 2.0 * math.pi * 5 / sr
    assert v["lfo_d"][3] == d and v["lfo_K"][3] == 0.02 / (2.0 * math.sin(d / 2.0))
    # the int16 boundary guard (ABI 6): the polynomial / Clenshaw voices WITHOUT FM carry their own list, shared lists stored once
    assert partials.dtype == N.PARTIAL_DTYPE and len(partials) == 2 + 29
    assert v["guard_count"].tolist() == [0, 0, 0, 0, 2, 29, 0] and v["guard_offset"][4] == 0 and v["guard_offset"][5] == 2
    assert partials["k"][:2].tolist() == [1.0, 3.0] and partials["amp"][:2].tolist() == [1.0, 0.5]
    per_t, const = G.guard_bounds(((1.0, 1.0), (3.0, 0.5)), voices[4].spec().harm_poly, None)
    assert v["guard_t"][4] == per_t == 2.5 * 2.0 ** -52 and 0.0 < v["guard_c"][4] < 1e-9 and v["guard_t"][3] == 0.0
    from synthesizer_amd import params
    params.int16_guard = False
    try:
        v2, _s, _c, p2 = G.pack_voices([G.Harmonics(330, [(1, 1), (3, 0.5)], samplerate=sr).spec()])
    finally:
        params.int16_guard = True
    assert v2["guard_count"][0] == 0 and len(p2) == 0


def test_harmonics_form_selection():
    sr = 48000
    assert G.Harmonics(100, [(k, 1.0 / k) for k in range(1, 17)], samplerate=sr).spec().harm_poly is not None
    assert G.Harmonics(100, [(17, 1.0)], samplerate=sr).spec().harm_dense is not None
    assert G.Harmonics(100, [(2.5, 1.0)], samplerate=sr).spec().harm_sparse is not None
    assert G.Harmonics(100, [(1, 1.0), (5000, 1.0)], samplerate=sr).spec().harm_sparse is not None
    assert G.Harmonics(100, [], samplerate=sr).spec().harm_sparse is not None


def test_series_polynomial_is_accurate():
    rng = np.random.default_rng(0)
    th = rng.uniform(0, 2 * np.pi, 20000)
    c, s = np.cos(th), np.sin(th)
    for amps in ([1.0 / k for k in range(1, 17)], [1.0] * 16, list(rng.uniform(-1, 1, 16)), [0.0] * 15 + [1.0]):
        p = G.series_polynomial(tuple(float(a) for a in amps))
        assert p is not None and len(p) == 16
        acc = np.full_like(c, p[0])
        for u in range(1, 16):
            acc = acc * c + p[u]
        want = sum(a * np.sin(k * th) for k, a in enumerate(amps, 1))
        assert np.max(np.abs(acc * s - want)) < 5e-11 * max(1.0, sum(abs(a) for a in amps))
    assert G.series_polynomial(tuple([0.0] * 16)) is None


def test_closed_form_lfo_selection():
    sr = 48000
    assert G.Sine(440, fm_lfo=G.Sine(5, 0.1, samplerate=sr), samplerate=sr).spec().fm_mode == N.SH_FM_SINE
    assert G.Sine(440, fm_lfo=G.Sawtooth(5, 0.1, samplerate=sr), samplerate=sr).spec().fm_mode == N.SH_FM_BUFFER
    assert G.Sine(440, fm_lfo=G.Sine(5, 0.1, fm_lfo=G.Sine(1, 0.1, samplerate=sr), samplerate=sr), samplerate=sr).spec().fm_mode == N.SH_FM_BUFFER
    assert G.Pulse(440, pwm_lfo=G.Sine(1, 0.1, samplerate=sr), samplerate=sr).spec().needs_pwm
    # an envelope over another envelope (or over a filter) is no single voice record: it renders block by block
    nested = G.EnvelopeFilter(G.EnvelopeFilter(G.Sine(1), 0, 0, 1, 1, 0), 0, 0, 1, 1, 0)
    with pytest.raises(NotImplementedError):
        nested.spec()
    assert nested.length is None and G.EnvelopeFilter(G.Sine(1), 0, 0, 1, 1, 0).spec().env is not None


def test_shard_ranges_partition_the_voice_table():
    for n in (1, 7, 8, 1024, 8192, 1000):
        for world in (1, 2, 3, 8):
            sizes = shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def test_sample_bookkeeping_without_gpu():
    raw = np.arange(-8, 8, dtype=np.int16)
    s = Sample.from_raw_frames(raw.tobytes(), 2, 8000, 2, name="x")
    assert len(s) == 8 and s.nchannels == 2 and s.samplewidth == 2 and s.samplerate == 8000
    assert s.duration == 8 / 8000 and s.frame_idx(0.0005) == 2 * 2 * 4
    assert s.get_frame_array().tolist() == raw.tolist()
    assert s.get_frames_numpy().shape == (8, 2)
    assert s.copy() == s and s.copy() is not s
    assert Sample.from_array(raw.tolist(), 8000, 2) == s
    assert Sample.from_array(np.array(raw), 8000, 2) == s
    assert s.maximum == 32767
    assert abs(s.get_frames_as_floats()[0] + 8 / 32768) < 1e-12
    with pytest.raises(ValueError):
        Sample.from_raw_frames(b"\0\0\0", 2, 8000, 1)
    with pytest.raises(RuntimeError):
        s.lock().add_silence(1.0)
    import io
    buf = io.BytesIO()
    Sample.from_raw_frames(raw.tobytes(), 2, 8000, 2).write_wav(buf)
    buf.seek(0)
    assert Sample(buf) == Sample.from_raw_frames(raw.tobytes(), 2, 8000, 2)


def test_note_tables():
    from synthesizer_amd import synth as S
    assert S.key_num("A", 4) == 49 and S.key_freq(49) == 440.0 and S.note_freq("A4") == 440.0
    assert abs(S.note_freq("C4") - 261.6256) < 1e-3 and abs(S.note_freq("C#", 3) - 138.5913) < 1e-3
    assert S.major_chord_keys("C", 4) == (40, 44, 47) and S.major_chord_keys("G", 4) == (47, 51, 54)
    with pytest.raises(ValueError):
        S.WaveSynth(samplewidth=3)


@pytest.mark.parametrize("args", [(0.0, 0.3, -1.0, 1.0), (0.5, -0.4, -1.0, 1.0), (0.0, 1e-5, -1.0, 1.0), (0.25, 0.0, -1.0, 1.0),
                                  (0.9, -3.3e-6, 0.0, 1.0), (2.0, 0.1, -1.0, 1.0), (-1.0, 0.1, -1.0, 1.0), (1e-3, 1e-9, -1.0, 1.0)])
def test_linear_table_replays_the_running_sum(args):
    """Linear: the (stopped) phase table gives the oracle's level at every sample, exactly."""
    from oracle import synth_oracle as O
    from synthesizer_amd.oscillators import Linear
    spec = Linear(*args, samplerate=48000).spec()
    n = 400000
    want = O.Linear(*args, samplerate=48000).take(n)
    tab = spec.carrier
    for k in list(range(0, 64)) + list(range(64, n, 997)) + [n - 1]:
        assert tab.value(k) == want[k], k
    assert tab.segments[0][0] == 0
    if want[-1] == want[-2]:                       # stopped: the table ends with a constant piece
        assert tab.segments[-1][2] == 0.0


def test_white_noise_spec():
    from synthesizer_amd.oscillators import WhiteNoise, EnvelopeFilter
    s = WhiteNoise(4410.0, 0.5, 0.1, samplerate=44100, seed=-1).spec()
    assert (s.kind, s.noise_hold, s.amplitude, s.bias) == (N.SH_NOISE, 10, 0.5, 0.1)
    v = G.pack_voices([EnvelopeFilter(WhiteNoise(100.0, samplerate=8000, seed=2 ** 64 + 7), 0.1, 0.1, 0.1, 0.5, 0.1).spec()])[0]
    assert int(v["noise_seed"][0]) == 7 and int(v["noise_hold"][0]) == 80 and v["env"]["enabled"][0] == 1
    with pytest.raises(ValueError):
        WhiteNoise(10000.0, samplerate=8000).spec()


def test_staggered_notes_workload_and_onset_records():
    """workloads.staggered_notes on the host: slots * notes voices in the order they start, every note a DelayFilter over an
    EnvelopeFilter over Harmonics whose record carries the onset (sh_voice::start_frame = int(samplerate * seconds), as the
    oracle's DelayFilter counts) -- the same voices from both modules, so one seed gives the two sides of a parity check."""
    from oracle import synth_oracle as O
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.oscillators import pack_voices
    from synthesizer_amd.workloads import staggered_notes
    sr, slots, notes = 48000, 32, 3
    gv, gg = staggered_notes(G, slots, sr, seed=5, period=0.5, notes=notes)
    ov, og = staggered_notes(O, slots, sr, seed=5, period=0.5, notes=notes)
    assert len(gv) == len(ov) == slots * notes and gg == og
    specs = [v.spec() for v in gv]
    onsets = [s.start_frame for s in specs]
    assert onsets == sorted(onsets) and onsets[0] == 0 and onsets[-1] < notes * 0.5 * sr
    for s, o in zip(specs, ov):
        want = int(sr * o._seconds) if isinstance(o, O.DelayFilter) else 0
        assert s.start_frame == want
        assert s.env is not None and s.harm_poly is not None                # a fused envelope, the polynomial form of the series
    packed = pack_voices(specs, gg)
    voices = packed[0]
    assert [int(x) for x in voices["start_frame"]] == onsets


def test_every_module_of_the_package_imports_and_compiles():
    """A syntax error in a module only the GPU tests import would otherwise wait for the GPU box: compile every source of the
    package and the root scripts, and import the package's modules (importing loads no library and touches no device)."""
    import importlib
    import py_compile
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    files = sorted((root / "synthesizer_amd").glob("*.py")) + [root / "bench.py", root / "__graft_entry__.py"] + sorted((root / "tools").glob("*.py"))
    for f in files:
        py_compile.compile(str(f), doraise=True)
    for name in ("_native", "build", "dist", "mixer", "oscillators", "params", "phasetable", "sample", "synth", "workloads"):
        importlib.import_module("synthesizer_amd." + name)


def test_eight_ranks_map_to_eight_distinct_gpu_ordinals(tmp_path):
    """sh_init(device != 0) has never run anywhere, so at least the plumbing that picks the ordinal is pinned on
    CPU: the launcher's environment of an 8-rank job (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE) gives ordinals
    0 .. 7, through the function and through the worker script's own argv / env path; bench.py picks the same way."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tests"))
    import multi_gpu_worker as W
    from synthesizer_amd import dist
    envs = [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"} for r in range(8)]
    got = [W.plumbing(e) for e in envs]
    assert [g["device"] for g in got] == list(range(8)) and all(g["world"] == 8 for g in got) and [g["rank"] for g in got] == list(range(8))
    # two nodes of four: the ordinal is the LOCAL rank
    assert [dist.device_for_rank({"RANK": str(r), "LOCAL_RANK": str(r % 4)}) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert dist.device_for_rank({}) == 0 and dist.device_for_rank({"RANK": "3"}) == 3
    for r in (0, 5, 7):
        env = dict(os.environ, **envs[r])
        env.pop("SYNTHHIP_DEVICE", None)
        p = subprocess.run([sys.executable, str(root / "tests" / "multi_gpu_worker.py"), "--plumbing"], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        assert json.loads(p.stdout.strip().splitlines()[-1]) == {"rank": r, "world": 8, "device": r}


def test_lfo_table_reproduces_the_reference_fm_loop():
    """oscillators.LfoTable: the carrier's angle of the reference's FM loop -- freq_j = f (1 + lfo_j), phase_correction += (freq_{j-1} -
    freq_j) t, t += inc, the LFO an oscillator of its own on an accumulated phase -- from per-piece closed forms over the joint pieces of
    the LFO's phase table and the time table, against the loop itself run sample by sample in float64.  A slow deep LFO, a fast one under
    a high carrier, a biased one; the ideal-line closed form of rounds 1-3 is shown to be off by orders of magnitude more."""
    import math
    from synthesizer_amd.oscillators import LfoTable
    sr = 48000
    inc = 2 * math.pi / sr
    for f, rate, depth, bias, lph, secs, tol in ((440.0, 0.01, 0.5, 0.0, 0.3, 20, 2e-9), (3520.0, 5.0, 0.5, 0.0, 0.3, 25, 2e-8), (880.0, 7.0, 0.05, 0.02, 0.0, 20, 2e-9)):
        d, a = 2 * math.pi * rate / sr, lph * 2 * math.pi
        tab = LfoTable(a, d, depth, bias, inc)
        rec = tab.records
        assert len(rec) == 2 * tab.pieces and np.isfinite(rec["t0"]).all() and np.isfinite(rec["dt"]).all()
        n0s = rec["n0"][0::2]
        n_max = secs * sr
        marks = set([1, 2, 1000, n_max // 3, n_max // 2, n_max - 1] + [int(n) for n in n0s if 0 < n < n_max] + [int(n) + 1 for n in n0s if n + 1 < n_max])
        T, tl, pc, fprev, worst, worst_ideal = 0.0, a, 0.0, None, 0.0, 0.0
        feff = f * (1.0 + bias)
        K0 = depth / (2 * math.sin(d / 2))
        for n in range(n_max):
            freq = f * (1.0 + (depth * math.sin(tl) + bias))
            if fprev is None:
                fprev = freq
            pc += (fprev - freq) * T
            fprev = freq
            if n in marks:
                want = T * freq + pc
                g = int(np.searchsorted(n0s, n, side="right") - 1)
                u, dl, Kp, Cp = float(rec["t0"][2 * g]), float(rec["dt"][2 * g]), float(rec["t0"][2 * g + 1]), float(rec["dt"][2 * g + 1])
                got = feff * T + (feff * inc) * (Kp * (Cp - math.cos(u + ((n - int(n0s[g])) - 0.5) * dl)))
                worst = max(worst, abs(got - want))
                ideal = f * T + (f * inc) * (K0 * (math.cos(a - d / 2) - math.cos(a + (n - 0.5) * d)) + bias * n)
                worst_ideal = max(worst_ideal, abs(ideal - want))
            T += inc
            tl += d
        assert worst <= tol, (f, rate, worst)
        assert worst_ideal > 20 * worst, (f, rate, worst_ideal, worst)


def test_lfo_table_on_random_parameters():
    """... and over a grid the fixed cases do not reach: four sample rates, both kinds of time step (radians / turns), LFO rates of
    0.005 .. 400 Hz, depths to 0.9, biases of either sign, phases anywhere: relative error of the angle against the reference loop."""
    import math
    from synthesizer_amd.oscillators import LfoTable
    rng = np.random.default_rng(1)
    worst = 0.0
    for _case in range(12):
        sr = int(rng.choice([22050, 44100, 48000, 96000]))
        inc = (1.0 / sr) if rng.integers(0, 2) else 2 * math.pi / sr
        f = float(np.exp(rng.uniform(np.log(30), np.log(8000))))
        rate = float(np.exp(rng.uniform(np.log(0.005), np.log(400))))
        depth, bias, lph = float(rng.uniform(0.001, 0.9)), float(rng.choice([0.0, rng.uniform(-0.2, 0.2)])), float(rng.uniform(0, 1))
        d, a = 2 * math.pi * rate / sr, lph * 2 * math.pi
        rec = LfoTable(a, d, depth, bias, inc).records
        n0s = rec["n0"][0::2]
        n_max = int(rng.integers(5000, 120000))
        marks = set(int(x) for x in rng.integers(1, n_max, 6)) | {n_max - 1}
        T, tl, pc, fprev, feff = 0.0, a, 0.0, None, f * (1.0 + bias)
        for n in range(n_max):
            freq = f * (1.0 + (depth * math.sin(tl) + bias))
            if fprev is None:
                fprev = freq
            pc += (fprev - freq) * T
            fprev = freq
            if n in marks:
                want = T * freq + pc
                g = int(np.searchsorted(n0s, n, side="right") - 1)
                u, dl, Kp, Cp = float(rec["t0"][2 * g]), float(rec["dt"][2 * g]), float(rec["t0"][2 * g + 1]), float(rec["dt"][2 * g + 1])
                got = feff * T + (feff * inc) * (Kp * (Cp - math.cos(u + ((n - int(n0s[g])) - 0.5) * dl)))
                worst = max(worst, abs(got - want) / max(1.0, abs(want)))
            T += inc
            tl += d
    assert worst <= 1e-13, worst


def test_time_step_weights_sum_to_the_accumulated_time():
    """oscillators.time_step_weights: the runs (offset, count, w - 1) of a block cover it without gaps, and sum (w inc) over them is the
    accumulated time's difference across the block to 1e-22 of it (rational arithmetic) -- pieces' ends and the one odd step in front of them
    included; both kinds of time step."""
    from fractions import Fraction
    from synthesizer_amd.oscillators import _table, time_step_weights
    for inc in (2 * np.pi / 48000, 1.0 / 44100):
        tab = _table(0.0, inc)
        ends = [int(n) for n in tab.records["n0"][1:40]]
        for start, n in [(0, 5000), (ends[25] - 100, 300), (ends[30] - 1, 2), (ends[30], 1), (1440000, 50000), (ends[20] - 3, ends[22] - ends[20] + 7)]:
            runs = time_step_weights(inc, start, n)
            pos = 0
            total = Fraction(0)
            for off, cnt, wm1 in runs:
                assert off == pos and cnt > 0
                pos += cnt
                total += cnt * (Fraction(wm1) + 1) * Fraction(inc)
            assert pos == n
            want = Fraction(tab.value(start + n)) - Fraction(tab.value(start))
            # (w - 1 is a float64 of ~1e-10: its rounding is 1e-26 of the sum)
            assert abs(total - want) <= Fraction(abs(float(want))) * Fraction(1, 10 ** 22), (inc, start, n, float(total - want))
