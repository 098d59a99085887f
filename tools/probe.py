#!/usr/bin/env python
"""Probes that back numbers DESIGN.md and the bench line still quote (GPU box, from the repo root):

    python tools/probe.py <name> [args]

    run-lengths us per block of ONE launch that renders B blocks (sh_bank_render_run), next to the two-stream pipeline of one-block launches
    staggered  bench.py's staggered_notes table (1024 players x 22 rounds, literal ADSR, tile-classified launches) with the players' instruments
    kinds      Fused render of a 1024-voice bank of MIXED lean kinds (Harmonics x16, FM Sine, Sine, Sawtooth, Square, Pulse; one-second blocks, steady
    late       Every oscillator kind far into a note (300 s at 48 kHz: 1.4e7 samples of accumulated phase) against the C oracle's float64 values: the plain
    fm-long    How long does an FM Sine voice stay within the contract (1e-6 RMS) of the reference's generator?  The reference adds
    job        BASELINE's literal job: 10 s of the 1024-voice additive bank from frame 0 in blocks of 48 000 (the first block is the note's

Each is the body of a former one-off script (rounds 3-4; their outputs are in profiles/r04_*); the other ~35 probes of those rounds
were deleted in round 5 -- what they measured is recorded in profiles/ and CHANGELOG.md, and git history keeps the scripts.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def probe_staggered(argv):
    """bench.py's staggered_notes table (1024 players x 22 rounds, literal ADSR, tile-classified launches) with the players' instruments
varied: all Harmonics x16, all FM Sine, every other player FM Sine, and Harmonics / FM Sine / Sawtooth in turn.  us per one-second block."""
    import os
    import statistics
    import sys
    import numpy as np
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    SR, slots, notes, nblocks = 48000, 1024, 22, 20
    _, f, amp, phase, gains = W._voice_params(slots, 0)
    rng = np.random.default_rng(3)
    fm = rng.uniform(0.5, 8.0, slots)
    depth = rng.uniform(0.0, 0.05, slots)
    harm = [(k, 1.0 / k) for k in range(1, 17)]
    e = W.ADSR


    def instrument(s, which):
        if which == 0:
            return G.Harmonics(float(f[s]), harm, amplitude=float(amp[s]), phase=float(phase[s]), samplerate=SR)
        if which == 1:
            return G.Sine(float(f[s]), float(amp[s]), phase=float(phase[s]), fm_lfo=G.Sine(float(fm[s]), float(depth[s]), samplerate=SR), samplerate=SR)
        return G.Sawtooth(float(f[s]), float(amp[s]), phase=float(phase[s]), samplerate=SR)


    for name, pick in (("all Harmonics", lambda s: 0), ("all FM Sine", lambda s: 1), ("Harmonics | FM Sine", lambda s: s & 1),
                       ("Harmonics | FM Sine | Sawtooth", lambda s: s % 3)):
        if len(argv) > 0 and argv[0] not in name:
            continue
        voices, vgains = [], []
        for k in range(notes):
            for s in range(slots):
                onset = (s / slots + k) * 1.0
                osc = G.EnvelopeFilter(instrument(s, pick(s)), e["attack"], e["decay"], e["sustain"], e["sustain_level"], e["release"])
                voices.append(G.DelayFilter(osc, onset) if onset else osc)
                vgains.append(gains[s])
        bank = VoiceBank(voices, gains=vgains)
        ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]

        def loop():
            for k in range(1, 3):
                bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
            N.timer_start()
            for k in range(3, nblocks + 1):
                bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
            return N.timer_stop() / (nblocks - 2)
        loop()
        N.sync()
        got = [loop() for _ in range(9)]
        N.sync()
        x = ring[0].download(np.float32, SR * 2).astype(np.float64)
        print("%-32s %6.1f us per block   checksum %.9f" % (name, statistics.median(got) * 1e3, float(np.abs(x).sum())))
        for b in ring:
            b.free()
        bank.close() if hasattr(bank, "close") else None


def probe_kinds(argv):
    """Fused render of a 1024-voice bank of MIXED lean kinds (Harmonics x16, FM Sine, Sine, Sawtooth, Square, Pulse; one-second blocks, steady
state): the lean kernel instantiated for all kinds (k_render_lean<.., LEAN_K_ALL, false>).  Prints us per block and a checksum."""
    import os
    import sys
    import numpy as np
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    SR = 48000
    rng = np.random.default_rng(0)
    f = np.exp(rng.uniform(np.log(55), np.log(3520), 1024))
    gains = [(0.01, 0.02)] * 1024
    harm = [(j, 1.0 / j) for j in range(1, 17)]
    makers = [lambda k: G.Harmonics(float(f[k]), harm, 0.5, samplerate=SR),
              lambda k: G.Sine(float(f[k]), 0.5, fm_lfo=G.Sine(5.0, 0.02, samplerate=SR), samplerate=SR),
              lambda k: G.Sine(float(f[k]), 0.5, samplerate=SR),
              lambda k: G.Sawtooth(float(f[k]), 0.5, samplerate=SR),
              lambda k: G.Square(float(f[k]), 0.5, samplerate=SR),
              lambda k: G.Pulse(float(f[k]), 0.5, pulsewidth=0.3, samplerate=SR)]
    CASES = (("mixed six kinds", lambda k: makers[k % 6](k)), ("harmonics + fm", lambda k: makers[k % 2](k)), ("saw + square + pulse", lambda k: makers[3 + k % 3](k)),
                       ("harmonics only", lambda k: makers[0](k)), ("harmonics, one saw", lambda k: makers[3 if k == 500 else 0](k)),
                       ("fm only", lambda k: makers[1](k)), ("fm, one saw", lambda k: makers[3 if k == 500 else 1](k)),
                       ("harmonics | fm halves", lambda k: makers[0 if k < 512 else 1](k)))
    for name, pick in CASES:
        if len(argv) > 0 and argv[0] not in name:
            continue
        bank = VoiceBank([pick(k) for k in range(1024)], gains=gains)
        ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
        for s in range(40):
            bank.render_device(SR, (100 + s) * SR, bus_f32=ring[s & 3])
        N.sync()
        best = 1e9
        for rep in range(5):
            N.timer_start()
            for s in range(40, 140):
                bank.render_device(SR, (100 + s) * SR, bus_f32=ring[s & 3])
            best = min(best, N.timer_stop() / 100)
        N.sync()
        got = ring[3].download(np.float32, SR * 2).astype(np.float64)
        print("%-24s %6.1f us per block   checksum %.9f" % (name, best * 1e3, float(np.abs(got).sum())))


def probe_late(argv):
    """Every oscillator kind far into a note (300 s at 48 kHz: 1.4e7 samples of accumulated phase) against the C oracle's float64 values: the plain
kinds, Harmonics in its three forms, FM under a Sine LFO on a turn-based carrier, a Pulse with a pwm_lfo, an envelope with a long sustain.
usage (GPU box): python tools/late_parity_probe.py"""
    import os
    import sys
    import numpy as np
    from oracle import synth_oracle as O
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    N.ensure_init(0)
    SR, blk = 48000, 16384
    first = 300 * SR


    def cases(m):
        h16 = [(k, 1.0 / k) for k in range(1, 17)]
        return [("Sine 1234.5 Hz", m.Sine(1234.5, 0.8, phase=0.1, samplerate=SR)),
                ("Sawtooth 1000 Hz (edges on samples)", m.Sawtooth(1000.0, 0.8, samplerate=SR)),
                ("Square 1000 Hz (edges on samples)", m.Square(1000.0, 0.8, samplerate=SR)),
                ("Pulse 777.7 Hz", m.Pulse(777.7, 0.8, pulsewidth=0.3, samplerate=SR)),
                ("Triangle 432.1 Hz", m.Triangle(432.1, 0.8, phase=0.4, samplerate=SR)),
                ("Harmonics x16 (polynomial)", m.Harmonics(440.0, h16, 0.5, samplerate=SR)),
                ("Harmonics 1 + 33 (Clenshaw)", m.Harmonics(200.0, [(1, 1.0), (33, 0.2)], 0.5, samplerate=SR)),
                ("Harmonics sparse (1, 7.5, 1000 non-integer)", m.Harmonics(100.0, [(1, 1.0), (7.5, 0.3)], 0.5, samplerate=SR)),
                ("Sawtooth under a Sine LFO (turn-based FM)", m.Sawtooth(880.0, 0.5, fm_lfo=m.Sine(5.0, 0.1, bias=0.01, samplerate=SR), samplerate=SR)),
                ("Pulse with pwm_lfo", m.Pulse(300.0, 0.5, pulsewidth=0.5, pwm_lfo=m.Sine(0.7, 0.3, bias=0.5, samplerate=SR), samplerate=SR)),
                ("Sine under an envelope with a 400 s sustain", m.EnvelopeFilter(m.Sine(660.0, 0.9, samplerate=SR), 0.01, 0.05, 400.0, 0.6, 0.2))]


    for (name, g), (_n, o) in zip(cases(G), cases(O)):
        try:
            want = CO.render(o, first + blk)[first:]
        except Exception as e:                                   # (what the C oracle does not know: the pure-Python one, 30 s in)
            short = 30 * SR
            want = np.array(o.take(short + blk), dtype=np.float64)[short:]
            got = g.render_f64(blk, start=short)
            print("%-48s  30 s in (Python oracle): max |err| %.3e  differing float32 %d of %d" % (name, float(np.max(np.abs(got - want))), int(np.sum(got.astype(np.float32) != want.astype(np.float32))), blk))
            continue
        got = g.render_f64(blk, start=first)
        print("%-48s 300 s in: max |err| %.3e  differing float32 %d of %d" % (name, float(np.max(np.abs(got - want))), int(np.sum(got.astype(np.float32) != want.astype(np.float32))), blk))


def probe_fm_long(argv):
    """How long does an FM Sine voice stay within the contract (1e-6 RMS) of the reference's generator?  The reference adds
phase_correction += (freq_previous - freq) * t sample by sample and evaluates sin(t * freq + phase_correction): two terms of ~f t radians each
whose rounding (ulp(f t) / 2 per addition, sqrt(n) of them) is part of ITS output; the closed form here has no such noise.  One carrier
(440 Hz and 3520 Hz) with a 5 Hz Sine LFO, render_f64 against the C oracle at 1 .. 300 s into the note.
usage (GPU box): python tools/fm_long_time_probe.py"""
    import os
    import sys
    import numpy as np
    from oracle import synth_oracle as O
    from oracle import c_oracle as CO
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    N.ensure_init(0)
    SR, blk = 48000, 16384
    for f in (440.0, 3520.0):
        for depth in (0.05, 0.5):
            g = G.Sine(f, 1.0, phase=0.2, fm_lfo=G.Sine(5.0, depth, phase=0.3, samplerate=SR), samplerate=SR)
            o = O.Sine(f, 1.0, phase=0.2, fm_lfo=O.Sine(5.0, depth, phase=0.3, samplerate=SR), samplerate=SR)
            want_all = CO.render(o, 300 * SR + blk)
            for secs in (1, 10, 30, 100, 300):
                first = secs * SR
                got = g.render_f64(blk, start=first)
                w = want_all[first:first + blk]
                print("carrier %6.0f Hz depth %.2f, %3d s in: max |err| %.3e rms %.3e" % (f, depth, secs, float(np.max(np.abs(got - w))), float(np.sqrt(np.mean((got - w) ** 2)))))


def probe_job(argv):
    """BASELINE's literal job: 10 s of the 1024-voice additive bank from frame 0 in blocks of 48 000 (the first block is the note's
attack, decay and a dozen binades of the phase sum), timed as a whole -- next to the steady state bench.py's passes measure."""
    import sys
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd.mixer import VoiceBank
    from synthesizer_amd.workloads import additive_voices
    N.ensure_init(0)
    SR = 48000
    voices, gains = additive_voices(G, 1024, SR, seed=0, adsr={"sustain": 1e6})
    bank = VoiceBank(voices, gains=gains)
    ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
    # clocks up on other frames
    for k in range(400):
        bank.render_device(SR, (100 + k) * SR, bus_f32=ring[k & 3])
    N.sync()
    best = 1e9
    for rep in range(20):
        N.timer_start()
        for k in range(10):
            bank.render_device(SR, k * SR, bus_f32=ring[k & 3])
        best = min(best, N.timer_stop())
        for k in range(50):                        # (keeps the clocks up between repetitions, on other frames)
            bank.render_device(SR, (600 + k) * SR, bus_f32=ring[k & 3])
        N.sync()
    print("10 s job from frame 0: %.1f us  = %.3f T voice-samples/s" % (best * 1e3, 1024 * 10 * SR / best / 1e9))


def probe_run_lengths(argv):
    """What ONE launch costs per block when it renders B blocks (sh_bank_render_run into a contiguous ring), launches one at a time -- a
    caller that does not stream -- next to the two-stream pipeline of one-block launches.  The headline's bank; also config 3's FM bank."""
    import statistics
    import numpy as np
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    SR = 48000
    for name, (v, g) in (("1024 x Harmonics16 + ADSR (headline)", W.additive_voices(G, 1024, SR, seed=0, partials=16, adsr={"sustain": 1e6})),
                         ("1024 x FM Sine (config 3)", W.fm_voices(G, 1024, SR, seed=1)),
                         ("64 x Harmonics16 + ADSR (config 2)", W.additive_voices(G, 64, SR, seed=0, partials=16, adsr={"sustain": 1e6}))):
        bank = VoiceBank(v, gains=g)
        ring4 = [N.DeviceBuffer(SR * 8) for _ in range(4)]
        pos = [100]
        for _ in range(300):                                     # clocks up
            bank.render_device(SR, pos[0] * SR, bus_f32=ring4[pos[0] & 3]); pos[0] += 1
        N.sync()
        N.timer_start()
        for _ in range(400):
            bank.render_device(SR, pos[0] * SR, bus_f32=ring4[pos[0] & 3]); pos[0] += 1
        streamed = N.timer_stop() / 400 * 1e3
        row = ["%-38s streamed, one block per launch on two streams: %6.2f us per block;  one launch at a time, B blocks per launch:" % (name, streamed)]
        for B in (1, 2, 4, 8, 16):
            ring = bank.make_ring(SR, B)
            got = []
            for rep in range(40):
                for _ in range(3):                               # (keeps the clocks up; on other frames, pipelined)
                    bank.render_device(SR, pos[0] * SR, bus_f32=ring4[pos[0] & 3]); pos[0] += 1
                N.sync()
                N.timer_start()
                bank.render_run(SR, B, pos[0] * SR, ring=ring)
                got.append(N.timer_stop() / B * 1e3)
                pos[0] += B
            row.append("  B = %2d: %6.2f us per block" % (B, statistics.median(got)))
        print("\n".join(row), flush=True)


PROBES = {"run-lengths": probe_run_lengths, "staggered": probe_staggered, "kinds": probe_kinds, "late": probe_late, "fm-long": probe_fm_long, "job": probe_job}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in PROBES:
        sys.exit(__doc__)
    PROBES[sys.argv[1]](sys.argv[2:])
