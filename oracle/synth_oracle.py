"""
CPU ORACLE for the synthplayer oscillator-bank + sample-mixing hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import it.  The product
package (synthesizer_amd/) never imports anything from oracle/.

PARITY STATUS: **unpinned** for the oscillator / envelope / quantise rows.
The upstream tree mounted at /root/reference contains only a relocation notice
(/root/reference/README.md:1-2): there is no synthplayer/oscillators.py,
sample.py, playback.py or tests/ to cite or to run.  Every formula below is a
restatement of the *publicly documented behaviour* of synthplayer 2.x as
recalled by the author ([RECALL] in SURVEY.md's legend) and of the grading
contract in BASELINE.json ([SPEC]).  When the real source is mounted, each
function below names the upstream symbol it has to be diffed against.

What IS pinned: the integer PCM rows (Sample.mix -> audioop.add,
Sample.resample -> audioop.ratecv).  Those delegate, upstream, to the CPython
standard library module ``audioop`` (Modules/audioop.c; CPython 3.10.12 is the
interpreter in this image).  oracle/pcm_oracle.py restates those two C
functions and tests/test_oracle_pcm.py checks the restatement against the live
``audioop`` module and against golden vectors generated from it
(tests/golden/make_golden.py).

Semantics restated here (all float64, one Python-level iteration per output
sample, exactly like the upstream generators):

* every oscillator keeps a running ``t`` that is *accumulated* (``t += inc``),
  never recomputed from an integer index.  This matters: the float64 rounding
  of the accumulation drifts by ~1e-5 rad over 10 s at 3.5 kHz, and it decides
  on which side of an edge a Square/Pulse sample lands.  The HIP path
  reproduces the accumulated value bit-for-bit (see DESIGN.md "phase tables").
* FM: ``freq = f*(1+lfo[i]); phase_correction += (freq_prev-freq)*t`` carried
  across samples and blocks (upstream: oscillators.py, class Sine/Sawtooth/...
  ``blocks()``, FM branch).
* EnvelopeFilter: four consecutive loops (attack, decay, sustain, release)
  gated by an accumulated ``time`` and an accumulated ``amp``.
* quantise: ``int(scale*v)`` (truncation toward zero), scale = 2**(8w-1)-1,
  packed into array('h') which raises OverflowError when out of range
  (upstream: sample.py Sample.from_osc_block / from_array).
"""
from __future__ import annotations

import itertools
from math import sin, pi, floor
from typing import Generator, Iterable, List, Optional, Sequence, Tuple

# upstream: synthplayer/params.py (module-level defaults)
norm_samplerate = 44100
norm_nchannels = 2
norm_samplewidth = 2
norm_osc_blocksize = 512

# ---- host-side VARIANTS of the recalled arithmetic (tools/pin_oracle.py --variants) ---------------------------------------------
# Where the recollection could be off by a last bit or a comparison operator, both readings exist, here and -- under the same names
# and values -- in the product (synthesizer_amd/params.py `variants`): the day the real package is importable, tools/pin_oracle.py
# re-runs every failing case under each reading and prints the one that matches; adopting it is a change of these defaults.
#   increment  "mul": 2 pi f / sr (f / sr for the turn-based kinds)       "div": rate = sr / f; 2 pi / rate (1 / rate)
#   square     "int2": -a if int(t * 2) % 2 else a                        "mod1": a if t % 1.0 < 0.5 else -a
#   pulse      "lt": a if t % 1.0 < pulsewidth else -a                    "le": ... <= pulsewidth
#   quantise   "trunc": int(scale * v)                                    "round": round(scale * v) (Python 3: half to even)
#   envelope   "lt": while time < phase_end                               "le": while time <= phase_end
VARIANT_CHOICES = {"increment": ("mul", "div"), "square": ("int2", "mod1"), "pulse": ("lt", "le"), "quantise": ("trunc", "round"),
                   "envelope": ("lt", "le")}
VARIANTS = {k: v[0] for k, v in VARIANT_CHOICES.items()}


def set_variants(**kw) -> dict:
    """Change readings (returns the previous table, for a try / finally); unknown names or values raise."""
    old = dict(VARIANTS)
    for k, v in kw.items():
        if k not in VARIANT_CHOICES or v not in VARIANT_CHOICES[k]:
            raise ValueError("variant %s=%r: choose from %r" % (k, v, VARIANT_CHOICES.get(k)))
        VARIANTS[k] = v
    return old


def _increment(frequency: float, samplerate: int, radians: bool) -> float:
    """The per-sample phase step of the non-FM branch of blocks()."""
    if VARIANTS["increment"] == "div":
        rate = samplerate / frequency
        return 2.0 * pi / rate if radians else 1.0 / rate
    return 2.0 * pi * frequency / samplerate if radians else frequency / samplerate


def _square(t: float, a: float) -> float:
    if VARIANTS["square"] == "mod1":
        return a if t % 1.0 < 0.5 else -a
    return -a if int(t * 2) % 2 else a


def _pulse_high(m: float, pw: float) -> bool:
    return m <= pw if VARIANTS["pulse"] == "le" else m < pw


def _env_before(time: float, end: float) -> bool:
    return time <= end if VARIANTS["envelope"] == "le" else time < end


class Oscillator:
    """upstream: oscillators.py class Oscillator (ABC with blocks())."""

    def __init__(self, samplerate: int = 0) -> None:
        self.samplerate = samplerate or norm_samplerate

    def blocks(self) -> Generator[List[float], None, None]:
        raise NotImplementedError

    # convenience for tests: n samples as a flat list
    def take(self, n: int) -> List[float]:
        out: List[float] = []
        gen = self.blocks()
        while len(out) < n:
            try:
                out.extend(next(gen))
            except StopIteration:
                break
        return out[:n]


def _fm_blocks(lfo: Oscillator) -> Generator[List[float], None, None]:
    return lfo.blocks()


class Sine(Oscillator):
    """upstream: oscillators.py class Sine."""

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase

    def blocks(self):
        if self.fm:
            phase_correction = self._phase * 2.0 * pi
            freq_previous = self.frequency
            increment = 2.0 * pi / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm)
            while True:
                block = []
                for fm in next(fm_blocks):
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    block.append(sin(t * freq + phase_correction) * self.amplitude + self.bias)
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, True)
            t = self._phase * 2.0 * pi
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    block.append(sin(t) * self.amplitude + self.bias)
                    t += increment
                yield block


class Sawtooth(Oscillator):
    """upstream: oscillators.py class Sawtooth (naive, not band-limited)."""

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase

    def blocks(self):
        if self.fm:
            phase_correction = self._phase
            freq_previous = self.frequency
            increment = 1.0 / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm)
            while True:
                block = []
                for fm in next(fm_blocks):
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    tt = t * freq + phase_correction
                    block.append(self.bias + self.amplitude * 2.0 * (tt - floor(0.5 + tt)))
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, False)
            t = self._phase
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    block.append(self.bias + self.amplitude * 2.0 * (t - floor(0.5 + t)))
                    t += increment
                yield block


class Square(Oscillator):
    """upstream: oscillators.py class Square (perfect square, 50% duty)."""

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase

    def blocks(self):
        if self.fm:
            phase_correction = self._phase
            freq_previous = self.frequency
            increment = 1.0 / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm)
            while True:
                block = []
                for fm in next(fm_blocks):
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    tt = t * freq + phase_correction
                    block.append(_square(tt, self.amplitude) + self.bias)
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, False)
            t = self._phase
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    block.append(_square(t, self.amplitude) + self.bias)
                    t += increment
                yield block


class Pulse(Oscillator):
    """upstream: oscillators.py class Pulse (pulse width optionally modulated)."""

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0,
                 pulsewidth: float = 0.1, bias: float = 0.0, fm_lfo: Optional[Oscillator] = None,
                 pwm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        assert 0 <= pulsewidth <= 1
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.pulsewidth = pulsewidth
        self.fm = fm_lfo
        self.pwm = pwm_lfo
        self._phase = phase

    def blocks(self):
        if self.fm or self.pwm:
            phase_correction = self._phase
            freq_previous = self.frequency
            increment = 1.0 / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm) if self.fm else itertools.repeat([0.0] * norm_osc_blocksize)
            pwm_blocks = (_fm_blocks(self.pwm) if self.pwm
                          else itertools.repeat([self.pulsewidth] * norm_osc_blocksize))
            while True:
                block = []
                for fm, pw in zip(next(fm_blocks), next(pwm_blocks)):
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    tt = t * freq + phase_correction
                    block.append((self.amplitude if _pulse_high(tt % 1.0, pw) else -self.amplitude) + self.bias)
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, False)
            t = self._phase
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    block.append((self.amplitude if _pulse_high(t % 1.0, self.pulsewidth) else -self.amplitude) + self.bias)
                    t += increment
                yield block


class Harmonics(Oscillator):
    """upstream: oscillators.py class Harmonics (additive sine series)."""

    def __init__(self, frequency: float, harmonics: Sequence[Tuple[int, float]], amplitude: float = 1.0,
                 phase: float = 0.0, bias: float = 0.0, fm_lfo: Optional[Oscillator] = None,
                 samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase
        self.harmonics = list(harmonics)

    def blocks(self):
        if self.fm:
            phase_correction = self._phase * 2.0 * pi
            freq_previous = self.frequency
            increment = 2.0 * pi / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm)
            while True:
                block = []
                for fm in next(fm_blocks):
                    h = 0.0
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    q = t * freq + phase_correction
                    for k, amp in self.harmonics:
                        h += sin(q * k) * amp
                    block.append(h * self.amplitude + self.bias)
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, True)
            t = self._phase * 2.0 * pi
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    h = 0.0
                    for k, amp in self.harmonics:
                        h += sin(t * k) * amp
                    block.append(h * self.amplitude + self.bias)
                    t += increment
                yield block


class EnvelopeFilter(Oscillator):
    """upstream: oscillators.py class EnvelopeFilter (ADSR volume envelope).

    Four consecutive phases gated by an accumulated ``time`` (+= 1/samplerate)
    with an accumulated ``amp`` inside attack/decay/release; after release one
    more sample is emitted if amp is still > 0; then silence forever, or the
    stream ends when stop_at_end is set.
    """

    def __init__(self, source: Oscillator, attack: float, decay: float, sustain: float,
                 sustain_level: float, release: float, stop_at_end: bool = False, cycle: bool = False) -> None:
        assert attack >= 0 and decay >= 0 and sustain >= 0 and release >= 0
        assert 0 <= sustain_level <= 1
        super().__init__(source.samplerate)
        self._source = source
        self._attack = attack
        self._decay = decay
        self._sustain = sustain
        self._sustain_level = sustain_level
        self._release = release
        self._stop_at_end = stop_at_end
        # SURVEY.md 8(a) row a8 lists ``cycle=False`` in the signature [RECALL]: the phases start over after the release
        # (``time`` and ``amp`` from zero), the source generator running on; the silence / stop_at_end tail is then never reached
        self._cycle = cycle

    def _samples(self) -> Generator[float, None, None]:
        src = itertools.chain.from_iterable(self._source.blocks())
        end_time_decay = self._attack + self._decay
        end_time_sustain = end_time_decay + self._sustain
        end_time_release = end_time_sustain + self._release
        increment = 1.0 / self.samplerate
        while True:
            time = 0.0
            if self._attack:
                amp_change = 1.0 / self._attack * increment
                amp = 0.0
                while _env_before(time, self._attack):
                    yield next(src) * amp
                    amp += amp_change
                    time += increment
            if self._decay:
                amp = 1.0
                amp_change = (self._sustain_level - 1.0) / self._decay * increment
                while _env_before(time, end_time_decay):
                    yield next(src) * amp
                    amp += amp_change
                    time += increment
            while _env_before(time, end_time_sustain):
                yield next(src) * self._sustain_level
                time += increment
            if self._release:
                amp = self._sustain_level
                amp_change = (-self._sustain_level) / self._release * increment
                while _env_before(time, end_time_release):
                    yield next(src) * amp
                    amp += amp_change
                    time += increment
                if amp > 0.0:
                    yield next(src) * amp
            if not self._cycle:
                break
        if not self._stop_at_end:
            while True:
                yield 0.0

    def blocks(self):
        samples = self._samples()
        while True:
            block = list(itertools.islice(samples, norm_osc_blocksize))
            if not block:
                return
            yield block


# ---------------------------------------------------------------------------
# quantise (upstream: sample.py Sample.from_osc_block -> from_array)
# ---------------------------------------------------------------------------

def quantise(block: Iterable[float], samplewidth: int = norm_samplewidth,
             amplitude_scale: Optional[float] = None) -> List[int]:
    """int(scale*v): truncation toward zero, OverflowError when out of range."""
    if amplitude_scale is None:
        amplitude_scale = 2 ** (8 * samplewidth - 1) - 1
    lo, hi = -(2 ** (8 * samplewidth - 1)), 2 ** (8 * samplewidth - 1) - 1
    out = []
    for v in block:
        i = round(amplitude_scale * v) if VARIANTS["quantise"] == "round" else int(amplitude_scale * v)
        if i < lo or i > hi:
            raise OverflowError("signed integer out of range for sample width %d" % samplewidth)
        out.append(i)
    return out


# ---------------------------------------------------------------------------
# float stereo bus (new in the MI355X build, [SPEC] BASELINE.json north_star):
# bus[n] = (sum_v gl_v*x_v[n], sum_v gr_v*x_v[n]), voices summed in table order.
# ---------------------------------------------------------------------------

def mix_bus(voices: Sequence[Sequence[float]], gains: Sequence[Tuple[float, float]]) -> List[Tuple[float, float]]:
    n = min(len(v) for v in voices)
    out = []
    for i in range(n):
        l = 0.0
        r = 0.0
        for v, (gl, gr) in zip(voices, gains):
            l += gl * v[i]
            r += gr * v[i]
        out.append((l, r))
    return out


# ---------------------------------------------------------------------------------------------------
# further oscillators and filters (all [RECALL], parity unpinned)
# ---------------------------------------------------------------------------------------------------

class Triangle(Oscillator):
    """upstream: oscillators.py class Triangle."""

    def __init__(self, frequency: float, amplitude: float = 1.0, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency = frequency
        self.amplitude = amplitude
        self.bias = bias
        self.fm = fm_lfo
        self._phase = phase

    def blocks(self):
        if self.fm:
            phase_correction = self._phase
            freq_previous = self.frequency
            increment = 1.0 / self.samplerate
            t = 0.0
            fm_blocks = _fm_blocks(self.fm)
            while True:
                block = []
                for fm in next(fm_blocks):
                    freq = self.frequency * (1.0 + fm)
                    phase_correction += (freq_previous - freq) * t
                    freq_previous = freq
                    tt = t * freq + phase_correction
                    block.append(4.0 * self.amplitude * (abs((tt + 0.75) % 1.0 - 0.5) - 0.25) + self.bias)
                    t += increment
                yield block
        else:
            increment = _increment(self.frequency, self.samplerate, False)
            t = self._phase
            while True:
                block = []
                for _ in range(norm_osc_blocksize):
                    block.append(4.0 * self.amplitude * (abs((t + 0.75) % 1.0 - 0.5) - 0.25) + self.bias)
                    t += increment
                yield block


class SquareH(Harmonics):
    """upstream: oscillators.py class SquareH (odd harmonics, 1/n)."""

    def __init__(self, frequency: float, num_harmonics: int = 16, amplitude: float = 0.9999, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        harmonics = [(n, 1.0 / n) for n in range(1, num_harmonics * 2, 2)]
        super().__init__(frequency, harmonics, amplitude, phase, bias, fm_lfo, samplerate)


class SawtoothH(Harmonics):
    """upstream: oscillators.py class SawtoothH (all harmonics, 1/n, phase + 0.5, flipped)."""

    def __init__(self, frequency: float, num_harmonics: int = 16, amplitude: float = 0.9999, phase: float = 0.0,
                 bias: float = 0.0, fm_lfo: Optional[Oscillator] = None, samplerate: int = 0) -> None:
        harmonics = [(n, 1.0 / n) for n in range(1, num_harmonics + 1)]
        super().__init__(frequency, harmonics, amplitude, phase + 0.5, bias, fm_lfo, samplerate)

    def blocks(self):
        for block in super().blocks():
            yield [self.bias * 2.0 - value for value in block]


class MixingFilter(Oscillator):
    """upstream: oscillators.py class MixingFilter."""

    def __init__(self, *sources: Oscillator) -> None:
        super().__init__(sources[0].samplerate)
        self._sources = sources

    def blocks(self):
        source_blocks = zip(*[src.blocks() for src in self._sources])
        for blocks in source_blocks:
            yield [sum(v) for v in zip(*blocks)]


class AmpModulationFilter(Oscillator):
    """upstream: oscillators.py class AmpModulationFilter."""

    def __init__(self, source: Oscillator, modulator: Oscillator) -> None:
        super().__init__(source.samplerate)
        self._source, self._modulator = source, modulator

    def blocks(self):
        for sb, mb in zip(self._source.blocks(), self._modulator.blocks()):
            yield [v * m for v, m in zip(sb, mb)]


class ClipFilter(Oscillator):
    """upstream: oscillators.py class ClipFilter."""

    def __init__(self, source: Oscillator, minimum: float = -1.0, maximum: float = 1.0) -> None:
        super().__init__(source.samplerate)
        self._source, self.min, self.max = source, minimum, maximum

    def blocks(self):
        for block in self._source.blocks():
            yield [max(min(v, self.max), self.min) for v in block]


class AbsFilter(Oscillator):
    """upstream: oscillators.py class AbsFilter."""

    def __init__(self, source: Oscillator) -> None:
        super().__init__(source.samplerate)
        self._source = source

    def blocks(self):
        from math import fabs
        for block in self._source.blocks():
            yield [fabs(v) for v in block]


class NullFilter(Oscillator):
    """upstream: oscillators.py class NullFilter."""

    def __init__(self, source: Oscillator) -> None:
        super().__init__(source.samplerate)
        self._source = source

    def blocks(self):
        yield from self._source.blocks()


class DelayFilter(Oscillator):
    """upstream: oscillators.py class DelayFilter."""

    def __init__(self, source: Oscillator, seconds: float) -> None:
        super().__init__(source.samplerate)
        self._source, self._seconds = source, seconds

    def _samples(self):
        src = itertools.chain.from_iterable(self._source.blocks())
        if self._seconds > 0.0:
            for _ in range(int(self.samplerate * self._seconds)):
                yield 0.0
        elif self._seconds < 0.0:
            for _ in range(int(-self.samplerate * self._seconds)):
                next(src)
        yield from src

    def blocks(self):
        samples = self._samples()
        while True:
            block = list(itertools.islice(samples, norm_osc_blocksize))
            if not block:
                return
            yield block


class Linear(Oscillator):
    """upstream: oscillators.py class Linear [RECALL]: the level is emitted, then incremented for as long as it lies
    strictly between min_value and max_value."""

    def __init__(self, startlevel: float, increment: float = 0.0, min_value: float = -1.0, max_value: float = 1.0,
                 samplerate: int = 0) -> None:
        super().__init__(samplerate)
        self._value, self._increment, self._min, self._max = startlevel, increment, min_value, max_value

    def blocks(self):
        value, increment, minv, maxv = self._value, self._increment, self._min, self._max
        while True:
            block = []
            for _ in range(norm_osc_blocksize):
                block.append(value)
                if minv < value < maxv:
                    value += increment
            yield block


_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    """Steele, Lea & Flood's SplitMix64 output function (public domain reference: xoshiro.di.unimi.it/splitmix64.c)."""
    z = x & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class WhiteNoise(Oscillator):
    """upstream: oscillators.py class WhiteNoise [RECALL]: ``random.uniform(-amplitude, amplitude) + bias`` drawn once per
    int(samplerate/frequency) samples and held.  Upstream draws from the global Mersenne Twister (not reproducible, and
    sequential); this build defines the draw as a counter-based generator instead -- value h uses
    u = (splitmix64(seed + h*0x9E3779B97F4A7C15) >> 11) * 2**-53 -- which is what this class restates.
    random.uniform(a, b) is a + (b-a)*random()."""

    def __init__(self, frequency: float, amplitude: float = 1.0, bias: float = 0.0, samplerate: int = 0, seed: int = 0) -> None:
        super().__init__(samplerate)
        self.frequency, self.amplitude, self.bias, self.seed = frequency, amplitude, bias, seed

    def blocks(self):
        cycles = int(self.samplerate / self.frequency)
        if cycles < 1:
            raise ValueError("whitenoise frequency cannot be bigger than the sample rate")
        a, b = -self.amplitude, self.amplitude
        n = 0
        while True:
            block = []
            for _ in range(norm_osc_blocksize):
                h = n // cycles
                u = (splitmix64(self.seed + h * 0x9E3779B97F4A7C15) >> 11) * 2.0 ** -53
                block.append((a + (b - a) * u) + self.bias)
                n += 1
            yield block


class EchoFilter(Oscillator):
    """upstream: oscillators.py class EchoFilter [RECALL]: the source plays alone for int(samplerate*after) samples; from
    there on `amount` copies of the remaining stream (itertools.tee) are mixed in, copy i delayed by
    int(samplerate*echo_delay) zeros (echo_delay accumulates `delay`) and scaled by the running product of `decay`."""

    def __init__(self, source: Oscillator, after: float, amount: int, delay: float, decay: float) -> None:
        super().__init__(source.samplerate)
        if decay < 0 or decay > 1:
            raise ValueError("decay should be 0-1")
        self._source, self._after, self._amount, self._delay, self._decay = source, after, amount, delay, decay
        self.echo_duration = after + amount * delay

    def _samples(self):
        src = itertools.chain.from_iterable(self._source.blocks())
        for _ in range(int(self.samplerate * self._after)):
            yield next(src)
        copies = itertools.tee(src, max(0, self._amount) + 1)
        streams = [copies[0]]
        amp = self._decay
        echo_delay = self._delay
        for echo in copies[1:]:
            zeros = [0.0] * int(self.samplerate * echo_delay)
            streams.append(itertools.chain(zeros, (lambda e, a: (v * a for v in e))(echo, amp)))
            echo_delay += self._delay
            amp *= self._decay
        for values in zip(*streams):
            yield sum(values)

    def blocks(self):
        samples = self._samples()
        while True:
            block = list(itertools.islice(samples, norm_osc_blocksize))
            if not block:
                return
            yield block
