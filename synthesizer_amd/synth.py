"""
WaveSynth: the façade of upstream ``synthplayer/synth.py`` (tree not
mounted, so names/signatures are as recalled): ``sine()``, ``square()``, ``square_h()``, ``triangle()``,
``sawtooth()``, ``sawtooth_h()``, ``pulse()``, ``harmonics()`` return a ``Sample`` of ``duration`` seconds,
the ``*_gen`` forms return the oscillator, plus the equal-temperament note tables.  It is a thin caller of
the oscillators and of ``Sample.from_osc_block`` -- no arithmetic of its own: the block is rendered on
the GPU in float64 (the reference's arithmetic), quantised on the GPU from that float64 block
(``int(scale*v)``, scale = 2**(8*width-1)-1) and stays in HBM.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from . import params
from .oscillators import (Harmonics, Linear, Oscillator, Pulse, Sawtooth, SawtoothH, Sine, Square, SquareH, Triangle,
                          WhiteNoise)
from .sample import Sample

__all__ = ["WaveSynth", "key_num", "key_freq", "note_freq", "octave_notes", "major_chords", "major_chord_keys"]

octave_notes = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
major_chords = {"C": ("C", "E", "G"), "D": ("D", "F#", "A"), "E": ("E", "G#", "B"), "F": ("F", "A", "C"),
                "G": ("G", "B", "D"), "A": ("A", "C#", "E"), "B": ("B", "D#", "F#")}


def key_num(note: str, octave: int) -> int:
    """Piano key number of a note: A4 (440 Hz) is key 49."""
    notes = {n: i for i, n in enumerate(octave_notes)}
    return (octave - 1) * 12 + notes[note.upper()] + 4


def key_freq(key_number: int, a4: float = 440.0) -> float:
    """Frequency of a piano key (equal temperament)."""
    return 2 ** ((key_number - 49) / 12) * a4


def note_freq(note: str, octave: Optional[int] = None, a4: float = 440.0) -> float:
    """Frequency of a note such as 'A4', 'C#3' or ('C#', 3)."""
    if octave is None:
        note, octave = note[:-1], int(note[-1:])
    return key_freq(key_num(note, octave), a4)


def major_chord_keys(rootnote: str, octave: int) -> Tuple[int, int, int]:
    keys = [key_num(n, octave) for n in major_chords[rootnote.upper()]]
    for i in (1, 2):
        if keys[i] < keys[i - 1]:
            keys[i] += 12
    return tuple(keys)


class WaveSynth:
    """Waveform sample synthesizer: every method renders ``duration`` seconds of an oscillator to a mono Sample."""

    def __init__(self, samplerate: int = 0, samplewidth: int = 0) -> None:
        self.samplerate = samplerate or params.norm_samplerate
        self.samplewidth = samplewidth or params.norm_samplewidth
        if self.samplewidth not in (1, 2, 4):
            raise ValueError("only samplewidth sizes 1, 2 and 4 are supported")

    # -- helpers -----------------------------------------------------------------------------------
    def _check(self, freq: float, amplitude: float, bias: float) -> None:
        assert freq <= self.samplerate / 2
        assert 0 <= amplitude <= 1.0
        assert -1 <= bias <= 1.0

    def to_sample(self, osc: Oscillator, duration: float) -> Sample:
        """``duration`` seconds of the oscillator, quantised to this synth's sample width."""
        n = int(self.samplerate * duration)
        limit = osc.length
        if limit is not None:
            n = min(n, limit)
        if n <= 0:
            return Sample(samplerate=self.samplerate, nchannels=1, samplewidth=self.samplewidth)
        # float64 block in HBM -> int(scale * v) on the device: the float32 storage format is never an intermediate
        block = osc._render_f64_device(0, n)
        return Sample.from_osc_device(block, n, self.samplerate, samplewidth=self.samplewidth)

    # -- oscillator factories (upstream: the *_gen methods) ------------------------------------------
    def sine_gen(self, frequency, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return Sine(frequency, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def square_gen(self, frequency, amplitude=0.75, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return Square(frequency, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def square_h_gen(self, frequency, num_harmonics=16, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return SquareH(frequency, num_harmonics, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def triangle_gen(self, frequency, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return Triangle(frequency, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def sawtooth_gen(self, frequency, amplitude=0.75, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return Sawtooth(frequency, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def sawtooth_h_gen(self, frequency, num_harmonics=16, amplitude=0.5, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return SawtoothH(frequency, num_harmonics, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    def pulse_gen(self, frequency, amplitude=0.75, phase=0.0, bias=0.0, pulsewidth=0.1, fm_lfo=None, pwm_lfo=None) -> Oscillator:
        assert 0 <= pulsewidth <= 1
        self._check(frequency, amplitude, bias)
        return Pulse(frequency, amplitude, phase, pulsewidth, bias, fm_lfo=fm_lfo, pwm_lfo=pwm_lfo, samplerate=self.samplerate)

    def harmonics_gen(self, frequency, harmonics: Sequence[Tuple[int, float]], amplitude=0.5, phase=0.0, bias=0.0, fm_lfo=None) -> Oscillator:
        self._check(frequency, amplitude, bias)
        return Harmonics(frequency, harmonics, amplitude, phase, bias, fm_lfo=fm_lfo, samplerate=self.samplerate)

    # -- Samples -----------------------------------------------------------------------------------
    def white_noise_gen(self, frequency, amplitude=0.9999, bias=0.0, seed: int = 0) -> Oscillator:
        """White noise, a new value every samplerate/frequency samples (`seed`: this build's counter-based generator)."""
        self._check(frequency, amplitude, bias)
        return WhiteNoise(frequency, amplitude, bias, samplerate=self.samplerate, seed=seed)

    def linear_gen(self, startlevel, increment=0.0, min_value=-1.0, max_value=1.0) -> Oscillator:
        """Linear ramp (a constant if increment is 0), staying within the given bounds."""
        return Linear(startlevel, increment, min_value, max_value, samplerate=self.samplerate)

    def sine(self, frequency, duration, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.sine_gen(frequency, amplitude, phase, bias, fm_lfo), duration)

    def square(self, frequency, duration, amplitude=0.75, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.square_gen(frequency, amplitude, phase, bias, fm_lfo), duration)

    def square_h(self, frequency, duration, num_harmonics=16, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.square_h_gen(frequency, num_harmonics, amplitude, phase, bias, fm_lfo), duration)

    def triangle(self, frequency, duration, amplitude=0.9999, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.triangle_gen(frequency, amplitude, phase, bias, fm_lfo), duration)

    def sawtooth(self, frequency, duration, amplitude=0.75, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.sawtooth_gen(frequency, amplitude, phase, bias, fm_lfo), duration)

    def sawtooth_h(self, frequency, duration, num_harmonics=16, amplitude=0.5, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.sawtooth_h_gen(frequency, num_harmonics, amplitude, phase, bias, fm_lfo), duration)

    def pulse(self, frequency, duration, amplitude=0.75, phase=0.0, bias=0.0, pulsewidth=0.1, fm_lfo=None, pwm_lfo=None) -> Sample:
        return self.to_sample(self.pulse_gen(frequency, amplitude, phase, bias, pulsewidth, fm_lfo, pwm_lfo), duration)

    def harmonics(self, frequency, duration, harmonics, amplitude=0.5, phase=0.0, bias=0.0, fm_lfo=None) -> Sample:
        return self.to_sample(self.harmonics_gen(frequency, harmonics, amplitude, phase, bias, fm_lfo), duration)


    def white_noise(self, frequency, duration, amplitude=0.9999, bias=0.0, seed: int = 0) -> Sample:
        return self.to_sample(self.white_noise_gen(frequency, amplitude, bias, seed), duration)

    def linear(self, duration, startlevel, increment=0.0, min_value=-1.0, max_value=1.0) -> Sample:
        return self.to_sample(self.linear_gen(startlevel, increment, min_value, max_value), duration)