"""
CPU ORACLE for the editing operations of ``Sample``: clip / split / join /
add_silence / delay, speed, at_volume, echo, envelope, modulate_amp, pan with an LFO, and of the level metering
(``level_db_peak`` / ``level_db_rms`` and the stateful ``LevelMeter``).

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/synth_oracle.py): only tests/ may import it.

``RefSample`` restates upstream ``synthplayer/sample.py`` class ``Sample`` for those methods the way upstream
writes them: plain ``bytes`` slicing for the editing operations, CPython's ``audioop`` (3.10.12, the live module)
for the arithmetic (``mul``, ``add``, ``ratecv``), and the per-sample Python expressions for fades and amplitude
modulation.  The upstream tree is not mounted at /root/reference (README.md:1-2), so the *composition* of each
method (which audioop calls, in which order, with which byte offsets) is [RECALL] and carries no file:line.

PARITY STATUS: arithmetic **pinned against the live dependency** (audioop itself is called here);
method composition **parity unpinned** until the upstream tree is mounted.
"""
from __future__ import annotations

import array
import audioop
import itertools
import math
from typing import Iterable, Optional, Sequence, Union

_TYPECODE = {1: "b", 2: "h", 4: "i"}


class RefSample:
    def __init__(self, frames: bytes, samplewidth: int, samplerate: int, nchannels: int) -> None:
        self.frames = bytes(frames)
        self.samplewidth = samplewidth
        self.samplerate = samplerate
        self.nchannels = nchannels

    # -- accessors ---------------------------------------------------------------------------------
    @property
    def duration(self) -> float:
        return len(self.frames) / self.samplerate / self.samplewidth / self.nchannels

    def __len__(self) -> int:
        return len(self.frames) // self.samplewidth // self.nchannels

    def frame_idx(self, seconds: float) -> int:
        return self.nchannels * self.samplewidth * int(self.samplerate * seconds)

    def get_frame_array(self) -> array.array:
        return array.array(_TYPECODE[self.samplewidth], self.frames)

    def copy(self) -> "RefSample":
        return RefSample(self.frames, self.samplewidth, self.samplerate, self.nchannels)

    # -- level metering ----------------------------------------------------------------------------
    def _db_level(self, rms_mode: bool):
        maxvalue = 2 ** (8 * self.samplewidth - 1)
        measure = audioop.rms if rms_mode else audioop.max
        if self.nchannels == 1:
            left = right = (measure(self.frames, self.samplewidth) + 1) / maxvalue
        else:
            left_frames = audioop.tomono(self.frames, self.samplewidth, 1, 0)
            right_frames = audioop.tomono(self.frames, self.samplewidth, 0, 1)
            left = (measure(left_frames, self.samplewidth) + 1) / maxvalue
            right = (measure(right_frames, self.samplewidth) + 1) / maxvalue
        return max(20.0 * math.log(left, 10), -60.0), max(20.0 * math.log(right, 10), -60.0)

    @property
    def level_db_peak(self):
        return self._db_level(False)

    @property
    def level_db_rms(self):
        return self._db_level(True)

    @property
    def level_db_peak_mono(self) -> float:
        maxvalue = 2 ** (8 * self.samplewidth - 1)
        return max(20.0 * math.log((audioop.max(self.frames, self.samplewidth) + 1) / maxvalue, 10), -60.0)

    @property
    def level_db_rms_mono(self) -> float:
        maxvalue = 2 ** (8 * self.samplewidth - 1)
        return max(20.0 * math.log((audioop.rms(self.frames, self.samplewidth) + 1) / maxvalue, 10), -60.0)

    # -- arithmetic (audioop) ------------------------------------------------------------------------
    def amplify(self, factor: float) -> "RefSample":
        self.frames = audioop.mul(self.frames, self.samplewidth, factor)
        return self

    def mix(self, other: "RefSample", other_seconds: Optional[float] = None, pad_shortest: bool = True) -> "RefSample":
        frames1 = self.frames
        frames2 = other.frames[:other.frame_idx(other_seconds)] if other_seconds else other.frames
        if pad_shortest:
            if len(frames1) < len(frames2):
                frames1 += b"\0" * (len(frames2) - len(frames1))
            elif len(frames2) < len(frames1):
                frames2 += b"\0" * (len(frames1) - len(frames2))
        self.frames = audioop.add(frames1, frames2, self.samplewidth)
        return self

    def mix_at(self, seconds: float, other: "RefSample", other_seconds: Optional[float] = None) -> "RefSample":
        if seconds == 0.0:
            return self.mix(other, other_seconds)
        start_frame_idx = self.frame_idx(seconds)
        if other_seconds:
            other_frames = other.frames[:other.frame_idx(other_seconds)]
        else:
            other_frames = other.frames
        # mix the overlapping part, keep the rest of both
        pre, to_mix, post = self._mix_split_frames(len(other_frames), start_frame_idx)
        self.frames = b""                                       # free memory (as upstream does)
        if len(to_mix) < len(other_frames):
            to_mix += b"\0" * (len(other_frames) - len(to_mix))
        mixed = audioop.add(to_mix, other_frames, self.samplewidth)
        self.frames = pre + mixed + post
        return self

    def _mix_split_frames(self, other_frames_length: int, start_frame_idx: int):
        self._mix_grow_if_needed(start_frame_idx, other_frames_length)
        pre = self.frames[:start_frame_idx]
        to_mix = self.frames[start_frame_idx:start_frame_idx + other_frames_length]
        post = self.frames[start_frame_idx + other_frames_length:]
        return pre, to_mix, post

    def _mix_grow_if_needed(self, start_frame_idx: int, other_length: int) -> None:
        required_length = start_frame_idx + other_length
        if required_length > len(self.frames):
            self.frames += b"\0" * (required_length - len(self.frames))

    def fadeout(self, seconds: float, target_volume: float = 0.0) -> "RefSample":
        seconds = min(seconds, self.duration)
        i = self.frame_idx(self.duration - seconds)
        begin = self.frames[:i]
        end = self.frames[i:]
        numsamples = len(end) / self.samplewidth
        decrease = 1.0 - target_volume
        faded = array.array(_TYPECODE[self.samplewidth], end)
        for k in range(int(numsamples)):
            faded[k] = int(faded[k] * (1.0 - k * decrease / numsamples))
        self.frames = begin + faded.tobytes()
        return self

    def fadein(self, seconds: float, start_volume: float = 0.0) -> "RefSample":
        seconds = min(seconds, self.duration)
        i = self.frame_idx(seconds)
        begin = self.frames[:i]
        end = self.frames[i:]
        numsamples = len(begin) / self.samplewidth
        increase = 1.0 - start_volume
        faded = array.array(_TYPECODE[self.samplewidth], begin)
        for k in range(int(numsamples)):
            faded[k] = int(faded[k] * (k * increase / numsamples + start_volume))
        self.frames = faded.tobytes() + end
        return self

    def resample(self, samplerate: int) -> "RefSample":
        if samplerate == self.samplerate:
            return self
        self.frames = audioop.ratecv(self.frames, self.samplewidth, self.nchannels, self.samplerate, samplerate, None)[0]
        self.samplerate = samplerate
        return self

    def chunked_frame_data(self, chunksize: int, repeat: bool = False, stopcondition=lambda: False):
        # (upstream's signature as recalled: a repeating generator also ends when stopcondition() turns true -- the mixer's
        # "this sample was removed" test)
        if repeat:
            bdata = self.frames
            if len(bdata) < chunksize:
                bdata = bdata * math.ceil(chunksize / len(bdata))
            length = len(bdata)
            bdata += bdata[:chunksize]
            mdata = memoryview(bdata)
            i = 0
            while not stopcondition():
                yield mdata[i: i + chunksize]
                i = (i + chunksize) % length
        else:
            mdata = memoryview(self.frames)
            i = 0
            while i < len(mdata) and not stopcondition():
                yield mdata[i: i + chunksize]
                i += chunksize

    # -- editing --------------------------------------------------------------------------------------
    def add_silence(self, seconds: float, at_start: bool = False) -> "RefSample":
        required_extra = self.frame_idx(seconds)
        if at_start:
            self.frames = b"\0" * required_extra + self.frames
        else:
            self.frames += b"\0" * required_extra
        return self

    def clip(self, start_seconds: float, end_seconds: float) -> "RefSample":
        assert end_seconds >= start_seconds
        start = self.frame_idx(start_seconds)
        end = self.frame_idx(end_seconds)
        if start != 0 or end != len(self.frames):
            self.frames = self.frames[start:end]
        return self

    def split(self, seconds: float) -> "RefSample":
        end = self.frame_idx(seconds)
        if end != len(self.frames):
            chopped = self.copy()
            chopped.frames = self.frames[end:]
            self.frames = self.frames[:end]
            return chopped
        return RefSample(b"", self.samplewidth, self.samplerate, self.nchannels)

    def join(self, other: "RefSample") -> "RefSample":
        assert (self.samplewidth, self.samplerate, self.nchannels) == (other.samplewidth, other.samplerate, other.nchannels)
        self.frames += other.frames
        return self

    def delay(self, seconds: float, keep_length: bool = False) -> "RefSample":
        if seconds > 0:
            if keep_length:
                num_frames = len(self)
                self.add_silence(seconds, at_start=True)
                self.clip(0, num_frames / self.samplerate)
            else:
                self.add_silence(seconds, at_start=True)
        elif seconds < 0:
            seconds = -seconds
            if keep_length:
                self.add_silence(seconds)
            self.clip(seconds, self.duration)
        return self

    def speed(self, speed: float) -> "RefSample":
        assert speed > 0
        if speed == 1.0:
            return self
        if speed > 10.0 or speed < 0.1:
            raise ValueError("speed must be between 0.1 and 10")
        self.frames = audioop.ratecv(self.frames, self.samplewidth, self.nchannels,
                                     int(self.samplerate * speed), self.samplerate, None)[0]
        return self

    def at_volume(self, volume: float) -> "RefSample":
        cpy = self.copy()
        cpy.amplify(volume)
        return cpy

    def echo(self, length: float, amount: int, delay: float, decay: float) -> "RefSample":
        if amount > 0:
            length = max(0, self.duration - length)
            echo = self.copy()
            echo.frames = self.frames[self.frame_idx(length):]
            echo_amp = decay
            for _ in range(amount):
                if echo_amp < 1.0 / (2 ** (8 * self.samplewidth - 1)):
                    break
                length += delay
                echo = echo.copy().amplify(echo_amp)
                self.mix_at(length, echo)
                echo_amp *= decay
        return self

    def envelope(self, attack: float, decay: float, sustainlevel: float, release: float) -> "RefSample":
        assert attack >= 0 and decay >= 0 and release >= 0
        assert 0 <= sustainlevel <= 1
        D = self.split(attack)
        S = D.split(decay)
        if sustainlevel < 1:
            S.amplify(sustainlevel)
        R = S.split(S.duration - release)
        if attack > 0:
            self.fadein(attack)
        if decay > 0:
            D.fadeout(decay, sustainlevel)
        if release > 0:
            R.fadeout(release)
        self.join(D).join(S).join(R)
        return self

    def stereo_mix(self, other: "RefSample", other_channel: str, other_mix_factor: float = 1.0, mix_at: float = 0.0,
                   other_seconds: Optional[float] = None) -> "RefSample":
        w = self.samplewidth
        if self.nchannels == 1:
            self.frames = audioop.tostereo(self.frames, w, *((0, 1) if other_channel == "L" else (1, 0)))
            self.nchannels = 2
        other = other.copy()
        other.frames = audioop.tostereo(other.frames, w, *((other_mix_factor, 0) if other_channel == "L" else (0, other_mix_factor)))
        other.nchannels = 2
        return self.mix_at(mix_at, other, other_seconds)

    def pan(self, panning: float = 0.0, lfo: Optional[Iterable[float]] = None) -> "RefSample":
        if lfo is None:
            lf, rf = (1 - panning) / 2, (1 + panning) / 2
            w = self.samplewidth
            if self.nchannels == 2:           # Sample.stereo on a stereo source: channels scaled apart, then recombined
                left = audioop.mul(audioop.tomono(self.frames, w, 1, 0), w, lf)
                right = audioop.mul(audioop.tomono(self.frames, w, 0, 1), w, rf)
                self.frames = audioop.add(audioop.tostereo(left, w, 1, 0), audioop.tostereo(right, w, 0, 1), w)
            else:
                self.frames = audioop.tostereo(self.frames, w, lf, rf)
            self.nchannels = 2
            return self
        lfo = iter(lfo)
        if self.nchannels == 2:
            right = array.array(_TYPECODE[self.samplewidth], audioop.tomono(self.frames, self.samplewidth, 0, 1))
            left = array.array(_TYPECODE[self.samplewidth], audioop.tomono(self.frames, self.samplewidth, 1, 0))
            stereo = self.get_frame_array()
            for i in range(len(right)):
                panning = next(lfo)
                stereo[i * 2] = int(left[i] * (1 - panning) / 2)
                stereo[i * 2 + 1] = int(right[i] * (1 + panning) / 2)
        else:
            mono = self.get_frame_array()
            stereo = mono + mono
            for i, sample in enumerate(mono):
                panning = next(lfo)
                stereo[i * 2] = int(sample * (1 - panning) / 2)
                stereo[i * 2 + 1] = int(sample * (1 + panning) / 2)
            self.nchannels = 2
        self.frames = stereo.tobytes()
        return self

    def modulate_amp(self, modulation_source: Union["RefSample", Sequence[float], Iterable[float]]) -> "RefSample":
        frames = self.get_frame_array()
        if isinstance(modulation_source, (RefSample, list, array.array)):
            if isinstance(modulation_source, RefSample):
                modulation_source = modulation_source.get_frame_array()
            biggest = max(max(modulation_source), abs(min(modulation_source)))
            actual_modulator = (v / biggest for v in itertools.cycle(modulation_source))
        else:
            actual_modulator = iter(modulation_source)           # an oscillator's chained blocks, or any iterator
        for i in range(len(frames)):
            frames[i] = int(frames[i] * next(actual_modulator))
        self.frames = frames.tobytes()
        return self


class RefLevelMeter:
    """Upstream ``LevelMeter`` ([RECALL]): level / held peak per channel, hold 0.4 s, fall 30 dB per second."""

    def __init__(self, rms_mode: bool = False, lowest: float = -60.0) -> None:
        self._rms = rms_mode
        self._lowest = lowest
        self.peak_left = self.peak_right = lowest
        self._hold_left = self._hold_right = 0.0
        self._time = 0.0

    def update(self, sample: RefSample):
        left, right = sample.level_db_rms if self._rms else sample.level_db_peak
        left, right = max(left, self._lowest), max(right, self._lowest)
        now = self._time + sample.duration
        if now - self._hold_left > 0.4:
            self.peak_left -= sample.duration * 30.0
        if left >= self.peak_left:
            self.peak_left, self._hold_left = left, now
        if now - self._hold_right > 0.4:
            self.peak_right -= sample.duration * 30.0
        if right >= self.peak_right:
            self.peak_right, self._hold_right = right, now
        self._time = now
        return left, self.peak_left, right, self.peak_right


class RefRealTimeMixer:
    """Upstream ``playback.py`` ``RealTimeMixer`` ([RECALL]), reduced to its arithmetic: every turn takes the next
    chunk of every active sample in the order they were added, pads a short one with silence, and folds them with
    ``audioop.add``; a sample whose chunks ran out is dropped; nothing playing = silence."""

    def __init__(self, chunksize: int, samplewidth: int = 2) -> None:
        self.chunksize = chunksize
        self.samplewidth = samplewidth
        self.active = {}
        self.counter = 0

    @staticmethod
    def _with_delay(chunks, chunk_delay, chunksize):
        for _ in range(chunk_delay):
            yield None                                    # not started yet: contributes nothing this turn
        yield from chunks

    def add_sample(self, sample: RefSample, repeat: bool = False, chunk_delay: int = 0) -> int:
        self.counter += 1
        self.active[self.counter] = self._with_delay(sample.chunked_frame_data(self.chunksize, repeat), chunk_delay, self.chunksize)
        return self.counter

    def remove_sample(self, sid: int) -> None:
        self.active.pop(sid, None)

    def chunks(self):
        silence = b"\0" * self.chunksize
        while True:
            to_mix = []
            for sid, gen in list(self.active.items()):
                try:
                    chunk = next(gen)
                except StopIteration:
                    del self.active[sid]
                    continue
                if chunk is None or len(chunk) == 0:
                    continue
                if len(chunk) < self.chunksize:
                    chunk = bytes(chunk) + silence[: self.chunksize - len(chunk)]
                to_mix.append(bytes(chunk))
            to_mix = to_mix or [silence]
            mixed = to_mix[0]
            for other in to_mix[1:]:
                mixed = audioop.add(mixed, other, self.samplewidth)
            yield mixed
