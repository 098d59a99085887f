"""
Sample: signed-integer PCM container with the hot-path operations of upstream
``synthplayer/sample.py`` (tree not mounted at /root/reference): ``from_osc_block`` (quantise),
``mix`` / ``mix_at`` (saturating add) and ``resample`` (linear interpolation), plus the
constructors/accessors a caller needs around them.  The arithmetic upstream delegates to CPython's
``audioop`` (add, ratecv) runs here as HIP kernels over PCM that stays resident in HBM; results are
bit-exact with ``audioop`` (tests/test_gpu_pcm.py).

Frames live either on the host (``bytes``) or on the device (``DeviceBuffer``); operations move them
to the device once and leave them there, accessors bring them back lazily.  There is no CPU
implementation of the arithmetic in this package.

Also here (the elementwise operations upstream delegates to audioop):
amplify / amplify_max / invert (``audioop.mul``), bias, reverse, mono / left / right (``tomono``),
stereo / pan (``tostereo``), normalize / make_16bit / make_32bit (``lin2lin``), peak / rms, fadein / fadeout.
Editing operations composed from those on device-resident PCM: clip / split / join / add_silence / delay,
speed (``ratecv``), at_volume, echo, envelope (ADSR), modulate_amp (sample- or oscillator-driven).
Level metering: ``level_db_peak`` / ``level_db_rms`` (both channels from ONE pass over the interleaved PCM, where
upstream makes two ``tomono`` copies and reads each) and the stateful ``LevelMeter``.
24-bit samples (width 3): everything upstream delegates to audioop; not the per-sample ``array`` operations (fades,
modulate_amp, pan with an lfo), which have no 24-bit form upstream either.
"""
from __future__ import annotations

import array
import ctypes as C
import math
import wave
from typing import BinaryIO, Iterable, Optional, Sequence, Tuple, Union

import numpy as np

from . import params
from . import _native as N

__all__ = ["Sample", "LevelMeter"]

_TYPECODE = {1: "b", 2: "h", 4: "i"}
_NPTYPE = {1: np.int8, 2: np.int16, 4: np.int32}


class Sample:
    """Audio sample data: interleaved little-endian signed PCM."""

    norm_samplerate = params.norm_samplerate
    norm_nchannels = params.norm_nchannels
    norm_samplewidth = params.norm_samplewidth

    def __init__(self, wave_file: Optional[Union[str, BinaryIO]] = None, name: str = "", samplerate: int = 0,
                 nchannels: int = 0, samplewidth: int = 0) -> None:
        self.name = name
        self.__locked = False
        self.__samplerate = samplerate or params.norm_samplerate
        self.__nchannels = nchannels or params.norm_nchannels
        self.__samplewidth = samplewidth or params.norm_samplewidth
        self.__frames: Optional[bytes] = b""
        self.__dev: Optional[N.DeviceBuffer] = None
        self.__dev_shared = False      # somebody else (a playing mixer source) reads __dev: never write it in place
        self.__nbytes = 0
        self.filename = None
        if wave_file:
            self.load_wav(wave_file)
            if isinstance(wave_file, str):
                self.filename = wave_file

    # -- storage -------------------------------------------------------------------------------
    def _set_host(self, frames: bytes) -> None:
        self.__frames = bytes(frames)
        self.__dev = None
        self.__dev_shared = False
        self.__nbytes = len(self.__frames)

    def _set_device(self, buf: N.DeviceBuffer, nbytes: int) -> None:
        if buf is not self.__dev:
            self.__dev_shared = False  # a fresh buffer is this Sample's own again
        self.__frames = None
        self.__dev = buf
        self.__nbytes = nbytes

    def _share_device(self) -> N.DeviceBuffer:
        """The device buffer, handed to a reader that keeps it (RealTimeMixer.add_sample streams from it): from now on every
        operation of this Sample writes a buffer of its own -- the in-place form of mix / mix_at is off until the data moves."""
        buf = self._device()
        self.__dev_shared = True
        return buf

    def _host(self) -> bytes:
        if self.__frames is None:
            self.__frames = self.__dev.download_bytes(self.__nbytes) if self.__nbytes else b""
        return self.__frames

    def _device(self) -> N.DeviceBuffer:
        if self.__dev is None:
            self.__dev = N.DeviceBuffer.from_bytes(self.__frames or b"")
        return self.__dev

    def to_device(self) -> "Sample":
        """Make the PCM resident in HBM (drops nothing; the host copy is kept until modified)."""
        self._device()
        return self

    @property
    def on_device(self) -> bool:
        return self.__dev is not None

    # -- constructors --------------------------------------------------------------------------
    @classmethod
    def from_raw_frames(cls, frames: Union[bytes, memoryview, bytearray], samplewidth: int, samplerate: int,
                        numchannels: int, name: str = "") -> "Sample":
        assert samplewidth in (1, 2, 3, 4) and numchannels >= 1 and samplerate > 1
        s = cls(name=name, samplerate=samplerate, nchannels=numchannels, samplewidth=samplewidth)
        frames = bytes(frames)
        if len(frames) % (samplewidth * numchannels):
            raise ValueError("frames data is not a whole number of frames")
        s._set_host(frames)
        return s

    @classmethod
    def from_array(cls, array_or_list: Union[Sequence[int], array.array, np.ndarray], samplerate: int,
                   numchannels: int, name: str = "") -> "Sample":
        if isinstance(array_or_list, np.ndarray):
            width = array_or_list.dtype.itemsize
            assert array_or_list.dtype.kind == "i" and width in (1, 2, 4)
            frames = np.ascontiguousarray(array_or_list).astype(array_or_list.dtype.newbyteorder("<")).tobytes()
        else:
            if isinstance(array_or_list, list):
                try:
                    array_or_list = array.array("h", array_or_list)       # OverflowError when out of range
                except OverflowError:
                    array_or_list = array.array("i", array_or_list)
            width = array_or_list.itemsize
            frames = array_or_list.tobytes()
        return cls.from_raw_frames(frames, width, samplerate, numchannels, name)

    @classmethod
    def from_osc_block(cls, block: Union[Iterable[float], np.ndarray], samplerate: int,
                       amplitude_scale: Optional[float] = None, samplewidth: int = 0) -> "Sample":
        """Quantise one oscillator block (mono): ``int(amplitude_scale * v)`` per sample, truncating
        toward zero, OverflowError when a value does not fit -- upstream Sample.from_osc_block."""
        width = samplewidth or params.norm_samplewidth
        if width not in (1, 2, 4):
            raise NotImplementedError("from_osc_block: sample width %d" % width)
        if amplitude_scale is None:
            amplitude_scale = 2 ** (8 * width - 1) - 1
        arr = block if isinstance(block, np.ndarray) else np.asarray(list(block), dtype=np.float64)
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        n = int(arr.size)
        s = cls(samplerate=samplerate, nchannels=1, samplewidth=width)
        if n == 0:
            return s
        src = N.DeviceBuffer.from_array(arr.reshape(-1))
        dst = N.DeviceBuffer(n * width)
        fn = N.lib().sh_quantize_f32 if arr.dtype == np.float32 else N.lib().sh_quantize_f64
        N.check(fn(src.handle, 0, n, float(amplitude_scale), width, dst.handle, 0))
        src.free()
        s._set_device(dst, n * width)
        return s

    @classmethod
    def from_osc_device(cls, block_f64: N.DeviceBuffer, n: int, samplerate: int,
                        amplitude_scale: Optional[float] = None, samplewidth: int = 0, free_block: bool = True) -> "Sample":
        """from_osc_block for a float64 block that already lives in HBM (an oscillator's ``_render_f64_device``): the
        quantiser reads the float64 samples the reference would have yielded, nothing is rounded to float32 on the way."""
        width = samplewidth or params.norm_samplewidth
        if width not in (1, 2, 4):
            raise NotImplementedError("from_osc_block: sample width %d" % width)
        if amplitude_scale is None:
            amplitude_scale = 2 ** (8 * width - 1) - 1
        s = cls(samplerate=samplerate, nchannels=1, samplewidth=width)
        if n:
            dst = N.DeviceBuffer(n * width)
            try:
                N.check(N.lib().sh_quantize_f64(block_f64.handle, 0, n, float(amplitude_scale), width, dst.handle, 0))
            finally:
                if free_block:
                    block_f64.free()
            s._set_device(dst, n * width)
        return s

    # -- accessors -------------------------------------------------------------------------------
    @property
    def samplewidth(self) -> int:
        return self.__samplewidth

    @property
    def samplerate(self) -> int:
        return self.__samplerate

    @samplerate.setter
    def samplerate(self, rate: int) -> None:
        assert rate > 0
        self.__samplerate = int(rate)

    @property
    def nchannels(self) -> int:
        return self.__nchannels

    @property
    def duration(self) -> float:
        return self.__nbytes / self.__samplerate / self.__samplewidth / self.__nchannels

    @property
    def maximum(self) -> int:
        return 2 ** (8 * self.__samplewidth - 1) - 1

    def __len__(self) -> int:
        """Number of frames."""
        return self.__nbytes // self.__samplewidth // self.__nchannels

    def __eq__(self, other) -> bool:
        if not isinstance(other, Sample):
            return False
        return (self.__samplewidth == other.__samplewidth and self.__samplerate == other.__samplerate and
                self.__nchannels == other.__nchannels and self._host() == other._host())

    def __repr__(self) -> str:
        return "<Sample '%s' at 0x%x, %g seconds, %d channels, %d bits, rate %d>" % (
            self.name, id(self), self.duration, self.__nchannels, 8 * self.__samplewidth, self.__samplerate)

    def frame_idx(self, seconds: float) -> int:
        """Byte index of the frame at the given time."""
        return self.__nchannels * self.__samplewidth * int(self.__samplerate * seconds)

    def view_frame_data(self) -> memoryview:
        return memoryview(self._host())

    def get_frame_array(self) -> array.array:
        if self.__samplewidth not in _TYPECODE:
            raise NotImplementedError("get_frame_array: sample width %d" % self.__samplewidth)
        return array.array(_TYPECODE[self.__samplewidth], self._host())

    def get_frames_numpy(self) -> np.ndarray:
        """[frames, channels] integer array (copy)."""
        if self.__samplewidth == 3:                   # 24-bit little endian -> int32
            b = np.frombuffer(self._host(), dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            return (v - ((v & 0x800000) << 1)).reshape(-1, self.__nchannels)
        return np.frombuffer(self._host(), dtype=_NPTYPE[self.__samplewidth]).reshape(-1, self.__nchannels).copy()

    def get_frames_as_floats(self) -> Sequence[float]:
        maxsize = 2 ** (8 * self.__samplewidth - 1)
        return [v / maxsize for v in self.get_frame_array()]

    def copy(self) -> "Sample":
        cpy = Sample(name=self.name, samplerate=self.__samplerate, nchannels=self.__nchannels,
                     samplewidth=self.__samplewidth)
        if self.__dev is not None and self.__frames is None:        # resident in HBM only: copy it there
            buf = N.DeviceBuffer(self.__nbytes)
            if self.__nbytes:
                N.check(N.lib().sh_buf_copy(buf.handle, 0, self.__dev.handle, 0, self.__nbytes))
            cpy._set_device(buf, self.__nbytes)
        else:
            cpy._set_host(self._host())
        cpy.filename = self.filename
        return cpy

    def chunked_frame_data(self, chunksize: int, repeat: bool = False, stopcondition=lambda: False):
        """Generator over the frame bytes in chunks of ``chunksize`` bytes (the last one of a one-shot sample may be
        shorter); with ``repeat`` the data wraps around forever and every chunk is full."""
        frames = self._host()
        if repeat:
            if not frames:
                return
            if len(frames) < chunksize:
                frames = frames * math.ceil(chunksize / len(frames))
            length = len(frames)
            mdata = memoryview(frames + frames[:chunksize])
            i = 0
            while not stopcondition():
                yield mdata[i: i + chunksize]
                i = (i + chunksize) % length
        else:
            mdata = memoryview(frames)
            i = 0
            while i < len(mdata) and not stopcondition():
                yield mdata[i: i + chunksize]
                i += chunksize

    def lock(self) -> "Sample":
        self.__locked = True
        return self

    def _check_writable(self) -> None:
        if self.__locked:
            raise RuntimeError("cannot modify a locked sample")

    # -- WAV in/out (host only, stdlib wave) -----------------------------------------------------
    def load_wav(self, file_or_stream: Union[str, BinaryIO]) -> "Sample":
        self._check_writable()
        with wave.open(file_or_stream) as w:
            if not 2 <= w.getsampwidth() <= 4:
                raise IOError("only supports sample sizes of 2, 3 or 4 bytes")
            if not 1 <= w.getnchannels() <= 2:
                raise IOError("only supports mono or stereo channels")
            self.__nchannels = w.getnchannels()
            self.__samplerate = w.getframerate()
            self.__samplewidth = w.getsampwidth()
            self._set_host(w.readframes(w.getnframes()))
        return self

    def write_wav(self, file_or_stream: Union[str, BinaryIO]) -> None:
        with wave.open(file_or_stream, "wb") as out:
            out.setparams((self.__nchannels, self.__samplewidth, self.__samplerate, 0, "NONE", "not compressed"))
            out.writeframes(self._host())

    # -- editing: slices and concatenations of device-resident PCM -------------------------------------
    def __assemble(self, parts: Sequence[tuple]) -> None:
        """frames = concatenation of parts; a part is (sample, first_byte, nbytes) or (None, 0, nbytes) for silence."""
        total = sum(p[2] for p in parts)
        dst = N.DeviceBuffer(total)
        L = N.lib()
        at = 0
        for src, first, nbytes in parts:
            if nbytes:
                if src is None:
                    dst.zero(at, nbytes)
                else:
                    N.check(L.sh_buf_copy(dst.handle, at, src._device().handle, first, nbytes))
            at += nbytes
        self._set_device(dst, total)

    def add_silence(self, seconds: float, at_start: bool = False) -> "Sample":
        """Add silence at the end (or at the start)."""
        self._check_writable()
        pad = self.frame_idx(seconds)
        if pad:
            me = (self, 0, self.__nbytes)
            keep = self.__dev          # keep the source alive while it is being copied
            self.__assemble([(None, 0, pad), me] if at_start else [me, (None, 0, pad)])
            del keep
        return self

    def clip(self, start_seconds: float, end_seconds: float) -> "Sample":
        """Keep only the given time range."""
        self._check_writable()
        assert end_seconds >= start_seconds
        start, end = self.frame_idx(start_seconds), self.frame_idx(end_seconds)
        if start != 0 or end != self.__nbytes:
            start, end, _ = slice(start, end).indices(self.__nbytes)      # byte-string slicing rules, as upstream
            keep = self.__dev
            self.__assemble([(self, start, max(0, end - start))])
            del keep
        return self

    def split(self, seconds: float) -> "Sample":
        """Keep the first part and return the chopped-off rest as a new sample."""
        self._check_writable()
        end = self.frame_idx(seconds)
        rest = Sample(name=self.name, samplerate=self.__samplerate, nchannels=self.__nchannels, samplewidth=self.__samplewidth)
        if end != self.__nbytes:
            end = slice(end, None).indices(self.__nbytes)[0]               # byte-string slicing rules, as upstream
            rest.__assemble([(self, end, self.__nbytes - end)])
            keep = self.__dev
            self.__assemble([(self, 0, end)])
            del keep
        return rest

    def join(self, other: "Sample") -> "Sample":
        """Append another sample to this one."""
        self._check_writable()
        assert self.samplewidth == other.samplewidth
        assert self.samplerate == other.samplerate
        assert self.nchannels == other.nchannels
        if other.__nbytes:
            keep = self.__dev
            self.__assemble([(self, 0, self.__nbytes), (other, 0, other.__nbytes)])
            del keep
        return self

    def delay(self, seconds: float, keep_length: bool = False) -> "Sample":
        """Delay the sample (insert silence at the start); a negative delay skips a bit from the start instead."""
        self._check_writable()
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

    def speed(self, speed: float) -> "Sample":
        """Change the playback speed (and the pitch) without changing the sample rate: frames are interpolated with
        ``audioop.ratecv(frames, width, nchannels, int(samplerate*speed), samplerate, None)``."""
        self._check_writable()
        assert speed > 0
        if speed == 1.0:
            return self
        if speed > 10.0 or speed < 0.1:
            raise ValueError("speed must be between 0.1 and 10")
        rate = self.__samplerate
        self.__samplerate = int(rate * speed)
        self.resample(rate)
        return self

    def at_volume(self, volume: float) -> "Sample":
        """A copy of the sample at the given volume 0..1 (works on locked samples: the original is untouched)."""
        cpy = self.copy()
        cpy.amplify(volume)
        return cpy

    def echo(self, length: float, amount: int, delay: float, decay: float) -> "Sample":
        """Add `amount` echos of the last `length` seconds, `delay` seconds apart, each `decay` times the previous
        volume.  Echos too quiet for the sample width are skipped."""
        self._check_writable()
        self._check_gpu_width("echo")
        if amount > 0:
            length = max(0, self.duration - length)
            echo = Sample(name=self.name, samplerate=self.__samplerate, nchannels=self.__nchannels, samplewidth=self.__samplewidth)
            first = slice(self.frame_idx(length), None).indices(self.__nbytes)[0]      # frames[frame_idx(length):], as upstream
            echo.__assemble([(self, first, self.__nbytes - first)])
            echo_amp = decay
            for _ in range(amount):
                if echo_amp < 1.0 / (2 ** (8 * self.__samplewidth - 1)):
                    break       # an echo nobody can hear
                length += delay
                echo = echo.copy().amplify(echo_amp)
                self.mix_at(length, echo)
                echo_amp *= decay
        return self

    def envelope(self, attack: float, decay: float, sustainlevel: float, release: float) -> "Sample":
        """Apply an ADSR volume envelope; attack, decay and release in seconds, sustainlevel a factor."""
        self._check_writable()
        assert attack >= 0 and decay >= 0 and release >= 0
        assert 0 <= sustainlevel <= 1
        D = self.split(attack)          # self is now the attack part
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

    def modulate_amp(self, modulation_source) -> "Sample":
        """Amplitude modulation: every sample becomes int(sample * factor).  The factors come from an oscillator
        (its block stream from the start), from another Sample or a sequence of numbers (cycled, scaled so that its
        largest absolute value is 1.0), or from any iterable of floats."""
        self._check_writable()
        self._check_gpu_width("modulate_amp")
        n = self.__nbytes // self.__samplewidth
        if not n:
            return self
        L = N.lib()
        from .oscillators import Oscillator
        if isinstance(modulation_source, Sample):
            modulation_source._check_gpu_width("modulate_amp")
            nmod = modulation_source.__nbytes // modulation_source.__samplewidth
            if not nmod:
                raise ValueError("modulation sample is empty")
            biggest = modulation_source.peak()
            mod = N.DeviceBuffer(nmod * 8)
            N.check(L.sh_pcm_to_f64(modulation_source._device().handle, nmod, modulation_source.__samplewidth, float(biggest), mod.handle))
        elif isinstance(modulation_source, Oscillator):
            nmod = n
            mod = modulation_source._render_f64_device(0, n)
        else:
            if isinstance(modulation_source, (list, tuple, array.array, np.ndarray)):
                values = np.asarray(modulation_source, dtype=np.float64)
                if not len(values):
                    raise ValueError("modulation sequence is empty")
                biggest = max(values.max(), abs(values.min()))
                values = values / biggest
            else:
                import itertools
                values = np.fromiter(itertools.islice(iter(modulation_source), n), dtype=np.float64)
                if len(values) < n:
                    raise ValueError("modulation iterator ran out after %d of %d samples" % (len(values), n))
            nmod = len(values)
            mod = N.DeviceBuffer(nmod * 8)
            mod.upload(values)
        dst = N.DeviceBuffer(self.__nbytes)
        N.check(L.sh_pcm_modulate(self._device().handle, self.__nbytes, self.__samplewidth, mod.handle, nmod, dst.handle))
        self._set_device(dst, self.__nbytes)
        return self

    # -- the hot path ----------------------------------------------------------------------------
    # 24-bit samples: every operation upstream hands to audioop (which reads 3-byte samples) runs on the GPU too; the ones
    # upstream does sample by sample through ``array`` (fades, modulate_amp, pan with an lfo, from_osc_block) have no 24-bit
    # form there either (no array typecode) and none here
    _WIDTH3_OK = frozenset(("mix", "mix_at", "amplify", "peak", "rms", "level_db", "bias", "reverse", "mono", "stereo", "normalize",
                            "make_32bit", "make_16bit", "resample", "echo"))

    def _check_gpu_width(self, what: str) -> None:
        if self.__samplewidth == 3 and what in self._WIDTH3_OK:
            return
        if self.__samplewidth not in (1, 2, 4):
            raise NotImplementedError("%s: %d-byte samples are not supported on the GPU path" % (what, self.__samplewidth))

    def mix(self, other: "Sample", other_seconds: Optional[float] = None, pad_shortest: bool = True) -> "Sample":
        """Mix another sample into this one (saturating add).  The shorter operand is zero-padded
        unless pad_shortest is False, in which case unequal lengths are an error (audioop.add)."""
        self._check_writable()
        assert self.samplewidth == other.samplewidth
        assert self.samplerate == other.samplerate
        assert self.nchannels == other.nchannels
        self._check_gpu_width("mix")
        n1 = self.__nbytes
        n2 = other.__nbytes if not other_seconds else min(other.__nbytes, other.frame_idx(other_seconds))
        if not pad_shortest and n1 != n2:
            raise ValueError("Lengths should be the same")
        self.__mix_region(other, 0, n2, max(n1, n2))
        return self

    def mix_at(self, seconds: float, other: "Sample", other_seconds: Optional[float] = None) -> "Sample":
        """Mix another sample into this one starting at the given time; grows as needed."""
        if seconds == 0.0:
            return self.mix(other, other_seconds)
        self._check_writable()
        assert self.samplewidth == other.samplewidth
        assert self.samplerate == other.samplerate
        assert self.nchannels == other.nchannels
        self._check_gpu_width("mix_at")
        start = self.frame_idx(seconds)
        n2 = other.frame_idx(other_seconds) if other_seconds else other.__nbytes
        n2 = min(n2, other.__nbytes)
        self.__mix_region(other, start, n2, max(self.__nbytes, start + n2))
        return self

    def __mix_region(self, other: "Sample", start: int, n2: int, total: int) -> None:
        """self[start:start+n2] = sat_add(self[start:start+n2] (zero-extended), other[:n2]); length -> total."""
        L = N.lib()
        n1 = self.__nbytes
        aliased = other is self and start != 0              # overlapping source and destination at shifted offsets inside one kernel
        if total == n1 and n1 and self._device().nbytes >= n1 and not self.__dev_shared and not aliased:
            # nothing grows (the mixer's common case: equal lengths, or mix_at inside the sample): add in place -- the kernel is
            # elementwise and declared without __restrict__ for exactly this -- 3 bytes moved per output byte instead of 5 (allocate,
            # copy self, add).  A Sample's device buffer is its own (copy() copies), so nobody else sees the write; a lock()ed
            # sample never gets here (_check_writable); one that a mixer streams from (_share_device) and a mix_at of the sample
            # into itself at an offset take the copying form below.
            if n2:
                N.check(L.sh_pcm_add(self.__dev.handle, start, other._device().handle, 0, n2, self.__samplewidth, self.__dev.handle, start))
            self._set_device(self.__dev, n1)            # (drops the host copy: it is stale now)
            return
        dst = N.DeviceBuffer(total)
        if total > n1:
            dst.zero(n1, total - n1)
        if n1:
            N.check(L.sh_buf_copy(dst.handle, 0, self._device().handle, 0, n1))
        if n2:
            N.check(L.sh_pcm_add(dst.handle, start, other._device().handle, 0, n2, self.__samplewidth, dst.handle, start))
        self._set_device(dst, total)

    # -- elementwise operations (upstream: thin wrappers over audioop) ---------------------------------
    def __unary(self, fn_name: str, out_nbytes: int, *args) -> "Sample":
        """frames = op(frames): run sh_<fn_name>(in, ..., out) into a fresh device buffer."""
        dst = N.DeviceBuffer(out_nbytes)
        N.check(getattr(N.lib(), fn_name)(self._device().handle, *args, dst.handle))
        self._set_device(dst, out_nbytes)
        return self

    def amplify(self, factor: float) -> "Sample":
        """Amplify (or attenuate, or with a negative factor invert) the sample: audioop.mul."""
        self._check_writable()
        self._check_gpu_width("amplify")
        n = self.__nbytes
        dst = N.DeviceBuffer(n)
        N.check(N.lib().sh_pcm_mul(self._device().handle, 0, n, self.__samplewidth, float(factor), dst.handle, 0))
        self._set_device(dst, n)
        return self

    def peak(self) -> int:
        """Maximum absolute sample value (audioop.max)."""
        self._check_gpu_width("peak")
        mx = C.c_uint32()
        N.check(N.lib().sh_pcm_stats(self._device().handle, self.__nbytes, self.__samplewidth, C.byref(mx), None))
        return int(mx.value)

    def rms(self) -> int:
        """Root mean square of the samples, truncated (audioop.rms)."""
        self._check_gpu_width("rms")
        n = self.__nbytes // self.__samplewidth
        if n == 0:
            return 0
        sq = C.c_double()
        N.check(N.lib().sh_pcm_stats(self._device().handle, self.__nbytes, self.__samplewidth, None, C.byref(sq)))
        from math import sqrt
        return int(sqrt(sq.value / float(n)))

    def _channel_stats(self) -> Tuple[Tuple[int, int], Tuple[float, float]]:
        """((max|L|, max|R|), (sum L^2, sum R^2)) of a stereo sample, one pass on the device."""
        mx = (C.c_uint32 * 2)()
        sq = (C.c_double * 2)()
        N.check(N.lib().sh_pcm_stats_stereo(self._device().handle, len(self), self.__samplewidth, mx, sq))
        return (int(mx[0]), int(mx[1])), (float(sq[0]), float(sq[1]))

    def __db_level(self, rms_mode: bool) -> Tuple[float, float]:
        self._check_gpu_width("level_db")
        maxvalue = 2 ** (8 * self.__samplewidth - 1)
        if self.__nchannels == 1:
            left = right = ((self.rms() if rms_mode else self.peak()) + 1) / maxvalue
        elif self.__nchannels == 2:
            (ml, mr), (sl, sr) = self._channel_stats()
            if rms_mode:
                n = len(self)
                ml, mr = (int(math.sqrt(sl / float(n))), int(math.sqrt(sr / float(n)))) if n else (0, 0)
            left, right = (ml + 1) / maxvalue, (mr + 1) / maxvalue
        else:
            raise ValueError("level metering needs a mono or stereo sample")
        # cut off at -60 dB instead of running down to -infinity
        return max(20.0 * math.log(left, 10), -60.0), max(20.0 * math.log(right, 10), -60.0)

    def __db_level_mono(self, rms_mode: bool) -> float:
        self._check_gpu_width("level_db")
        maxvalue = 2 ** (8 * self.__samplewidth - 1)
        level = ((self.rms() if rms_mode else self.peak()) + 1) / maxvalue
        return max(20.0 * math.log(level, 10), -60.0)

    @property
    def level_db_peak(self) -> Tuple[float, float]:
        """Peak level in dB (0 = full scale) of the (left, right) channel, not lower than -60."""
        return self.__db_level(False)

    @property
    def level_db_rms(self) -> Tuple[float, float]:
        """RMS level in dB of the (left, right) channel, not lower than -60."""
        return self.__db_level(True)

    @property
    def level_db_peak_mono(self) -> float:
        return self.__db_level_mono(False)

    @property
    def level_db_rms_mono(self) -> float:
        return self.__db_level_mono(True)

    def amplify_max(self) -> "Sample":
        """Amplify to the maximum volume without clipping."""
        self._check_writable()
        max_amp = self.peak()
        max_target = 2 ** (8 * self.__samplewidth - 1) - 2
        if max_amp > 0:
            self.amplify(max_target / max_amp)
        return self

    def invert(self) -> "Sample":
        return self.amplify(-1)

    def bias(self, bias: int) -> "Sample":
        """Add a constant to every sample (wrapping, like audioop.bias)."""
        self._check_writable()
        self._check_gpu_width("bias")
        return self.__unary("sh_pcm_bias", self.__nbytes, self.__nbytes, self.__samplewidth, int(bias))

    def reverse(self) -> "Sample":
        """Reverse the sound (audioop.reverse: the order of the samples, channels included)."""
        self._check_writable()
        self._check_gpu_width("reverse")
        return self.__unary("sh_pcm_reverse", self.__nbytes, self.__nbytes, self.__samplewidth)

    def mono(self, left_factor: float = 1.0, right_factor: float = 1.0) -> "Sample":
        """Stereo -> mono with per-channel factors (audioop.tomono)."""
        self._check_writable()
        if self.__nchannels == 1:
            return self
        if self.__nchannels != 2:
            raise ValueError("sample must be stereo or mono already")
        self._check_gpu_width("mono")
        nframes = len(self)
        self.__unary("sh_pcm_tomono", nframes * self.__samplewidth, nframes, self.__samplewidth, float(left_factor), float(right_factor))
        self.__nchannels = 1
        return self

    def left(self) -> "Sample":
        return self.mono(1.0, 0)

    def right(self) -> "Sample":
        return self.mono(0, 1.0)

    def stereo(self, left_factor: float = 1.0, right_factor: float = 1.0) -> "Sample":
        """Mono -> stereo with per-channel factors (audioop.tostereo); a stereo sample gets its channels scaled."""
        self._check_writable()
        self._check_gpu_width("stereo")
        if self.__nchannels == 2:
            # upstream: left().amplify(lf) mixed with right().amplify(rf) -> (fbound(L*lf), fbound(R*rf))
            right = self.copy().right().amplify(right_factor).stereo(0, 1.0)
            self.left().amplify(left_factor).stereo(1.0, 0)
            return self.mix(right)
        if self.__nchannels != 1:
            raise ValueError("sample must be mono or stereo already")
        nframes = len(self)
        self.__unary("sh_pcm_tostereo", nframes * 2 * self.__samplewidth, nframes, self.__samplewidth, float(left_factor), float(right_factor))
        self.__nchannels = 2
        return self

    def stereo_mix(self, other: "Sample", other_channel: str, other_mix_factor: float = 1.0, mix_at: float = 0.0,
                   other_seconds: Optional[float] = None) -> "Sample":
        """Mix a mono sample into the left ("L") or right ("R") channel of this one, scaled by ``other_mix_factor``,
        starting at ``mix_at`` seconds.  A mono self first becomes the opposite channel of a stereo sample."""
        self._check_writable()
        assert other.nchannels == 1
        assert other.samplerate == self.__samplerate
        assert other.samplewidth == self.__samplewidth
        assert other_channel in ("L", "R")
        if self.__nchannels == 1:
            if other_channel == "L":
                self.stereo(left_factor=0, right_factor=1)
            else:
                self.stereo(left_factor=1, right_factor=0)
        other = other.copy()
        if other_channel == "L":
            other.stereo(left_factor=other_mix_factor, right_factor=0)
        else:
            other.stereo(left_factor=0, right_factor=other_mix_factor)
        return self.mix_at(mix_at, other, other_seconds)

    def pan(self, panning: float = 0.0, lfo=None) -> "Sample":
        """Linear stereo panning, -1.0 (left) .. 1.0 (right); the sample becomes stereo.  With an ``lfo`` (an
        oscillator, or any iterable of floats: one value per frame) the position follows it instead: the left side of
        frame i becomes int(l * (1 - p) / 2), the right side int(r * (1 + p) / 2)."""
        if lfo is None:
            assert -1.0 <= panning <= 1.0
            left_volume = (1.0 - panning) / 2.0
            right_volume = (1.0 + panning) / 2.0
            return self.stereo(left_volume, right_volume)     # a stereo source keeps its channels apart, like the lfo form
        self._check_writable()
        self._check_gpu_width("pan")
        if self.__nchannels not in (1, 2):
            raise ValueError("pan needs a mono or stereo sample")
        n = len(self)
        from .oscillators import Oscillator
        if isinstance(lfo, Oscillator):
            mod = lfo._render_f64_device(0, n) if n else N.DeviceBuffer(0)
        else:
            import itertools
            values = np.fromiter(itertools.islice(iter(lfo), n), dtype=np.float64)
            if len(values) < n:
                raise ValueError("pan lfo ran out after %d of %d frames" % (len(values), n))
            mod = N.DeviceBuffer(n * 8)
            if n:
                mod.upload(values)
        nbytes = n * 2 * self.__samplewidth
        dst = N.DeviceBuffer(nbytes)
        N.check(N.lib().sh_pcm_pan_lfo(self._device().handle, n, self.__samplewidth, self.__nchannels, mod.handle, dst.handle))
        self.__nchannels = 2
        self._set_device(dst, nbytes)
        return self

    def __lin2lin(self, new_width: int) -> None:
        n = self.__nbytes // self.__samplewidth
        self.__unary("sh_pcm_lin2lin", n * new_width, n, self.__samplewidth, new_width)
        self.__samplewidth = new_width

    def normalize(self) -> "Sample":
        """Bring the sample to the default rate, width and channel count (params.norm_*)."""
        self._check_writable()
        self._check_gpu_width("normalize")
        self.resample(params.norm_samplerate)
        if self.__samplewidth != params.norm_samplewidth:
            self.__lin2lin(params.norm_samplewidth)
        if self.__nchannels == 1 and params.norm_nchannels == 2:
            self.stereo(1, 1)
        return self

    def make_32bit(self, scale_amplitude: bool = True) -> "Sample":
        """Convert to 32-bit samples; without scaling the integer values are kept as they were."""
        self._check_writable()
        self._check_gpu_width("make_32bit")
        old = self.__samplewidth
        if old != 4:
            self.__lin2lin(4)
            if not scale_amplitude:
                self.amplify(1.0 / 2 ** (8 * (4 - old)))
        return self

    def make_16bit(self, maximize_amplitude: bool = True) -> "Sample":
        """Convert to 16-bit samples, optionally maximising the amplitude first."""
        self._check_writable()
        self._check_gpu_width("make_16bit")
        assert self.__samplewidth >= 2
        if maximize_amplitude:
            self.amplify_max()
        if self.__samplewidth > 2:
            self.__lin2lin(2)
        return self

    def __fade(self, first_byte: int, nbytes: int, fadeout: bool, slope: float, offset: float) -> None:
        n = self.__nbytes
        dst = N.DeviceBuffer(n)
        L = N.lib()
        if n:
            N.check(L.sh_buf_copy(dst.handle, 0, self._device().handle, 0, n))
        if nbytes:
            N.check(L.sh_pcm_fade(self._device().handle, first_byte, nbytes, self.__samplewidth, 1 if fadeout else 0,
                                  float(slope), float(offset), dst.handle, first_byte))
        self._set_device(dst, n)

    def fadeout(self, seconds: float, target_volume: float = 0.0) -> "Sample":
        """Fade the end of the sample out to the target volume: int(sample_i * (1 - i*decrease/numsamples))."""
        self._check_writable()
        self._check_gpu_width("fadeout")
        seconds = min(seconds, self.duration)
        i = self.frame_idx(self.duration - seconds)
        self.__fade(i, self.__nbytes - i, True, 1.0 - target_volume, 0.0)
        return self

    def fadein(self, seconds: float, start_volume: float = 0.0) -> "Sample":
        """Fade the start of the sample in from the start volume: int(sample_i * (i*increase/numsamples + start))."""
        self._check_writable()
        self._check_gpu_width("fadein")
        seconds = min(seconds, self.duration)
        i = self.frame_idx(seconds)
        self.__fade(0, i, False, 1.0 - start_volume, start_volume)
        return self

    def resample(self, samplerate: int) -> "Sample":
        """Resample to a different rate without changing pitch/duration: linear interpolation with
        the arithmetic of ``audioop.ratecv(frames, width, nchannels, rate, samplerate, None)``."""
        self._check_writable()
        if samplerate == self.__samplerate:
            return self
        self._check_gpu_width("resample")
        L = N.lib()
        nin = len(self)
        nout = L.sh_resample_out_frames(nin, self.__samplerate, samplerate)
        fb = self.__samplewidth * self.__nchannels
        dst = N.DeviceBuffer(nout * fb)
        out_frames = C.c_size_t()
        if nin:
            N.check(L.sh_resample(self._device().handle, nin, self.__nchannels, self.__samplewidth, 0,
                                  self.__samplerate, samplerate, dst.handle, C.byref(out_frames)))
        self._set_device(dst, nout * fb)
        self.__samplerate = samplerate
        return self


class LevelMeter:
    """Sound level tracker on the decibel scale (0 dB = full scale) with peak hold: a peak stays for 0.4 s of audio,
    then falls by 30 dB per second until the level catches up.  ``update`` takes one chunk; the real-time mixer feeds
    it every mixed chunk.  (Upstream ``synthplayer/sample.py`` class ``LevelMeter``, [RECALL], tree not mounted.)"""

    def __init__(self, rms_mode: bool = False, lowest: float = -60.0) -> None:
        assert -60.0 <= lowest < 0.0
        self._rms = rms_mode
        self._lowest = lowest
        self.reset()

    def reset(self) -> None:
        self.peak_left = self.peak_right = self._lowest
        self._peak_left_hold = self._peak_right_hold = 0.0
        self.level_left = self.level_right = self._lowest
        self._time = 0.0

    def update(self, sample: Sample) -> Tuple[float, float, float, float]:
        """Feed one chunk; returns (level left, peak left, level right, peak right)."""
        left, right = sample.level_db_rms if self._rms else sample.level_db_peak
        left = max(left, self._lowest)
        right = max(right, self._lowest)
        time = self._time + sample.duration
        if (time - self._peak_left_hold) > 0.4:
            self.peak_left -= sample.duration * 30.0
        if left >= self.peak_left:
            self.peak_left = left
            self._peak_left_hold = time
        if (time - self._peak_right_hold) > 0.4:
            self.peak_right -= sample.duration * 30.0
        if right >= self.peak_right:
            self.peak_right = right
            self._peak_right_hold = time
        self.level_left = left
        self.level_right = right
        self._time = time
        return left, self.peak_left, right, self.peak_right

    def print(self, bar_width: int = 60, stereo: bool = False) -> None:
        """One line of text bars for the current levels (carriage return, no newline)."""
        def bar(level: float, peak: float, width: int) -> str:
            filled = int(width * (level - self._lowest) / -self._lowest)
            mark = min(width - 1, max(0, int(width * (peak - self._lowest) / -self._lowest)))
            cells = ["#"] * filled + ["-"] * (width - filled)
            cells[mark] = ":"
            return "".join(cells)
        if stereo:
            half = bar_width // 2
            print(" %s| L-R |%s" % (bar(self.level_left, self.peak_left, half)[::-1], bar(self.level_right, self.peak_right, half)), end="\r")
        else:
            level, peak = (self.level_left + self.level_right) / 2, (self.peak_left + self.peak_right) / 2
            print(" %d dB |%s| 0 dB" % (int(self._lowest), bar(level, peak, bar_width)), end="\r")
