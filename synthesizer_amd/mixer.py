"""
The mixer sum bus.

Two shapes of the same operation:

* ``VoiceBank``: N oscillator voices (the classes of oscillators.py) -> stereo bus.  The voice table
  lives in HBM; ``render`` runs the fused generate-and-mix kernel (no per-voice PCM ever reaches
  HBM), ``generate`` + ``mix_bus`` is the reference-shaped two-step form (voices materialised as
  float32 PCM, then summed) whose mix step is HBM-bound.
* ``mix_samples``: the reference's real-time mixer rule over integer PCM chunks (upstream
  playback.py mixer loop: ``mixed = audioop.add(mixed, chunk, width)`` for every active voice, in
  order) -- an order-dependent chain of saturating adds, reproduced bit-exactly.
* ``RealTimeMixer``: that loop itself, chunk by chunk: samples are added while it runs, may repeat or start a few
  chunks late, and every ``next(chunks())`` folds the current chunk of every active sample, read in place in HBM
  (``sh_mix_chain_gather_i16``: a pointer table instead of a staging copy), into one chunk of output.

``pan`` follows the linear law used throughout: gain_l = (1 - pan) / 2, gain_r = (1 + pan) / 2.
"""
from __future__ import annotations

import ctypes as C
import gc
import threading
from typing import Callable, Dict, Generator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native as N
from . import params
from .oscillators import Oscillator, VoiceSpec, _pwm_widths, _table, pack_voices, time_step_weights
from .sample import Sample

__all__ = ["VoiceBank", "RealTimeMixer", "mix_samples", "pan_gains"]


def pan_gains(pan: float) -> Tuple[float, float]:
    return (1.0 - pan) / 2.0, (1.0 + pan) / 2.0


class _RowMatrix:
    """The per-launch float64 rows of a VoiceBank (sh_bank_render_rows): rows 0 .. nfm-1 hold the running sums of the fm_lfo
    modulators (rendered, then scanned in place by ONE batched launch, the carries staying on the device from block to
    block), the rows after them pulse widths / rendered voices.  Sources that are single closed-form records are rendered by
    one launch per group (a bank of modulators, sh_bank_generate_f64); anything else -- a modulator that is itself
    modulated by an arbitrary oscillator, a filter graph -- renders into its row on its own."""

    def __init__(self, fm_sources: Sequence[Oscillator], other_sources: Sequence[Oscillator], samplerate: int,
                 other_roles: Optional[Sequence[str]] = None, fm_incs: Optional[Sequence[float]] = None) -> None:
        self.nfm = len(fm_sources)
        self.nrows = len(fm_sources) + len(other_sources)
        self.samplerate = samplerate
        self.fm_rows: List[int] = []
        self.other_rows: List[int] = []
        self._banks: List[Tuple[N.Bank, int, bool]] = []        # (bank of closed-form sources, first row, fm rows?)
        self._pwm_bank_rows: List[int] = []                     # rows of pulse widths among them (params.variants["pulse"])
        self._singles: List[Tuple[Oscillator, int, str]] = []   # (source, row, "fm" | "pwm" | "voice")

        def place(sources, first_row, out_rows, roles):
            bankable, single = [], []
            is_fm = roles is None
            for k, m in enumerate(sources):
                if m.samplerate != samplerate:
                    raise ValueError("a modulator must run at the sample rate of its voice")
                try:
                    sp = m.spec()
                    ok = sp.fm_mode != N.SH_FM_BUFFER and not sp.needs_pwm and m.length is None
                except NotImplementedError:
                    sp, ok = None, False
                (bankable if ok else single).append((k, m, sp))
            rows = [0] * len(sources)
            r = first_row
            if bankable:
                self._banks.append((N.Bank(*pack_voices([sp for _, _, sp in bankable])), r, is_fm))
                for k, _m, _sp in bankable:
                    rows[k] = r
                    if not is_fm and roles[k] == "pwm":
                        self._pwm_bank_rows.append(r)
                    r += 1
            for k, m, _sp in single:
                rows[k] = r
                self._singles.append((m, r, "fm" if is_fm else roles[k]))
                r += 1
            out_rows.extend(rows)

        place(fm_sources, 0, self.fm_rows, None)
        # the time step inc of every fm row's CARRIER (2 pi / sr for Sine / Harmonics, 1 / sr for the turn-based kinds): the row is
        # weighted with the accumulated time's actual steps over inc in front of the scan (oscillators.time_step_weights); runs of
        # neighbouring rows with the same inc are weighted by one call
        self._row_inc: List[float] = [0.0] * self.nfm
        if fm_incs is not None:
            for k, r in enumerate(self.fm_rows):
                self._row_inc[r] = float(fm_incs[k])
        place(other_sources, self.nfm, self.other_rows, list(other_roles) if other_roles is not None else ["voice"] * len(other_sources))
        self._buf: Optional[N.DeviceBuffer] = None
        self._stride = 0
        self._carry: Optional[N.DeviceBuffer] = None
        if self.nfm:
            self._carry = N.DeviceBuffer(self.nfm * 8)
            self._carry.zero()
        self._pos = 0                                           # the carries hold L(_pos)

    def _render(self, start: int, n: int, fm_only: bool = False) -> None:
        """Rows of frames [start, start + n).  fm_only: a replay of the running sums on the way to a later block -- only the fm
        rows carry state from block to block, the others are not touched."""
        L = N.lib()
        for bank, row0, is_fm in self._banks:
            if is_fm or not fm_only:
                N.check(L.sh_bank_generate_f64(bank.handle, start, n, self._buf.handle, row0, self._stride))
        if not fm_only:
            for row in self._pwm_bank_rows:                         # (closed-form pwm sources rendered by the bank launch above)
                _pwm_widths(self._buf, row * self._stride, n)
        for m, row, role in self._singles:
            if fm_only and role != "fm":
                continue
            # a VOICE that ends (an envelope with stop_at_end over a filter graph, a filter over a finite source) is silent from
            # there on, like a fused stop_at_end voice of the same bank; a modulator that ends before its carrier is an error,
            # as upstream (the carrier's next() on it raises)
            avail = n
            if role == "voice" and m.length is not None:
                avail = max(0, min(n, m.length - start))
            if avail:
                m._render_device(start, avail, out_f64=self._buf.view(row * self._stride * 8, avail * 8))
                if role == "pwm":
                    _pwm_widths(self._buf, row * self._stride, avail)
            if avail < n:
                N.check(L.sh_ew_f64(N.SH_EW_FILL, None, 0, None, 0, n - avail, 0.0, 0.0, self._buf.handle, row * self._stride + avail, None, 0, None))
        if self.nfm:
            r0 = 0
            while r0 < self.nfm:
                r1 = r0 + 1
                while r1 < self.nfm and self._row_inc[r1] == self._row_inc[r0]:
                    r1 += 1
                if self._row_inc[r0] != 0.0:
                    runs = time_step_weights(self._row_inc[r0], start, n)
                    if len(runs) == 1 and runs[0][:2] == (0, n):          # one piece of the time table: rows r0 .. r1 - 1 in one call
                        if runs[0][2] != 0.0:
                            o = r0 * self._stride
                            N.check(L.sh_ew_f64(N.SH_EW_AXPY, self._buf.handle, o, self._buf.handle, o, (r1 - r0 - 1) * self._stride + n,
                                                runs[0][2], 0.0, self._buf.handle, o, None, 0, None))
                    else:                                                  # a piece of the time table ends inside the block (rare)
                        for r in range(r0, r1):
                            for off, cnt, wm1 in runs:
                                if wm1 != 0.0:
                                    o = r * self._stride + off
                                    N.check(L.sh_ew_f64(N.SH_EW_AXPY, self._buf.handle, o, self._buf.handle, o, cnt, wm1, 0.0,
                                                        self._buf.handle, o, None, 0, None))
                r0 = r1
            N.check(L.sh_scan_rows_f64(self._buf.handle, 0, self.nfm, n, self._stride, self._carry.handle))
            self._pos = start + n

    def fill(self, start: int, n: int) -> Tuple[N.DeviceBuffer, int]:
        if self._buf is None or n > self._stride:
            self._stride = max(n, 4096)
            self._buf = N.DeviceBuffer(self.nrows * self._stride * 8)
        if self.nfm and start != self._pos:                     # random access into a recurrence: replay the sums up to `start`
            if start < self._pos:
                self._carry.zero()
                self._pos = 0
            while self._pos < start:
                self._render(self._pos, min(self._stride, start - self._pos), fm_only=True)
        self._render(start, n)
        return self._buf, self._stride


class VoiceBank:
    """N oscillator voices summed to one stereo bus on the GPU."""

    def __init__(self, voices: Sequence[Oscillator], gains: Optional[Sequence[Tuple[float, float]]] = None,
                 pans: Optional[Sequence[float]] = None) -> None:
        if not voices:
            raise ValueError("a voice bank needs at least one voice")
        if gains is not None and pans is not None:
            raise ValueError("give gains or pans, not both")
        if pans is not None:
            gains = [pan_gains(p) for p in pans]
        if gains is None:
            gains = [(1.0, 1.0)] * len(voices)
        if len(gains) != len(voices):
            raise ValueError("one (left, right) gain pair per voice")
        rates = {v.samplerate for v in voices}
        if len(rates) != 1:
            raise ValueError("all voices of a bank must share one sample rate")
        self.samplerate = rates.pop()
        self.nvoices = len(voices)
        self.gains = [(float(l), float(r)) for l, r in gains]
        # Voices that read a ROW of a per-launch float64 matrix: a carrier whose fm_lfo is not a closed-form Sine (the row is
        # the running sum of the modulator), a Pulse with a pwm_lfo (pulse width per sample), or a voice that is no single
        # record at all -- a filter graph, an envelope over one -- whose samples are rendered into its row (SH_BUFFER).
        specs: List[VoiceSpec] = []
        fm_src: List[Tuple[int, Oscillator]] = []           # (voice, modulator): rows 0 .. nfm-1, scanned in place
        fm_incs: List[float] = []                           # ... and the time step of the voice (the carrier) itself
        other_src: List[Tuple[str, int, Oscillator]] = []   # ("pwm" | "voice", voice, source): the rows after them
        # (the collector off while the records are made: a table of 200 000 notes is a million live objects, and every generation-2
        # pass walks them all -- 2.5 s with it, 1.35 s without; nothing in the loop makes a cycle)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            for i, v in enumerate(voices):
                try:
                    sp = v.spec()
                except NotImplementedError:
                    sp = VoiceSpec(kind=N.SH_BUFFER, amplitude=1.0, bias=0.0, fm_mode=N.SH_FM_NONE, carrier=_table(0.0, 0.0))
                    other_src.append(("voice", i, v))
                else:
                    if sp.fm_mode == N.SH_FM_BUFFER:
                        fm_src.append((i, v._fm_source()))
                        fm_incs.append(sp.fm_inc)
                    if sp.needs_pwm:
                        other_src.append(("pwm", i, v._pwm_source()))
                specs.append(sp)
            self._packed = pack_voices(specs, self.gains)
        finally:
            if gc_was_on:
                gc.enable()
        self._bank = N.Bank(*self._packed)
        self._gains_dev: Optional[N.DeviceBuffer] = None
        self._pan_dev: Optional[N.DeviceBuffer] = None
        self._rows: Optional[_RowMatrix] = None
        if fm_src or other_src:
            fm_row = np.full(self.nvoices, -1, dtype=np.int32)
            pwm_row = np.full(self.nvoices, -1, dtype=np.int32)
            self._rows = _RowMatrix([m for _, m in fm_src], [m for _, _, m in other_src], self.samplerate,
                                    other_roles=[what for what, _, _ in other_src], fm_incs=fm_incs)
            for (i, _m), r in zip(fm_src, self._rows.fm_rows):
                fm_row[i] = r
            for (what, i, _m), r in zip(other_src, self._rows.other_rows):
                (pwm_row if what == "pwm" else fm_row)[i] = r
            N.check(N.lib().sh_bank_set_rows(self._bank.handle, fm_row.ctypes.data, pwm_row.ctypes.data))

    # -- fused path ------------------------------------------------------------------------------
    def render_device(self, nframes: int, start: int = 0, bus_f32: Optional[N.DeviceBuffer] = None,
                      bus_f64: Optional[N.DeviceBuffer] = None) -> N.DeviceBuffer:
        """Fused generate-and-mix into device buffers (float32 frames x 2 and/or float64 frames x 2)."""
        if bus_f32 is None and bus_f64 is None:
            bus_f32 = N.DeviceBuffer(nframes * 8)
        if self._rows is not None and nframes:
            rows, stride = self._rows.fill(start, nframes)
            N.check(N.lib().sh_bank_render_rows(self._bank.handle, start, nframes, rows.handle, stride,
                                                bus_f32.handle if bus_f32 is not None else None,
                                                bus_f64.handle if bus_f64 is not None else None))
            return bus_f32 if bus_f32 is not None else bus_f64
        N.check(N.lib().sh_bank_render(self._bank.handle, start, nframes,
                                       bus_f32.handle if bus_f32 is not None else None,
                                       bus_f64.handle if bus_f64 is not None else None))
        return bus_f32 if bus_f32 is not None else bus_f64

    def render_run(self, nframes: int, nblocks: int, start: int = 0, ring: Optional[Sequence[N.DeviceBuffer]] = None,
                   pcm_ring: Optional[Sequence[N.DeviceBuffer]] = None, scale: float = 32767.0) -> Sequence[N.DeviceBuffer]:
        """``nblocks`` consecutive blocks of ``nframes`` frames in ONE call: block k goes to ``ring[k % len(ring)]`` (float32 stereo)
        and / or ``pcm_ring[k % len(pcm_ring)]`` (saturated int16 stereo).  Ring buffers that are consecutive windows of one
        allocation (``VoiceBank.make_ring``) are filled by one launch per stretch: a run of blocks of a small bank costs one
        launch, not one per block.  Without a ring: a fresh contiguous one of ``nblocks`` float32 buffers.  Returns the ring."""
        if self._rows is not None:
            raise NotImplementedError("render_run: a bank with modulation rows renders block by block (render_device)")
        if ring is None and pcm_ring is None:
            ring = self.make_ring(nframes, nblocks)
        n32 = len(ring) if ring is not None else 0
        n16 = len(pcm_ring) if pcm_ring is not None else 0
        if n32 and n16 and n32 != n16:
            raise ValueError("the float32 and the PCM ring must have the same number of slots")
        nring = n32 or n16
        a32 = (C.c_void_p * nring)(*[b.handle for b in ring]) if n32 else None
        a16 = (C.c_void_p * nring)(*[b.handle for b in pcm_ring]) if n16 else None
        N.check(N.lib().sh_bank_render_run(self._bank.handle, start, nframes, nblocks, a32, a16, nring, float(scale)))
        return ring if ring is not None else pcm_ring

    @staticmethod
    def make_ring(nframes: int, nslots: int, bytes_per_frame: int = 8) -> Sequence[N.DeviceBuffer]:
        """``nslots`` bus buffers of ``nframes`` frames that are consecutive windows of ONE allocation (8 bytes per frame: float32
        stereo; 4: int16 stereo PCM): ``render_run`` fills neighbouring slots with one launch."""
        whole = N.DeviceBuffer(nslots * nframes * bytes_per_frame)
        return [whole.view(k * nframes * bytes_per_frame, nframes * bytes_per_frame) for k in range(nslots)]

    def render(self, nframes: int, start: int = 0) -> np.ndarray:
        """Stereo bus as a [nframes, 2] float32 array."""
        if nframes == 0:
            return np.zeros((0, 2), dtype=np.float32)
        buf = self.render_device(nframes, start)
        out = buf.download(np.float32, nframes * 2).reshape(nframes, 2)
        buf.free()
        return out

    def render_pcm_device(self, nframes: int, start: int = 0, scale: float = 32767.0,
                          pcm: Optional[N.DeviceBuffer] = None) -> N.DeviceBuffer:
        """Fused generate-and-mix delivered as saturated int16 stereo PCM (nframes x 2) in a device buffer: the fold of
        the voice groups' partial buses quantises as it goes, so consecutive blocks stay pipelined (no quantise kernel
        between them).  The buffer is complete once any other library call (or ``_native.sync()``) has been made."""
        if pcm is None:
            pcm = N.DeviceBuffer(nframes * 4)
        if self._rows is not None:                       # modulation rows: float32 bus first, then the saturating quantiser
            bus = self.render_device(nframes, start)
            N.check(N.lib().sh_quantize_clip_f32(bus.handle, nframes * 2, float(scale), pcm.handle))
            bus.free()
            return pcm
        N.check(N.lib().sh_bank_render_pcm(self._bank.handle, start, nframes, float(scale), pcm.handle))
        return pcm

    def render_sample(self, nframes: int, start: int = 0, scale: float = 32767.0) -> Sample:
        """Stereo bus quantised to an int16 Sample (saturating: a bus can exceed full scale)."""
        s = Sample(samplerate=self.samplerate, nchannels=2, samplewidth=2)
        if nframes == 0:
            return s
        s._set_device(self.render_pcm_device(nframes, start, scale), nframes * 4)
        return s

    # -- two-step (reference-shaped) path -----------------------------------------------------------
    def generate_device(self, nframes: int, start: int = 0, out: Optional[N.DeviceBuffer] = None,
                        stride: Optional[int] = None) -> N.DeviceBuffer:
        """Every voice as float32 PCM in HBM, voice-major: out[v*stride + i]."""
        stride = nframes if stride is None else stride
        if out is None:
            out = N.DeviceBuffer(self.nvoices * stride * 4)
        if self._rows is not None and nframes:
            # voices that read rows of the launch's float64 matrix (arbitrary fm_lfo / pwm_lfo, filter graphs): the same matrix
            # the fused render fills, read by the general materialisation kernel
            rows, rstride = self._rows.fill(start, nframes)
            N.check(N.lib().sh_bank_generate_rows(self._bank.handle, start, nframes, rows.handle, rstride, out.handle, stride))
            return out
        N.check(N.lib().sh_bank_generate(self._bank.handle, start, nframes, out.handle, stride))
        return out

    def generate(self, nframes: int, start: int = 0) -> np.ndarray:
        buf = self.generate_device(nframes, start)
        out = buf.download(np.float32, self.nvoices * nframes).reshape(self.nvoices, nframes)
        buf.free()
        return out

    # -- two-step, integer: the route upstream itself takes (oscillator block -> Sample.from_osc_block -> mixer) ---------------
    def generate_i16_device(self, nframes: int, start: int = 0, scale: float = 32767.0, out: Optional[N.DeviceBuffer] = None,
                            stride: Optional[int] = None, check: bool = True) -> Tuple[N.DeviceBuffer, int]:
        """Every voice as int16 PCM in HBM, voice-major: out[v*stride + i] = int(scale * sample) -- ``Sample.from_osc_block`` of
        every voice's block in one launch (OverflowError where a sample does not fit).  Returns (buffer, stride); the stride is
        ``nframes`` rounded up to a multiple of 64 unless given (it must be even).  ``check=False`` (a stream of blocks): the
        call only enqueues; ``VoiceBank.overflow_check()`` raises for every block since the last check at once."""
        stride = (nframes + 63) & ~63 if stride is None else stride
        if out is None:
            out = N.DeviceBuffer(max(self.nvoices * stride * 2, 4))
        if nframes == 0:
            return out, stride
        if params.variants["quantise"] == "round":
            # the other reading of the quantiser (round half to even): the fused epilogue truncates, so every voice goes through
            # float64 rows and the library's quantiser, which follows the option (a contingency path: 8 B per voice-sample on the way)
            if self._rows is not None:
                raise NotImplementedError("generate_i16_device under the round() quantise variant: banks with modulation rows")
            r64 = N.DeviceBuffer(self.nvoices * stride * 8)
            N.check(N.lib().sh_bank_generate_f64(self._bank.handle, start, nframes, r64.handle, 0, stride))
            try:
                for v in range(self.nvoices):
                    N.check(N.lib().sh_quantize_f64(r64.handle, v * stride, nframes, float(scale), 2, out.handle, v * stride))
            finally:
                r64.free()
            return out, stride
        if self._rows is not None:
            rows, rstride = self._rows.fill(start, nframes)
            N.check(N.lib().sh_bank_generate_rows_i16(self._bank.handle, start, nframes, rows.handle, rstride, float(scale), out.handle, stride))
        else:
            fn = N.lib().sh_bank_generate_i16 if check else N.lib().sh_bank_generate_i16_async
            N.check(fn(self._bank.handle, start, nframes, float(scale), out.handle, stride))
        return out, stride

    @staticmethod
    def overflow_check() -> None:
        """OverflowError if a ``generate_i16_device(..., check=False)`` since the last check met a sample that does not fit."""
        N.check(N.lib().sh_overflow_check())

    def voice_samples(self, nframes: int, start: int = 0, scale: float = 32767.0) -> List[Sample]:
        """The voices as mono int16 Samples (views of one matrix in HBM): ``[Sample.from_osc_block(voice block) ...]``."""
        buf, stride = self.generate_i16_device(nframes, start, scale)
        out = []
        for v in range(self.nvoices):
            s = Sample(samplerate=self.samplerate, nchannels=1, samplewidth=2)
            if nframes:
                s._set_device(buf.view(v * stride * 2, nframes * 2), nframes * 2)
                s._share_device()                              # (a window of the shared matrix: never written in place)
            out.append(s)
        return out

    def mixdown_i16_device(self, nframes: int, start: int = 0, scale: float = 32767.0,
                           out: Optional[N.DeviceBuffer] = None, check: bool = True, two_step: bool = False) -> N.DeviceBuffer:
        """The mono mixdown the reference's mixer makes of the voices: every voice quantised (``from_osc_block``), then
        ``mixed = audioop.add(mixed, voice, 2)`` down the voices in order.  nframes int16 samples.  The library folds the samples
        into the chain where they are made wherever it can (``sh_bank_mixdown_i16``); ``two_step=True`` materialises the int16 rows
        and runs the chain kernel over them -- the same bytes."""
        if out is None:
            out = N.DeviceBuffer(max(nframes * 2, 4))
        if nframes == 0:
            return out
        if two_step or self._rows is not None or params.variants["quantise"] == "round":
            rows, stride = self.generate_i16_device(nframes, start, scale, check=check)
            N.check(N.lib().sh_mix_chain_i16(rows.handle, self.nvoices, stride, nframes, out.handle))
            rows.free()
            return out
        fn = N.lib().sh_bank_mixdown_i16 if check else N.lib().sh_bank_mixdown_i16_async
        N.check(fn(self._bank.handle, start, nframes, float(scale), out.handle))
        return out

    def pan_factors_device(self) -> N.DeviceBuffer:
        if self._pan_dev is None:
            self._pan_dev = N.DeviceBuffer.from_array(np.asarray(self.gains, dtype=np.float64).reshape(-1))
        return self._pan_dev

    def mixdown_stereo_i16_device(self, nframes: int, start: int = 0, scale: float = 32767.0,
                                  out: Optional[N.DeviceBuffer] = None) -> N.DeviceBuffer:
        """The stereo mixdown the reference's mixer makes: every voice quantised (``from_osc_block``), placed with
        ``Sample.stereo(left, right)`` (``audioop.tostereo`` with the voice's gains as factors), then the saturating
        ``audioop.add`` chain in voice order.  nframes x 2 int16, interleaved."""
        rows, stride = self.generate_i16_device(nframes, start, scale)
        if out is None:
            out = N.DeviceBuffer(max(nframes * 4, 4))
        if nframes:
            N.check(N.lib().sh_mix_chain_pan_i16(rows.handle, self.nvoices, stride, nframes, self.pan_factors_device().handle, out.handle))
        rows.free()
        return out

    def gains_device(self) -> N.DeviceBuffer:
        if self._gains_dev is None:
            self._gains_dev = N.DeviceBuffer.from_array(np.asarray(self.gains, dtype=np.float32).reshape(-1))
        return self._gains_dev

    def mix_device(self, voices: N.DeviceBuffer, nframes: int, stride: Optional[int] = None,
                   bus_f32: Optional[N.DeviceBuffer] = None) -> N.DeviceBuffer:
        """Sum materialised voices to the stereo bus (HBM-bound: 4*N + 8 bytes per frame)."""
        stride = nframes if stride is None else stride
        if bus_f32 is None:
            bus_f32 = N.DeviceBuffer(nframes * 8)
        N.check(N.lib().sh_mix_bus_f32(voices.handle, self.nvoices, stride, nframes,
                                       self.gains_device().handle, bus_f32.handle))
        return bus_f32

    def render_two_step(self, nframes: int, start: int = 0) -> np.ndarray:
        v = self.generate_device(nframes, start)
        bus = self.mix_device(v, nframes)
        out = bus.download(np.float32, nframes * 2).reshape(nframes, 2)
        v.free()
        bus.free()
        return out


def mix_bus(voices: np.ndarray, gains: Sequence[Tuple[float, float]]) -> np.ndarray:
    """[nvoices, nframes] float32 voices -> [nframes, 2] float32 bus (host arrays in and out)."""
    voices = np.ascontiguousarray(voices, dtype=np.float32)
    nv, nf = voices.shape
    vb = N.DeviceBuffer.from_array(voices)
    gb = N.DeviceBuffer.from_array(np.asarray(gains, dtype=np.float32).reshape(-1))
    bus = N.DeviceBuffer(max(nf, 1) * 8)
    N.check(N.lib().sh_mix_bus_f32(vb.handle, nv, nf, nf, gb.handle, bus.handle))
    out = bus.download(np.float32, nf * 2).reshape(nf, 2)
    for b in (vb, gb, bus):
        b.free()
    return out


def _gather(sources: Sequence[Tuple[N.DeviceBuffer, int, int]], nsamples: int, out: N.DeviceBuffer, width: int = 2,
            out_sample_off: int = 0) -> None:
    """out[out_sample_off : +nsamples] = ordered saturating fold of (buffer, first sample, samples available) sources of
    `width`-byte samples: mixed = audioop.add(mixed, chunk, width) down the list."""
    n = len(sources)
    bufs = (C.c_void_p * max(n, 1))(*[b.handle for b, _o, _n in sources])
    offs = (C.c_size_t * max(n, 1))(*[o for _b, o, _n in sources])
    lens = (C.c_uint32 * max(n, 1))(*[min(k, 0xFFFFFFFF) for _b, _o, k in sources])
    N.check(N.lib().sh_mix_chain_gather(bufs, offs, lens, n, nsamples, width, out.handle, out_sample_off))


def mix_samples(samples: Sequence[Sample], name: str = "mix") -> Sample:
    """The real-time mixer's fold over whole samples: pad every voice with silence to the longest,
    then ``mixed = add(mixed, voice)`` in the given order, saturating at every step."""
    if not samples:
        raise ValueError("nothing to mix")
    first = samples[0]
    for s in samples[1:]:
        assert s.samplewidth == first.samplewidth and s.samplerate == first.samplerate and s.nchannels == first.nchannels
    width = first.samplewidth
    nbytes = max(len(s) for s in samples) * width * first.nchannels
    out = Sample(name=name, samplerate=first.samplerate, nchannels=first.nchannels, samplewidth=width)
    if nbytes == 0:
        return out
    nsamples = nbytes // width
    dst = N.DeviceBuffer(nbytes)
    _gather([(s._device(), 0, len(s) * s.nchannels) for s in samples], nsamples, dst, width)
    out._set_device(dst, nbytes)
    return out


class _MixSource:
    """One playing sample: its PCM in HBM (a repeating one carries one extra chunk of its own start behind its end,
    so every chunk is one contiguous range), the play position, and how many silent chunks come first."""

    def __init__(self, name: str, buf: N.DeviceBuffer, nbytes: int, loop_bytes: int, delay: int) -> None:
        self.name = name
        self.buf = buf
        self.nbytes = nbytes            # one-shot: length; repeating: length of the looped data (before the extra chunk)
        self.loop_bytes = loop_bytes    # 0 = one-shot
        self.pos = 0
        self.delay = delay
        self.acquired = False           # ordered behind its producers on the mixer's real-time lane (RealTimeMixer.chunks)


class RealTimeMixer:
    """Real-time sample mixer: samples play from the moment they are added; every turn of ``chunks()`` yields
    ``chunksize`` bytes, the saturating sum -- in the order the samples were added -- of the current chunk of every
    active sample.  A sample that ran out is dropped (``all_played_callback`` fires when the last one goes); with
    nothing playing the mixer yields silence.  Samples must be in the mixer's format (``samplewidth`` bytes per sample, 16-bit
    by default; 8-, 24- and 32-bit mixers fold with the same rule, ``audioop.add(mixed, chunk, samplewidth)``).  Mirrors upstream ``synthplayer/playback.py`` ``RealTimeMixer`` ([RECALL], tree not
    mounted): ``add_sample`` / ``remove_sample`` / ``clear_sources`` / ``chunks``.

    The PCM stays in HBM: ``add_sample`` uploads (or reuses the resident buffer of) the sample once, a chunk turn is
    one kernel over a pointer table plus ``chunksize`` bytes back to the host.  ``chunks_device()`` yields the device
    buffer instead, for a consumer that keeps going on the GPU (level metering, resampling)."""

    use_lane = True           # chunks() on the mixer's real-time lane; False: through the library's lock and stream (rounds 1-5; the A/B of tests/test_gpu_realtime_mixer.py)

    def __init__(self, chunksize: int, all_played_callback: Optional[Callable[[], None]] = None, samplewidth: int = 2) -> None:
        if samplewidth not in (1, 2, 3, 4):
            raise ValueError("samplewidth must be 1, 2, 3 or 4")
        if chunksize <= 0 or chunksize % samplewidth:
            raise ValueError("chunksize must be a positive whole number of samples")
        self.chunksize = chunksize
        self.samplewidth = samplewidth
        self.all_played_callback = all_played_callback or (lambda: None)
        self.add_lock = threading.Lock()
        self.chunks_mixed = 0
        self.active_samples: Dict[int, _MixSource] = {}
        self.sample_counter = 0
        self._out = [N.DeviceBuffer(chunksize), N.DeviceBuffer(chunksize)]      # the consumer may still hold the last one
        self._lane: Optional[N.RtLane] = None                                    # created by the first chunks() turn

    def add_sample(self, sample: Sample, repeat: bool = False, chunk_delay: int = 0, sid: Optional[int] = None) -> int:
        """Start playing a sample; returns its id.  ``repeat`` loops it forever, ``chunk_delay`` holds it back."""
        if sample.samplewidth != self.samplewidth:
            raise ValueError("sample width must be %d" % self.samplewidth)
        nbytes = len(sample) * sample.samplewidth * sample.nchannels
        if repeat and nbytes:
            # upstream: data repeated up to at least one chunk, then one more chunk of its start appended
            reps = -(-self.chunksize // nbytes) if nbytes < self.chunksize else 1
            loop = nbytes * reps
            buf = N.DeviceBuffer(loop + self.chunksize)
            src = sample._device()
            for r in range(reps):
                N.check(N.lib().sh_buf_copy(buf.handle, r * nbytes, src.handle, 0, nbytes))
            N.check(N.lib().sh_buf_copy(buf.handle, loop, buf.handle, 0, self.chunksize))
            source = _MixSource(sample.name, buf, loop, loop, chunk_delay)
        else:
            # the Sample's own buffer, read in place: _share_device() tells the Sample never to write it in place from now on
            # (Sample.mix adds in place when nothing grows -- it would change, and race with, the audio being streamed)
            source = _MixSource(sample.name, sample._share_device(), nbytes, 0, chunk_delay)
        with self.add_lock:
            self.sample_counter += 1
            sid = sid or self.sample_counter
            self.active_samples[sid] = source
            return sid

    def remove_sample(self, sid_or_name: Union[int, str]) -> None:
        with self.add_lock:
            if isinstance(sid_or_name, int):
                self.active_samples.pop(sid_or_name, None)
            else:
                for sid in [i for i, src in self.active_samples.items() if src.name == sid_or_name]:
                    del self.active_samples[sid]

    def clear_sources(self) -> None:
        with self.add_lock:
            self.active_samples.clear()
        self.all_played_callback()

    def _advance(self, lane: Optional[N.RtLane] = None) -> list:
        """One turn of the play positions: the (buffer, first sample, samples available) of every source that sounds in this chunk.
        With a real-time lane: a source the lane has not seen yet is ordered behind its producers first (sh_rt_acquire)."""
        with self.add_lock:
            active = list(self.active_samples.items())
        if lane is not None:
            for _sid, src in active:
                if not src.acquired:
                    lane.acquire(src.buf)
                    src.acquired = True
        sources = []
        finished = []
        for sid, src in active:
            if src.delay > 0:
                src.delay -= 1
                continue
            w = self.samplewidth
            if src.loop_bytes:
                sources.append((src.buf, src.pos // w, self.chunksize // w))
                src.pos = (src.pos + self.chunksize) % src.loop_bytes
            elif src.pos < src.nbytes:
                n = min(self.chunksize, src.nbytes - src.pos)
                sources.append((src.buf, src.pos // w, n // w))
                src.pos += self.chunksize
            else:
                finished.append(sid)
        if finished:
            with self.add_lock:
                for sid in finished:
                    self.active_samples.pop(sid, None)
                empty = not self.active_samples
            if empty:
                self.all_played_callback()
        return sources

    def _turn(self) -> N.DeviceBuffer:
        sources = self._advance()
        out = self._out[self.chunks_mixed & 1]
        _gather(sources, self.chunksize // self.samplewidth, out, self.samplewidth)
        self.chunks_mixed += 1
        return out

    def _turn_host(self) -> bytes:
        """One chunk on the mixer's REAL-TIME LANE (sh_rt_*): a stream, a lock and buffers of the mixer's own -- the turn neither takes the
        library's lock nor queues behind what other threads have enqueued (a bank streaming its blocks, Sample operations); the same
        fold kernels as ``_turn``: bit-identical chunks.  A source is ordered behind its producers once, before its first turn."""
        if self._lane is None:
            self._lane = N.RtLane(self.chunksize, max_sources=32768)
        sources = self._advance(self._lane)
        chunk = self._lane.mix_turn(sources, self.chunksize // self.samplewidth, self.samplewidth)
        self.chunks_mixed += 1
        return chunk

    def chunks_device(self) -> Generator[N.DeviceBuffer, None, None]:
        """Endless stream of mixed chunks left in HBM (valid until the turn after the next)."""
        while True:
            yield self._turn()

    def chunks(self) -> Generator[memoryview, None, None]:
        """Endless stream of mixed chunks, ``chunksize`` bytes each (on the mixer's real-time lane: see ``_turn_host``)."""
        while True:
            yield memoryview(self._turn_host() if self.use_lane else self._turn().download_bytes(self.chunksize))
