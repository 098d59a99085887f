#!/usr/bin/env python
"""bench.py -- headline benchmark of the oscillator-bank + mixer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1.  Started by a launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process IS a
rank; started plainly with --gpus N > 1 it becomes the launcher: it starts N copies of itself, one per GPU ordinal, with that
environment, and rank 0 prints the one JSON line.  Either way the ranks talk through synthesizer_amd.dist.Rendezvous (a TCP star
next to MASTER_PORT for data, a page of /dev/shm for the barriers that bracket the timed passes): no torch anywhere.

Workload (BASELINE.json metric): 1024 additive voices PER GPU -- each a
Harmonics oscillator with 16 partials a_k = 1/k under an ADSR envelope, f log-uniform in [55, 3520] Hz,
random phase/pan, seed 0 -- mixed to one float32 stereo bus at 48 kHz.  One step = one block of 48 000
frames (1 s of audio) rendered by the fused generate-and-mix kernel; steps render consecutive blocks and
the envelope's sustain spans the whole run (no silent voices).  With N > 1 the voice table is sharded
across ranks and the float64 partial buses are summed to rank 0 by RCCL: `--scaling weak` (default)
N x 1024 voices (configs[3] at N = 8), `--scaling strong` 8192 voices whatever N.

Timing: W untimed warm-up steps, then PASSES of exactly K steps, each pass bracketed by barrier +
device synchronisation on both sides and clocked on every rank (maximum over ranks).  Passes repeat
(consecutive blocks of the same stream) until the timed passes add up to >= --min-seconds (5.5 s): one
pass of the driver's K = 20 lasts under a millisecond, less than the GPU's clock governor needs to
leave its idle state (the first ~50 ms of a run are up to 25 % slower), so a single pass measures the
governor, not the kernel.  `value` / `ms_per_step` are those of the MEDIAN pass (`passes` lists count,
median, min, first); a step stays one block.

value = voice-samples mixed per second over all ranks / 1e6 ("Msamples/s"), inputs (the voice table)
resident in HBM before the timed region.  The same run also times the other BASELINE configs
(`configs`), the reference-shaped two-step path (voices materialised as float32 PCM in HBM, then the
HBM-bound mixer kernel), the PCM rows (resample = configs[4], integer mixer chain, audioop.add ...) and
the CPU oracle (pure-Python generators; its C restatement on one core and on all host cores) on a bounded sample -- reported
beside, never as `value`.  Also beside: `staggered_notes` (a table of 22 528 notes that do not move in lock-step, one-second blocks
and 4096-frame chunks), `config4` at N = 1 (the 8192-voice table; one rank's share through the RCCL ring) and `verified` (the last
timed block against a fresh single render, bit for bit).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SR = 48000
VOICES_PER_GPU = 1024
STRONG_VOICES = 8192           # BASELINE configs[3]
PARTIALS = 16
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_PEAK_TOPS = 256 * 4 * 16 * 2.4 / 1e3   # float64 VALU issue peak: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s
                                            # (an FMA, a MUL and an ADD each take one lane-op slot; 78.6 TFLOP/s spec = 2 x this)
ADSR_BENCH = {"sustain": 1.0e6}             # attack 0.01, decay 0.05, release 0.2 as SURVEY 8(d); the sustain spans any run


def launch_ranks(argv, n: int, dry: bool = False) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks -- this file again, rank r on GPU ordinal r
    (synthesizer_amd.dist.rank_env: RANK = LOCAL_RANK = r, WORLD_SIZE = N, MASTER_ADDR 127.0.0.1, a free MASTER_PORT) -- and wait for
    them.  stdout / stderr are inherited: rank 0's JSON line is this process's output.  The first rank that fails takes the others
    with it (exactly the processes started here); the exit status is that rank's."""
    import subprocess
    from synthesizer_amd import dist
    port = dist.free_port()
    procs = []
    for r in range(n):
        env = dist.rank_env(r, n, port)
        env["SYNTHHIP_BENCH_LAUNCHED"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, alive = 0, list(procs)
    try:
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    print("bench.py: rank %d exited with status %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
                    for q in alive:
                        q.terminate()
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class _DryNative:
    """SYNTHHIP_BENCH_DRY_RUN=1 (tests/test_bench_launch.py, no GPU): what main() needs of synthesizer_amd._native to walk the launch /
    rendezvous / device check / timed passes / gather plumbing of an N-rank run on CPU.  Renders nothing; the line it leads to says
    "data": "dry-run" and is no measurement."""

    class _Lib:
        @staticmethod
        def sh_version():
            return b"dry-run (no GPU)"

    def __init__(self):
        self._dev = 0
        self._t0 = 0.0

    def ensure_init(self, device=0):
        self._dev = int(device)

    def device_info(self):
        return {"name": "dry-run", "arch": "none", "device": self._dev}

    def device_pci(self):
        if os.environ.get("SYNTHHIP_BENCH_DRY_SAME_PCI") == "1":      # (the test of the device check: two ranks on ONE GPU must not pass)
            return "dry0:00:00.0"
        return "dry0:%02x:00.0" % self._dev

    def lib(self):
        return self._Lib

    def sync(self):
        pass

    def timer_start(self):
        self._t0 = time.perf_counter()

    def timer_stop(self):
        return (time.perf_counter() - self._t0) * 1e3


class _DryBackend:
    """DistVoiceBank's device side for the dry run: buffers are labels, a render is a 20-us pause, a reduce is nothing."""

    def nslots(self):
        return 4

    def alloc(self, nbytes):
        return ("buf", nbytes)

    def view(self, buf, offset, nbytes):
        return ("view", offset, nbytes)

    def render(self, nframes, start, bus_f32, bus_f64):
        t = time.perf_counter() + 20e-6
        while time.perf_counter() < t:
            pass

    def wait_slot(self, slot):
        pass

    def reduce_async(self, bus_f64, nvalues, root, bus_f32, slot):
        pass

    def sync(self):
        pass


def build_voices(n_total: int):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    return W.additive_voices(G, n_total, SR, seed=0, partials=PARTIALS, adsr=ADSR_BENCH)


def _cpu_worker_init():
    from oracle import synth_oracle  # noqa: F401  (import cost paid before the timed map)
    from synthesizer_amd import workloads  # noqa: F401


def _cpu_shard(args):
    lo, hi, frames = args
    from oracle import synth_oracle as O
    from synthesizer_amd import workloads as W
    voices, gains = W.additive_voices(O, VOICES_PER_GPU, SR, seed=0, partials=PARTIALS, adsr=ADSR_BENCH)
    bus = O.mix_bus([v.take(frames) for v in voices[lo:hi]], gains[lo:hi])
    return (len(bus), float(sum(l + r for l, r in bus)))           # the partial bus stays in the worker: only a checksum travels


def cpu_baseline(frames: int, all_cores: bool = True):
    """The oracle (pure-Python generator restatement of the reference path) on ONE core, timed on a
    bounded sample of the same workload: all 1024 voices x `frames` frames, mixed to the stereo bus."""
    from oracle import synth_oracle as O
    from synthesizer_amd import workloads as W
    voices, gains = W.additive_voices(O, VOICES_PER_GPU, SR, seed=0, partials=PARTIALS, adsr=ADSR_BENCH)
    t0 = time.perf_counter()
    blocks = [v.take(frames) for v in voices]
    O.mix_bus(blocks, gains)
    dt = time.perf_counter() - t0
    out = {"value": VOICES_PER_GPU * frames / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "1024 voices (the bench's: Harmonics x16 + ADSR attack 0.01 / decay 0.05 / sustain spanning the run, level 0.6) "
                     "x the first %d frames (%.3f s of audio: attack, decay and the start of the sustain), pure-Python oracle "
                     "generators + float bus sum, %.1f s wall" % (frames, frames / SR, dt)}
    # ... and a window that matches the phase the GPU's timed region sits in: the sustain plateau -- the generators
    # run 3072 frames (attack and decay, untimed) and the NEXT `frames` frames are timed; same voices, same interpreter, one core
    try:
        skip_blocks = -(-3072 // O.norm_osc_blocksize)
        gens = [v.blocks() for v in voices]
        for g_ in gens:
            for _ in range(skip_blocks):
                next(g_)
        t0 = time.perf_counter()
        nb = -(-frames // O.norm_osc_blocksize)
        blocks = []
        for g_ in gens:
            row = []
            for _ in range(nb):
                row.extend(next(g_))
            blocks.append(row[:frames])
        O.mix_bus(blocks, gains)
        dtp = time.perf_counter() - t0
        out["sustain_window"] = {"value": VOICES_PER_GPU * frames / dtp / 1e6, "unit": "Msamples/s", "cores": 1, "wall_s": dtp,
                                 "sample": "the same voices, frames %d .. %d of the note (the constant-gain plateau the GPU's timed blocks lie on), "
                                           "pure-Python generators + float bus sum" % (skip_blocks * O.norm_osc_blocksize, skip_blocks * O.norm_osc_blocksize + frames)}
    except Exception as e:
        out["sustain_window"] = {"error": str(e)}
    # the same sample through the C restatement of the oracle (oracle/oracle.c, one core): what a compiled single-threaded
    # CPU implementation of this arithmetic does -- a fairer yardstick than the interpreter
    try:
        import numpy as np
        from oracle import c_oracle as CO
        CO.render(voices[0], 16)                     # build / load the shared object outside the timed region
        t0 = time.perf_counter()
        CO.mix_bus(np.stack([CO.render(v, frames) for v in voices]), gains)
        dtc = time.perf_counter() - t0
        out["c_port"] = {"value": VOICES_PER_GPU * frames / dtc / 1e6, "unit": "Msamples/s", "cores": 1, "wall_s": dtc}
    except Exception as e:                           # no C compiler on the box: the Python number stands alone
        out["c_port"] = {"error": str(e)}
    # ... and the C restatement over all host cores (threads: ctypes drops the GIL around the C call; voices dealt to the threads,
    # the partial buses of the threads added at the end -- timed): the yardstick a maintainer with a compiler and a many-core box
    # would reach for first, and the honest denominator for "how much faster is the GPU than this host"
    if all_cores and "error" not in out["c_port"]:
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncores = os.cpu_count() or 1
            nthr = max(1, min(ncores, 64, VOICES_PER_GPU))
            cframes = 8 * frames
            shares = [voices[VOICES_PER_GPU * i // nthr:VOICES_PER_GPU * (i + 1) // nthr] for i in range(nthr)]
            gshares = [gains[VOICES_PER_GPU * i // nthr:VOICES_PER_GPU * (i + 1) // nthr] for i in range(nthr)]

            def share(k):
                return CO.mix_bus(np.stack([CO.render(v, cframes) for v in shares[k]]), gshares[k])
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(lambda k: CO.render(shares[k][0], 64), range(nthr)))      # threads up
                t0 = time.perf_counter()
                parts = list(ex.map(share, range(nthr)))
                bus = parts[0]
                for p_ in parts[1:]:
                    bus = bus + p_
                dtt = time.perf_counter() - t0
            out["c_port_all_cores"] = {"value": VOICES_PER_GPU * cframes / dtt / 1e6, "unit": "Msamples/s", "cores": nthr, "host_cpu_count": ncores,
                                       "wall_s": dtt, "frames": cframes,
                                       "note": "oracle/oracle.c over %d threads (one share of the 1024 voices each, %d frames), partial buses added; "
                                               "16 libm sin per voice-sample" % (nthr, cframes)}
        except Exception as e:
            out["c_port_all_cores"] = {"error": str(e)}
    # BASELINE.md's optional leg: the same pure-Python sample over all host cores (one process per core, voices split evenly)
    if all_cores:
        try:
            import multiprocessing as mp
            ncores = os.cpu_count() or 1
            nproc = max(1, min(ncores, 64, VOICES_PER_GPU))
            aframes = 2 * frames                                       # twice the single-core sample: the workers' share stays ~1 s
            bounds = [(VOICES_PER_GPU * i // nproc, VOICES_PER_GPU * (i + 1) // nproc, aframes) for i in range(nproc)]
            with mp.get_context("spawn").Pool(nproc, initializer=_cpu_worker_init) as pool:
                pool.map_async(_cpu_shard, [(i, i + 1, 64) for i in range(nproc)], chunksize=1).get(timeout=180)    # every worker up and warm
                t0 = time.perf_counter()
                pool.map_async(_cpu_shard, bounds, chunksize=1).get(timeout=300)     # (a pool whose workers die would hang a plain map)
                dta = time.perf_counter() - t0
            out["all_cores"] = {"value": VOICES_PER_GPU * aframes / dta / 1e6, "unit": "Msamples/s", "cores": nproc,
                                "host_cpu_count": ncores, "wall_s": dta, "frames": aframes,
                                "note": "multiprocessing over voices (spawn), one process per core up to 64, each renders and mixes its "
                                        "share of the 1024 voices over %d frames; the final sum of the partial buses is not timed" % aframes}
        except Exception as e:
            out["all_cores"] = {"error": str(e)}
    return out


def committed_profile():
    """The newest committed rocprofv3 summary: HBM bytes per dispatch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) and
    the SQ instruction counters per dispatch (profiles/rNN_traffic.json, profiles/rNN_counters.json, both written by
    tools/summarize_profiles.py), with the hash of the kernel sources they were measured on."""
    out = {"traffic": {}, "counters": {}, "source": None, "source_hash": None}
    tr = sorted((ROOT / "profiles").glob("r*_traffic.json"))
    if tr:
        try:
            out["traffic"] = json.loads(tr[-1].read_text())
            out["source"] = tr[-1].name
        except Exception:
            pass
    cn = sorted((ROOT / "profiles").glob("r*_counters.json"))
    if cn:
        try:
            c = json.loads(cn[-1].read_text())
            out["source_hash"] = c.get("_meta", {}).get("source_hash")
            out["counters"] = {k: v for k, v in c.items() if not k.startswith("_")}
            out["counters_source"] = cn[-1].name
        except Exception:
            pass
    return out


def _by_prefix(table, prefix):
    for k, v in table.items():
        if k.startswith(prefix):
            return v
    return None


def steady(N, call, min_seconds=0.05, reps=5, max_loops=400, spread=None):
    """Average time (ms, HIP events on the library stream) of `call` once the clocks are up: loops of `reps` calls until
    the loops add up to min_seconds; the median loop counts (spread: a dict that receives min / max / loops)."""
    call()
    N.sync()
    loops, total = [], 0.0
    while (total < min_seconds or len(loops) < 3) and len(loops) < max_loops:
        N.timer_start()
        for _ in range(reps):
            call()
        ms = N.timer_stop()
        loops.append(ms / reps)
        total += ms / 1e3
    if spread is not None:
        spread.update({"min_ms": min(loops), "max_ms": max(loops), "loops": len(loops), "timed_s": total})
    return statistics.median(loops)


def pcm_rows(N):
    """HBM-bound rows of the path, each timed with HIP events on resident buffers."""
    import ctypes
    import numpy as np
    L = N.lib()
    rows = {}

    def row(name, call, nbytes, min_seconds=0.02, **extra):
        sp = {}
        ms = steady(N, lambda: N.check(call()), min_seconds=min_seconds, reps=3, spread=sp)
        rows[name] = dict({"ms": ms, "bytes": nbytes, "GBps": nbytes / (ms / 1e3) / 1e9,
                           "frac_hbm": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}, **extra)
        if min_seconds >= 0.1:      # a row timed like the headline: the spread says what a difference between two runs means
            rows[name].update({"frac_hbm_min": nbytes / (sp["max_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, "frac_hbm_max": nbytes / (sp["min_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                               "loops": sp["loops"], "timed_s": sp["timed_s"]})
        return ms

    # configs[4]: 8-channel, 10-minute float32 PCM, 96 kHz -> 44.1 kHz (1.84 GB in, 0.85 GB out)
    nch, in_frames = 8, 96000 * 600
    nout = L.sh_resample_out_frames(in_frames, 96000, 44100)
    src = N.DeviceBuffer(in_frames * nch * 4)
    dst = N.DeviceBuffer(nout * nch * 4)
    chunk = np.random.default_rng(0).uniform(-1, 1, 1 << 24).astype(np.float32)
    for off in range(0, src.nbytes, chunk.nbytes):
        src.upload(chunk[:min(len(chunk), (src.nbytes - off) // 4)], off)
    for width, is_float, name in ((4, 1, "resample_f32_8ch_600s_96k_to_44k1"), (2, 0, "resample_i16_8ch_1200s_96k_to_44k1")):
        frames = in_frames if is_float else in_frames * 2       # same byte count as the float case
        nout_w = L.sh_resample_out_frames(frames, 96000, 44100)
        ms = row(name, lambda: L.sh_resample(src.handle, frames, nch, width, is_float, 96000, 44100, dst.handle, None),
                 (frames + nout_w) * nch * width)
        rows[name]["out_Mframes_per_s"] = nout_w / (ms / 1e3) / 1e6
    # 16-bit mono / stereo, the shapes Sample.resample sees in practice (WaveSynth output, loaded WAVs): 900 MB in
    for name, nch_, inr, outr in (("resample_i16_mono_44k1_to_48k_900MB", 1, 44100, 48000),
                                  ("resample_i16_stereo_44k1_to_48k_900MB", 2, 44100, 48000),
                                  ("resample_i16_stereo_96k_to_44k1_900MB", 2, 96000, 44100)):
        frames = 450_000_000 // nch_
        nout_m = L.sh_resample_out_frames(frames, inr, outr)
        big = N.DeviceBuffer(nout_m * 2 * nch_)
        row(name, lambda: L.sh_resample(src.handle, frames, nch_, 2, 0, inr, outr, big.handle, None), (frames + nout_m) * 2 * nch_,
            min_seconds=0.25 if nch_ == 1 else 0.02)
        big.free()
    # mixer chain: 1024 int16 voices x 10 s stereo (saturating fold in voice order), 2N+2 bytes per sample
    nv, nsamples = 1024, 48000 * 2 * 10
    chunks = N.DeviceBuffer(nv * nsamples * 2)
    pcm = (np.random.default_rng(1).integers(-3000, 3000, 1 << 24)).astype(np.int16)
    for off in range(0, chunks.nbytes, pcm.nbytes):
        chunks.upload(pcm[:min(len(pcm), (chunks.nbytes - off) // 2)], off)
    mixed = N.DeviceBuffer(nsamples * 2)
    nbytes = (2 * nv + 2) * nsamples
    row("mix_chain_i16_1024v_10s_stereo", lambda: L.sh_mix_chain_i16(chunks.handle, nv, nsamples, nsamples, mixed.handle), nbytes)
    # the same fold through the pointer table (RealTimeMixer / mix_samples: every source read where it lives)
    bufs = (ctypes.c_void_p * nv)(*[chunks.handle] * nv)
    offs = (ctypes.c_size_t * nv)(*[v * nsamples for v in range(nv)])
    lens = (ctypes.c_uint32 * nv)(*[nsamples] * nv)
    row("mix_chain_gather_i16_1024v_10s_stereo", lambda: L.sh_mix_chain_gather_i16(bufs, offs, lens, nv, nsamples, mixed.handle, 0), nbytes)
    # one real-time turn: 64 sources x 4096-byte chunk (latency-bound: table upload + one small kernel)
    small = 2048
    rows["mixer_turn_64src_4KB_chunk"] = {"ms": steady(N, lambda: N.check(L.sh_mix_chain_gather_i16(bufs, offs, lens, 64, small, mixed.handle, 0)),
                                                        min_seconds=0.01, reps=50)}
    # Sample.from_osc_block: float32 -> int16 with the overflow check (the call returns after reading the flag back)
    nq = 300_000_000
    row("quantize_f32_to_i16_1200MB", lambda: L.sh_quantize_f32(src.handle, 0, nq, 32767.0, 2, dst.handle, 0), 6 * nq)
    nq64 = 150_000_000
    row("quantize_f64_to_i16_1200MB", lambda: L.sh_quantize_f64(src.handle, 0, nq64, 32767.0, 2, dst.handle, 0), 10 * nq64,
        note="the float32 buffer read as float64 bit patterns: tiny magnitudes, same traffic")
    # Sample.mix: saturating add of two 900 MB int16 buffers (3 bytes moved per byte of output)
    n = 900_000_000
    row("pcm_add_i16_900MB", lambda: L.sh_pcm_add(chunks.handle, 0, chunks.handle, n, n, 2, src.handle, 0), 3 * n)
    # ... and the METHOD a user calls: Sample.mix of two equal-length samples adds in place (3 bytes moved per
    # output byte, as the kernel row above; rounds 1-3 allocated, copied self and added: 5) -- 900 MB, and the one-second stereo
    # sample of SURVEY 8(a) row a10 (192 KB: a launch, not a bandwidth)
    from synthesizer_amd.sample import Sample
    sa = Sample(samplerate=48000, nchannels=2, samplewidth=2)
    sa._set_device(src.view(0, n), n)
    sb = Sample(samplerate=48000, nchannels=2, samplewidth=2)
    sb._set_device(chunks.view(0, n), n)
    row("sample_mix_method_i16_900MB", lambda: (sa.mix(sb), 0)[1], 3 * n, note="Sample.mix(other), equal lengths, both resident: in place")
    n1s = 48000 * 2 * 2
    s1 = Sample(samplerate=48000, nchannels=2, samplewidth=2)
    s1._set_device(src.view(n, n1s), n1s)
    s2 = Sample(samplerate=48000, nchannels=2, samplewidth=2)
    s2._set_device(chunks.view(n, n1s), n1s)
    rows["sample_mix_1s_stereo"] = {"ms": steady(N, lambda: s1.mix(s2), min_seconds=0.01, reps=50), "bytes": 3 * n1s,
                                    "note": "Sample.mix of two one-second 48 kHz stereo int16 samples (192 KB each): one launch, latency-bound"}
    # Sample.amplify (audioop.mul), Sample.mono (audioop.tomono), peak/rms
    row("pcm_mul_i16_900MB", lambda: L.sh_pcm_mul(chunks.handle, 0, n, 2, 0.7071, src.handle, 0), 2 * n)
    row("pcm_tomono_i16_900MB", lambda: L.sh_pcm_tomono(chunks.handle, n // 4, 2, 0.5, 0.5, src.handle), n + n // 2)
    row("pcm_stats_i16_900MB", lambda: L.sh_pcm_stats(chunks.handle, n, 2, ctypes.byref(ctypes.c_uint32()), ctypes.byref(ctypes.c_double())), n)
    row("pcm_stats_stereo_i16_900MB", lambda: L.sh_pcm_stats_stereo(chunks.handle, n // 4, 2, (ctypes.c_uint32 * 2)(), (ctypes.c_double * 2)()), n)
    del sa, sb, s1, s2                      # (their buffers are windows of src / chunks)
    for b in (src, dst, chunks, mixed):
        b.free()
    return rows


def steady_render_kernel(name):
    """Is this kernel name (as rocprofv3 prints it, shortened by tools/summarize_profiles.py) one of the render kernels of a STEADY block?
    k_render_lean<W, F, M, KINDS, SEG> with SEG = false, k_render_general<W, F, M, KIND> with KIND != 1 (GEN_SEG), k_render_combined,
    k_render_tiles -- not the kernels of a segmented transition launch (the first block of a profiled run)."""
    m = re.match(r"k_render_(lean|general|combined|tiles)<([^>]*)>", name)
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "lean":
        return args[-1] in ("false", "0")
    if m.group(1) == "general":
        return args[-1] != "1"
    return True


PROFILE_RUN = 10      # blocks per launch in the per-config profiling passes (bench.py --only-config X: runs of PROFILE_RUN blocks ONLY, so that the
                      # per-dispatch averages rocprofv3 reports are averages over dispatches of one size)


def config_roofline(prof, tag, nvoices, ms_per_block):
    """Roofline of a BASELINE config's render launch from the committed counters of ITS profiling pass (profiles/rNN_counters.json,
    key "<tag>:<kernel>"; tools/profile_round.sh runs bench.py --only-config <tag> under rocprofv3, which renders in runs of PROFILE_RUN
    one-second blocks per launch and nothing else): float64 lane-operations per voice-sample = (FMA + MUL + ADD wave-instructions) x 64 /
    voice-samples per dispatch, against the issue peak at the row's time per block; HBM traffic per block against the algorithmic bytes."""
    best, name = None, None
    for k_, v_ in prof["counters"].items():
        if k_.startswith(tag + ":") and steady_render_kernel(k_.split(":", 1)[1]) and (best is None or v_.get("SQ_INSTS_VALU_FMA_F64", 0) > best.get("SQ_INSTS_VALU_FMA_F64", 0)):
            best, name = v_, k_             # (not the segmented transition launch of the loop's first block: that is not the steady state)
    if not best or not all(k in best for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
        return {"bound": "valu_f64", "note": "no counters for this config in profiles/ (tools/profile_round.sh writes them)"}
    per_dispatch = float(nvoices) * SR * PROFILE_RUN
    ops = (best["SQ_INSTS_VALU_FMA_F64"] + best["SQ_INSTS_VALU_MUL_F64"] + best["SQ_INSTS_VALU_ADD_F64"]) * 64.0 / per_dispatch
    achieved = float(nvoices) * SR * ops / (ms_per_block / 1e3) / 1e12
    traffic = prof["traffic"].get(name, {}).get("hbm_bytes")
    algorithmic = 8.0 * SR
    return {"kernel": name.split(":", 1)[1], "bound": "valu_f64", "ops_per_voice_sample": ops, "achieved": achieved, "peak": FP64_PEAK_TOPS,
            "unit": "T f64 lane-ops/s", "frac": achieved / FP64_PEAK_TOPS, "avg_launch_ms": ms_per_block,
            "profiled_dispatch": "%d voices x %d frames (a run of %d blocks in one launch)" % (nvoices, SR * PROFILE_RUN, PROFILE_RUN),
            "traffic": traffic / PROFILE_RUN if traffic else None, "algorithmic_bytes": algorithmic,
            "traffic_over_algorithmic": traffic / PROFILE_RUN / algorithmic if traffic else None,
            "source": prof.get("counters_source")}


def staggered_roofline(prof, sounding_voice_samples, ms):
    """The literal-ADSR row's roofline: float64 lane-operations of BOTH kernels of a tile-classified launch
    (k_render_tiles: the lean pairs; k_render_general<.., 2>: the general pairs) per SOUNDING voice-sample -- 0.76 of the players sound at
    any time -- from the counters of the row's own profiling pass (profiles/rNN_counters.json, keys "staggered:<kernel>"), against the
    issue peak; and the kernels' own durations from the same pass (two launches in flight: a kernel's duration is not the time per block)."""
    ks = {k_.split(":", 1)[1]: v_ for k_, v_ in prof["counters"].items() if k_.startswith("staggered:") and steady_render_kernel(k_.split(":", 1)[1])}
    need = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")
    ks = {k_: v_ for k_, v_ in ks.items() if all(n in v_ for n in need)}
    if not ks:
        return {"bound": "valu_f64", "note": "no counters for this row in profiles/ (tools/profile_round.sh writes them)"}
    f64 = sum(sum(v_[n] for n in need) for v_ in ks.values())
    valu = sum(v_.get("SQ_INSTS_VALU", 0) for v_ in ks.values())
    ops = f64 * 64.0 / sounding_voice_samples
    achieved = sounding_voice_samples * ops / (ms / 1e3) / 1e12
    durations = {}
    try:
        import csv
        newest = sorted((ROOT / "profiles").glob("r*_staggered_kernel_stats.csv"))[-1]
        for r in csv.DictReader(open(newest)):
            m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Name"])
            if m and m.group(1).startswith("k_render_"):
                durations[m.group(1)] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    except Exception:
        pass
    return {"kernels": sorted(ks), "bound": "valu_f64", "ops_per_sounding_voice_sample": ops,
            "valu_instructions_per_sounding_voice_sample": valu * 64.0 / sounding_voice_samples if valu else None,
            "achieved": achieved, "peak": FP64_PEAK_TOPS, "unit": "T f64 lane-ops/s", "frac": achieved / FP64_PEAK_TOPS,
            "sounding_voice_samples_per_block": sounding_voice_samples, "avg_launch_ms": ms,
            "kernel_durations_under_rocprofv3": durations, "source": prof.get("counters_source"),
            "note": "frac counts the float64 operations of the SOUNDING voice-samples only (0.76 of 1024 players x frames); the counters are per "
                    "dispatch of a steady block; kernel durations: two launches in flight, so a kernel's duration is about twice the time per block"}


def staggered_row(N, F, prof=None):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    import ctypes
    slots, notes, nblocks = VOICES_PER_GPU, 22, 20
    t0 = time.perf_counter()
    voices, gains = W.staggered_notes(G, slots, SR, seed=0, partials=PARTIALS, period=1.0, notes=notes)
    bank = VoiceBank(voices, gains=gains)
    build_s = time.perf_counter() - t0
    ring = [N.DeviceBuffer(F * 8) for _ in range(4)]
    pos = [0]

    def step(k):
        bank.render_device(F, k * F, bus_f32=ring[k & 3])

    def loop(first_timed):
        """Blocks 1 .. 20 of the piece (block 0 has half its notes yet to start); the clock runs from block `first_timed`.  The jump
        back to block 1 is a standing start: its first two launches resolve their record and tile sets in front of the render."""
        for k in range(1, first_timed):
            step(k)
        N.timer_start()
        for k in range(first_timed, nblocks + 1):
            step(k)
        return N.timer_stop() / (nblocks + 1 - first_timed)

    def median_loop(first_timed, min_seconds=0.2):
        loop(first_timed)
        N.sync()
        got, total = [], 0.0
        while total < min_seconds or len(got) < 3:
            got.append(loop(first_timed))
            total += got[-1] * (nblocks + 1 - first_timed) / 1e3
        return statistics.median(got)

    ms = median_loop(3)             # blocks 3 .. 20: a stream of blocks, timed like the headline's (a pass of eighteen between two syncs)
    ms_cold = median_loop(1)        # ... and the twenty blocks from the standing start
    # ... and the same table in real-time chunks: 4096 frames (85 ms of audio) per render call, eight seconds of the piece per loop
    cf = 4096
    cring = [N.DeviceBuffer(cf * 8) for _ in range(4)]
    cpos = [0]

    def chunk():
        k = cpos[0] % (8 * SR // cf)
        bank.render_device(cf, 3 * SR + k * cf, bus_f32=cring[k & 3])
        cpos[0] += 1
    ms_chunk = steady(N, chunk, min_seconds=0.05, reps=8 * SR // cf)
    return {"players": slots, "voices_in_the_table": len(voices), "ms_per_step": ms, "value": slots * F / (ms / 1e3) / 1e6, "unit": "Msamples/s",
            "roofline": staggered_roofline(prof if prof is not None else committed_profile(), slots * F * 0.76, ms),
            "from_a_standing_start_ms_per_step": ms_cold,
            "realtime_chunks_4096": {"ms_per_chunk": ms_chunk, "x_real_time": cf / SR / (ms_chunk / 1e3),
                                     "note": "the same table rendered 4096 frames per call (tile-classified launches at every block length)"},
            "sounding_fraction": 0.76, "host_build_s": build_s,
            "blocks_per_loop": nblocks,
            "note": "1024 players x 22 rounds: every note a voice of its own with an onset (DelayFilter fused into the record), Harmonics x16 under "
                    "the literal ADSR (attack 0.01, decay 0.05, sustain 0.5 at 0.6, release 0.2); blocks of one second, 1 .. 20 of the piece over and "
                    "over; ms_per_step: blocks 3 .. 20 of every loop (a stream of blocks between two syncs, like a pass of the headline), "
                    "from_a_standing_start: all twenty, the jump back to block 1 included (two launches resolve their record and tile sets in "
                    "front of the render); counted as 1024 voice-samples per frame although a quarter of the players is between notes at any time; "
                    "tile-classified launches (lean per (voice, 512-frame tile) pair -- one piece and one line, a corner, up to three pieces, or a "
                    "walk along the voice's table for the first tiles of a note -- general code for what is left)"}


def job_row(N, local, F, step0, prof, with_run=True):
    """BASELINE's job taken literally -- 10 s from frame 0 in blocks of F, the first block being the notes' attack, decay and a dozen
    binades of the phase sum (a segmented launch) -- best of 12, other frames rendered in between (the clocks stay up, block 0's records
    go cold).  roofline: the float64 lane-operations of ALL the job's kernels (profiles/rNN_counters.json, keys "job:<kernel>", from
    tools/profile_round.sh's `--only-config job` pass: block 0's four kernels once, the steady kernel nine times) over the job's time,
    and what block 0's kernels take one by one (profiles/rNN_job_kernel_stats.csv)."""
    import csv
    ring = [N.DeviceBuffer(F * 8) for _ in range(4)]
    job_ms = float("inf")
    for rep in range(12):
        N.sync()
        N.timer_start()
        for k in range(10):
            local.render_device(F, k * F, bus_f32=ring[k & 3])
        job_ms = min(job_ms, N.timer_stop())
        for k in range(30):
            local.render_device(F, (step0 + k) * F, bus_f32=ring[k & 3])
    N.sync()
    # the same job asked for in ONE call (sh_bank_render_run: the ten blocks are one launch, cut into segments where the notes' envelopes and
    # phase sums demand it -- the lean kernel once over all of them, the general code over the first segment only)
    cont = local.make_ring(F, 10)
    run_ms = float("inf")
    for rep in range(12 if with_run else 0):       # (not in the profiling pass: its launch would be averaged with block 0's kernels of the same name)
        N.sync()
        N.timer_start()
        local.render_run(F, 10, 0, ring=cont)
        run_ms = min(run_ms, N.timer_stop())
        for k in range(30):
            local.render_device(F, (step0 + k) * F, bus_f32=ring[k & 3])
    N.sync()
    for b_ in ring:
        b_.free()
    del cont
    nv = local.nvoices
    block_by_block_ms = job_ms
    in_one_run = run_ms < job_ms
    job_ms = min(job_ms, run_ms)
    row = {"blocks": 10, "ms": job_ms, "value": nv * 10.0 * F / (job_ms / 1e3) / 1e6, "unit": "Msamples/s", "us_per_block": job_ms * 100.0,
           "how": "one sh_bank_render_run call (one launch)" if in_one_run else "ten sh_bank_render calls",
           "ten_calls_ms": block_by_block_ms, "one_run_ms": run_ms,
           "note": "ten consecutive blocks starting at frame 0 (first block of the notes included), best of 12, both ways"}
    ops = {k.split(":", 1)[1]: (v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64.0
           for k, v in prof["counters"].items() if k.startswith("job:")}
    steady_k = [k for k in ops if steady_render_kernel(k)]
    block0_k = [k for k in ops if k.startswith("k_render_") and not steady_render_kernel(k)]
    roof = {"bound": "valu_f64"}
    if steady_k and block0_k:
        total = 9.0 * max(ops[k] for k in steady_k) + sum(ops[k] for k in block0_k)
        roof.update({"lane_ops_per_job": total, "ops_per_voice_sample": total / (nv * 10.0 * F), "achieved": total / (job_ms / 1e3) / 1e12,
                     "peak": FP64_PEAK_TOPS, "unit": "T f64 lane-ops/s", "frac": total / (job_ms / 1e3) / 1e12 / FP64_PEAK_TOPS,
                     "kernels": {"steady (x9)": steady_k, "block 0 (x1)": block0_k}, "source": prof.get("counters_source")})
    else:
        roof["note"] = "no counters of a job pass in profiles/ (tools/profile_round.sh: bench.py --only-config job)"
    stats = sorted((ROOT / "profiles").glob("r*_job_kernel_stats.csv"))
    if stats:
        split = {}
        for r in csv.DictReader(open(stats[-1])):
            m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Name"])
            name = m.group(1) if m else r["Name"][:40]
            if not steady_render_kernel(name) and (name.startswith("k_render_") or name.startswith("k_prepare") or name.startswith("k_seg") or name.startswith("k_bus")):
                split[name] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
        roof["block0_kernels"] = split
        roof["block0_source"] = stats[-1].name
    row["roofline"] = roof
    return row


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: what a native library prints on stdout (librccl's version banner
    at communicator creation) must not end up beside the ONE JSON line this script owes its caller."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def config4_rows(N, K):
    """BASELINE configs[3] as far as ONE GPU goes: (a) the whole 8192-voice table on this GPU (what --scaling strong reports at
    N = 1); (b) one rank's share -- 1024 voices -- driven down the multi-rank path: DistVoiceBank's slot ring, float64 partial
    buses, ncclReduce on a 1-rank communicator on the communication stream, the rounding to float32 behind it -- per block, with
    and without the exchange."""
    from synthesizer_amd import dist
    out = {}
    voices, gains = build_voices(STRONG_VOICES)
    bank = dist.DistVoiceBank(voices, gains, 0, 1)
    pos = [5]

    def step():
        bank.render_device(SR, pos[0] * SR)
        pos[0] += 1
    for _ in range(4):
        step()
    ms = steady(N, step, min_seconds=0.1, reps=K)
    out["strong_8192v_n1"] = {"voices": STRONG_VOICES, "ms_per_step": ms, "value": STRONG_VOICES * SR / (ms / 1e3) / 1e6, "unit": "Msamples/s",
                              "note": "the whole configs[3] table on one GPU (bench.py --scaling strong --gpus 1), float32 bus, steady state"}
    del bank
    try:
        with _stdout_to_stderr():
            dist.init(0, 1, broadcast=lambda payload, rank, world, n: payload)
        local_v, local_g = voices[:VOICES_PER_GPU], gains[:VOICES_PER_GPU]
        ring = dist.DistVoiceBank(local_v, local_g, 0, 1, batch=8)
        ring.world, ring.batch = 2, 8                 # the multi-rank code path on a 1-rank communicator: same calls, the reduce is the identity
        pos = [5]

        def ring_step():
            ring.render_device(SR, pos[0] * SR)
            pos[0] += 1
        for _ in range(16):
            ring_step()
        # (steady(): loops of 4 batches between HIP events until the clocks are up, median loop -- a single shot after the pause
        # in which the bank was built measures the clock governor; timer_stop ends the run: one drain per 4 batches stays in)
        with_x = steady(N, ring_step, min_seconds=0.15, reps=32)
        ring.flush()
        N.sync()
        # the same renders (float64 partial bus out) without the exchange
        bufs = [N.DeviceBuffer(SR * 16) for _ in range(4)]
        p2 = [5]

        def local_step():
            ring.local.render_device(SR, p2[0] * SR, bus_f32=None, bus_f64=bufs[p2[0] & 3])
            p2[0] += 1
        without_x = steady(N, local_step, min_seconds=0.1, reps=K)
        # one blocking reduce of a batch message, back to back
        nval = 8 * SR * 2
        msg = N.DeviceBuffer(nval * 8)
        msg.zero()
        red = steady(N, lambda: N.check(N.lib().sh_dist_reduce_bus(msg.handle, nval, 0)), min_seconds=0.02, reps=5)
        out["rank_local_1024v"] = {"voices": VOICES_PER_GPU, "ms_per_step_with_exchange": with_x, "ms_per_step_render_only_f64_bus": without_x,
                                   "blocking_reduce_ms_per_batch_of_8": red, "reduce_message_bytes": nval * 8,
                                   "value": VOICES_PER_GPU * SR / (with_x / 1e3) / 1e6, "unit": "Msamples/s",
                                   "note": "one rank's share of configs[3] through DistVoiceBank(batch=8)'s slot ring: render -> float64 partial bus -> "
                                           "ncclReduce (1-rank communicator, communication stream; enqueued a few launches after the slot filled up, behind marks on both render "
                                           "streams: the run of pipelined renders is never ended) -> float32; HIP events, loops of 32 blocks"}
        for b_ in bufs:
            b_.free()
        msg.free()
    except Exception as e:                               # no librccl on the box: the render-only row stands alone
        out["rank_local_1024v"] = {"error": str(e)}
    finally:
        try:
            dist.shutdown()
        except Exception:
            pass
    out["note"] = "the 8-GPU run itself: python bench.py --gpus 8 [--scaling strong] (or the same under torch.distributed.run / any launcher that sets RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*)"
    return out


def config_rows(N, prof=None, only=None, K=20):
    """The BASELINE configs other than the headline, each on its own steady-state loop (one MI355X)."""
    import numpy as np
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    rows = {}
    prof = prof or {"counters": {}, "traffic": {}}

    def _config1_row(G):
        # configs[0]: single 440 Hz Sine, 1 s @ 44.1 kHz mono, delivered to a host array (the call includes the D2H copy)
        osc = G.Sine(440, samplerate=44100)
        osc.render(44100)
        t0 = time.perf_counter()
        for _ in range(200):
            osc.render(44100, start=0)
        dt = (time.perf_counter() - t0) / 200
        return {"config1_sine_440Hz_1s_44k1_mono_to_host": {"ms": dt * 1e3, "Msamples_per_s": 44100 / dt / 1e6,
                                                            "note": "Oscillator.render -> numpy float32 on the host, wall clock, PCIe copy included"}}

    def bank_row(name, tag, voices, gains, note):
        bank = VoiceBank(voices, gains=gains)
        ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
        pos = [0]

        def step():
            bank.render_device(SR, pos[0] * SR, bus_f32=ring[pos[0] & 3])
            pos[0] += 1
        ms, host_us = float("inf"), float("nan")
        if only is None:                    # (a profiling pass -- bench.py --only-config X -- renders in runs only: dispatches of ONE size)
            for _ in range(8):
                step()
            ms = steady(N, step, min_seconds=0.1, reps=20)
            t0 = time.perf_counter()
            for _ in range(400):
                step()
            host_us = (time.perf_counter() - t0) / 400 * 1e6
            N.sync()
        else:
            pos[0] = 30
        # the same stream of blocks asked for RUN blocks at a time (sh_bank_render_run: one call, and -- the ring being windows of one
        # allocation -- one launch per run): what BASELINE's job "10 s in blocks of 48 000" is for a caller that knows it wants 10 blocks
        RUN = PROFILE_RUN
        cont = bank.make_ring(SR, RUN)

        def run():
            bank.render_run(SR, RUN, pos[0] * SR, ring=cont)
            pos[0] += RUN
        for _ in range(3):
            run()
        sp = {}
        ms_run = steady(N, run, min_seconds=0.1, reps=8, spread=sp) / RUN
        t0 = time.perf_counter()
        for _ in range(100):
            run()
        host_run_us = (time.perf_counter() - t0) / (100 * RUN) * 1e6
        N.sync()
        # ms_per_1s_block = ONE sh_bank_render per block (comparable with rounds 1-4): the run-of-blocks figure is a row of its own.
        # (A profiling pass -- --only-config -- renders in runs only: its row's headline figure is then the run's, and says so.)
        per_block = ms if only is None else ms_run
        rows[name] = {"voices": len(voices), "ms_per_1s_block": per_block, "Msamples_per_s": len(voices) * SR / (per_block / 1e3) / 1e6,
                      "realtime_factor": 1e3 / per_block,
                      "how": "one sh_bank_render per block into a ring of 4 buffers" if only is None
                             else ("runs of %d blocks per call (profiling pass: dispatches of one size)" % RUN),
                      "host_enqueue_us_per_block": host_us if only is None else host_run_us, "note": note,
                      "runs_of_%d_blocks" % RUN: {"ms_per_1s_block": ms_run, "Msamples_per_s": len(voices) * SR / (ms_run / 1e3) / 1e6,
                                                  "host_enqueue_us_per_block": host_run_us,
                                                  "min_ms": sp["min_ms"] / RUN, "max_ms": sp["max_ms"] / RUN,
                                                  "how": "sh_bank_render_run into a contiguous ring: one call and one launch per %d blocks" % RUN,
                                                  "roofline": config_roofline(prof, tag, len(voices), ms_run)},
                      "roofline": config_roofline(prof, tag, len(voices), per_block)}
        for b in ring:
            b.free()
    if only in (None, "config1"):
        rows.update(_config1_row(G))
    if only in (None, "config2"):
        v, g = W.additive_voices(G, 64, SR, seed=0, partials=PARTIALS, adsr=ADSR_BENCH)
        bank_row("config2_additive_64v_adsr_48k_stereo", "config2", v, g,
                 "64 Harmonics x16 voices + ADSR -> float32 stereo bus into a ring of 4 buffers; latency-bound (64 voices x 48 000 frames are 2 us of "
                 "arithmetic): consecutive blocks alternate between two streams, the records of the block two launches on are resolved "
                 "by a workgroup of their own")
    if only in (None, "config3"):
        v, g = W.fm_voices(G, 1024, SR, seed=1)
        bank_row("config3_fm_1024v_48k_stereo", "config3", v, g, "1024 Sine carriers, each with a Sine fm_lfo (closed-form running sum), -> float32 stereo bus")
    if only in (None, "mixed"):
        # not a BASELINE config: Harmonics and FM Sine voices in ONE bank (what a patch with both kinds of instrument asks for): the lean
        # lists hold the kinds in runs, one loop per run (tools/probe.py kinds has the other mixes)
        va, ga = W.additive_voices(G, 512, SR, seed=0, partials=PARTIALS, adsr=ADSR_BENCH)
        vf, gf = W.fm_voices(G, 512, SR, seed=1)
        v = [x for pair in zip(va, vf) for x in pair]
        g = [x for pair in zip(ga, gf) for x in pair]
        bank_row("mixed_512_additive_512_fm_48k_stereo", "mixed", v, g,
                 "512 Harmonics x16 voices + ADSR interleaved with 512 FM Sine voices in one bank -> float32 stereo bus")
        r = rows["mixed_512_additive_512_fm_48k_stereo"]
        if "config3_fm_1024v_48k_stereo" in rows:
            r["x_config3"] = r["ms_per_1s_block"] / rows["config3_fm_1024v_48k_stereo"]["ms_per_1s_block"]
    if only in (None, "config4"):
        rows["config4_8192v_8gpu"] = config4_rows(N, K)
    if only is None:
        rows["config5_resample_8ch_600s_96k_to_44k1"] = {"note": "pcm_rows.resample_f32_8ch_600s_96k_to_44k1"}
    return rows


COMPACT_LIMIT = 4096           # bytes of the ONE line on stdout (the driver's parser gave up on a 22.8 KB line)


def write_detail(out: dict, path=None):
    """Every side row, note and per-pass figure of the run goes to a file (default bench_detail.json beside this script, and a
    copy under gpurun_out/ when that directory exists so that a gpurun call brings it home); stdout carries compact_line() only."""
    targets = [Path(path)] if path else [ROOT / "bench_detail.json"] + ([ROOT / "gpurun_out" / "bench_detail.json"] if (ROOT / "gpurun_out").is_dir() else [])
    written = None
    for t in targets:
        try:
            t.write_text(json.dumps(out, indent=1) + "\n")
            written = written or t
        except OSError as e:          # (a read-only checkout: the line on stdout is what matters)
            print("bench.py: could not write %s: %s" % (t, e), file=sys.stderr)
    if written is None:
        return None
    try:
        return str(written.relative_to(ROOT))
    except ValueError:
        return str(written)


def _rnd(x, digits=6):
    """Floats of the compact line to `digits` significant digits (the detail file keeps them whole)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _rnd(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rnd(v, digits) for v in x]
    return x


def compact_line(out: dict, detail_path=None) -> str:
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline`, and a handful of side figures -- <= COMPACT_LIMIT bytes.
    Everything else of `out` (notes, passes, configs, pcm_rows, two_step*, staggered_notes, job_*) is in the detail file."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d is not None and k in d}
    r = out.get("roofline") or {}
    hbm = r.get("hbm") or {}
    line = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    if line.get("data", "").startswith("dry-run"):
        line["data"] = "dry-run (no GPU: plumbing only, not a measurement)"
    cfg = out.get("config") or {}
    line["config"] = pick(cfg, ("workload", "voices_total", "voices_this_rank", "frames_per_step", "samplerate", "voice_shards", "parallelism"))
    two = out.get("two_step") or {}
    mix_frac = (two.get("roofline_mix") or {}).get("frac")
    gen_frac = (two.get("roofline_generate") or {}).get("frac")
    # the metric asks for "% HBM roofline": the fused headline kernel owes HBM 8 B per frame, so its binding roof is float64 VALU issue;
    # the HBM-regime pair (SURVEY 8(d) regime i: voices materialised, then mixed) is roofline_hbm_regime below
    line["roofline"] = dict(pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "ops_per_voice_sample", "traffic", "avg_launch_ms", "launches_in_flight", "profile_stale", "sin_evals_per_s")),
                            algorithmic_bytes=hbm.get("algorithmic_bytes"), traffic_over_algorithmic=hbm.get("traffic_over_algorithmic"),
                            hbm=pick(hbm, ("achieved", "frac", "peak", "unit")),
                            why="fused generate-and-mix writes 8 B per frame: HBM idle by construction, float64 VALU issue binds (SURVEY 8(d) regime ii)")
    if two:
        line["roofline_hbm_regime"] = {"mix_frac": mix_frac, "generate_frac": gen_frac, "mix_kernel": "k_mix_bus_direct", "generate_kernel": "k_generate_lean_harm<16, float>",
                                       "mix_GBps": (two.get("roofline_mix") or {}).get("achieved"), "generate_GBps": (two.get("roofline_generate") or {}).get("achieved"),
                                       "peak_GBps": HBM_PEAK_GBS, "two_step_value": two.get("value"),
                                       "note": "two-step path (float32 voice rows in HBM, then the mixer): north_star's >= 0.60 of HBM is met by the mix half only"}
    cb = out.get("cpu_baseline")
    if cb:
        c = pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = (cb.get("sample") or "")[:160]
        if cb.get("c_port"):
            c["c_port"] = pick(cb["c_port"], ("value", "cores"))
        if cb.get("c_port_all_cores"):
            c["c_port_all_cores"] = pick(cb["c_port_all_cores"], ("value", "cores"))
        line["cpu_baseline"] = c
        line["speedup_vs_cpu_baseline"] = out.get("speedup_vs_cpu_baseline")
    if out.get("verified") is not None:
        line["verified"] = pick(out["verified"], ("ok", "block_start_frame", "checksum_f64"))
    rc = out.get("rccl") or {}
    line["rccl"] = dict(pick(rc, ("world", "rank", "version", "communicator")),
                        ranks=[pick(g, ("rank", "device", "pci", "rccl_world")) for g in rc.get("ranks", [])])
    if out.get("per_rank"):
        line["per_rank"] = pick(out["per_rank"], ("render_us_per_step", "blocking_reduce_us_per_call", "reduce_message_bytes", "blocks_per_reduce",
                                                  "step_us_with_exchange", "exposed_exchange_us_per_step"))
    ps = out.get("passes") or {}
    line["passes"] = pick(ps, ("count", "timed_region_s", "min_ms_per_step", "max_ms_per_step"))
    # a handful of side figures (one number each; the rows themselves are in the detail file)
    side = {}
    cfgs = out.get("configs") or {}
    for key, name in (("config2_additive_64v_adsr_48k_stereo", "config2_ms_per_1s_block"), ("config3_fm_1024v_48k_stereo", "config3_ms_per_1s_block")):
        if key in cfgs:
            side[name] = cfgs[key].get("ms_per_1s_block")
            run = cfgs[key].get("runs_of_%d_blocks" % PROFILE_RUN)
            if run:
                side[name.replace("_ms_per_1s_block", "_ms_per_block_in_runs_of_%d" % PROFILE_RUN)] = run.get("ms_per_1s_block")
    c1 = cfgs.get("config1_sine_440Hz_1s_44k1_mono_to_host")
    if c1:
        side["config1_ms_to_host"] = c1.get("ms")
    pr = out.get("pcm_rows") or {}
    for key, name in (("resample_f32_8ch_600s_96k_to_44k1", "config5_resample_f32_frac_hbm"), ("resample_i16_mono_44k1_to_48k_900MB", "resample_i16_mono_frac_hbm"),
                      ("resample_i16_stereo_44k1_to_48k_900MB", "resample_i16_stereo_frac_hbm"), ("mix_chain_i16_1024v_10s_stereo", "mix_chain_i16_frac_hbm")):
        if key in pr:
            side[name] = pr[key].get("frac_hbm")
    if "resample_f32_8ch_600s_96k_to_44k1" in pr:
        side["config5_oracle"] = "float32 path: no reference oracle (audioop is integer-only; SURVEY a12) -- checked against this build's float restatement"
    for key, name in (("int16_stream", "int16_stream_ms_per_step"), ("run_of_blocks", "run_of_blocks_ms_per_step"), ("staggered_notes", "staggered_ms_per_step")):
        if key in out:
            side[name] = out[key].get("ms_per_step")
    g16 = (out.get("two_step_i16") or {}).get("int16_guard")
    if g16:
        side["int16_exact_rows_ms"] = g16.get("rows_ms")
        side["int16_exact_fused_mixdown_ms"] = g16.get("fused_mixdown_ms")
        side["int16_guard_cost_x"] = [g16.get("rows_x"), g16.get("fused_mixdown_x")]
    if "lone_call" in out:
        side["lone_call_us_per_block"] = out["lone_call"].get("us_per_block")
    if side:
        line["side"] = side
    line["library"] = out.get("library")
    line["launcher"] = out.get("launcher")
    line["control_channel"] = out.get("control_channel")
    line["detail"] = detail_path
    text = json.dumps(_rnd(line), separators=(",", ":"))
    for drop in ("side", "per_rank", "control_channel", "passes", "roofline_hbm_regime"):     # (never needed at today's sizes: a guard, not a habit)
        if len(text.encode()) <= COMPACT_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(_rnd(line), separators=(",", ":"))
    assert len(text.encode()) <= COMPACT_LIMIT, len(text)
    return text


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=SR, help="frames per step (block size)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 1024 voices per GPU; strong: 8192 voices in total (BASELINE configs[3]) whatever --gpus")
    ap.add_argument("--min-seconds", type=float, default=5.5,
                    help="passes of K steps repeat until they add up to this much timed work (5.5 s: longer than the 5 s at which the driver samples amd-smi, "
                         "so that its independent busy / power signal sees the timed region -- VERDICT r03; the clock governor needs 0.05 s)")
    ap.add_argument("--max-passes", type=int, default=40000)
    ap.add_argument("--cpu-frames", type=int, default=12288, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-all-cores", action="store_true")
    ap.add_argument("--no-pcm-rows", action="store_true", help="skip the resample / integer-mix rows")
    ap.add_argument("--no-two-step", action="store_true")
    ap.add_argument("--no-runs", action="store_true", help="skip the run_of_blocks row (tools/profile_round.sh: its launches of K blocks would be "
                                                           "averaged with the one-block launches of the same kernel in rocprofv3's per-dispatch counters)")
    ap.add_argument("--no-configs", action="store_true", help="skip the rows of the other BASELINE configs")
    ap.add_argument("--only-config", choices=("config2", "config3", "config4", "staggered", "mixed", "job"), default=None,
                    help="run ONLY that config's row and print it (tools/profile_round.sh: one rocprofv3 pass per config)")
    ap.add_argument("--detail", default=None, help="where the detail rows go (default: bench_detail.json beside this script, and gpurun_out/ when it exists)")
    ap.add_argument("--reduce-batch", type=int, default=8, help="blocks per RCCL reduce when --gpus > 1 (also the length of a run of pipelined renders)")
    args = ap.parse_args()

    dry = os.environ.get("SYNTHHIP_BENCH_DRY_RUN") == "1"
    if dry:
        args.no_two_step = args.no_configs = args.no_pcm_rows = True
        args.cpu_frames = 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes one (the ranks are copies of this command with the launcher's environment)
        if not dry:
            from synthesizer_amd import _native as N0
            seen = N0.lib().sh_device_count()            # (0 without a GPU; never fails)
            if seen < args.gpus:
                print("bench.py: --gpus %d, but this node shows %d GPU%s" % (args.gpus, seen, "" if seen == 1 else "s"), file=sys.stderr)
                return 3
        return launch_ranks(sys.argv[1:], args.gpus, dry)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE is %d (one process per GPU: start it plainly, or under a launcher with "
              "--nproc-per-node %d)" % (args.gpus, world, args.gpus), file=sys.stderr)
        return 2
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from synthesizer_amd import build as B
    from synthesizer_amd import dist
    if dry:
        N = _DryNative()
    else:
        from synthesizer_amd import _native as N
    # the ranks' control channel: data over a TCP star beside MASTER_PORT, barriers through a page of shared memory
    rdzv = dist.Rendezvous(rank, world)
    N.ensure_init(dist.device_for_rank())          # the GPU ordinal = LOCAL_RANK (one process per GPU)
    info = N.device_info()
    if world > 1 and not dry:
        with _stdout_to_stderr():          # (librccl prints a version banner on stdout at communicator creation: not beside the JSON line)
            dist.init(rank, world, broadcast=rdzv.as_broadcast())
    # what RCCL itself saw (ncclCommCount / ncclCommUserRank), read back BEFORE anything is timed: a line that claims N GPUs must
    # come from a communicator of N ranks, each on a GPU of its own
    rccl = dist.comm_info() if not dry else {"communicator": world > 1, "world": world, "rank": rank, "version": "dry-run", "librccl_loaded": False}
    me = {"rank": rank, "local_rank": local_rank, "device": info["device"], "pci": N.device_pci(), "rccl_rank": rccl["rank"], "rccl_world": rccl["world"]}
    gathered_me = rdzv.gather(me)
    bad = (rccl["world"] != world or rccl["rank"] != rank or info["device"] != dist.device_for_rank()
           or len({g["pci"] for g in gathered_me}) != world)
    if bad:
        print("bench.py: RCCL / device check failed on rank %d: WORLD_SIZE %d, RCCL says %s, ranks %s" % (rank, world, rccl, gathered_me), file=sys.stderr)
        return 3

    K, Wm, F = args.steps, args.warmup, args.frames
    if args.only_config == "staggered":
        print(json.dumps({"only_config": "staggered", "library": N.lib().sh_version().decode(), "configs": {"staggered_notes": staggered_row(N, F)}}), flush=True)
        return 0
    if args.only_config == "job":
        voices, gains = build_voices(VOICES_PER_GPU)
        from synthesizer_amd.mixer import VoiceBank
        local = VoiceBank(voices, gains=gains)
        warm = [N.DeviceBuffer(F * 8) for _ in range(4)]
        for k in range(60):                                    # (clocks up, on the frames the job's in-between renders use)
            local.render_device(F, (Wm + k) * F, bus_f32=warm[k & 3])
        print(json.dumps({"only_config": "job", "library": N.lib().sh_version().decode(),
                          "configs": {"job_from_frame_0": job_row(N, local, F, Wm, committed_profile(), with_run=False)}}), flush=True)
        return 0
    if args.only_config:
        rows = config_rows(N, committed_profile(), only=args.only_config, K=K)
        print(json.dumps({"only_config": args.only_config, "library": N.lib().sh_version().decode(), "configs": rows}), flush=True)
        return 0
    total_voices = STRONG_VOICES if args.scaling == "strong" else VOICES_PER_GPU * world
    voices, gains = build_voices(total_voices)
    bank = dist.DistVoiceBank(voices, gains, rank, world, batch=args.reduce_batch, backend=_DryBackend() if dry else None)
    local_voices = bank.hi - bank.lo
    voice_shards = rdzv.gather([int(bank.lo), int(bank.hi)])      # every rank's [lo, hi) of the voice table (rank 0 prints them: they must tile it)
    L = N.lib()

    def barrier():
        N.sync()
        rdzv.barrier()

    allmax = rdzv.allmax

    # ---- fused path (headline) ----
    for s in range(Wm):
        bank.render_device(F, s * F)
    bank.flush()
    barrier()
    passes = []               # (wall s, HIP-event ms) of each pass of K steps, max over ranks
    step0, timed = Wm, 0.0
    host_enqueue = 0.0
    while True:
        barrier()
        t0 = time.perf_counter()
        N.timer_start()
        for s in range(K):
            bank.render_device(F, (step0 + s) * F)
        bank.flush()
        host_enqueue = time.perf_counter() - t0      # host time to enqueue all K steps (launches are asynchronous)
        ev_ms = N.timer_stop()                       # HIP events on the library stream (also synchronises it)
        barrier()
        wall = time.perf_counter() - t0
        wall, ev_ms = allmax(wall, ev_ms)
        passes.append((wall, ev_ms))
        step0 += K
        timed += wall
        if (timed >= args.min_seconds and len(passes) >= 3) or len(passes) >= args.max_passes:
            break
    # self-check: the last block of the last timed pass (left in the bank's bus buffer by the pipelined run) against a fresh
    # single render of the same block (records by a prepare kernel, fold by k_bus_combine: none of the run's machinery)
    verified = None
    if world == 1 and not dry:
        import numpy as np
        last = (step0 - 1) * F
        got = bank._bus32[0].download(np.float32, F * 2)
        chk = N.DeviceBuffer(F * 8)
        bank.local.render_device(F, last, bus_f32=chk)
        ref = chk.download(np.float32, F * 2)
        chk.free()
        verified = {"ok": bool(np.array_equal(got, ref) and np.isfinite(got).all() and float(np.abs(got).max()) > 1e-3),
                    "block_start_frame": int(last), "abs_max": float(np.abs(got).max()),
                    "checksum_f64": float(got.astype(np.float64).sum()),
                    "note": "last timed block, downloaded after the timed region, == a fresh single render of that block (bit for bit)"}
    walls = sorted(p[0] for p in passes)
    wall = statistics.median(walls)
    ev_ms = statistics.median(sorted(p[1] for p in passes))

    prof = committed_profile()
    src_hash = B.source_hash()
    prof_stale = prof["source_hash"] is not None and prof["source_hash"] != src_hash

    voice_samples = float(total_voices) * F * K
    value = voice_samples / wall / 1e6
    kern_s = ev_ms / 1e3 / K         # time per block of the stream of launches: HIP events over the median pass / K
    fused_bytes = 8.0 * F            # algorithmic: one float32 stereo frame written per output frame
    # float64 VALU lane-operations per voice-sample of the render kernel on this workload, from the committed rocprofv3
    # counters: (SQ_INSTS_VALU_FMA_F64 + MUL_F64 + ADD_F64) wave-instructions x 64 lanes / voice-samples per dispatch
    # (the render kernel = the k_render_* instantiation with the most float64 FMAs: the lean kernel of a split launch)
    # (the instantiations of a segmented transition launch -- MODE 6, 7, 8: block 0 of the profiled run -- are not the steady state)
    steady_render = steady_render_kernel
    render_name, render_counters = None, None
    for k_, v_ in prof["counters"].items():
        if steady_render(k_) and (render_counters is None or v_.get("SQ_INSTS_VALU_FMA_F64", 0) > render_counters.get("SQ_INSTS_VALU_FMA_F64", 0)):
            render_name, render_counters = k_, v_
    if render_counters and all(k in render_counters for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
        fma, mul, add = (render_counters[k] for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
        per_launch = float(VOICES_PER_GPU * SR)      # the profiled dispatches: 1024 voices x 48 000 frames each (tools/profile_round.sh)
        lane_ops = (fma + mul + add) * 64.0 / per_launch
        flops_per_vs = (2.0 * fma + mul + add) * 64.0 / per_launch
        ops_source = prof.get("counters_source")
    else:
        lane_ops, flops_per_vs = 25.97, 25.97 + 16448494 * 64.0 / (VOICES_PER_GPU * SR)     # profiles/r01_summary.md
        ops_source = "profiles/r01_summary.md (literal: no rNN_counters.json committed)"
    valu_achieved = local_voices * F * lane_ops / kern_s / 1e12
    # HBM traffic of one block: every render dispatch of it (lean + general-lists kernel of a split launch)
    render_traffic = [v_["hbm_bytes"] for k_, v_ in prof["traffic"].items() if steady_render(k_)]
    traffic_bytes = sum(render_traffic) if render_traffic else None
    out = {
        "metric": "Msamples/sec mixed to stereo bus, 1024-voice additive @48kHz",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not dry else "dry-run (no GPU: launch / rendezvous / timing plumbing only -- not a measurement)",
        "config": {"workload": "%d-voice additive (Harmonics x%d partials + ADSR) -> float32 stereo bus, 48 kHz, "
                               "fused generate-and-mix, block %d frames" % (total_voices, PARTIALS, F),
                   "voices_total": total_voices, "voices_this_rank": local_voices, "frames_per_step": F, "samplerate": SR,
                   "voice_shards": voice_shards,
                   "adsr": "attack 0.01 s, decay 0.05 s, sustain level 0.6 held for the whole run (SURVEY 8(d)'s 0.5 s sustain would leave "
                           "every block after the first silent), release 0.2 s",
                   "parallelism": ("voice-shard x%d (%s scaling), pipelined RCCL reduce of float64 partial buses every %d blocks"
                                   % (world, args.scaling, bank.batch)) if world > 1 else "single GPU"},
        "passes": {"count": len(passes), "steps_per_pass": K, "timed_region_s": timed,
                   "median_ms_per_step": wall * 1e3 / K, "min_ms_per_step": walls[0] * 1e3 / K, "max_ms_per_step": walls[-1] * 1e3 / K,
                   "first_pass_ms_per_step": passes[0][0] * 1e3 / K,
                   "value_of_min_pass": voice_samples / walls[0] / 1e6,
                   "note": "every pass = exactly K consecutive blocks between barrier + device sync on both sides, max over ranks; "
                           "value / ms_per_step are the median pass's; the first passes run while the clock governor ramps up"},
        "frames_per_s": F * K / wall,
        "host_enqueue_ms_per_step": host_enqueue * 1e3 / K,
        "realtime_factor": F * K / wall / SR,
        "device": info["name"] or "AMD Instinct MI355X", "arch": info["arch"],
        "library": L.sh_version().decode(),
        "rccl": {"world": rccl["world"], "rank": rccl["rank"], "version": rccl["version"], "communicator": rccl["communicator"],
                 "ranks": gathered_me,
                 "note": "world / rank: ncclCommCount / ncclCommUserRank of this process's communicator, read back before the timed region "
                         "(no communicator at N = 1: world 1 by definition); ranks: every rank's GPU ordinal and PCI bus id -- the run "
                         "aborts (exit 3) unless RCCL's world equals WORLD_SIZE and the PCI ids are distinct"},
        "verified": verified,
        "roofline": {
            "kernel": render_name or "k_render_lean<4, 8, 4, 0, false>", "bound": "valu_f64",
            "kernel_note": "k_render_lean<WAVES, FPL, MINW, KINDS, SEG>: <4, 16, 3, 0, false> = the lean Harmonics kernel of a split launch (table lookup + "
                           "rotation + three-term recurrence + four Horner chains at a time, sixteen frames per lane, three waves per SIMD; <4, 8, 4, ..> for "
                           "shorter blocks and smaller banks; KINDS 2: FM Sine banks); k_render_general<4, 4, 4, 0> = the "
                           "general-lists kernel that follows it on the same stream where a launch can hold a voice that needs the general code",
            "clock_note": "peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz; under this load the chip holds 2.12-2.18 GHz (s_memtime / s_memrealtime per "
                          "wavefront, profiles/r04_headline_phases.md): at that clock the stream of launches issues within 6-8 % of what the SIMDs can",
            "achieved": valu_achieved, "peak": FP64_PEAK_TOPS, "unit": "T f64 lane-ops/s", "frac": valu_achieved / FP64_PEAK_TOPS,
            "ops_per_voice_sample": lane_ops, "ops_source": ops_source,
            "flops": {"achieved_TFLOPs": local_voices * F * flops_per_vs / kern_s / 1e12, "peak_TFLOPs": 2 * FP64_PEAK_TOPS,
                      "frac": local_voices * F * flops_per_vs / kern_s / 1e12 / (2 * FP64_PEAK_TOPS),
                      "note": "the same work counted in FLOPs (FMA = 2) against the 78.6 TFLOP/s vector figure: lower, because "
                              "MUL and ADD fill an issue slot with one FLOP"},
            "traffic": traffic_bytes, "traffic_source": prof["source"],
            # SURVEY 8(d) regime ii asks for the sine evaluations too: the algorithm's -- one per partial and voice-sample (the reference calls
            # math.sin PARTIALS times per sample; the kernel gets them from one table lookup per sixteen frames, a recurrence and a Horner chain)
            "sin_evals_per_s": local_voices * F * PARTIALS / kern_s,
            "avg_launch_ms": kern_s * 1e3,
            "launches_in_flight": 1 if os.environ.get("SYNTHHIP_NO_OVERLAP") == "1" else 2,
            "why": "the fused kernel writes 8 B per output frame and reads only the voice table: HBM is idle by construction "
                   "(SURVEY 8(d) regime ii); what binds is float64 VALU issue (one lane-op slot per FMA / MUL / ADD)",
            "hbm": {"bound": "hbm", "achieved": fused_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": fused_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": fused_bytes,
                    "traffic": traffic_bytes,
                    "traffic_over_algorithmic": (traffic_bytes / fused_bytes) if traffic_bytes else None,
                    "note": "measured traffic is the float64 partial buses of the voice groups (16 B x frames x groups written, "
                            "then read by the fold two launches on) next to the 8 B per frame of the float32 bus"},
            "profile_stale": prof_stale,
            "profile_note": ("profiles/ counters were measured on kernel sources %s, this library is %s: traffic / ops per voice-sample "
                             "may have drifted" % (prof["source_hash"], src_hash)) if prof_stale else
                            ("profiles/ counters and this library share source hash %s" % src_hash if prof["source_hash"] else
                             "profiles/ carries no source hash (pre-round-2 summary)"),
            "timing_note": "avg_launch_ms = HIP events over the median pass / K (time per launch of the stream of launches); "
                           "consecutive launches overlap pairwise on two streams, so one kernel's own start-to-end duration "
                           "(rocprofv3 kernel trace) is about twice that -- profiles/ sets both against the serialised run "
                           "(SYNTHHIP_NO_OVERLAP=1), where they coincide",
        },
    }

    # ---- multi-GPU: what each rank spends on rendering and what the exchange adds ----
    if world > 1 and not dry:
        ring = [N.DeviceBuffer(F * 16) for _ in range(4)]
        pos = [step0]

        def local_step():
            bank.local.render_device(F, pos[0] * F, bus_f32=None, bus_f64=ring[pos[0] & 3])
            pos[0] += 1
        render_ms = steady(N, local_step, min_seconds=0.1, reps=K)
        nval = bank.batch * F * 2
        msg = N.DeviceBuffer(nval * 8)
        msg.zero()
        barrier()
        N.timer_start()
        for _ in range(20):
            N.check(L.sh_dist_reduce_bus(msg.handle, nval, 0))
        reduce_ms = N.timer_stop() / 20
        barrier()
        gathered = rdzv.gather([float(render_ms), float(reduce_ms)])
        out["per_rank"] = {"render_us_per_step": [float(g[0]) * 1e3 for g in gathered],
                           "blocking_reduce_us_per_call": [float(g[1]) * 1e3 for g in gathered],
                           "reduce_message_bytes": nval * 8, "blocks_per_reduce": bank.batch,
                           "step_us_with_exchange": wall * 1e6 / K,
                           "exposed_exchange_us_per_step": wall * 1e6 / K - max(float(g[0]) for g in gathered) * 1e3,
                           "note": "render: this rank's shard alone, no collective; blocking reduce: ncclReduce of one batch message "
                                   "back to back on the main stream (the pipelined path hides it behind the next renders)"}
        for b_ in ring:
            b_.free()
        msg.free()

    # ---- the same stream of blocks delivered as int16 PCM (what a player consumes): quantised by the fold itself ----
    if world == 1:
        ring = [N.DeviceBuffer(F * 4) for _ in range(4)]
        pos = [step0]

        def pcm_step():
            bank.local.render_pcm_device(F, pos[0] * F, pcm=ring[pos[0] & 3])
            pos[0] += 1
        pcm_ms = steady(N, pcm_step, min_seconds=0.1, reps=K)
        out["int16_stream"] = {"ms_per_step": pcm_ms, "value": local_voices * F / (pcm_ms / 1e3) / 1e6, "unit": "Msamples/s",
                               "note": "sh_bank_render_pcm into a ring of 4 buffers: saturated int16 stereo straight from the "
                                       "partial-bus fold, launches pipelined like the headline's"}
        for b_ in ring:
            b_.free()

    # ---- BASELINE's job taken literally: 10 s from frame 0 in blocks of F -- block 0 is the notes' attack, decay and a dozen
    # binades of the phase sum (a segmented launch); the passes above measure the steady state
    if world == 1 and not dry:
        out["job_from_frame_0"] = job_row(N, bank.local, F, step0, prof)
        out["job_from_frame_0"]["x_steady"] = out["job_from_frame_0"]["ms"] / 10.0 / out["ms_per_step"]

    # ---- the same stream asked for K blocks per call (sh_bank_render_run into a ring of K windows of one allocation: one crossing
    # of the ABI and one launch per K blocks): what a latency-bound caller gains, and what the host then spends per block ----
    if world == 1 and not dry and not args.no_runs:
        cont = bank.local.make_ring(F, K)
        rpos = [step0]

        def run_k():
            bank.local.render_run(F, K, rpos[0] * F, ring=cont)
            rpos[0] += K
        for _ in range(2):
            run_k()
        sp_run = {}
        run_ms = steady(N, run_k, min_seconds=0.3, reps=4, spread=sp_run) / K
        t0 = time.perf_counter()
        for _ in range(50):
            run_k()
        run_host = (time.perf_counter() - t0) / (50 * K)
        N.sync()
        out["run_of_blocks"] = {"blocks_per_call": K, "ms_per_step": run_ms, "min_ms_per_step": sp_run["min_ms"] / K, "max_ms_per_step": sp_run["max_ms"] / K,
                                "value": local_voices * F / (run_ms / 1e3) / 1e6, "unit": "Msamples/s",
                                "host_enqueue_ms_per_step": run_host * 1e3, "x_headline": run_ms / (wall * 1e3 / K),
                                "note": "K blocks per call (VoiceBank.render_run): a launch of K x %d frames renders them -- fewer voice groups per "
                                        "launch (fewer float64 partial-bus planes), one tail per K blocks instead of one per block" % F}
        del cont

    # ---- a render call that stands alone (a caller that does not stream: VERDICT r05 item 3): the device synchronised in front of it, a few
    # pipelined renders before that so that the clocks are what a busy caller sees; HIP events around the one call ----
    if world == 1 and not dry and not args.no_runs:
        ring = [N.DeviceBuffer(F * 8) for _ in range(4)]
        lpos = [step0 + 7 * K]
        got = []
        for _ in range(30):
            for _ in range(3):
                bank.local.render_device(F, lpos[0] * F, bus_f32=ring[lpos[0] & 3])
                lpos[0] += 1
            N.sync()
            N.timer_start()
            bank.local.render_device(F, lpos[0] * F, bus_f32=ring[lpos[0] & 3])
            got.append(N.timer_stop() * 1e3)
            lpos[0] += 1
        got.sort()
        out["lone_call"] = {"us_per_block": got[len(got) // 2], "min_us": got[0], "max_us": got[-1], "calls": len(got), "x_headline": got[len(got) // 2] / (wall * 1e6 / K),
                            "note": "one sh_bank_render of one block with nothing in flight beside it: its own kernel (wavefront priorities falling with progress, "
                                    "LaunchArgs::alone) + k_bus_combine + launch latency; a streaming caller pays ms_per_step, a caller that asks for runs of blocks run_of_blocks"}
        for b_ in ring:
            b_.free()

    # ---- notes that do not move in lock-step: 1024 players re-triggering SURVEY 8(d)'s literal note (0.76 s of sound) every second,
    # onsets spread uniformly over the second; one-second blocks 1 .. 20 of the piece ----
    if world == 1 and not args.no_configs:
        out["staggered_notes"] = staggered_row(N, F, prof)
        out["staggered_notes"]["x_headline"] = out["staggered_notes"]["ms_per_step"] / out["ms_per_step"]     # (same run, same box, same protocol)

    # ---- two-step path on rank 0's shard: materialise (generate) + HBM-bound mix ----
    if not args.no_two_step:
        nv = bank.local.nvoices
        F2 = F * 10 if nv * F * 10 * 4 <= (8 << 30) else F     # 10 s blocks: 1.97 GB of voices, far past the 256 MB L3
        vbuf = N.DeviceBuffer(nv * F2 * 4)
        bus = N.DeviceBuffer(F2 * 8)
        # (timed over 0.25 s each, with min / max, like the 16-bit mono resample row: 0.03 s x 3 launches said 0.49, 0.54 and 0.57 of HBM
        #  for the same kernel on three occasions)
        sp_gen, sp_mix = {}, {}
        gen_ms = steady(N, lambda: bank.local.generate_device(F2, Wm * F, out=vbuf), min_seconds=0.25, reps=3, spread=sp_gen)
        mix_ms = steady(N, lambda: bank.local.mix_device(vbuf, F2, bus_f32=bus), min_seconds=0.25, reps=3, spread=sp_mix)
        gen0_ms = steady(N, lambda: bank.local.generate_device(F2, 0, out=vbuf), min_seconds=0.03, reps=3)     # rows that start with the notes
        mix_bytes = (4.0 * nv + 8.0) * F2
        gen_bytes = 4.0 * nv * F2
        tm = _by_prefix(prof["traffic"], "k_mix_bus")
        tg = _by_prefix(prof["traffic"], "k_generate_lean_harm<16, float") or _by_prefix(prof["traffic"], "k_generate_lean_harm<16>")
        out["two_step"] = {
            "frames_per_launch": F2, "voices": nv,
            "value": nv * F2 / ((gen_ms + mix_ms) / 1e3) / 1e6, "unit": "Msamples/s",
            "from_frame_0": {"generate_ms": gen0_ms, "value": nv * F2 / ((gen0_ms + mix_ms) / 1e3) / 1e6, "unit": "Msamples/s",
                             "note": "the same rows starting with the notes (attack, decay, the first binades of the phase sum: the head of the rows in unequal segments)"},
            "roofline_mix": {"kernel": "k_mix_bus_direct<8,4>", "bound": "hbm", "achieved": mix_bytes / (mix_ms / 1e3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mix_bytes / (mix_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": tm["hbm_bytes"] if tm else None, "avg_launch_ms": mix_ms, "bytes_per_frame": 4 * nv + 8,
                             "algorithmic_bytes": mix_bytes},
            "roofline_generate": {"kernel": "k_generate_lean_harm<16, float> (+ k_prepare_segments; k_generate_lists<4, false> where a segment holds general or silent voices)", "bound": "hbm", "achieved": gen_bytes / (gen_ms / 1e3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gen_bytes / (gen_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": tg["hbm_bytes"] if tg else None, "avg_launch_ms": gen_ms, "bytes_per_voice_sample": 4,
                                  "algorithmic_bytes": gen_bytes, "min_ms": sp_gen["min_ms"], "max_ms": sp_gen["max_ms"], "loops": sp_gen["loops"]},
        }
        out["two_step"]["roofline_mix"].update({"min_ms": sp_mix["min_ms"], "max_ms": sp_mix["max_ms"], "loops": sp_mix["loops"]})
        bus.free()
        # ---- the same pair in the reference's own sample format: every voice as int16 (Sample.from_osc_block's quantiser in the
        # generate kernels' epilogue: 2 B per voice-sample reach HBM), then the mixer's saturating audioop.add chain over the rows --
        # mono, and with every voice placed by Sample.stereo(l, r) (audioop.tostereo in registers: the stereo rows never exist) ----
        stride16 = (F2 + 63) & ~63
        rows16 = vbuf.view(0, nv * stride16 * 2) if nv * stride16 * 2 <= vbuf.nbytes else N.DeviceBuffer(nv * stride16 * 2)
        mono16 = N.DeviceBuffer(F2 * 2)
        st16 = N.DeviceBuffer(F2 * 4)
        fac = bank.local.pan_factors_device()
        sp_g16, sp_c16, sp_p16 = {}, {}, {}
        g16_ms = steady(N, lambda: bank.local.generate_i16_device(F2, Wm * F, out=rows16, stride=stride16, check=False),
                        min_seconds=0.25, reps=3, spread=sp_g16)
        bank.local.overflow_check()
        c16_ms = steady(N, lambda: N.check(N.lib().sh_mix_chain_i16(rows16.handle, nv, stride16, F2, mono16.handle)),
                        min_seconds=0.25, reps=3, spread=sp_c16)
        p16_ms = steady(N, lambda: N.check(N.lib().sh_mix_chain_pan_i16(rows16.handle, nv, stride16, F2, fac.handle, st16.handle)),
                        min_seconds=0.25, reps=3, spread=sp_p16)
        sp_f16 = {}
        f16_ms = steady(N, lambda: bank.local.mixdown_i16_device(F2, Wm * F, out=mono16, check=False), min_seconds=0.25, reps=3, spread=sp_f16)
        bank.local.overflow_check()
        fused_stretches = N.lib().sh_get_option(N.SH_INFO_LAST_MIXDOWN_FUSED)
        g16_bytes, c16_bytes, p16_bytes = 2.0 * nv * F2, (2.0 * nv + 2.0) * F2, (2.0 * nv + 4.0) * F2
        tg16 = _by_prefix(prof["traffic"], "k_generate_lean_harm<16, short")
        cg16 = _by_prefix(prof["counters"], "k_generate_lean_harm<16, short")
        tc16, tp16 = _by_prefix(prof["traffic"], "k_mix_chain_direct_s"), _by_prefix(prof["traffic"], "k_mix_chain_pan_direct_s")

        def hbm_roof(kernel, nbytes, ms, sp, traffic, per_unit_key, per_unit):
            return {"kernel": kernel, "bound": "hbm", "achieved": nbytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic["hbm_bytes"] if traffic else None,
                    "avg_launch_ms": ms, per_unit_key: per_unit, "algorithmic_bytes": nbytes,
                    "min_ms": sp["min_ms"], "max_ms": sp["max_ms"], "loops": sp["loops"]}
        rg16 = hbm_roof("k_generate_lean_harm<16, short> (+ k_prepare_segments; k_generate_lists<4, false, short> where a segment holds general or silent voices)",
                        g16_bytes, g16_ms, sp_g16, tg16, "bytes_per_voice_sample", 2)
        # the materialisation does the headline's arithmetic per voice-sample plus the quantiser: with 2 B per sample to store it is
        # float64-VALU-bound, not HBM-bound -- the fraction of the issue peak beside the HBM figure the row is asked for
        if cg16 and all(k in cg16 for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
            ops16 = (cg16["SQ_INSTS_VALU_FMA_F64"] + cg16["SQ_INSTS_VALU_MUL_F64"] + cg16["SQ_INSTS_VALU_ADD_F64"]) * 64.0 / (nv * F2)
            rg16["valu_f64"] = {"ops_per_voice_sample": ops16, "achieved": nv * F2 * ops16 / (g16_ms / 1e3) / 1e12, "peak": FP64_PEAK_TOPS,
                                "unit": "T f64 lane-ops/s", "frac": nv * F2 * ops16 / (g16_ms / 1e3) / 1e12 / FP64_PEAK_TOPS,
                                "all_valu_busy": (cg16.get("SQ_INSTS_VALU", 0) * 4.0 / 1024 / 2.4e9 / (g16_ms / 1e3)) or None}
        # the mono mixdown without the rows (sh_bank_mixdown_i16): the samples enter the chain where they are made; what reaches HBM is one
        # (a, L | U) map per frame and (64-voice chunk, half) -- 16 B x planes per frame, written once, read once -- and 2 B per frame of result
        cf16 = _by_prefix(prof["counters"], "k_generate_lean_harm<16, short, true")
        tf16 = _by_prefix(prof["traffic"], "k_generate_lean_harm<16, short, true")
        tcomb = _by_prefix(prof["traffic"], "k_mixdown_combine")
        fused_row = {"kernel": "k_generate_lean_harm<16, short, true> + k_mixdown_combine", "ms": f16_ms, "min_ms": sp_f16["min_ms"], "max_ms": sp_f16["max_ms"],
                     "loops": sp_f16["loops"], "value": nv * F2 / (f16_ms / 1e3) / 1e6, "unit": "Msamples/s", "fused_stretches_of_last_call": fused_stretches,
                     "x_two_step_i16": (g16_ms + c16_ms) / f16_ms,
                     "hbm": {"bound": "hbm", "algorithmic_bytes": 2.0 * F2, "achieved": 2.0 * F2 / (f16_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 2.0 * F2 / (f16_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": (tf16["hbm_bytes"] + tcomb["hbm_bytes"]) if tf16 and tcomb else None,
                             "note": "2 B per output sample is all the route owes HBM: like the fused float path it is bound by float64 issue, below"}}
        if cf16 and all(k in cf16 for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
            opsf = (cf16["SQ_INSTS_VALU_FMA_F64"] + cf16["SQ_INSTS_VALU_MUL_F64"] + cf16["SQ_INSTS_VALU_ADD_F64"]) * 64.0 / (nv * F2)
            fused_row["roofline"] = {"bound": "valu_f64", "ops_per_voice_sample": opsf, "achieved": nv * F2 * opsf / (f16_ms / 1e3) / 1e12, "peak": FP64_PEAK_TOPS,
                                     "unit": "T f64 lane-ops/s", "frac": nv * F2 * opsf / (f16_ms / 1e3) / 1e12 / FP64_PEAK_TOPS,
                                     "all_valu_busy": (cf16.get("SQ_INSTS_VALU", 0) * 4.0 / 1024 / 2.4e9 / (f16_ms / 1e3)) or None}
        else:
            fused_row["roofline"] = {"bound": "valu_f64", "note": "no counters for the fold kernel in profiles/ yet"}
        out["two_step_i16"] = {
            "frames_per_launch": F2, "voices": nv, "scale": 32767.0,
            "value": nv * F2 / ((g16_ms + c16_ms) / 1e3) / 1e6, "unit": "Msamples/s",
            "value_stereo": nv * F2 / ((g16_ms + p16_ms) / 1e3) / 1e6,
            "x_float_two_step": (gen_ms + mix_ms) / (g16_ms + c16_ms),
            "note": "the route upstream itself takes: oscillator block -> Sample.from_osc_block (int(32767 v)) -> [Sample.stereo(l, r)] -> "
                    "the mixer's audioop.add chain in voice order; byte for byte against the oracle + live audioop in tests/test_gpu_int_mixdown.py",
            "roofline_generate": rg16,
            "roofline_mix": hbm_roof("k_mix_chain_direct_s<4, 4, 8> (four samples per lane, eight row loads in flight)", c16_bytes, c16_ms, sp_c16, tc16, "bytes_per_sample", 2 * nv + 2),
            "fused_mono_mixdown": fused_row,
            "roofline_mix_stereo": hbm_roof("k_mix_chain_pan_direct_s<4, 4, 8> (audioop.tostereo per voice in registers: ten float64-rate operations per frame -- VALU-bound, not HBM-bound)", p16_bytes, p16_ms, sp_p16, tp16, "bytes_per_frame", 2 * nv + 4),
        }
        # ---- what bit-exactness costs on these routes (round 6: the int16 boundary guard is the default): the same bank built without guard
        # lists (params.int16_guard = False: rounds 1-5, one sample in 10^6 .. 2 10^5 a step off the reference late in a note)
        if rank == 0 and world == 1:
            from synthesizer_amd import params as _params
            from synthesizer_amd.mixer import VoiceBank as _VB
            _params.int16_guard = False
            try:
                v_ng, g_ng = build_voices(nv)
                plain = _VB(v_ng, gains=g_ng)
            finally:
                _params.int16_guard = True
            ng16_ms = steady(N, lambda: plain.generate_i16_device(F2, Wm * F, out=rows16, stride=stride16, check=False), min_seconds=0.15, reps=3)
            nf16_ms = steady(N, lambda: plain.mixdown_i16_device(F2, Wm * F, out=mono16, check=False), min_seconds=0.15, reps=3)
            plain.overflow_check()
            out["two_step_i16"]["int16_guard"] = {
                "rows_ms": g16_ms, "rows_ms_without": ng16_ms, "rows_x": g16_ms / ng16_ms,
                "fused_mixdown_ms": f16_ms, "fused_mixdown_ms_without": nf16_ms, "fused_mixdown_x": f16_ms / nf16_ms,
                "note": "default: polynomial-Harmonics samples whose int(scale v) is in doubt are redone term by term (sh_voice::guard_*): equal to the "
                        "oracle's int16 at any time into a note (tests/test_gpu_guard.py: 0 of 67 M samples differ 10 s and 300 s in; without the guard 1 and 12); "
                        "term by term throughout (params.exact_harmonics = True) costs 9-10 x: tools/exact_cost.py"}
        for b_ in (mono16, st16, rows16, vbuf):       # (rows16 may be a window of vbuf: freed before it)
            b_.free()

    # ---- the other BASELINE configs (rank 0, N = 1) ----
    if rank == 0 and world == 1 and not args.no_configs:
        out["configs"] = config_rows(N, prof, K=K)

    # ---- the integer / float PCM rows (rank 0): Sample.resample (configs[4]) and the mixer chain ----
    if rank == 0 and not args.no_pcm_rows:
        out["pcm_rows"] = pcm_rows(N)

    # ---- CPU baseline (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline(args.cpu_frames, all_cores=not args.no_cpu_all_cores)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if out["rccl"]["version"] is None and not dry:
        out["rccl"]["version"] = dist.comm_info()["version"]     # (N = 1: librccl.so is only loaded by the configs[3] row's 1-rank ring)
    out["launcher"] = ("bench.py itself (python bench.py --gpus %d)" % world) if os.environ.get("SYNTHHIP_BENCH_LAUNCHED") == "1" else \
                      ("an external launcher (RANK / WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else "none (one process)")
    out["control_channel"] = {"data": "TCP star beside MASTER_PORT" if world > 1 else None,
                              "barrier": ("shared memory (/dev/shm), spin" if rdzv._slots is not None else "TCP") if world > 1 else None}
    if world > 1:
        barrier()
        if not dry:
            dist.shutdown()
    rdzv.close()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # C stdio of the loaded libraries (RCCL's banner) goes out first
        sys.stderr.flush()
        detail_path = write_detail(out, args.detail)
        line = compact_line(out, detail_path)
        print(line, flush=True)                 # last, so that nothing a library prints on teardown follows it
    return 0


if __name__ == "__main__":
    sys.exit(main())
