#!/usr/bin/env python
"""bench.py -- headline benchmark of the oscillator-bank + mixer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): 1024 additive voices PER GPU -- each a
Harmonics oscillator with 16 partials a_k = 1/k under an ADSR envelope, f log-uniform in [55, 3520] Hz,
random phase/pan, seed 0 -- mixed to one float32 stereo bus at 48 kHz.  One step = one block of 48 000
frames (1 s of audio) rendered by the fused generate-and-mix kernel; steps render consecutive blocks and
the envelope spans the whole run (no silent voices).  With N > 1 the voice table (N x 1024 voices, weak
scaling; configs[3] at N = 8) is sharded across ranks and the float64 partial buses are summed to rank 0
by RCCL every step.

value = voice-samples mixed per second over all ranks / 1e6 ("Msamples/s"), inputs (the voice table)
resident in HBM before the timed region.  The same run also times (a) the reference-shaped two-step
path (voices materialised as float32 PCM in HBM, then the HBM-bound mixer kernel) and (b) the CPU
oracle (pure-Python generators, 1 core) on a bounded sample -- reported beside, never as `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SR = 48000
VOICES_PER_GPU = 1024
PARTIALS = 16
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_PEAK_GOPS = 256 * 4 * 16 * 2.4   # float64 FMA lanes/s (78.6 TFLOP/s spec / 2), in G lane-ops/s


def build_voices(n_total: int, seconds_total: float):
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    # envelope spans the whole run: attack 0.01, decay 0.05, release 0.2, sustain fills the rest
    return W.additive_voices(G, n_total, SR, seed=0, partials=PARTIALS,
                             adsr={"sustain": max(0.0, seconds_total - 0.26)})


def cpu_baseline(frames: int):
    """The oracle (pure-Python generator restatement of the reference path) on ONE core, timed on a
    bounded sample of the same workload: all 1024 voices x `frames` frames, mixed to the stereo bus."""
    from oracle import synth_oracle as O
    from synthesizer_amd import workloads as W
    voices, gains = W.additive_voices(O, VOICES_PER_GPU, SR, seed=0, partials=PARTIALS)
    t0 = time.perf_counter()
    blocks = [v.take(frames) for v in voices]
    O.mix_bus(blocks, gains)
    dt = time.perf_counter() - t0
    out = {"value": VOICES_PER_GPU * frames / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "1024 voices x %d frames (%.3f s of audio), pure-Python oracle generators + float bus sum, "
                     "%.1f s wall" % (frames, frames / SR, dt)}
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
    return out


def measured_traffic():
    """HBM bytes per dispatch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, profiles/rNN_traffic.json written by tools/summarize_profiles.py); None when absent."""
    best = None
    for p in sorted((ROOT / "profiles").glob("r*_traffic.json")):
        best = p
    if best is None:
        return {}, None
    try:
        return json.loads(best.read_text()), best.name
    except Exception:
        return {}, None


def pcm_rows(N):
    """HBM-bound rows of the path, each timed with HIP events on resident buffers."""
    import ctypes
    import numpy as np
    L = N.lib()
    rows = {}
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
        for _ in range(2):
            N.check(L.sh_resample(src.handle, frames, nch, width, is_float, 96000, 44100, dst.handle, None))
        N.sync()
        reps = 5
        N.timer_start()
        for _ in range(reps):
            N.check(L.sh_resample(src.handle, frames, nch, width, is_float, 96000, 44100, dst.handle, None))
        ms = N.timer_stop() / reps
        nbytes = (frames + nout_w) * nch * width
        rows[name] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / (ms / 1e3) / 1e9, "frac_hbm": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                      "out_Mframes_per_s": nout_w / (ms / 1e3) / 1e6}
    # 16-bit mono / stereo, the shapes Sample.resample sees in practice (WaveSynth output, loaded WAVs): 900 MB in
    for name, nch, inr, outr in (("resample_i16_mono_44k1_to_48k_900MB", 1, 44100, 48000),
                                 ("resample_i16_stereo_44k1_to_48k_900MB", 2, 44100, 48000),
                                 ("resample_i16_stereo_96k_to_44k1_900MB", 2, 96000, 44100)):
        frames = 450_000_000 // nch
        nout_m = L.sh_resample_out_frames(frames, inr, outr)
        big = N.DeviceBuffer(nout_m * 2 * nch)
        for _ in range(2):
            N.check(L.sh_resample(src.handle, frames, nch, 2, 0, inr, outr, big.handle, None))
        N.sync()
        N.timer_start()
        for _ in range(5):
            N.check(L.sh_resample(src.handle, frames, nch, 2, 0, inr, outr, big.handle, None))
        ms = N.timer_stop() / 5
        nbytes = (frames + nout_m) * 2 * nch
        rows[name] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / (ms / 1e3) / 1e9,
                      "frac_hbm": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
        big.free()
    # mixer chain: 1024 int16 voices x 10 s stereo (saturating fold in voice order), 2N+2 bytes per sample
    nv, nsamples = 1024, 48000 * 2 * 10
    chunks = N.DeviceBuffer(nv * nsamples * 2)
    pcm = (np.random.default_rng(1).integers(-3000, 3000, 1 << 24)).astype(np.int16)
    for off in range(0, chunks.nbytes, pcm.nbytes):
        chunks.upload(pcm[:min(len(pcm), (chunks.nbytes - off) // 2)], off)
    mixed = N.DeviceBuffer(nsamples * 2)
    for _ in range(2):
        N.check(L.sh_mix_chain_i16(chunks.handle, nv, nsamples, nsamples, mixed.handle))
    N.sync()
    N.timer_start()
    for _ in range(5):
        N.check(L.sh_mix_chain_i16(chunks.handle, nv, nsamples, nsamples, mixed.handle))
    ms = N.timer_stop() / 5
    nbytes = (2 * nv + 2) * nsamples
    rows["mix_chain_i16_1024v_10s_stereo"] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / (ms / 1e3) / 1e9,
                                              "frac_hbm": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    # the same fold through the pointer table (RealTimeMixer / mix_samples: every source read where it lives)
    bufs = (ctypes.c_void_p * nv)(*[chunks.handle] * nv)
    offs = (ctypes.c_size_t * nv)(*[v * nsamples for v in range(nv)])
    lens = (ctypes.c_uint32 * nv)(*[nsamples] * nv)
    for _ in range(2):
        N.check(L.sh_mix_chain_gather_i16(bufs, offs, lens, nv, nsamples, mixed.handle, 0))
    N.sync()
    N.timer_start()
    for _ in range(5):
        N.check(L.sh_mix_chain_gather_i16(bufs, offs, lens, nv, nsamples, mixed.handle, 0))
    ms = N.timer_stop() / 5
    rows["mix_chain_gather_i16_1024v_10s_stereo"] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / (ms / 1e3) / 1e9,
                                                     "frac_hbm": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    # one real-time turn: 64 sources x 4096-byte chunk (latency-bound: table upload + one small kernel)
    small = 2048
    for _ in range(3):
        N.check(L.sh_mix_chain_gather_i16(bufs, offs, lens, 64, small, mixed.handle, 0))
    N.sync()
    N.timer_start()
    for _ in range(200):
        N.check(L.sh_mix_chain_gather_i16(bufs, offs, lens, 64, small, mixed.handle, 0))
    rows["mixer_turn_64src_4KB_chunk"] = {"ms": N.timer_stop() / 200}
    # Sample.from_osc_block: float32 -> int16 with the overflow check (the call returns after reading the flag back)
    nq = 300_000_000
    for _ in range(2):
        N.check(L.sh_quantize_f32(src.handle, 0, nq, 32767.0, 2, dst.handle, 0))
    N.timer_start()
    for _ in range(5):
        N.check(L.sh_quantize_f32(src.handle, 0, nq, 32767.0, 2, dst.handle, 0))
    ms = N.timer_stop() / 5
    rows["quantize_f32_to_i16_1200MB"] = {"ms": ms, "bytes": 6 * nq, "GBps": 6 * nq / (ms / 1e3) / 1e9,
                                          "frac_hbm": 6 * nq / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    # Sample.mix: saturating add of two 900 MB int16 buffers (3 bytes moved per byte of output)
    n = 900_000_000
    N.timer_start()
    for _ in range(5):
        N.check(L.sh_pcm_add(chunks.handle, 0, chunks.handle, n, n, 2, src.handle, 0))
    ms = N.timer_stop() / 5
    rows["pcm_add_i16_900MB"] = {"ms": ms, "bytes": 3 * n, "GBps": 3 * n / (ms / 1e3) / 1e9, "frac_hbm": 3 * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    # Sample.amplify (audioop.mul), Sample.mono (audioop.tomono), peak/rms
    n = 900_000_000
    for name, call, moved in (
            ("pcm_mul_i16_900MB", lambda: L.sh_pcm_mul(chunks.handle, 0, n, 2, 0.7071, src.handle, 0), 2 * n),
            ("pcm_tomono_i16_900MB", lambda: L.sh_pcm_tomono(chunks.handle, n // 4, 2, 0.5, 0.5, src.handle), n + n // 2),
            ("pcm_stats_i16_900MB", lambda: L.sh_pcm_stats(chunks.handle, n, 2, ctypes.byref(ctypes.c_uint32()), ctypes.byref(ctypes.c_double())), n),
            ("pcm_stats_stereo_i16_900MB", lambda: L.sh_pcm_stats_stereo(chunks.handle, n // 4, 2, (ctypes.c_uint32 * 2)(), (ctypes.c_double * 2)()), n)):
        N.check(call())
        N.sync()
        N.timer_start()
        for _ in range(5):
            N.check(call())
        ms = N.timer_stop() / 5
        rows[name] = {"ms": ms, "bytes": moved, "GBps": moved / (ms / 1e3) / 1e9, "frac_hbm": moved / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    for b in (src, dst, chunks, mixed):
        b.free()
    return rows


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=SR, help="frames per step (block size)")
    ap.add_argument("--cpu-frames", type=int, default=12288, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-pcm-rows", action="store_true", help="skip the resample / integer-mix rows")
    ap.add_argument("--no-two-step", action="store_true")
    ap.add_argument("--reduce-batch", type=int, default=8, help="blocks per RCCL reduce when --gpus > 1 (also the length of a run of pipelined renders)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs one process per GPU (torch.distributed.run)" % args.gpus, file=sys.stderr)
            return 2
    td = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as td      # plumbing only: barrier + max over ranks + id broadcast
        td.init_process_group("gloo", rank=rank, world_size=world)

    from synthesizer_amd import _native as N
    from synthesizer_amd import dist
    N.ensure_init(local_rank)
    info = N.device_info()
    if world > 1:
        dist.init(rank, world)

    K, Wm, F = args.steps, args.warmup, args.frames
    total_voices = VOICES_PER_GPU * world
    seconds_total = (K + Wm) * F / SR
    voices, gains = build_voices(total_voices, seconds_total)
    bank = dist.DistVoiceBank(voices, gains, rank, world, batch=args.reduce_batch)
    L = N.lib()

    def barrier():
        N.sync()
        if td is not None:
            td.barrier()

    # ---- fused path (headline) ----
    for s in range(Wm):
        bank.render_device(F, s * F)
    bank.flush()
    barrier()
    t0 = time.perf_counter()
    N.timer_start()
    for s in range(K):
        bank.render_device(F, (Wm + s) * F)
    bank.flush()
    host_enqueue = time.perf_counter() - t0      # host time to enqueue all K steps (launches are asynchronous)
    ev_ms = N.timer_stop()           # HIP events on the library stream (also synchronises it)
    barrier()
    wall = time.perf_counter() - t0
    if td is not None:
        import torch
        t = torch.tensor([wall, ev_ms], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])

    traffic, traffic_src = measured_traffic()

    def traffic_of(prefix):
        for k, v in traffic.items():
            if k.startswith(prefix):
                return v["hbm_bytes"]
        return None

    voice_samples = float(total_voices) * F * K
    value = voice_samples / wall / 1e6
    kern_s = ev_ms / 1e3 / K         # average duration of one block (k_locate + k_bank_render [+ reduce/finalize])
    fused_bytes = 8.0 * F            # algorithmic: one float32 stereo frame written per output frame
    # float64 VALU lane-operations per voice-sample of k_bank_render on this workload, from rocprofv3:
    # (SQ_INSTS_VALU_FMA_F64 + MUL_F64 + ADD_F64) * 64 / voice-samples per launch = 20.34 M * 64 / 49.152 M
    # (profiles/r01_summary.md, "SQ instruction mix"; all VALU: 33.0 per voice-sample, prepare step included)
    harm_lane_ops = 26.5
    out = {
        "metric": "Msamples/sec mixed to stereo bus, 1024-voice additive @48kHz",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d-voice additive (Harmonics x%d partials + ADSR) -> float32 stereo bus, 48 kHz, "
                               "fused generate-and-mix, block %d frames" % (total_voices, PARTIALS, F),
                   "voices_per_gpu": VOICES_PER_GPU, "frames_per_step": F, "samplerate": SR,
                   "parallelism": ("voice-shard x%d, pipelined RCCL reduce of float64 partial buses every %d blocks" % (world, bank.batch)) if world > 1 else "single GPU"},
        "frames_per_s": F * K / wall,
        "host_enqueue_ms_per_step": host_enqueue * 1e3 / K,
        "realtime_factor": F * K / wall / SR,
        "device": info["name"] or "AMD Instinct MI355X", "arch": info["arch"],
        "roofline": {
            "kernel": "k_bank_render<4,4,4,1>", "bound": "hbm",
            "achieved": fused_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": fused_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of("k_bank_render"),
            "traffic_source": traffic_src,
            "note": "fused kernel writes 8 B per output frame: VALU(float64)-bound by construction, see valu",
            "valu": {"unit": "G f64 lane-ops/s", "achieved": VOICES_PER_GPU * F * harm_lane_ops / kern_s / 1e9,
                     "peak": FP64_PEAK_GOPS, "frac": VOICES_PER_GPU * F * harm_lane_ops / kern_s / 1e9 / FP64_PEAK_GOPS,
                     "ops_per_voice_sample": harm_lane_ops},
            "avg_launch_ms": kern_s * 1e3,
            "launches_in_flight": 1 if os.environ.get("SYNTHHIP_NO_OVERLAP") == "1" else 2,
            "timing_note": "avg_launch_ms = HIP events over the timed region / launches (time per launch of the stream of "
                           "launches); consecutive launches overlap pairwise on two streams, so one kernel's own start-to-end "
                           "duration (rocprofv3 kernel trace) is about twice that -- profiles/r01_summary.md sets both against "
                           "the serialised run (SYNTHHIP_NO_OVERLAP=1), where they coincide",
        },
    }

    # ---- the same stream of blocks delivered as int16 PCM (what a player consumes): quantised by the fold itself ----
    if world == 1:
        ring = [N.DeviceBuffer(F * 4) for _ in range(4)]
        for s in range(Wm):
            bank.local.render_pcm_device(F, s * F, pcm=ring[s & 3])
        N.sync()
        N.timer_start()
        for s in range(K):
            bank.local.render_pcm_device(F, (Wm + s) * F, pcm=ring[s & 3])
        pcm_ms = N.timer_stop() / K
        out["int16_stream"] = {"ms_per_step": pcm_ms, "value": VOICES_PER_GPU * F / (pcm_ms / 1e3) / 1e6, "unit": "Msamples/s",
                               "note": "sh_bank_render_pcm into a ring of 4 buffers: saturated int16 stereo straight from the "
                                       "partial-bus fold, launches pipelined like the headline's"}
        for b_ in ring:
            b_.free()

    # ---- two-step path on rank 0's shard: materialise (generate) + HBM-bound mix ----
    if not args.no_two_step:
        nv = bank.local.nvoices
        F2 = F * 10 if nv * F * 10 * 4 <= (8 << 30) else F     # 10 s blocks: 1.97 GB of voices, far past the 256 MB L3
        vbuf = N.DeviceBuffer(nv * F2 * 4)
        bus = N.DeviceBuffer(F2 * 8)
        reps = max(3, min(20, K // 5))
        for _ in range(2):
            bank.local.generate_device(F2, Wm * F, out=vbuf)
            bank.local.mix_device(vbuf, F2, bus_f32=bus)
        N.sync()
        N.timer_start()
        for r in range(reps):
            bank.local.generate_device(F2, Wm * F, out=vbuf)
        gen_ms = N.timer_stop() / reps
        N.timer_start()
        for r in range(reps):
            bank.local.mix_device(vbuf, F2, bus_f32=bus)
        mix_ms = N.timer_stop() / reps
        mix_bytes = (4.0 * nv + 8.0) * F2
        gen_bytes = 4.0 * nv * F2
        out["two_step"] = {
            "frames_per_launch": F2, "voices": nv,
            "value": nv * F2 / ((gen_ms + mix_ms) / 1e3) / 1e6, "unit": "Msamples/s",
            "roofline_mix": {"kernel": "k_mix_bus_direct<8,4>", "bound": "hbm", "achieved": mix_bytes / (mix_ms / 1e3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mix_bytes / (mix_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic_of("k_mix_bus"), "avg_launch_ms": mix_ms, "bytes_per_frame": 4 * nv + 8,
                             "algorithmic_bytes": mix_bytes},
            "roofline_generate": {"kernel": "k_generate", "bound": "hbm", "achieved": gen_bytes / (gen_ms / 1e3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gen_bytes / (gen_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": traffic_of("k_generate"), "avg_launch_ms": gen_ms, "bytes_per_voice_sample": 4,
                                  "algorithmic_bytes": gen_bytes},
        }
        vbuf.free()
        bus.free()

    # ---- the integer / float PCM rows (rank 0): Sample.resample (configs[4]) and the mixer chain ----
    if rank == 0 and not args.no_pcm_rows:
        out["pcm_rows"] = pcm_rows(N)

    # ---- CPU baseline (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline(args.cpu_frames)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if world > 1:
        barrier()
        dist.shutdown()
        td.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # C stdio of the loaded libraries (RCCL's banner) goes out first
        sys.stderr.flush()
        print(json.dumps(out), flush=True)      # last, so that nothing a library prints on teardown follows it
    return 0


if __name__ == "__main__":
    sys.exit(main())
