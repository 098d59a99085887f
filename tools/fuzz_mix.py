"""One-off sweep: mixer.mix_samples (the saturating fold in voice order) and Sample.mix against the live audioop.add."""
import audioop
import sys

sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd.mixer import mix_samples
from synthesizer_amd.sample import Sample

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
DT = {1: np.int8, 2: np.int16, 4: np.int32}
bad = 0
for case in range(200):
    width = int(rng.choice([1, 2, 2, 2, 4]))
    nv = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 64, 65, 200]))
    n = int(rng.choice([1, 2, 7, 8, 9, 63, 511, 512, 513, 4097, 20001]))
    info = np.iinfo(DT[width])
    scale = float(rng.choice([1.0, 0.5, 0.05]))
    chunks = [(rng.integers(info.min, info.max + 1, n, dtype=np.int64) * scale).astype(DT[width]) for _ in range(nv)]
    for c in chunks[:3]:
        c[:min(n, 4)] = np.array([info.max, info.min, info.max, info.min], dtype=DT[width])[:min(n, 4)]
    want = chunks[0].tobytes()
    for c in chunks[1:]:
        want = audioop.add(want, c.tobytes(), width)
    got = mix_samples([Sample.from_raw_frames(c.tobytes(), width, 8000, 1) for c in chunks])
    if bytes(got.view_frame_data()) != want:
        bad += 1
        print("MISMATCH chain", width, nv, n)
    a = Sample.from_raw_frames(chunks[0].tobytes(), width, 8000, 1)
    if nv > 1:
        a.mix(Sample.from_raw_frames(chunks[1].tobytes(), width, 8000, 1))
        if bytes(a.view_frame_data()) != audioop.add(chunks[0].tobytes(), chunks[1].tobytes(), width):
            bad += 1
            print("MISMATCH add", width, n)
print("cases 200 mismatches", bad)
