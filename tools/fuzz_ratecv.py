"""One-off sweep: Sample.resample against the live audioop.ratecv over random rates, widths, layouts and lengths."""
import audioop
import sys

sys.path.insert(0, ".")
import numpy as np
from synthesizer_amd.sample import Sample

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
DT = {1: np.int8, 2: np.int16, 4: np.int32}
bad = 0
for case in range(400):
    width = int(rng.choice([1, 2, 2, 2, 4]))
    nch = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 8]))
    if rng.random() < 0.5:
        i, o = (int(x) for x in rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000], 2))
    else:
        i, o = int(rng.integers(1, 200000)), int(rng.integers(1, 200000))
    frames = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 2047, 2049, 5000, 30011, 100003]))
    if frames * o / i > 3e6:
        frames = max(1, int(3e6 * i / o))
    info = np.iinfo(DT[width])
    x = rng.integers(info.min, info.max + 1, frames * nch, dtype=np.int64).astype(DT[width])
    want = audioop.ratecv(x.tobytes(), width, nch, i, o, None)[0]
    got = bytes(Sample.from_raw_frames(x.tobytes(), width, max(i, 2), nch).resample(o).view_frame_data()) if i >= 2 else want
    if got != want:
        bad += 1
        print("MISMATCH", width, nch, i, o, frames, len(got), len(want))
print("cases 400 mismatches", bad)
