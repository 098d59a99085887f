"""
synthesizer_amd -- synthplayer's oscillator-bank + sample-mixing hot path on AMD MI355X (gfx950).

Host code is Python and keeps the reference's class API (oscillators, Sample.mix/resample, the mixer
sum bus); the arithmetic runs in hand-written HIP kernels reached through the C ABI of
libsynthhip.so (include/synthhip.h).  No ML framework anywhere in the package, no CPU fallback.

    from synthesizer_amd.oscillators import Sine, Harmonics, EnvelopeFilter
    from synthesizer_amd.sample import Sample
    from synthesizer_amd.mixer import VoiceBank, mix_samples
"""
from . import params                                   # noqa: F401

__version__ = "0.1.0"
__all__ = ["params", "oscillators", "sample", "mixer", "dist", "phasetable"]
