"""Global defaults, same names as upstream synthplayer/params.py (module-level globals)."""

norm_samplerate = 44100
norm_nchannels = 2
norm_samplewidth = 2
norm_frames_per_chunk = norm_samplerate // 30
norm_osc_blocksize = 512

# ---- not upstream (INTEGRATION.md "Extensions") --------------------------------------------------------------------------------
# Harmonics voices whose partials are all integers <= 16 are evaluated as sin(t) * P(cos t) (15 FMAs per sample instead of 16 sines; a
# Clenshaw recurrence for denser lists): that is sum_k a_k sin(k t) of the EXACT products k t, while the reference rounds every t * k
# before its sine -- the two part ways as t grows (1.6e-10 after 300 s at 48 kHz: inside the 1e-6 float contract by four orders of
# magnitude, but where int(scale * v) of such a sample sits that close to an integer the two truncate to neighbouring integers: one
# int16 sample in 1e6 .. 2e5).
#   None (default)  a single oscillator (blocks(), render*, Sample.from_osc*: launch-bound whatever the form) sums term by term,
#                   sin(fl(t * k)) * a_k in list order like the reference's loop; a VoiceBank keeps the fast forms for the float bus
#                   and, on its int16 routes (generate_i16 / mixdown_i16 / voice_samples), redoes term by term exactly the samples
#                   whose integer is in doubt (`int16_guard`, sh_voice::guard_* in include/synthhip.h): equal integers at any time
#                   into a note at ~1.1 x the arithmetic (tools/exact_cost.py: term by term throughout costs 7-13 x)
#   True            every Harmonics voice built from now on sums term by term everywhere (the float bus too)
#   False           the fast forms everywhere, single oscillators included (the guard still protects a bank's int16 routes)
exact_harmonics = None
# False: voices built from now on carry no guard list -- a bank's int16 routes quantise the fast forms as they are (rounds 1-5; an A/B switch)
int16_guard = True

# Readings of the recalled arithmetic that differ by a last bit or a comparison operator (same names and values as
# oracle/synth_oracle.py VARIANTS; tools/pin_oracle.py --variants tells which reading the real package follows -- adopting it is a
# change of these defaults, not of any kernel: the device reproduces whatever accumulated phase, boundaries and records the host
# code below describes).  Read when an oscillator's record is built / a block is quantised.
#   increment  "mul": 2 pi f / sr (f / sr for the turn-based kinds)       "div": rate = sr / f; 2 pi / rate (1 / rate)
#   square     "int2": -a if int(t * 2) % 2 else a                        "mod1": a if t % 1.0 < 0.5 else -a   (a Pulse record, width 0.5)
#   pulse      "lt": a if t % 1.0 < pulsewidth else -a                    "le": ... <= pulsewidth              (the width's successor)
#   quantise   "trunc": int(scale * v)                                    "round": round(scale * v), half to even (sh_set_option)
#   envelope   "lt": while time < phase_end                               "le": while time <= phase_end        (the boundaries move)
variant_choices = {"increment": ("mul", "div"), "square": ("int2", "mod1"), "pulse": ("lt", "le"), "quantise": ("trunc", "round"),
                   "envelope": ("lt", "le")}
variants = {k: v[0] for k, v in variant_choices.items()}


def set_variants(**kw) -> dict:
    """Change readings (returns the previous table, for a try / finally); unknown names or values raise.  Objects built before keep the
    records they were built with."""
    old = dict(variants)
    for k, v in kw.items():
        if k not in variant_choices or v not in variant_choices[k]:
            raise ValueError("variant %s=%r: choose from %r" % (k, v, variant_choices.get(k)))
        variants[k] = v
    if "quantise" in kw:
        from . import _native
        _native.set_quantise_round(variants["quantise"] == "round")
    return old
