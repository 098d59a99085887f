"""Global defaults, same names as upstream synthplayer/params.py (module-level globals)."""

norm_samplerate = 44100
norm_nchannels = 2
norm_samplewidth = 2
norm_frames_per_chunk = norm_samplerate // 30
norm_osc_blocksize = 512

# ---- not upstream (INTEGRATION.md "Extensions") --------------------------------------------------------------------------------
# Harmonics voices whose partials are all integers <= 16 are evaluated as sin(t) * P(cos t) (15 FMAs per sample instead of 16 sines):
# that is sum_k a_k sin(k t) of the EXACT products k t, while the reference rounds every t * k before its sine -- the two part ways
# as t grows (1.6e-10 after 300 s at 48 kHz: inside the 1e-6 float contract by four orders of magnitude, but one int16 sample in
# 2e5 then truncates to the neighbouring integer).  True: every Harmonics voice built from now on sums its partials term by term,
# sin(fl(t * k)) * a_k in list order like the reference's loop -- bit-faithful at any time into a note, at ~8 x the arithmetic.
exact_harmonics = False

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
