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
