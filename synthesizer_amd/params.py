"""Global defaults, same names as upstream synthplayer/params.py (module-level globals)."""

norm_samplerate = 44100
norm_nchannels = 2
norm_samplewidth = 2
norm_frames_per_chunk = norm_samplerate // 30
norm_osc_blocksize = 512
