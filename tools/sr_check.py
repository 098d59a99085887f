"""The GPU parity tests of round 4's new loops (lean lists and tile lists in runs by kind, the FM-only lean kernel, FM Sine tile pairs on the
lean lists' arithmetic) at sample rates other than the 48 kHz the test files fix: 44.1, 96 and 22.05 kHz (other time tables, other piece ends).
usage (GPU box): python tools/sr_check.py"""
import sys
sys.path.insert(0, ".")
import tests.test_gpu_onsets as T
import tests.test_gpu_bank as TB
for sr in (44100, 96000, 22050):
    T.SR = sr
    TB.SR = sr
    T.test_fm_notes_in_runs_of_a_tile_list(None)
    T.test_notes_of_every_plain_kind_tile_by_tile(None)
    TB.test_lean_lists_in_runs_by_kind(None)
    TB.test_fm_only_lean_kernel_in_a_steady_window(None)
    print("sr", sr, "ok")
