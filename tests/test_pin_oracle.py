"""tools/pin_oracle.py: the recipe that pins oracle/synth_oracle.py and oracle/sample_oracle.py against the real
synthplayer package the first time it is importable in the build container.  Here it must either report the
reference absent (exit 3 -- today: /root/reference is a two-line README) or, where the package exists, find NO
difference (exit 0); differences (exit 1) fail this test.  The outcome is printed so that the round's log records it."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_pin_oracle_outcome(tmp_path):
    out = tmp_path / "pin.json"
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "pin_oracle.py"), "--quick", "--json", str(out)],
                       capture_output=True, text=True, timeout=900)
    outcome = json.loads(out.read_text())
    print("pin_oracle: rc %d, status %r, reference %r" % (p.returncode, outcome["status"], outcome["reference"]))
    assert p.returncode in (0, 3), p.stdout[-3000:] + p.stderr[-2000:]
    if p.returncode == 3:
        assert outcome["status"] == "reference absent" and outcome["reference"] is None
        assert "UNPINNED" in p.stdout
    else:
        assert outcome["status"] == "pinned" and outcome["differences"] == []


def test_case_grid_runs_against_the_oracle_itself():
    """The seed grid must at least construct and run with the oracle's own classes on both sides (no difference by
    construction): a typo in the grid must not wait for the day the reference shows up."""
    sys.path.insert(0, str(ROOT / "tools"))
    import pin_oracle as P
    from oracle import synth_oracle as O
    report = []
    P.diff_oscillators(O, O, True, report)
    assert report == []
    P.diff_params(O, O, report)
    assert report == []
    assert P.take(O.Sine(440, samplerate=8000), 10, skip=1000) == O.Sine(440, samplerate=8000).take(1010)[1000:]
    assert len(P.oscillator_cases(False)) > 200
