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


def test_api_audit_exits_3_here_and_its_code_runs_against_the_oracle():
    """--api: the signature audit of the reference's public classes against synthesizer_amd's.  Reference absent: exit 3.  So that the
    audit itself cannot rot it runs here with the ORACLE standing in for the reference: the oracle restates upstream's classes with
    upstream's signatures as recalled, so the product's constructors must at least agree with those -- names, order, defaults."""
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "pin_oracle.py"), "--api"], capture_output=True, text=True, timeout=300)
    assert p.returncode in (0, 1, 3), p.stderr[-2000:]
    if p.returncode == 3:
        assert "UNPINNED" in p.stdout
    sys.path.insert(0, str(ROOT / "tools"))
    import pin_oracle as P
    from oracle import synth_oracle as O
    from oracle import sample_oracle as SO
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import sample as GS
    osc_classes = [n for n in ("Sine", "Sawtooth", "Square", "Pulse", "Harmonics", "Triangle", "SquareH", "SawtoothH", "Linear", "WhiteNoise",
                               "EnvelopeFilter", "MixingFilter", "AmpModulationFilter", "ClipFilter", "AbsFilter", "NullFilter",
                               "DelayFilter", "EchoFilter") if hasattr(O, n)]
    assert len(osc_classes) >= 12
    rows = P.audit_api(O, G, osc_classes)
    ctor = [r for r in rows if r["symbol"].endswith(".__init__")]
    assert len(ctor) == len(osc_classes)
    bad = [r for r in ctor if r["status"] != "equal"]
    assert not bad, bad                                   # constructors: same names, order, defaults as the oracle's
    # Sample: every public method of the oracle's RefSample exists on the product's Sample with the same parameters
    rows = P.audit_api(SO, GS, ["RefSample"], rename={"RefSample": "Sample"})
    methods = [r for r in rows if not r["symbol"].endswith(".__init__")]
    assert len(methods) >= 20
    def names(sig):
        return [(n, k) for n, k, _d in sig] if sig else sig
    # (defaults that are function objects -- chunked_frame_data's stopcondition lambda -- have no comparable repr: names and kinds there)
    bad = [r for r in methods if r["status"] == "missing" or (r["status"] == "differs" and names(r.get("ref")) != names(r.get("ours")))]
    assert not bad, bad
    assert P.print_api_table(rows) is not None
