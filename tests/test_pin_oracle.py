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


def _oracle_copy(name, **variants):
    """A second instance of oracle/synth_oracle.py (its own VARIANTS table) standing in for a reference that follows other readings."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, str(ROOT / "oracle" / "synth_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.set_variants(**variants)
    return mod


def test_variants_recipe_names_the_reading_a_reference_follows():
    """--variants: against a stand-in reference that computes increments as 2 pi / (sr / f), compares Pulse with <= and quantises with
    round(), the recipe must report differences AND say which readings make the oracle equal -- per case and in the verdict."""
    sys.path.insert(0, str(ROOT / "tools"))
    import pin_oracle as P
    from oracle import synth_oracle as O
    assert O.VARIANTS == {k: v[0] for k, v in O.VARIANT_CHOICES.items()}
    assert len(P.variant_flips(O)) == 5 + 10
    ref = _oracle_copy("ref_like_div", increment="div", pulse="le")
    report = []
    P.diff_oscillators(ref, O, True, report, variants=True)
    assert report, "a reference with other increments must differ somewhere"
    assert O.VARIANTS == {k: v[0] for k, v in O.VARIANT_CHOICES.items()}          # the recipe restores the table
    by_case = {r["case"]: r for r in report}
    # a plain Sine from phase 0 differs through its increment alone (from phase 0.3 the two increments, one ulp apart, round to the same
    # running sums for thousands of samples: the grid's phase-0 cases are the ones that see it) ...
    sine = by_case["NullFilter"]
    assert {"increment": "div"} in sine["variants_that_match"] and {"pulse": "le"} not in sine["variants_that_match"]
    assert {"pulse": "le"} in by_case["Pulse width on a sample"]["variants_that_match"]
    # ... every differing case is explained by increment=div, alone or with pulse=le
    for r in report:
        assert any(f.get("increment") == "div" for f in r["variants_that_match"]) or any(f == {"pulse": "le"} for f in r["variants_that_match"]), r
    verdict = P.variant_verdict(report)
    assert verdict["adopt"].get("increment") == "div" and not verdict["unexplained"], verdict
    # the quantiser: a stand-in that rounds
    class RoundingSample:
        @staticmethod
        def from_osc_block(block, rate, samplewidth=2):
            class R:
                def get_frame_array(self_inner):
                    return [round((2 ** (8 * samplewidth - 1) - 1) * v) for v in block]
            return R()
    block = [0.9999 * __import__("math").sin(0.01 * i) for i in range(3000)] + [1.0, -1.0, 0.0, 1234.0 / 32767.0]
    want = list(RoundingSample.from_osc_block(block, 22050, 2).get_frame_array())
    assert want != list(O.quantise(block, 2))
    assert {"quantise": "round"} in P.explain_by_variants(None, None, O, None, quantise_block=(block, 2, want))


def test_every_variant_reading_changes_some_case_of_the_grid():
    """A reading no case of the grid can tell from the default would make the recipe blind to it."""
    sys.path.insert(0, str(ROOT / "tools"))
    import pin_oracle as P
    from oracle import synth_oracle as O
    for k, v in (("increment", "div"), ("envelope", "le")):
        ref = _oracle_copy("ref_like_%s" % k, **{k: v})
        report = []
        P.diff_oscillators(ref, O, True, report)
        assert report, (k, v)
    # Square int2 / mod1 differ for negative t only; Pulse < / <= where t % 1 lands ON the width: cases made for them
    import numpy as np
    for flip, make in (({"square": "mod1"}, lambda m: m.Square(440.0, 0.8, -0.3, samplerate=48000)),
                       ({"pulse": "le"}, lambda m: m.Pulse(375.0, 0.8, 0.0, 0.25, samplerate=48000))):
        a = np.array(P.take(make(O), 2048))
        old = O.set_variants(**flip)
        try:
            b = np.array(P.take(make(O), 2048))
        finally:
            O.set_variants(**old)
        assert not np.array_equal(a, b), flip
