#!/usr/bin/env python
"""Pin the oscillator / Sample oracle against the REAL synthplayer package -- the moment it is importable.

    python tools/pin_oracle.py [--regen] [--quick] [--json PATH]
    python tools/pin_oracle.py --api [--json PATH]      signatures only: names, order, kinds, defaults of every public class / method

The tree mounted at /root/reference holds a two-line relocation notice (README.md:1-2), so today this script
finds nothing to diff against and exits 3 ("reference absent"); `parity: unpinned` stays in oracle/synth_oracle.py's
header and in DESIGN.md.  In a build container where `import synthplayer` works (a clone of the CodeBerg repository
or the PyPI sdist mounted at /root/reference, or installed) it

  1. diffs every class of oracle/synth_oracle.py against the class of the same name in synthplayer.oscillators over
     a seed grid -- kinds x {plain, fm_lfo, pwm_lfo} x phases x three sample rates, the first blocks AND the blocks
     around sample 2**20 (late enough for the accumulated `t += increment` to have drifted from n*increment) -- and
     demands EQUAL float64 samples (the oracle claims to restate the arithmetic, not to approximate it);
  2. diffs oracle/sample_oracle.RefSample method by method against synthplayer.sample.Sample on random PCM
     (bytes must be equal), and the quantiser oracle.quantise against Sample.from_osc_block;
  3. diffs the module constants the oracle copied from synthplayer.params;
  4. with --regen, rewrites tests/golden/osc_*.np* FROM THE REAL PACKAGE (with a header array saying so), which turns
     the "guards the oracle against accidental edits" files into reference-pinned vectors.

Exit codes: 0 everything equal (oracle pinned), 1 differences found (listed), 3 reference absent.
The reference never travels to the GPU box: this runs in the build container only, and only its OUTPUTS (the golden
vectors) are committed.  SURVEY.md Appendix B lists the nine questions this settles; DESIGN.md section 9 lists the
assumption the oracle currently makes for each.
"""
from __future__ import annotations

import argparse
import itertools
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def find_reference():
    """synthplayer as installed, or from a tree mounted at /root/reference (this container only)."""
    try:
        import synthplayer                                   # noqa: F401
        return "installed"
    except ImportError:
        pass
    ref = Path("/root/reference")
    for cand in (ref, ref / "synthesizer", ref / "src"):
        if (cand / "synthplayer" / "oscillators.py").exists():
            sys.path.insert(0, str(cand))
            try:
                import synthplayer                           # noqa: F401
                return str(cand)
            except ImportError:
                sys.path.pop(0)
    return None


def take(osc, n, skip=0):
    """n samples of a blocks() generator after skipping `skip` (both sides: lists of Python floats)."""
    gen = osc.blocks()
    out = []
    seen = 0
    while len(out) < n:
        try:
            b = next(gen)
        except StopIteration:
            break
        if seen + len(b) <= skip:
            seen += len(b)
            continue
        lo = max(0, skip - seen)
        out.extend(b[lo:])
        seen += len(b)
    return out[:n]


def oscillator_cases(quick: bool):
    """(name, constructor(module) -> oscillator) over the seed grid."""
    rates = (44100, 48000, 22050) if not quick else (48000,)
    phases = (0.0, 0.3, -0.25) if not quick else (0.3,)
    cases = []
    for sr, ph in itertools.product(rates, phases):
        for kind in ("Sine", "Triangle", "Square", "Sawtooth"):
            for f in (440.0, 1000.0, 55.5, 441.0):          # (441 Hz at 48 kHz: f / sr != 1 / (sr / f) -- the turn-based kinds' increment readings part)
                cases.append(("%s f=%g sr=%d ph=%g" % (kind, f, sr, ph),
                              lambda m, kind=kind, f=f, sr=sr, ph=ph: getattr(m, kind)(f, 0.8, ph, 0.1, samplerate=sr)))
                cases.append(("%s f=%g sr=%d ph=%g fm" % (kind, f, sr, ph),
                              lambda m, kind=kind, f=f, sr=sr, ph=ph: getattr(m, kind)(f, 0.8, ph, 0.0, fm_lfo=m.Sine(5.0, 0.05, samplerate=sr), samplerate=sr)))
        cases.append(("Pulse sr=%d ph=%g" % (sr, ph), lambda m, sr=sr, ph=ph: m.Pulse(441.0, 0.7, ph, 0.25, 0.05, samplerate=sr)))
        cases.append(("Pulse pwm sr=%d ph=%g" % (sr, ph),
                      lambda m, sr=sr, ph=ph: m.Pulse(441.0, 0.7, ph, 0.25, 0.0, pwm_lfo=m.Sine(2.0, 0.2, bias=0.4, samplerate=sr), samplerate=sr)))
        cases.append(("Pulse fm+pwm sr=%d" % sr,
                      lambda m, sr=sr, ph=ph: m.Pulse(441.0, 0.7, ph, 0.25, 0.0, fm_lfo=m.Triangle(3.0, 0.1, samplerate=sr),
                                                      pwm_lfo=m.Sine(2.0, 0.2, bias=0.4, samplerate=sr), samplerate=sr)))
        harm = [(k, 1.0 / k) for k in range(1, 17)]
        cases.append(("Harmonics x16 sr=%d ph=%g" % (sr, ph), lambda m, sr=sr, ph=ph: m.Harmonics(220.0, harm, 0.5, ph, samplerate=sr)))
        cases.append(("Harmonics sparse fm sr=%d" % sr,
                      lambda m, sr=sr, ph=ph: m.Harmonics(220.0, [(1, 1.0), (2.5, 0.3), (7, 0.1)], 0.5, ph, fm_lfo=m.Sine(4.0, 0.02, samplerate=sr), samplerate=sr)))
        cases.append(("SquareH sr=%d" % sr, lambda m, sr=sr, ph=ph: m.SquareH(220.0, 9, 0.8, ph, samplerate=sr)))
        cases.append(("SawtoothH sr=%d" % sr, lambda m, sr=sr, ph=ph: m.SawtoothH(220.0, 9, 0.8, ph, samplerate=sr)))
    sr = 48000
    cases += [
        ("Linear up", lambda m: m.Linear(-0.5, 1e-4, -1.0, 0.25, samplerate=sr)),
        ("Linear const", lambda m: m.Linear(0.3, samplerate=sr)),
        # cases made to tell the readings of oracle.synth_oracle.VARIANTS apart (--variants): phase boundaries that fall ON a sample
        # (1 / 32768 is a float64: the accumulated time is exact), a Square on a negative phase, a Pulse whose t % 1 lands on its width
        ("Envelope boundaries on samples", lambda m: m.EnvelopeFilter(m.Sine(512.0, samplerate=32768), 1.0 / 64, 1.0 / 128, 1.0 / 64, 0.5, 1.0 / 128)),
        ("Square negative phase", lambda m: m.Square(440.0, 0.8, -0.3, 0.0, samplerate=sr)),
        ("Pulse width on a sample", lambda m: m.Pulse(375.0, 0.8, 0.0, 0.25, 0.0, samplerate=sr)),
        ("Envelope ADSR", lambda m: m.EnvelopeFilter(m.Sine(440.0, samplerate=sr), 0.01, 0.02, 0.03, 0.6, 0.02)),
        ("Envelope stop_at_end", lambda m: m.EnvelopeFilter(m.Square(440.0, samplerate=sr), 0.005, 0.0, 0.01, 0.5, 0.01, stop_at_end=True)),
        ("Envelope no release", lambda m: m.EnvelopeFilter(m.Sawtooth(440.0, samplerate=sr), 0.0, 0.01, 0.02, 0.7, 0.0)),
        # (SURVEY 8(a) row a8 recalls `cycle=False` in the signature: if the real class has no such keyword this case reports it)
        ("Envelope cycle", lambda m: m.EnvelopeFilter(m.Sine(440.0, samplerate=sr), 0.005, 0.005, 0.01, 0.5, 0.01, cycle=True)),
        ("MixingFilter", lambda m: m.MixingFilter(m.Sine(440.0, 0.4, samplerate=sr), m.Square(220.0, 0.3, samplerate=sr), m.Triangle(110.0, 0.2, samplerate=sr))),
        ("AmpModulationFilter", lambda m: m.AmpModulationFilter(m.Sine(440.0, samplerate=sr), m.Sine(3.0, 0.5, bias=0.5, samplerate=sr))),
        ("ClipFilter", lambda m: m.ClipFilter(m.Sine(440.0, samplerate=sr), -0.3, 0.6)),
        ("AbsFilter", lambda m: m.AbsFilter(m.Sawtooth(440.0, samplerate=sr))),
        ("NullFilter", lambda m: m.NullFilter(m.Sine(440.0, samplerate=sr))),
        ("DelayFilter +", lambda m: m.DelayFilter(m.Sine(440.0, samplerate=sr), 0.01)),
        ("DelayFilter -", lambda m: m.DelayFilter(m.Sine(440.0, samplerate=sr), -0.01)),
        ("EchoFilter", lambda m: m.EchoFilter(m.EnvelopeFilter(m.Sine(440.0, samplerate=sr), 0.01, 0.01, 0.02, 0.6, 0.01, stop_at_end=True), 0.02, 3, 0.015, 0.6)),
    ]
    return cases


def variant_flips(O):
    """Every single alternative reading of oracle.synth_oracle.VARIANTS, then every pair of them: [{name: value, ...}, ...]."""
    singles = [{k: v} for k, choices in O.VARIANT_CHOICES.items() for v in choices[1:]]
    pairs = [dict(a, **b) for a, b in itertools.combinations(singles, 2) if set(a) != set(b)]
    return singles + pairs


def explain_by_variants(make, ref, O, windows, quantise_block=None):
    """A case whose values differ: re-run the ORACLE side under every alternative reading (and pair of readings) and return those
    under which it equals the reference -- "which variant" instead of "different".  make(module) builds the case; windows =
    [(skip, count)].  quantise_block: instead of an oscillator case, (block, width) through Sample.from_osc_block / O.quantise."""
    import numpy as np
    matching = []
    for flip in variant_flips(O):
        old = O.set_variants(**flip)
        try:
            ok = True
            if quantise_block is not None:
                block, width, want = quantise_block
                ok = want == list(O.quantise(block, width))
            else:
                a, b = make(ref), make(O)
                for skip, cnt in windows:
                    x, y = np.array(take(a, cnt, skip), dtype=np.float64), np.array(take(b, cnt, skip), dtype=np.float64)
                    if x.shape != y.shape or not np.array_equal(x, y):
                        ok = False
                        break
        except Exception:
            ok = False
        finally:
            O.set_variants(**old)
        if ok:
            matching.append(flip)
    return matching


def diff_oscillators(ref, O, quick, report, very_late=False, variants=False):
    import numpy as np
    n = 2048
    late = (1 << 20) - 1024                     # a window that straddles sample 2**20
    for name, make in oscillator_cases(quick):
        try:
            a, b = make(ref), make(O)
        except Exception as e:                  # a class or keyword the real package does not have: that IS a finding
            report.append({"case": name, "error": "constructor: %r" % (e,)})
            continue
        windows = [(0, n)]
        if not quick and ("fm" not in name or "Sine f=440" in name) and "Envelope" not in name and "Echo" not in name:
            windows.append((late, n))
        # FM far into a note (2^22 samples: 87 s at 48 kHz): where a restatement that sums the LFO along the ideal lines a + j d and
        # j inc -- not over the LFO's own accumulated phase and the accumulated time steps -- has drifted by 1e-6 .. 1e-4 (a t^2 law:
        # round 4 found the product's closed form there).  Minutes of pure-Python generator per case: only when asked for
        if very_late and name.endswith(" fm") and ("f=440 " in name or "f=1000 " in name) and "sr=48000 ph=0.3" in name:
            windows.append(((1 << 22) - 1024, n))
        failed = []
        for skip, cnt in windows:
            try:
                x, y = np.array(take(a, cnt, skip), dtype=np.float64), np.array(take(b, cnt, skip), dtype=np.float64)
            except Exception as e:
                report.append({"case": name, "window": skip, "error": "blocks(): %r" % (e,)})
                continue
            if x.shape != y.shape:
                report.append({"case": name, "window": skip, "len_ref": int(x.size), "len_oracle": int(y.size)})
                failed.append(report[-1])
            elif not np.array_equal(x, y):
                d = np.abs(x - y)
                report.append({"case": name, "window": skip, "differing": int(np.sum(x != y)), "max_abs": float(d.max()),
                               "first": int(np.argmax(x != y))})
                failed.append(report[-1])
        if variants and failed:
            # which reading of the recalled arithmetic makes the oracle EQUAL the reference on every window of this case?
            match = explain_by_variants(make, ref, O, windows)
            for row in failed:
                row["variants_that_match"] = match


def diff_samples(refS, RefSample, O, report, variants=False):
    import numpy as np
    rng = np.random.default_rng(2024)

    def pair(width, nch, nframes, rate=22050):
        info = {1: np.int8, 2: np.int16, 4: np.int32}[width]
        ii = np.iinfo(info)
        raw = (rng.integers(ii.min, ii.max + 1, nframes * nch, dtype=np.int64) * 0.4).astype(info).tobytes()
        return refS.from_raw_frames(raw, width, rate, nch), RefSample(raw, width, rate, nch)

    def frames_of(s):
        return bytes(s.view_frame_data()) if hasattr(s, "view_frame_data") else s.frames

    ops = [
        ("amplify 0.5", lambda s: s.amplify(0.5)), ("amplify 1.7", lambda s: s.amplify(1.7)),
        ("fadeout", lambda s: s.fadeout(0.05, 0.1)), ("fadein", lambda s: s.fadein(0.05, 0.2)),
        ("resample 48000", lambda s: s.resample(48000)), ("resample 8000", lambda s: s.resample(8000)),
        ("add_silence", lambda s: s.add_silence(0.01)), ("add_silence start", lambda s: s.add_silence(0.01, True)),
        ("clip", lambda s: s.clip(0.01, 0.05)), ("delay", lambda s: s.delay(0.02)), ("delay keep", lambda s: s.delay(0.02, True)),
        ("delay negative", lambda s: s.delay(-0.02)), ("speed 1.26", lambda s: s.speed(1.26)), ("at_volume", lambda s: s.at_volume(0.7)),
        ("echo", lambda s: s.echo(0.1, 3, 0.05, 0.6)), ("envelope", lambda s: s.envelope(0.02, 0.02, 0.5, 0.03)),
    ]
    for width, nch in ((2, 1), (2, 2), (4, 2), (1, 1)):
        for name, op in ops:
            a, b = pair(width, nch, 4000)
            try:
                op(a), op(b)
                if frames_of(a) != frames_of(b):
                    report.append({"case": "Sample.%s w%d ch%d" % (name, width, nch), "bytes_differ": True,
                                   "len_ref": len(frames_of(a)), "len_oracle": len(frames_of(b))})
            except Exception as e:
                report.append({"case": "Sample.%s w%d ch%d" % (name, width, nch), "error": repr(e)})
    for other_at in (None, 0.03):
        a, b = pair(2, 2, 4000)
        c, d = pair(2, 2, 2500)
        if other_at is None:
            a.mix(c), b.mix(d)
        else:
            a.mix_at(other_at, c), b.mix_at(other_at, d)
        if frames_of(a) != frames_of(b):
            report.append({"case": "Sample.mix at=%r" % (other_at,), "bytes_differ": True})
    # the quantiser
    block = [0.9999 * __import__("math").sin(0.01 * i) for i in range(3000)] + [1.0, -1.0, 0.0, 1234.0 / 32767.0]
    for width in (2, 4):
        want = list(refS.from_osc_block(block, 22050, samplewidth=width).get_frame_array())
        if want != list(O.quantise(block, width)):
            report.append({"case": "from_osc_block width %d" % width, "differs": True})
            if variants:
                report[-1]["variants_that_match"] = explain_by_variants(None, None, O, None, quantise_block=(block, width, want))


def variant_verdict(report):
    """What the differing cases say together: the readings that appear in EVERY differing case's list of matching flips (adopt these:
    oracle.synth_oracle.VARIANTS and synthesizer_amd/params.py variants), and the cases no reading explains."""
    rows = [r for r in report if "variants_that_match" in r]
    unexplained = sorted({r["case"] for r in rows if not r["variants_that_match"]})
    votes = {}
    for r in rows:
        for flip in r["variants_that_match"]:
            if len(flip) == 1:
                (k, v), = flip.items()
                votes.setdefault((k, v), set()).add(r["case"])
    adopt = {k: v for (k, v), cases in votes.items()}
    summary = ("no value differences" if not rows else
               "%d differing case(s); single readings that make cases equal: %s; unexplained by any reading or pair: %s"
               % (len({r["case"] for r in rows}), ", ".join("%s=%s (%d cases)" % (k, v, len(c)) for (k, v), c in sorted(votes.items())) or "none",
                  ", ".join(unexplained) or "none"))
    return {"adopt": adopt, "unexplained": unexplained, "summary": summary}


def diff_params(refP, O, report):
    for name in ("norm_samplerate", "norm_nchannels", "norm_samplewidth", "norm_osc_blocksize"):
        if getattr(refP, name, None) != getattr(O, name):
            report.append({"case": "params.%s" % name, "ref": getattr(refP, name, None), "oracle": getattr(O, name)})


def regenerate_golden(ref):
    """tests/golden/osc_*.np*, same keys as tests/golden/make_golden.py::osc_vectors, from the real package."""
    import numpy as np
    out_dir = ROOT / "tests" / "golden"
    sine = np.array(take(ref.Sine(440, samplerate=44100), 44100), dtype=np.float64)
    np.save(out_dir / "osc_sine440_44k1.npy", sine[np.r_[0:4096, 40004:44100]].copy())
    sr = 48000
    out = {"_source": np.array("synthplayer (the real package), written by tools/pin_oracle.py --regen")}
    out["saw"] = np.array(take(ref.Sawtooth(1000, 0.8, phase=0.1, bias=0.05, samplerate=sr), 2048))
    out["square"] = np.array(take(ref.Square(1000, samplerate=sr), 2048))
    out["pulse"] = np.array(take(ref.Pulse(441, pulsewidth=0.25, samplerate=sr), 2048))
    out["harm"] = np.array(take(ref.Harmonics(220, [(k, 1.0 / k) for k in range(1, 17)], 0.5, samplerate=sr), 2048))
    out["fm_sine"] = np.array(take(ref.Sine(440, fm_lfo=ref.Sine(5, 0.03, samplerate=sr), samplerate=sr), 2048))
    out["adsr"] = np.array(take(ref.EnvelopeFilter(ref.Sine(440, samplerate=sr), 0.01, 0.01, 0.01, 0.6, 0.01), 2048))
    from synthplayer.sample import Sample
    out["quant"] = np.array(Sample.from_osc_block(list(out["harm"] * 0.5), sr).get_frame_array(), dtype=np.int16)
    np.savez_compressed(out_dir / "osc_misc.npz", **out)
    return ["osc_sine440_44k1.npy", "osc_misc.npz"]


# ---- API audit: "drops in behind synthplayer.oscillators / Sample / the mixer" is a claim about SIGNATURES too ----
API_MODULES = (("synthplayer.oscillators", "synthesizer_amd.oscillators", None),
               ("synthplayer.sample", "synthesizer_amd.sample", ("Sample", "LevelMeter")),
               ("synthplayer.synth", "synthesizer_amd.synth", ("WaveSynth",)),
               ("synthplayer.playback", "synthesizer_amd.mixer", ("RealTimeMixer",)))


def _sig(fn):
    """(name, kind, default) per parameter -- names, order, kinds and defaults are what a caller's code depends on."""
    import inspect
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    out = []
    for p_ in sig.parameters.values():
        d = "<required>" if p_.default is inspect.Parameter.empty else repr(p_.default)
        out.append((p_.name, p_.kind.name, d))
    return out


def _public_members(cls):
    import inspect
    names = ["__init__"] + sorted(n for n, v in vars(cls).items() if not n.startswith("_"))
    out = {}
    for n in names:
        v = inspect.getattr_static(cls, n, None)
        if v is None:
            continue
        if isinstance(v, property):
            out[n] = ("property", None)
        elif isinstance(v, (staticmethod, classmethod)):
            out[n] = (type(v).__name__, _sig(v.__func__))
        elif callable(v):
            out[n] = ("method", _sig(v))
        else:
            out[n] = ("attribute", None)
    return out


def audit_api(ref_mod, our_mod, class_names=None, rename=None):
    """One row per public class and per public member of it in `ref_mod`: status equal / differs / missing in `our_mod`.
    `rename`: reference class name -> our class name (the oracle's RefSample stands for Sample in the self-test)."""
    import inspect
    rename = rename or {}
    rows = []
    if class_names is None:
        class_names = sorted(n for n, v in vars(ref_mod).items()
                             if inspect.isclass(v) and not n.startswith("_") and getattr(v, "__module__", None) == ref_mod.__name__)
    for cname in class_names:
        rc = getattr(ref_mod, cname, None)
        if rc is None:
            continue
        oc = getattr(our_mod, rename.get(cname, cname), None)
        if oc is None:
            rows.append({"symbol": "%s.%s" % (ref_mod.__name__, cname), "status": "missing"})
            continue
        rm, om = _public_members(rc), _public_members(oc)
        for mname, (kind, sig) in rm.items():
            sym = "%s.%s.%s" % (ref_mod.__name__, cname, mname)
            if mname not in om and not hasattr(oc, mname):
                rows.append({"symbol": sym, "status": "missing", "ref": sig})
                continue
            if mname not in om:                      # inherited on our side: take it from the MRO
                v = inspect.getattr_static(oc, mname)
                okind = "property" if isinstance(v, property) else "method"
                osig = None if isinstance(v, property) else _sig(v.__func__ if isinstance(v, (staticmethod, classmethod)) else v)
            else:
                okind, osig = om[mname]
            same = (kind == okind or {kind, okind} <= {"method", "staticmethod", "classmethod"} and kind == okind) and sig == osig
            rows.append({"symbol": sym, "status": "equal" if same else "differs", "ref": sig, "ours": osig,
                         "ref_kind": kind, "our_kind": okind})
    return rows


def print_api_table(rows):
    bad = [r for r in rows if r["status"] != "equal"]
    print("API audit: %d symbols, %d equal, %d differ, %d missing" % (len(rows), len(rows) - len(bad),
                                                                      sum(r["status"] == "differs" for r in bad), sum(r["status"] == "missing" for r in bad)))
    for r in rows:
        if r["status"] == "equal":
            continue
        print("  %-8s %s" % (r["status"], r["symbol"]))
        if r.get("ref") is not None or r.get("ours") is not None:
            print("           ref : %s" % (r.get("ref"),))
            print("           ours: %s" % (r.get("ours"),))
    return bad


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--api", action="store_true", help="audit inspect.signature of every public class / method of synthplayer.oscillators, "
                                                       "sample.Sample, synth.WaveSynth, playback.RealTimeMixer against synthesizer_amd's")
    ap.add_argument("--regen", action="store_true", help="rewrite tests/golden/osc_*.np* from the real package")
    ap.add_argument("--quick", action="store_true", help="one sample rate / phase, no late windows")
    ap.add_argument("--very-late", action="store_true", help="FM cases also in a window at sample 2^22 (minutes of generator time per case)")
    ap.add_argument("--variants", action="store_true",
                    help="for every case whose values differ, re-run the oracle under each alternative reading of oracle.synth_oracle.VARIANTS "
                         "(increment mul / div, Square int2 / mod1, Pulse < / <=, quantise int / round, envelope < / <=; singly and in pairs) and "
                         "print the readings under which it EQUALS the reference; ends with the readings that explain every such case")
    ap.add_argument("--json", default=None, help="write the outcome here as JSON")
    args = ap.parse_args()
    where = find_reference()
    outcome = {"reference": where, "status": None, "differences": []}
    if where is None:
        outcome["status"] = "reference absent"
        msg = ("reference absent: synthplayer is not importable and /root/reference holds no synthplayer/oscillators.py "
               "(README.md:1-2 only) -- oracle parity stays UNPINNED")
        print(msg)
        if args.json:
            Path(args.json).write_text(json.dumps(outcome, indent=1) + "\n")
        return 3
    if args.api:
        import importlib
        rows = []
        for ref_name, our_name, classes in API_MODULES:
            try:
                rows += audit_api(importlib.import_module(ref_name), importlib.import_module(our_name), classes)
            except ImportError as e:
                rows.append({"symbol": ref_name, "status": "missing", "ref": repr(e)})
        bad = print_api_table(rows)
        outcome["status"] = "api equal" if not bad else "api differences"
        outcome["api"] = rows
        if args.json:
            Path(args.json).write_text(json.dumps(outcome, indent=1) + "\n")
        return 0 if not bad else 1
    from oracle import synth_oracle as O
    from oracle.sample_oracle import RefSample
    import synthplayer.oscillators as ref_osc
    import synthplayer.params as ref_params
    import synthplayer.sample as ref_sample
    report = outcome["differences"]
    diff_params(ref_params, O, report)
    diff_oscillators(ref_osc, O, args.quick, report, very_late=args.very_late, variants=args.variants)
    diff_samples(ref_sample.Sample, RefSample, O, report, variants=args.variants)
    if args.variants:
        outcome["variant_verdict"] = variant_verdict(report)
        print("variants: %s" % outcome["variant_verdict"]["summary"])
    if args.regen:
        outcome["regenerated"] = regenerate_golden(ref_osc)
    outcome["status"] = "pinned" if not report else "differences"
    print("reference: %s; %d difference(s)" % (where, len(report)))
    for r in report[:200]:
        print("  ", r)
    if args.json:
        Path(args.json).write_text(json.dumps(outcome, indent=1) + "\n")
    return 0 if not report else 1


if __name__ == "__main__":
    sys.exit(main())
