"""Turn the rocprofv3 CSVs collected on the GPU box (gpurun_out/prof_rNN/) into the committed summaries
under profiles/.  Usage: python tools/summarize_profiles.py gpurun_out/prof_r01 r01

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streaming reads (MI355X_MICROARCH.md section HBM): the corrected read bytes are 2x the raw
value for the streaming kernels here; both are shown."""
import collections
import csv
import re
import shutil
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path("profiles")
out.mkdir(exist_ok=True)


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


shutil.copy(src / "stats_kernel_stats.csv", out / ("%s_kernel_stats.csv" % tag))
lines = ["# rocprofv3 summary %s" % tag, "",
         "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-configs --min-seconds 0.25` "
         "(PMC passes: separate runs with `--pmc ...`, `--steps 30 --warmup 100 --min-seconds 0`; tools/profile_round.sh).", "",
         "## Kernel durations (kernel-trace --stats)", "",
         "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(src / "stats_kernel_stats.csv")):
    lines.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))


serial = src / "serial_kernel_stats.csv"
if serial.exists():
    shutil.copy(serial, out / ("%s_kernel_stats_serial.csv" % tag))
    lines += ["", "### k_render_lean: duration of a kernel vs time per launch", "",
              "The default run pipelines consecutive render launches over two HIP streams (CHANGELOG.md item 16): "
              "two kernels are in flight at any time, so the start-to-end duration of ONE kernel in the table above is about "
              "twice the time the stream of launches needs per launch -- the figure `bench.py` reports from HIP events over the "
              "timed region (`roofline.avg_launch_ms`) and the one that throughput follows.  With the launches serialised on one "
              "stream (`SYNTHHIP_NO_OVERLAP=1`, same command with `--no-pcm-rows --no-two-step`) the kernel-trace average IS the "
              "time per launch and agrees with that run's HIP-event figure:", "",
              "| run | kernel-trace avg us (k_render_lean) | calls | bench.py HIP events, us per launch | Msamples/s |", "|---|---|---|---|---|"]
    import json as _json

    def bench_line(path):
        try:
            last = [l for l in path.read_text().splitlines() if l.startswith("{")][-1]
            return _json.loads(last)
        except Exception:
            return None

    def render_row(csv_path):
        for r in csv.DictReader(open(csv_path)):
            if "k_render_lean" in r["Name"] or "k_bank_render" in r["Name"]:
                return float(r["AverageNs"]) / 1e3, r["Calls"]
        return float("nan"), "?"
    for label, csv_path, bpath in (("two streams (default)", src / "stats_kernel_stats.csv", src.parent / ("%s_bench.json" % src.name)),
                                   ("one stream (SYNTHHIP_NO_OVERLAP=1)", serial, src.parent / ("%s_bench_serial.json" % src.name))):
        avg, calls = render_row(csv_path)
        b = bench_line(bpath)
        lines.append("| %s | %.1f | %s | %s | %s |" % (label, avg, calls,
                                                     ("%.1f" % (b["roofline"]["avg_launch_ms"] * 1e3)) if b else "?",
                                                     ("%.0f" % b["value"]) if b else "?"))
    sb = bench_line(src.parent / ("%s_bench_serial.json" % src.name))
    if sb:
        (out / ("%s_bench_serial_under_rocprof.json" % tag)).write_text(_json.dumps(sb) + "\n")


def counters(fname, prefix=""):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    p = src / fname
    if not p.exists():
        return agg, meta
    for r in csv.DictReader(open(p)):
        k = prefix + short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    return agg, meta


lines += ["", "## HBM traffic per dispatch (PMC: FETCH_SIZE, WRITE_SIZE; separate passes)", "",
          "| kernel | FETCH_SIZE KiB (raw) | read MB (x2 gfx950 correction) | WRITE_SIZE KiB | write MB |", "|---|---|---|---|---|"]
f, _ = counters("fetch_counter_collection.csv")
w, _ = counters("write_counter_collection.csv")
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fv = f[k].get("FETCH_SIZE", [0])
    wv = w[k].get("WRITE_SIZE", [0])
    fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
    lines.append("| %s | %.0f | %.1f | %.0f | %.1f |" % (k, fa, fa * 1024 * 2 / 1e6, wa, wa * 1024 / 1e6))

for fname, title in (("sq1_counter_collection.csv", "SQ instruction mix"), ("sq2_counter_collection.csv", "SQ stalls / LDS")):
    agg, meta = counters(fname)
    if not agg:
        continue
    lines += ["", "## %s (average per dispatch)" % title, ""]
    for k in sorted(agg):
        if not k.startswith("k_"):
            continue
        lines.append("**%s** grid %s wg %s vgpr %s sgpr %s lds %s" % ((k,) + meta[k]))
        lines.append("")
        lines.append("| counter | value |")
        lines.append("|---|---|")
        for cn, v in sorted(agg[k].items()):
            lines.append("| %s | %.0f |" % (cn, sum(v) / len(v)))
        lines.append("")
# ---- the other BASELINE configs: passes of their own (bench.py --only-config), keys "configN:<kernel>" ----
import json
cfg_traffic, cfg_counters = {}, {}
for cfg in ("config2", "config3", "staggered", "mixed", "job"):
    st = src / ("%s_stats_kernel_stats.csv" % cfg)
    if not st.exists():
        continue
    shutil.copy(st, out / ("%s_%s_kernel_stats.csv" % (tag, cfg)))
    row_json = src.parent / ("%s_%s.json" % (src.name, cfg))
    row = None
    try:
        row = json.loads([l for l in row_json.read_text().splitlines() if l.startswith("{")][-1])
    except Exception:
        pass
    lines += ["", "## %s: `rocprofv3 ... -- python bench.py --only-config %s`" % (cfg, cfg), ""]
    if row:
        for name, r in row.get("configs", {}).items():
            if "us_per_block" in r:
                lines.append("bench row under rocprofv3: **%s** %.2f us per block over the ten blocks of the job (HIP events)" % (name, r["us_per_block"]))
            elif "ms_per_1s_block" in r or "ms_per_step" in r:
                lines.append("bench row under rocprofv3: **%s** %.2f us per 1-s block (HIP events), host enqueue %.1f us per block" %
                             (name, r.get("ms_per_1s_block", r.get("ms_per_step")) * 1e3, r.get("host_enqueue_us_per_block", float("nan"))))
        lines.append("")
    lines += ["| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(st)):
        lines.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    cf, _ = counters("%s_fetch_counter_collection.csv" % cfg)
    cw, _ = counters("%s_write_counter_collection.csv" % cfg)
    lines += ["", "| kernel | read MB (FETCH_SIZE x2) | write MB |", "|---|---|---|"]
    for k in sorted(set(cf) | set(cw)):
        if not k.startswith("k_"):
            continue
        fv, wv = cf[k].get("FETCH_SIZE", [0]), cw[k].get("WRITE_SIZE", [0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        lines.append("| %s | %.2f | %.2f |" % (k, fa * 1024 * 2 / 1e6, wa * 1024 / 1e6))
        cfg_traffic["%s:%s" % (cfg, k)] = {"fetch_size_kib_raw": fa, "write_size_kib": wa, "read_bytes_corrected_x2": fa * 1024 * 2,
                                           "write_bytes": wa * 1024, "hbm_bytes": fa * 1024 * 2 + wa * 1024}
    for fname in ("%s_sq1_counter_collection.csv" % cfg, "%s_sq2_counter_collection.csv" % cfg):
        agg, meta = counters(fname)
        for k in sorted(agg):
            if not k.startswith("k_render_"):
                continue
            cfg_counters.setdefault("%s:%s" % (cfg, k), {}).update({cn: sum(v) / len(v) for cn, v in agg[k].items()})
    for k in sorted(cfg_counters):
        if k.startswith(cfg + ":"):
            lines += ["", "**%s**" % k, "", "| counter | value |", "|---|---|"]
            for cn, v in sorted(cfg_counters[k].items()):
                lines.append("| %s | %.0f |" % (cn, v))
(out / ("%s_summary.md" % tag)).write_text("\n".join(lines) + "\n")
# machine-readable traffic per dispatch (bytes), read by bench.py for roofline.traffic
traffic = dict(cfg_traffic)
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fv = f[k].get("FETCH_SIZE", [0])
    wv = w[k].get("WRITE_SIZE", [0])
    fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
    traffic[k] = {"fetch_size_kib_raw": fa, "write_size_kib": wa,
                  "read_bytes_corrected_x2": fa * 1024 * 2, "write_bytes": wa * 1024,
                  "hbm_bytes": fa * 1024 * 2 + wa * 1024}
(out / ("%s_traffic.json" % tag)).write_text(json.dumps(traffic, indent=1) + "\n")
# machine-readable SQ counters per dispatch (averages), read by bench.py for the float64 lane-ops per voice-sample;
# _meta.source_hash = hash of the kernel sources the profiled library was built from (tools/profile_round.sh records it)
allc = dict(cfg_counters)
for fname in ("sq1_counter_collection.csv", "sq2_counter_collection.csv"):
    agg, _m = counters(fname)
    for k, cs in agg.items():
        if k.startswith("k_"):
            allc.setdefault(k, {}).update({cn: sum(v) / len(v) for cn, v in cs.items()})
hash_file = src / "source_hash.txt"
allc["_meta"] = {"source_hash": hash_file.read_text().strip() if hash_file.exists() else None,
                 "dispatch": "bench.py default workload: 1024 voices x 48000 frames per render dispatch"}
(out / ("%s_counters.json" % tag)).write_text(json.dumps(allc, indent=1) + "\n")
bench = src.parent / ("%s_bench.json" % src.name)
if bench.exists():
    last = [l for l in bench.read_text().splitlines() if l.startswith("{")]
    if last:
        (out / ("%s_bench_under_rocprof.json" % tag)).write_text(last[-1] + "\n")
print((out / ("%s_summary.md" % tag)).read_text())
