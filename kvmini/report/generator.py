"""HTML report generator (``kvmini report`` / ``report_generator.py``).

Three modes, exactly one per call (reference ``report_generator.py:842-847``):
``--input results.json`` (single run: metric cards, cold/warm/overall latency
bars, cost bars, head-room classification, pre-warm break-even, recommendations,
deep link to the trace of the request nearest p95 plus a static
``traces/view.html``), ``--grid-sweep sweep.csv`` (concurrency x max_tokens
heat-maps for p95 / RPS / cost, per-pattern bars, best-of cards) and
``--partition-matrix matrix.csv`` (MI355X partition profiles; ``--mig-matrix`` is
accepted as an alias).  A positional ``results.json`` also works (the reference's
README documents that form without implementing it).  Charts are inline SVG —
no matplotlib / seaborn at report time.  Threshold rules for recommendations and
head-room are the reference's (``report_generator.py:199-245,315-395``).
"""

from __future__ import annotations

import argparse
import csv
import html
import json
import os
import sys
from typing import Any, Dict, List, Optional, Sequence, Tuple

from kvmini.utils import load_json

CSS = """
body{font-family:system-ui,-apple-system,Segoe UI,sans-serif;margin:2em auto;max-width:1100px;color:#1b1f24;padding:0 1em}
h1{border-bottom:3px solid #d6336c;padding-bottom:.3em} h2{margin-top:1.6em;color:#343a40}
.cards{display:grid;grid-template-columns:repeat(auto-fill,minmax(200px,1fr));gap:12px}
.card{border:1px solid #dee2e6;border-radius:8px;padding:12px 14px;background:#f8f9fa}
.card .v{font-size:1.5em;font-weight:600} .card .l{color:#6c757d;font-size:.85em}
table{border-collapse:collapse;margin:.6em 0} td,th{border:1px solid #ced4da;padding:5px 10px;text-align:right} th{background:#e9ecef}
td:first-child,th:first-child{text-align:left} .ok{color:#2b8a3e} .warn{color:#e67700} .bad{color:#c92a2a}
ul.recs li{margin:.35em 0} .muted{color:#868e96;font-size:.85em} svg text{font-family:inherit}
"""


def format_number(v: Any, digits: int = 2, unit: str = "") -> str:
    if v is None:
        return "N/A"
    try:
        f = float(v)
    except (TypeError, ValueError):
        return str(v)
    if f != f:
        return "N/A"
    if abs(f) >= 1000:
        s = f"{f:,.0f}"
    elif abs(f) >= 1:
        s = f"{f:.{digits}f}"
    elif f == 0:
        s = "0"
    else:
        s = f"{f:.{max(digits, 4)}f}"
    return s + unit


# ---- SVG helpers -----------------------------------------------------------------------
def svg_bar_chart(groups: Sequence[str], series: Dict[str, Sequence[Optional[float]]], title: str, unit: str = "", width: int = 640, height: int = 300) -> str:
    colors = ["#d6336c", "#1c7ed6", "#2b8a3e", "#e67700", "#7048e8"]
    vals = [v for s in series.values() for v in s if v is not None]
    vmax = max(vals) if vals else 1.0
    vmax = vmax or 1.0
    ml, mb, mt = 56, 40, 34
    pw, ph = width - ml - 12, height - mb - mt
    n_g, n_s = max(1, len(groups)), max(1, len(series))
    gw = pw / n_g
    bw = gw * 0.8 / n_s
    out = [f"<svg viewBox='0 0 {width} {height}' width='{width}' height='{height}' role='img'><text x='{width / 2}' y='18' text-anchor='middle' font-weight='600'>{html.escape(title)}</text>"]
    for i in range(5):
        y = mt + ph - ph * i / 4
        out.append(f"<line x1='{ml}' y1='{y:.1f}' x2='{ml + pw}' y2='{y:.1f}' stroke='#e9ecef'/><text x='{ml - 6}' y='{y + 4:.1f}' text-anchor='end' font-size='11'>{format_number(vmax * i / 4, 1)}</text>")
    for gi, g in enumerate(groups):
        for si, (name, data) in enumerate(series.items()):
            v = data[gi] if gi < len(data) else None
            if v is None:
                continue
            h = ph * v / vmax
            x = ml + gi * gw + gw * 0.1 + si * bw
            out.append(f"<rect x='{x:.1f}' y='{mt + ph - h:.1f}' width='{bw * 0.92:.1f}' height='{h:.1f}' fill='{colors[si % len(colors)]}'><title>{html.escape(name)} {html.escape(str(g))}: {format_number(v)}{unit}</title></rect>")
            out.append(f"<text x='{x + bw * 0.46:.1f}' y='{mt + ph - h - 3:.1f}' text-anchor='middle' font-size='10'>{format_number(v, 1)}</text>")
        out.append(f"<text x='{ml + gi * gw + gw / 2:.1f}' y='{height - mb + 16}' text-anchor='middle' font-size='12'>{html.escape(str(g))}</text>")
    for si, name in enumerate(series):
        out.append(f"<rect x='{ml + si * 130}' y='{height - 14}' width='10' height='10' fill='{colors[si % len(colors)]}'/><text x='{ml + si * 130 + 14}' y='{height - 5}' font-size='11'>{html.escape(name)}</text>")
    out.append("</svg>")
    return "".join(out)


def svg_heatmap(rows: Sequence[Any], cols: Sequence[Any], values: Dict[Tuple[Any, Any], Optional[float]], title: str, lower_is_better: bool = True, row_label: str = "", col_label: str = "") -> str:
    cw, ch, ml, mt = 92, 38, 90, 52
    width, height = ml + cw * len(cols) + 20, mt + ch * len(rows) + 30
    vals = [v for v in values.values() if v is not None]
    lo, hi = (min(vals), max(vals)) if vals else (0.0, 1.0)
    out = [f"<svg viewBox='0 0 {width} {height}' width='{width}' height='{height}' role='img'><text x='{width / 2}' y='18' text-anchor='middle' font-weight='600'>{html.escape(title)}</text>"]
    out.append(f"<text x='{ml + cw * len(cols) / 2}' y='36' text-anchor='middle' font-size='11'>{html.escape(col_label)}</text>")
    for j, c in enumerate(cols):
        out.append(f"<text x='{ml + j * cw + cw / 2}' y='{mt - 4}' text-anchor='middle' font-size='12'>{html.escape(str(c))}</text>")
    for i, r in enumerate(rows):
        out.append(f"<text x='{ml - 8}' y='{mt + i * ch + ch / 2 + 4}' text-anchor='end' font-size='12'>{html.escape(row_label)} {html.escape(str(r))}</text>")
        for j, c in enumerate(cols):
            v = values.get((r, c))
            if v is None:
                fill, label = "#f1f3f5", "–"
            else:
                t = 0.5 if hi == lo else (v - lo) / (hi - lo)
                if not lower_is_better:
                    t = 1 - t
                red, green = int(60 + 180 * t), int(200 - 120 * t)
                fill, label = f"rgb({red},{green},90)", format_number(v, 1)
            out.append(f"<rect x='{ml + j * cw}' y='{mt + i * ch}' width='{cw - 2}' height='{ch - 2}' fill='{fill}'/><text x='{ml + j * cw + cw / 2}' y='{mt + i * ch + ch / 2 + 4}' text-anchor='middle' font-size='12' fill='#fff'>{label}</text>")
    out.append("</svg>")
    return "".join(out)


def svg_scatter(points: Sequence[Tuple[Optional[float], Optional[float], str, bool]], title: str, xlabel: str, ylabel: str, width: int = 460, height: int = 320) -> str:
    """Scatter of ``(x, y, label, highlighted)``; highlighted points (frontier
    members) are drawn as labelled stars."""
    pts = [(float(x), float(y), lab, hl) for x, y, lab, hl in points if x is not None and y is not None and x == x and y == y and abs(x) != float("inf") and abs(y) != float("inf")]
    ml, mb, mt = 62, 44, 30
    pw, ph = width - ml - 16, height - mb - mt
    out = [f"<svg viewBox='0 0 {width} {height}' width='{width}' height='{height}' role='img'><text x='{width / 2}' y='18' text-anchor='middle' font-weight='600'>{html.escape(title)}</text>"]
    out.append(f"<rect x='{ml}' y='{mt}' width='{pw}' height='{ph}' fill='none' stroke='#ced4da'/>")
    out.append(f"<text x='{ml + pw / 2}' y='{height - 8}' text-anchor='middle' font-size='11'>{html.escape(xlabel)}</text>")
    out.append(f"<text x='14' y='{mt + ph / 2}' text-anchor='middle' font-size='11' transform='rotate(-90 14 {mt + ph / 2})'>{html.escape(ylabel)}</text>")
    if pts:
        xs, ys = [p[0] for p in pts], [p[1] for p in pts]
        x0, x1, y0, y1 = min(xs), max(xs), min(ys), max(ys)
        dx, dy = (x1 - x0) or abs(x1) or 1.0, (y1 - y0) or abs(y1) or 1.0
        x0, x1, y0, y1 = x0 - 0.08 * dx, x1 + 0.08 * dx, y0 - 0.08 * dy, y1 + 0.08 * dy
        for i in range(5):
            gx, gy = ml + pw * i / 4, mt + ph - ph * i / 4
            out.append(f"<text x='{gx:.1f}' y='{mt + ph + 14}' text-anchor='middle' font-size='10'>{format_number(x0 + (x1 - x0) * i / 4, 2)}</text>")
            out.append(f"<text x='{ml - 5}' y='{gy + 3:.1f}' text-anchor='end' font-size='10'>{format_number(y0 + (y1 - y0) * i / 4, 2)}</text>")
        for x, y, lab, hl in pts:
            cx, cy = ml + pw * (x - x0) / (x1 - x0), mt + ph - ph * (y - y0) / (y1 - y0)
            if hl:
                out.append(f"<text x='{cx:.1f}' y='{cy + 5:.1f}' text-anchor='middle' font-size='16' fill='#c92a2a'>★<title>{html.escape(lab)}</title></text><text x='{cx + 8:.1f}' y='{cy - 6:.1f}' font-size='10'>{html.escape(lab)}</text>")
            else:
                out.append(f"<circle cx='{cx:.1f}' cy='{cy:.1f}' r='5' fill='#1c7ed6' fill-opacity='.7'><title>{html.escape(lab)}</title></circle>")
    out.append("</svg>")
    return "".join(out)


# ---- analysis helpers --------------------------------------------------------------------
def compute_prewarm_breakeven(results: Dict[str, Any], cost_file: Optional[str] = None) -> Dict[str, Any]:
    """RPS at which keeping one replica warm costs as much as the cold starts it
    avoids (the reference's toy model: cold rate assumed proportional to 1/RPS)."""
    cold = results.get("cold_start_count") or 0
    w = results.get("window") or {}
    window_s = w.get("seconds") or ((w.get("end") or 0) - (w.get("start") or 0))
    warm_p95, cold_p95 = results.get("warm_p95_ms") or 0, results.get("cold_p95_ms") or 0
    cost_per_req = results.get("cost_per_request")
    penalty_s = max(0.0, (cold_p95 - warm_p95) / 1000.0) if cold_p95 and warm_p95 else None
    cold_rate = cold / window_s if window_s else None
    gpu_hourly = (results.get("cost_breakdown") or {}).get("gpu_hourly")
    if cost_file:
        try:
            from kvmini.cost.pricing import load_pricing

            p = load_pricing(cost_file)
            gpu_hourly = p.gpu_hourly((results.get("cost_breakdown") or {}).get("gpu_product_label"))
        except Exception:
            pass
    breakeven, notes = None, []
    if penalty_s and cold_rate and cost_per_req and gpu_hourly:
        cost_cold_per_s = cold_rate * cost_per_req
        if cost_cold_per_s > 0:
            breakeven = cost_cold_per_s * (results.get("throughput_rps") or 1.0) / (gpu_hourly / 3600.0)
            notes.append("Cold rate assumed inversely proportional to RPS (toy model).")
    else:
        if gpu_hourly is None:
            notes.append("GPU hourly cost unavailable; provide --cost-file for the pre-warm estimate.")
        if penalty_s is None:
            notes.append("Insufficient cold/warm p95 data for a penalty estimate.")
        if cost_per_req is None:
            notes.append("cost_per_request not present; run `kvmini cost`.")
    return {"penalty_seconds": penalty_s, "cold_rate_per_s": cold_rate, "gpu_hourly_cost": gpu_hourly, "breakeven_rps_estimate": breakeven, "notes": notes}


def classify_headroom(results: Dict[str, Any]) -> Dict[str, Any]:
    gpu_util = results.get("gpu_util_avg") or 0
    err = results.get("error_rate") or 0
    p95, warm = results.get("p95_ms") or 0, results.get("warm_p95_ms") or 0
    cold = results.get("cold_start_count") or 0
    rtt, s3 = results.get("network_rtt_p95_ms"), results.get("s3_avg_MBps")
    if err > 0.05:
        cls, reason, hint = "Scheduler-bound", "high_error_rate", "Elevated errors; likely queuing/throttling or timeouts."
    elif cold > 0 and warm and p95 > 2 * warm:
        cls, reason, hint = "Scheduler-bound", "cold_start_penalty", "Cold starts cause latency spikes; consider a warm pool and autoscaler settings."
    elif gpu_util < 50 and ((rtt and rtt > 300) or (s3 and s3 < 20)):
        cls, reason, hint = "I/O-bound", "network_or_storage_tail", "Network RTT high or storage slow; optimise the I/O path."
    elif gpu_util >= 80:
        cls, reason, hint = "Compute-bound", "high_gpu_util", "GPU activity high; consider batching/tuning or more GPUs."
    else:
        cls, reason, hint = "Unknown", "balanced_or_unknown", "Signals mixed; collect more traces/metrics."
    return {"classification": cls, "hint": hint, "reason": reason, "gpu_util_avg": gpu_util, "network_rtt_p95_ms": rtt, "s3_avg_MBps": s3, "error_rate": err}


def generate_recommendations(results: Dict[str, Any]) -> List[str]:
    recs: List[str] = []
    p95, err, util = results.get("p95_ms") or 0, results.get("error_rate"), results.get("gpu_util_avg")
    if p95 > 2000:
        recs.append("🔴 **P95 latency is high (>2s).** Raise the concurrency target per replica or look for resource bottlenecks.")
    elif 0 < p95 < 500:
        recs.append("✅ **Excellent P95 latency** (<500ms). The current configuration performs well.")
    if err is not None:
        if err > 0.05:
            recs.append("🔴 **High error rate** (>5%). Check timeouts, resource limits and model loading.")
        elif err < 0.01:
            recs.append("✅ **Low error rate** (<1%). Reliability is good.")
    if util is not None:
        if util < 50:
            recs.append("💡 **GPU underutilised** (<50%). Use a smaller partition (DPX/QPX/CPX) or raise batch size/concurrency.")
        elif util > 90:
            recs.append("⚠️ **GPU highly utilised** (>90%). Traffic spikes will need more capacity.")
        else:
            recs.append("✅ **Good GPU utilisation** (50-90%).")
    cold, c95, w95 = results.get("cold_start_count") or 0, results.get("cold_p95_ms"), results.get("warm_p95_ms")
    if cold > 0 and c95 and w95:
        m = c95 / w95
        if m > 3:
            recs.append(f"🔴 **Cold starts are expensive** ({m:.1f}x slower). Pre-warm a pool or lengthen the scale-to-zero grace period.")
        elif m > 1.5:
            recs.append(f"⚠️ **Moderate cold-start penalty** ({m:.1f}x slower). Check whether the traffic pattern justifies scale-to-zero.")
    c1k = results.get("cost_per_1k_tokens")
    if c1k:
        if c1k > 0.1:
            recs.append(f"💰 **High cost per 1K tokens** (${c1k:.4f}). Raise utilisation or quantise (FP8 is native on MI355X).")
        elif c1k < 0.01:
            recs.append(f"✅ **Efficient cost per 1K tokens** (${c1k:.4f}).")
    e1k = results.get("energy_wh_per_1k_tokens")
    if e1k:
        if e1k > 50:
            recs.append(f"⚡ **High energy use** ({e1k:.1f} Wh/1K tokens). Consider a power cap or larger batches.")
        elif e1k < 10:
            recs.append(f"✅ **Efficient energy use** ({e1k:.3f} Wh/1K tokens).")
    return recs


def _md_inline(s: str) -> str:
    out, bold = [], False
    for part in html.escape(s).split("**"):
        out.append(f"<b>{part}</b>" if bold else part)
        bold = not bold
    return "".join(out)


# ---- single-run report -------------------------------------------------------------------------
def _p95_trace_link(results_path: str, results: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    run_dir = os.path.dirname(os.path.abspath(results_path))
    csv_path = os.path.join(run_dir, "requests.csv")
    p95 = results.get("p95_ms")
    if not os.path.exists(csv_path) or not p95:
        return None
    best = None
    with open(csv_path, newline="") as f:
        for row in csv.DictReader(f):
            try:
                lat = float(row["latency_ms"])
            except (KeyError, ValueError):
                continue
            if row.get("trace_id") and (best is None or abs(lat - p95) < abs(best[0] - p95)):
                best = (lat, row["trace_id"], row.get("id"))
    if best is None:
        return None
    return {"latency_ms": best[0], "trace_id": best[1], "request_id": best[2]}


def write_trace_viewer(run_dir: str) -> Optional[str]:
    """Static ``traces/view.html``: pick a trace id, see its spans as a waterfall."""
    tdir = os.path.join(run_dir, "traces")
    tpath = os.path.join(tdir, "traces.json")
    if not os.path.exists(tpath):
        return None
    out = os.path.join(tdir, "view.html")
    page = """<!DOCTYPE html><html><head><meta charset='utf-8'><title>Trace viewer</title>
<style>body{font-family:system-ui,sans-serif;margin:2em}.bar{height:16px;background:#1c7ed6;border-radius:3px;min-width:2px}.row{display:grid;grid-template-columns:240px 1fr 90px;gap:8px;align-items:center;margin:3px 0;font-size:13px}.lane{background:#f1f3f5;position:relative;height:16px}.lane .bar{position:absolute;top:0}</style>
</head><body><h2>Trace viewer</h2><p>Trace: <input id='tid' size='40'> <button onclick='show()'>Show</button> <span id='n'></span></p><div id='out'></div>
<script>
let spans=[];
fetch('traces.json').then(r=>r.json()).then(d=>{spans=d.resourceSpans[0].scopeSpans[0].spans;document.getElementById('n').textContent=spans.length+' spans loaded';
const q=new URLSearchParams(location.search).get('trace');if(q){document.getElementById('tid').value=q;show();}});
function show(){const id=document.getElementById('tid').value.trim();const s=spans.filter(x=>x.traceId===id).sort((a,b)=>Number(a.startTimeUnixNano)-Number(b.startTimeUnixNano));
const o=document.getElementById('out');o.innerHTML='';if(!s.length){o.textContent='no spans for this trace id';return;}
const t0=Number(s[0].startTimeUnixNano),t1=Math.max(...s.map(x=>Number(x.endTimeUnixNano)));const span=Math.max(1,t1-t0);
for(const x of s){const a=Number(x.startTimeUnixNano)-t0,b=Number(x.endTimeUnixNano)-Number(x.startTimeUnixNano);
const r=document.createElement('div');r.className='row';r.innerHTML=`<div>${x.name}</div><div class='lane'><div class='bar' style='left:${100*a/span}%;width:${Math.max(0.3,100*b/span)}%'></div></div><div>${(b/1e6).toFixed(2)} ms</div>`;o.appendChild(r);}}
</script></body></html>"""
    with open(out, "w") as f:
        f.write(page)
    return out


def generate_single_report(results_path: str, output: str, cost_file: Optional[str] = None) -> str:
    r = load_json(results_path)
    cards = [
        ("P50 latency", format_number(r.get("p50_ms"), 1, " ms")),
        ("P95 latency", format_number(r.get("p95_ms"), 1, " ms")),
        ("P99 latency", format_number(r.get("p99_ms"), 1, " ms")),
        ("TTFT (mean)", format_number(r.get("time_to_first_token_ms"), 1, " ms")),
        ("Throughput", format_number(r.get("throughput_rps"), 2, " RPS")),
        ("Tokens / s", format_number(r.get("tokens_per_sec"), 0)),
        ("Error rate", format_number((r.get("error_rate") or 0) * 100, 2, " %")),
        ("Cost / 1K tokens", "$" + format_number(r.get("cost_per_1k_tokens"), 4)),
        ("Energy / 1K tokens", format_number(r.get("energy_wh_per_1k_tokens"), 4, " Wh")),
        ("GPU activity", format_number(r.get("gpu_util_avg"), 1, " %")),
        ("GPU power", format_number(r.get("gpu_power_watts_avg"), 0, " W")),
        ("Cold starts", format_number(r.get("cold_start_count"), 0)),
    ]
    so = r.get("structured_output") or {}
    if so.get("checked"):  # structured-output runs: answers checked against the schema they were asked for
        cards.append(("Schema-valid answers", f"{so.get('schema_valid', 0)} / {so['checked']}"))
    lat_chart = svg_bar_chart(
        ["p50", "p95", "p99"],
        {
            "overall": [r.get("p50_ms"), r.get("p95_ms"), r.get("p99_ms")],
            "warm": [r.get("warm_p50_ms"), r.get("warm_p95_ms"), r.get("warm_p99_ms")],
            "cold": [r.get("cold_p50_ms"), r.get("cold_p95_ms"), r.get("cold_p99_ms")],
        },
        "Latency (ms): overall / warm / cold",
        " ms",
    )
    cost_chart = svg_bar_chart(
        ["per request", "per 1K tokens"],
        {
            "overall": [r.get("cost_per_request"), r.get("cost_per_1k_tokens")],
            "warm": [r.get("warm_cost_per_request"), r.get("warm_cost_per_1k_tokens")],
            "cold": [r.get("cold_cost_per_request"), r.get("cold_cost_per_1k_tokens")],
        },
        "Cost (USD)",
        " $",
    )
    hist_html = ""
    for key, title in (("ttfb_histogram", "Time to first byte (ms)"), ("per_token_time_histogram", "Per-token time (ms)"), ("itl_histogram", "Inter-token latency (ms)")):
        h = r.get(key) or {}
        if h.get("counts"):
            labels = [format_number(b, 1) for b in h["bins"]]
            hist_html += svg_bar_chart(labels, {"count": h["counts"]}, title, width=900, height=240)
    head = classify_headroom(r)
    pre = compute_prewarm_breakeven(r, cost_file)
    recs = generate_recommendations(r)
    link = _p95_trace_link(results_path, r)
    viewer = write_trace_viewer(os.path.dirname(os.path.abspath(results_path)))
    trace_html = ""
    if link and viewer:
        trace_html = f"<h2>Trace deep-link</h2><p>Request #{html.escape(str(link['request_id']))} ({link['latency_ms']:.1f} ms) is the one closest to p95: <a href='traces/view.html?trace={link['trace_id']}'>open its trace</a>.</p>"
    w = r.get("window") or {}
    req = r.get("requests") or {}
    doc = [
        f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>kvmini benchmark report</title><style>{CSS}</style></head><body>",
        "<h1>LLM serving benchmark report</h1>",
        f"<p class='muted'>{req.get('success', '?')}/{req.get('total', '?')} requests succeeded over {format_number(w.get('seconds'), 1)} s · source: {html.escape(os.path.abspath(results_path))}</p>",
        "<div class='cards'>" + "".join(f"<div class='card'><div class='v'>{v}</div><div class='l'>{html.escape(l)}</div></div>" for l, v in cards) + "</div>",
        "<h2>Latency</h2>" + lat_chart,
        "<h2>Cost</h2>" + cost_chart,
        ("<h2>Token timing</h2>" + hist_html) if hist_html else "",
        f"<h2>Head-room</h2><p><b>{head['classification']}</b> — {html.escape(head['hint'])} <span class='muted'>({head['reason']})</span></p>",
        "<h2>Pre-warm break-even</h2><table><tr><th>Cold penalty (s)</th><th>Cold starts / s</th><th>GPU $/h</th><th>Break-even RPS</th></tr>"
        f"<tr><td>{format_number(pre['penalty_seconds'], 3)}</td><td>{format_number(pre['cold_rate_per_s'], 4)}</td><td>{format_number(pre['gpu_hourly_cost'], 2)}</td><td>{format_number(pre['breakeven_rps_estimate'], 2)}</td></tr></table>"
        + "".join(f"<p class='muted'>{html.escape(n)}</p>" for n in pre["notes"]),
        "<h2>Recommendations</h2><ul class='recs'>" + "".join(f"<li>{_md_inline(x)}</li>" for x in recs) + "</ul>" if recs else "",
        trace_html,
        "<p class='muted'>Generated by kvmini.</p></body></html>",
    ]
    os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
    with open(output, "w") as f:
        f.write("\n".join(doc))
    return output


# ---- sweep reports -----------------------------------------------------------------------------
def _read_csv(path: str) -> List[Dict[str, str]]:
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def _f(v: Any) -> Optional[float]:
    try:
        x = float(v)
        return x if x == x else None
    except (TypeError, ValueError):
        return None


def generate_grid_report(csv_path: str, output: str) -> str:
    rows = [r for r in _read_csv(csv_path) if _f(r.get("p95_ms")) is not None]
    concs = sorted({int(float(r["concurrency"])) for r in rows})
    toks = sorted({int(float(r["max_tokens"])) for r in rows})
    patterns = sorted({r.get("pattern", "steady") for r in rows})
    parts = [f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>Grid sweep report</title><style>{CSS}</style></head><body><h1>Grid sweep report</h1><p class='muted'>{len(rows)} successful cells from {html.escape(csv_path)}</p>"]
    if rows:
        best_p95 = min(rows, key=lambda r: _f(r["p95_ms"]) or 1e18)
        best_rps = max(rows, key=lambda r: _f(r.get("throughput_rps")) or 0)
        cost_rows = [r for r in rows if _f(r.get("cost_per_1k_tokens"))]
        cards = [
            ("Lowest p95", f"{format_number(_f(best_p95['p95_ms']), 1)} ms", f"c={best_p95['concurrency']} t={best_p95['max_tokens']} {best_p95.get('pattern', '')}"),
            ("Highest RPS", format_number(_f(best_rps.get("throughput_rps")), 2), f"c={best_rps['concurrency']} t={best_rps['max_tokens']} {best_rps.get('pattern', '')}"),
        ]
        if cost_rows:
            bc = min(cost_rows, key=lambda r: _f(r["cost_per_1k_tokens"]) or 1e18)
            cards.append(("Lowest $/1K tokens", "$" + format_number(_f(bc["cost_per_1k_tokens"]), 5), f"c={bc['concurrency']} t={bc['max_tokens']} {bc.get('pattern', '')}"))
        parts.append("<div class='cards'>" + "".join(f"<div class='card'><div class='v'>{v}</div><div class='l'>{html.escape(l)} — {html.escape(d)}</div></div>" for l, v, d in cards) + "</div>")
    for metric, title, lower in (("p95_ms", "P95 latency (ms)", True), ("throughput_rps", "Throughput (RPS)", False), ("cost_per_1k_tokens", "Cost per 1K tokens ($)", True)):
        for pat in patterns:
            vals: Dict[Tuple[Any, Any], Optional[float]] = {}
            for r in rows:
                if r.get("pattern", "steady") == pat:
                    vals[(int(float(r["concurrency"])), int(float(r["max_tokens"])))] = _f(r.get(metric))
            if any(v is not None for v in vals.values()):
                parts.append(svg_heatmap(concs, toks, vals, f"{title} — {pat}", lower, "c", "max_tokens"))
    if patterns and rows:
        def avg(pat: str, key: str) -> Optional[float]:
            xs = [_f(r.get(key)) for r in rows if r.get("pattern", "steady") == pat]
            xs = [x for x in xs if x is not None]
            return sum(xs) / len(xs) if xs else None

        parts.append("<h2>By traffic pattern</h2>" + svg_bar_chart(patterns, {"mean p95 (ms)": [avg(p, "p95_ms") for p in patterns]}, "Mean p95 by pattern") + svg_bar_chart(patterns, {"mean RPS": [avg(p, "throughput_rps") for p in patterns]}, "Mean RPS by pattern"))
    if rows:
        cols = ["concurrency", "max_tokens", "pattern", "p50_ms", "p95_ms", "p99_ms", "throughput_rps", "tokens_per_sec", "error_rate", "ttft_p95_ms", "cost_per_1k_tokens"]
        body = "".join(
            "<tr>" + "".join(f"<td>{html.escape(str(r.get(c, '')))}</td>" if c in ("concurrency", "max_tokens", "pattern") else f"<td>{format_number(_f(r.get(c)), 4)}</td>" for c in cols)
            + f"<td><a href='{html.escape(r.get('run_dir', ''))}/results.json'>{html.escape(r.get('run_dir', ''))}</a></td></tr>"
            for r in sorted(rows, key=lambda r: _f(r["p95_ms"]) or 1e18))
        parts.append("<h2>All cells (by p95)</h2><table><tr>" + "".join(f"<th>{c}</th>" for c in cols) + "<th>run</th></tr>" + body + "</table>")
    parts.append("<p class='muted'>Generated by kvmini.</p></body></html>")
    os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
    with open(output, "w") as f:
        f.write("\n".join(parts))
    return output


def generate_partition_report(csv_path: str, output: str) -> str:
    rows = _read_csv(csv_path)
    names = [r.get("profile", "?") for r in rows]
    cols = ["p50_ms", "p95_ms", "throughput_rps", "Wh_per_1k_tokens", "cost_per_1k_tokens", "error_rate"]
    parts = [f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>Partition matrix</title><style>{CSS}</style></head><body><h1>MI355X partition matrix</h1><p class='muted'>{html.escape(csv_path)}</p>"]
    parts.append(svg_bar_chart(names, {"p50 (ms)": [_f(r.get("p50_ms")) for r in rows], "p95 (ms)": [_f(r.get("p95_ms")) for r in rows]}, "Latency by partition profile"))
    parts.append(svg_bar_chart(names, {"RPS": [_f(r.get("throughput_rps")) for r in rows]}, "Throughput by partition profile"))
    parts.append(svg_bar_chart(names, {"$ / 1K tokens": [_f(r.get("cost_per_1k_tokens")) for r in rows]}, "Cost by partition profile"))
    parts.append("<table><tr><th>profile</th>" + "".join(f"<th>{c}</th>" for c in cols) + "</tr>" + "".join("<tr><td>" + html.escape(r.get("profile", "?")) + "</td>" + "".join(f"<td>{format_number(_f(r.get(c)), 4)}</td>" for c in cols) + "</tr>" for r in rows) + "</table>")
    parts.append("<p class='muted'>Generated by kvmini.</p></body></html>")
    os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
    with open(output, "w") as f:
        f.write("\n".join(parts))
    return output


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini report", description="Generate HTML reports")
    ap.add_argument("results", nargs="?", default=None, help="results.json (same as --input)")
    ap.add_argument("--input", default=None, help="results.json of a single run")
    ap.add_argument("--grid-sweep", default=None, help="grid sweep CSV")
    ap.add_argument("--partition-matrix", "--mig-matrix", dest="partition_matrix", default=None, help="partition matrix CSV")
    ap.add_argument("--output", default=None)
    ap.add_argument("--cost-file", default=None)
    a = ap.parse_args(argv)
    single = a.input or a.results
    modes = [m for m in (single, a.grid_sweep, a.partition_matrix) if m]
    if len(modes) != 1:
        print("ERROR: specify exactly one of --input, --grid-sweep, --partition-matrix", file=sys.stderr)
        return 2
    try:
        if single:
            out = generate_single_report(single, a.output or os.path.join(os.path.dirname(os.path.abspath(single)), "report.html"), a.cost_file)
        elif a.grid_sweep:
            out = generate_grid_report(a.grid_sweep, a.output or os.path.splitext(a.grid_sweep)[0] + "_report.html")
        else:
            out = generate_partition_report(a.partition_matrix, a.output or os.path.splitext(a.partition_matrix)[0] + "_report.html")
    except (OSError, ValueError, KeyError) as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    print(f"Report written: {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
