#!/usr/bin/env python3
"""Markdown table of the configuration-size parity figures from a test log (profiles/rNN_parity_report.jsonl, written
by tests/parity.py:report while `pytest -m gpu` runs).  usage: parity_table.py <report.jsonl>"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
seen = {}
for r in rows:  # last occurrence of every tag
    seen[r["tag"]] = r
cols = [("flipped", "flipped px"), ("render_rel_median", "render rel. median"), ("render_rel_p99", "p99"),
        ("depth_rel", "max depth rel. (agreeing px)"), ("v_viewmat", "v_viewmat vs f64 (max over seeds)"),
        ("v_viewmat_deterministic", "same, deterministic backward (no atomics)"),
        ("v_viewmat_vs_f32_oracle", "vs the oracle's f32 build"), ("v_viewmat_f32_oracle_vs_f64", "f32 oracle vs f64 (floor)")]
print("| configuration | " + " | ".join(c[1] for c in cols) + " |")
print("|---|" + "---|" * len(cols))
for tag, r in seen.items():
    if not any(k in r for k in ("v_viewmat_vs_f32_oracle",)):
        continue
    print(f"| {tag} | " + " | ".join(f"{r[k]:.1e}" if k in r else "" for k, _ in cols) + " |")
