#!/usr/bin/env python3
"""Two lines about a bench.py JSON line (any workload): the headline and, when present, the stage table / sweep / roofline.
    python tools/bench_brief.py gpurun_out/<tag>/bench_<name>.json"""
import json
import sys


def main(path):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:                      # the bench died: its .err file says why
        print("  no line:", e)
        return
    rl = d.get("roofline") or {}
    cb = d.get("cpu_baseline") or {}
    print("  %.1f %s, %.3f ms per step, n_gpus %s; roofline frac %s step_frac %s; cpu %s %s" % (
        d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], rl.get("frac") and round(rl["frac"], 3), rl.get("step_frac") and round(rl["step_frac"], 3),
        cb.get("value") and round(cb["value"], 2), cb.get("parity_ok", "")))
    for extra in ("value_resident", "io", "points_only_pairs_per_s"):
        v = d.get(extra, d.get("config", {}).get(extra))
        if v is not None:
            print("  %s: %s" % (extra, json.dumps(v)[:300]))
    if d.get("stages"):
        print("  stages (ms):", {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
    for S, r in (d.get("sweep") or {}).items():
        print("  S=%s: %.1f frames/s, %.3f ms per time-step, p50 %.3f p99 %.3f %s" % (
            S, r["frames_per_s"], r["ms_per_time_step"], r["latency_ms"]["p50"], r["latency_ms"]["p99"], json.dumps(r.get("wall_split_ms_per_step"))))
    if d.get("latency_ms"):
        print("  latency:", json.dumps(d["latency_ms"])[:400])


if __name__ == "__main__":
    main(sys.argv[1])
