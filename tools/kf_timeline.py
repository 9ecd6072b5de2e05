"""Timeline of ONE batch-1 stereo keyframe (airfe_stereo_keyframe) from a rocprofv3 kernel trace: every dispatch with its start offset, duration, the gap
to the previous dispatch on the same queue, per-kernel totals and the idle time on the critical queue.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python /root/repo/tools/kf_timeline.py run     (on an MI355X)
    python tools/kf_timeline.py report <dir>            (reads the *_kernel_trace.csv below <dir>)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    from airslam_amd import api, synth, weights
    W, H = 752, 480
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe"),
                      lightglue=weights.synthetic_lightglue(1234), max_batch=2, enc_chunk=2, max_keypoints=400, image_width=W, image_height=H,
                      precision=1, matcher_precision=1)
    left, right = synth.stereo_pair(H, W, 1000)
    for _ in range(int(os.environ.get("KF_REPS", "12"))):
        k = ctx.stereo_keyframe(left, right)
    print(len(k["idx"]), "matches", len(k["linesL"]), "lines")
    ctx.close()


def report(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    reps = int(os.environ.get("KF_REPS", "12"))
    per = len(rows) // reps
    last = rows[-per:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = {}
    tot = {}
    print(f"{per} dispatches per keyframe; last keyframe:")
    print(f"{'start us':>9s} {'dur us':>7s} {'gap us':>7s} {'queue':>5s}  kernel")
    for r in last:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")
        name = r["Kernel_Name"].split("(")[0].replace("void airfe::", "").replace("airfe::", "")[:90]
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        tot.setdefault(name, [0, 0.0])
        tot[name][0] += 1
        tot[name][1] += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f} {q:>5s}  {name}")
    span = (max(int(r["End_Timestamp"]) for r in last) - t0) / 1e3
    busy = sum(v[1] for v in tot.values())
    print(f"\nspan {span:.1f} us, sum of kernel durations {busy:.1f} us (two queues overlap)")
    for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {us:8.1f} us {n:3d} x  {name}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
