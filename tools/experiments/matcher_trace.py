"""Which launch of the LightGlue forward first leaves the majority result?  

The keyframe step at the bench size (64 synthetic stereo pairs) with airfe_debug_trace on: every launch of the matcher is followed by a
checksum of what it wrote, in units of 16 token rows.  Phase 1 runs N steps and, for every step whose per-slot digests differ from the
reference (the result two of the first three steps agree on), prints the FIRST slot that differs and the units in it.  Phase 2 repeats the
forward up to the slot named most often (airfe_debug_trace_stop), so that the buffer can be read as that launch left it, and prints the
elements that differ from the reference content.

    python tools/experiments/matcher_trace.py [N1=600] [N2=600] [mode=stereo|match]      (AIRFE_OVERLAP_LINES=1 raises the fault rate)
"""
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from airslam_amd import api, synth, weights
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)

N1 = int(sys.argv[1]) if len(sys.argv) > 1 else 600
N2 = int(sys.argv[2]) if len(sys.argv) > 2 else 600
MODE = sys.argv[3] if len(sys.argv) > 3 else "stereo"
B, K, H = 64, 400, 4
dev = torch.device("cuda", 0)
ctx = api.Context(tuning=tuning_from_env(), superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", lightglue=weights.synthetic_lightglue(1234),
                  max_batch=B, enc_chunk=64, max_keypoints=K)
NP = ctx.np_rows
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
o = dict(fl=z(B, K, 259), fr=z(B, K, 259), nl=z(B, dt=torch.int32), nr=z(B, dt=torch.int32), lines=z(2 * B, 1024, 4, dt=torch.float64),
         nlines=z(2 * B, dt=torch.int32), junc=z(B, 1024, 259), njunc=z(B, dt=torch.int32), idx=z(B, K, 2, dt=torch.int32), sc=z(B, K),
         nm=z(B, dt=torch.int32), found=z(3 * B, dt=torch.int32))


def step():
    if MODE == "stereo":
        ctx.stereo_plnet_batch_dev(L, R, o["fl"], o["fr"], o["nl"], o["nr"], o["lines"], o["nlines"], o["junc"], o["njunc"], o["idx"], o["sc"], o["nm"], o["found"])
    else:  # the matcher alone on the features of the first step
        ctx.match_lightglue_batch_dev(o["fl"], o["nl"], o["fr"], o["nr"], o["idx"], o["sc"], o["nm"])


def where(name, u, uw):
    """unit index -> text, by the buffer's layout"""
    kind = name.rsplit(".", 1)[1]
    if kind in ("x32", "xb", "o", "md", "msg"):
        r = 16 * u
        return "seq %d rows %d-%d" % (r // NP, r % NP, r % NP + 15)
    if kind in ("q", "k"):
        rb = NP // 16
        sh, b = divmod(u, rb)
        return "seq %d head %d rows %d-%d" % (sh // H, sh % H, 16 * b, 16 * b + 15)
    if kind == "vt":
        sh, d = divmod(u, 64)
        return "seq %d head %d d %d" % (sh // H, sh % H, d)
    if kind == "z":
        r = 16 * u
        return "seq %d rows %d-%d" % (r // NP, r % NP, r % NP + 15)
    if kind == "sim":
        p, b = divmod(u, NP // 16)
        return "pair %d rows %d-%d" % (p, 16 * b, 16 * b + 15)
    return "pair %d" % u


if MODE != "stereo":
    MODE, keep = "stereo", MODE
    step(); ctx.sync()
    MODE = keep
ctx.trace(True)


def reference(nread):
    """the result two of three runs agree on: (digests, table)"""
    runs = []
    for _ in range(3):
        step()
        runs.append(ctx.trace_read(table=True))
    for i in range(3):
        for j in range(i + 1, 3):
            if np.array_equal(runs[i][1], runs[j][1]):
                return runs[i]
    raise SystemExit("no two of the first three runs agree")


ref_dig, ref_tab = reference(3)
slots = ctx.trace_slots()
print("%d slots, %d units per step; overlap_lines=%s" % (len(slots), len(ref_tab), os.environ.get("AIRFE_OVERLAP_LINES", "0")), flush=True)
first = collections.Counter()
ndev = 0
for i in range(N1):
    step()
    dig, _ = ctx.trace_read()
    if np.array_equal(dig, ref_dig):
        continue
    ndev += 1
    _, tab = ctx.trace_read(table=True)
    bad = np.nonzero(dig != ref_dig)[0]
    s0 = int(bad[0])
    name, off, units, uw = slots[s0]
    du = np.nonzero(tab[off:off + units] != ref_tab[off:off + units])[0]
    first[name] += 1
    if ndev <= 12:
        print("step %d: %d of %d slots differ, first = #%d %s: %d units: %s" % (
            i, len(bad), len(slots), s0, name, len(du), "; ".join(where(name, int(u), uw) for u in du[:6])), flush=True)
print("phase 1: %d of %d steps outside the reference; first deviating slot: %s" % (ndev, N1, dict(first) or "none"), flush=True)
if not first or N2 <= 0:
    sys.exit(0)

# ---- phase 2: element-level differences in the buffer the most frequent slot covers
kinds = collections.Counter()
for n, c in first.items():
    kinds[n.split(".", 1)[1]] += c            # "self.attn.o" etc. whatever the layer
kind = kinds.most_common(1)[0][0]
target = next(i for i, s in enumerate(slots) if s[0].split(".", 1)[1] == kind and first.get(s[0], 0) > 0)
name, off, units, uw = slots[target]
dt = {"x32": np.float32, "z": np.float32, "sim": np.float32, "rowlse": np.float32, "collse": np.float32, "rowval": np.float32,
      "rowarg": np.int32, "colarg": np.int32}.get(name.rsplit(".", 1)[1], np.float16)
print("phase 2: stopping behind slot #%d %s (%s)" % (target, name, np.dtype(dt).name), flush=True)
ctx.trace_stop(target)
bufs = []
for _ in range(3):
    step(); ctx.sync()
    bufs.append(ctx.trace_buffer(target, dt))
refbuf = next((bufs[i] for i in range(3) for j in range(i + 1, 3) if np.array_equal(bufs[i].view(np.uint8), bufs[j].view(np.uint8))), None)
if refbuf is None:
    raise SystemExit("phase 2: no two of three runs agree")
step()
d2, _ = ctx.trace_read()
ref2 = d2.copy()            # digests up to the stop slot (assumed good if equal to phase 1's prefix)
if not np.array_equal(ref2[:target + 1], ref_dig[:target + 1]):
    print("  (first stopped run is itself outside the reference)")
    ref2[:target + 1] = ref_dig[:target + 1]
nd2 = 0
for i in range(N2):
    step()
    dig, _ = ctx.trace_read()
    if np.array_equal(dig[:target + 1], ref2[:target + 1]):
        continue
    nd2 += 1
    bad = np.nonzero(dig[:target + 1] != ref2[:target + 1])[0]
    if int(bad[0]) != target:
        print("  step %d: deviates earlier, at #%d %s" % (i, int(bad[0]), slots[int(bad[0])][0]))
        continue
    buf = ctx.trace_buffer(target, dt)
    ne = np.nonzero(buf.view(np.uint16 if dt == np.float16 else np.uint32) != refbuf.view(np.uint16 if dt == np.float16 else np.uint32))[0]
    if nd2 <= 10:
        per_unit = uw * 4 // np.dtype(dt).itemsize
        us = np.unique(ne // per_unit)
        a, b = refbuf[ne].astype(np.float64), buf[ne].astype(np.float64)
        print("  step %d: %d elements differ in %d units (%s); flat index %d..%d; max |diff| %.3e, max |ref| %.3e; nonfinite new %d" % (
            i, len(ne), len(us), "; ".join(where(name, int(u), uw) for u in us[:4]), int(ne.min()), int(ne.max()), float(np.abs(a - b).max()),
            float(np.abs(a).max()), int((~np.isfinite(b)).sum())))
        cols = 256 if name.rsplit(".", 1)[1] in ("x32", "xb", "o", "md", "msg") else (64 if name.rsplit(".", 1)[1] in ("q", "k") else NP)
        rows = collections.Counter((ne // cols).tolist())
        print("     rows (flat // %d): %s" % (cols, sorted(rows.items())[:24]))
        print("     cols of the first row: %s" % (ne[ne // cols == ne[0] // cols] % cols).tolist()[:64])
        print("     first values ref/new: %s" % [(float(x), float(y)) for x, y in zip(a[:6], b[:6])])
        if name.rsplit(".", 1)[1] in ("q", "k") and nd2 <= 4:
            # which arithmetic slip gives the wrong value?  un-rotate the reference pair and try the candidates
            names = [t[0] for t in ctx.trace_slots()]
            rc = ctx.trace_buffer(names.index("L0.prep.rc"), np.float32).reshape(-1, 32)
            rs = ctx.trace_buffer(names.index("L0.prep.rs"), np.float32).reshape(-1, 32)
            for fi in ne[:16].tolist():
                d = fi % 64; n = (fi // 64) % NP; sh = fi // (64 * NP); tok = (sh // H) * NP + n
                de = d & ~1
                r0, r1 = float(refbuf[fi - d + de]), float(refbuf[fi - d + de + 1])
                c, sn = float(rc[tok, de // 2]), float(rs[tok, de // 2])
                x0, x1 = r0 * c + r1 * sn, r1 * c - r0 * sn
                cm, sm = float(rc[tok, de // 2 - 1]), float(rs[tok, de // 2 - 1])
                cand = {"ref": float(refbuf[fi]), "x0": x0, "x1": x1, "x0c+x1s": x0 * c + x1 * sn, "x0c": x0 * c, "x0c-x0s": x0 * c - x0 * sn,
                        "x0c'-x1s'": x0 * cm - x1 * sm, "x0c'-x1s": x0 * cm - x1 * sn, "x0c-x1s'": x0 * c - x1 * sm, "-x1s": -x1 * sn, "x1c-x0s": x1 * c - x0 * sn}
                new = float(buf[fi])
                best = sorted(cand.items(), key=lambda kv: abs(kv[1] - new))[:3]
                print("       d %2d tok %5d: new %+.5f ref %+.5f  c %+.4f s %+.4f x0 %+.5f x1 %+.5f  x0c-new %+.5f (x1s %+.5f, x0s %+.5f) nearest: %s" % (
                    d, tok, new, cand["ref"], c, sn, x0, x1, x0 * c - new, x1 * sn, x0 * sn, ", ".join("%s %+.5f" % kv for kv in best)))
print("phase 2: %d of %d stopped runs outside the reference" % (nd2, N2))
