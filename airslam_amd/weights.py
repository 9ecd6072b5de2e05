"""Weight containers for the airfe front end.

The reference keeps its weights inside ONNX files that TensorRT turns into
``*.engine`` caches on first run (``src/plnet.cpp:24-196,587-643``,
``src/super_point.cpp:18-85``, ``src/light_glue.cpp:20-106``,
``src/super_glue.cpp:20-120``).  Our equivalent is a flat *pack* file: named
fp32 tensors in PyTorch ``state_dict`` naming and layout.  ``libairfe.so`` reads
the pack at ``airfe_create`` time and re-packs the tensors into its MFMA slab
layout (≙ engine build); nothing here knows about the device layout.

Five of the six ONNX files are absent from the reference checkout
(``.MISSING_LARGE_BLOBS``), so this module also owns the *synthetic* weight
generator used for throughput runs and HIP-vs-oracle parity (same tensors on
both sides).  Shapes follow the public SuperPoint / LightGlue / SuperGlue
definitions (marked UNVERIFIED-UPSTREAM there).

Pack file format (little endian)::

    char[8]  magic  = b"AIRFEPK1"
    u32      count
    count x { u32 name_len; char name[name_len]; u32 ndim; u32 dims[ndim];
              f32 data[prod(dims)] }
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC = b"AIRFEPK1"

Spec = List[Tuple[str, Tuple[int, ...]]]


# --------------------------------------------------------------------------- specs
def superpoint_spec() -> Spec:
    """SuperPoint v1 VGG encoder + detector/descriptor heads."""
    chans = [("conv1a", 1, 64), ("conv1b", 64, 64), ("conv2a", 64, 64), ("conv2b", 64, 64),
             ("conv3a", 64, 128), ("conv3b", 128, 128), ("conv4a", 128, 128), ("conv4b", 128, 128),
             ("convPa", 128, 256), ("convDa", 128, 256)]
    spec: Spec = []
    for name, cin, cout in chans:
        spec.append((f"{name}.weight", (cout, cin, 3, 3)))
        spec.append((f"{name}.bias", (cout,)))
    spec += [("convPb.weight", (65, 256, 1, 1)), ("convPb.bias", (65,)),
             ("convDb.weight", (256, 256, 1, 1)), ("convDb.bias", (256,))]
    return spec


def plnet_line_spec() -> Spec:
    """PLNet stage-0 LINE branch on the shared trunk (SURVEY.md Appendix A.1 contract; HAWPv3-style head, UNVERIFIED-UPSTREAM —
    plnet_s0.onnx is absent from the reference checkout): a 3x3 conv on the conv3a features and one fused 1x1 head whose 145
    output channels are loi_features (128) | md0 md1 md2 dis res | jloc0 jloc1 | joff_x joff_y | thin0..3 | aux0..3."""
    return [("line.conv1.weight", (128, 128, 3, 3)), ("line.conv1.bias", (128,)),
            ("line.head.weight", (145, 128)), ("line.head.bias", (145,))]


LG_LAYERS = 9
LG_DIM = 256
LG_HEADS = 4


def lightglue_spec(n_layers: int = LG_LAYERS) -> Spec:
    """LightGlue (SuperPoint flavour) — SURVEY.md C.2."""
    d = LG_DIM
    spec: Spec = [("posenc.Wr.weight", (32, 2))]
    for i in range(n_layers):
        s = f"transformers.{i}.self_attn"
        spec += [(f"{s}.Wqkv.weight", (3 * d, d)), (f"{s}.Wqkv.bias", (3 * d,)),
                 (f"{s}.out_proj.weight", (d, d)), (f"{s}.out_proj.bias", (d,)),
                 (f"{s}.ffn.0.weight", (2 * d, 2 * d)), (f"{s}.ffn.0.bias", (2 * d,)),
                 (f"{s}.ffn.1.weight", (2 * d,)), (f"{s}.ffn.1.bias", (2 * d,)),
                 (f"{s}.ffn.3.weight", (d, 2 * d)), (f"{s}.ffn.3.bias", (d,))]
        c = f"transformers.{i}.cross_attn"
        spec += [(f"{c}.to_qk.weight", (d, d)), (f"{c}.to_qk.bias", (d,)),
                 (f"{c}.to_v.weight", (d, d)), (f"{c}.to_v.bias", (d,)),
                 (f"{c}.to_out.weight", (d, d)), (f"{c}.to_out.bias", (d,)),
                 (f"{c}.ffn.0.weight", (2 * d, 2 * d)), (f"{c}.ffn.0.bias", (2 * d,)),
                 (f"{c}.ffn.1.weight", (2 * d,)), (f"{c}.ffn.1.bias", (2 * d,)),
                 (f"{c}.ffn.3.weight", (d, 2 * d)), (f"{c}.ffn.3.bias", (d,))]
    # only the last layer's assignment head is evaluated at inference (no early exit)
    a = f"log_assignment.{n_layers - 1}"
    spec += [(f"{a}.matchability.weight", (1, d)), (f"{a}.matchability.bias", (1,)),
             (f"{a}.final_proj.weight", (d, d)), (f"{a}.final_proj.bias", (d,))]
    return spec


SG_LAYERS = 18
SG_RES_GAIN, SG_KENC_GAIN, SG_FINAL_DIAG = 0.06, 0.1, 16.0     # structured synthetic weights (synthetic_superglue)


def superglue_spec(n_layers: int = SG_LAYERS) -> Spec:
    """SuperGlue — SURVEY.md C.3.  BatchNorm is stored folded (inference form)."""
    d = 256
    spec: Spec = []
    enc = [3, 32, 64, 128, 256, d]
    for i in range(len(enc) - 1):
        spec += [(f"kenc.encoder.{i}.weight", (enc[i + 1], enc[i])), (f"kenc.encoder.{i}.bias", (enc[i + 1],))]
    for i in range(n_layers):
        g = f"gnn.layers.{i}"
        for p in ("attn.proj.0", "attn.proj.1", "attn.proj.2", "attn.merge"):
            spec += [(f"{g}.{p}.weight", (d, d)), (f"{g}.{p}.bias", (d,))]
        spec += [(f"{g}.mlp.0.weight", (2 * d, 2 * d)), (f"{g}.mlp.0.bias", (2 * d,)),
                 (f"{g}.mlp.3.weight", (d, 2 * d)), (f"{g}.mlp.3.bias", (d,))]
    spec += [("final_proj.weight", (d, d)), ("final_proj.bias", (d,)), ("bin_score", (1,))]
    return spec


def plnet_s1_spec() -> Spec:
    """PLNet stage-1 LOI head, shapes decoded from output/plnet_s1.onnx."""
    return [("fc2.0.weight", (128, 496)), ("fc2.0.bias", (128,)),
            ("fc2.2.weight", (128, 128)), ("fc2.2.bias", (128,)),
            ("fc2.4.weight", (128, 128)), ("fc2.4.bias", (128,)),
            ("fc2_res.0.weight", (128, 240)), ("fc2_res.0.bias", (128,)),
            ("fc2_head.weight", (2, 128)), ("fc2_head.bias", (2,)),
            # the graph's [1,1,30] constant `linspace(0,1,32)[1:-1]` (torch fp32 values, NOT (j+1)/31 rounded)
            ("sample_t", (30,))]


def linspace_t() -> np.ndarray:
    """torch.linspace(0, 1, 32)[1:-1] in fp32, bit-exact (matches initializer onnx::Mul_1141 of plnet_s1.onnx)."""
    step = np.float32(np.float32(1.0) / np.float32(31))
    i = np.arange(32)
    v = np.where(i < 16, (step * i.astype(np.float32)).astype(np.float32),
                 (np.float32(1) - step * (31 - i).astype(np.float32)).astype(np.float32))
    return v[1:-1].astype(np.float32)


# ----------------------------------------------------------------------- synthetic
def _fan_in(shape: Tuple[int, ...]) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def synthetic(spec: Spec, seed: int = 1234, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded He-uniform weights, small uniform biases; LayerNorm gains near 1."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in spec:
        if name.endswith(".bias"):
            w = rng.uniform(-0.05, 0.05, size=shape)
        elif len(shape) == 1:  # LayerNorm weight / scalar parameters
            w = rng.uniform(0.9, 1.1, size=shape)
        else:
            bound = gain * np.sqrt(6.0 / _fan_in(shape))
            w = rng.uniform(-bound, bound, size=shape)
        out[name] = w.astype(np.float32)
    return out


def _whiten_descriptor_head(w: Dict[str, np.ndarray], seed: int, lam: float = 1.0, size: int = 512) -> None:
    """Data-dependent initialisation of the descriptor head (LSUV-style: Mishkin & Matas, "All you need is a good init").

    A randomly initialised VGG trunk produces dense descriptors that are almost collinear (cosine 0.9 +- 0.1 between ANY
    two cells: one common-mode direction carries the norm), so no matcher can tell keypoints apart and every end-to-end run
    ends with zero matches.  One forward pass of the seeded trunk over a seeded calibration image gives the mean and the
    covariance C of the raw `convDb` outputs; `convDb` is then composed with the regularised whitening
    A = (C + lam * mean(eig C) * I)^-1/2 and the mean is folded into its bias.  After this the descriptors of a synthetic
    stereo pair have cosine ~0 +- 0.2 between unrelated cells and ~0.9 between corresponding ones — what a trained
    SuperPoint delivers — and the SHAPES, FLOPs and kernels are unchanged.  torch is used as plumbing for the conv stack."""
    import torch
    import torch.nn.functional as Fn

    from . import synth
    # the calibration image goes the way every frame goes: a 752x480 synthetic frame resized to the internal 512x512
    img = synth.gabor_image(480, 752, seed + 4321).astype(np.float32) / np.float32(255.0)
    with torch.no_grad():
        x = Fn.interpolate(torch.from_numpy(img)[None, None], size=(size, size), mode="bilinear", align_corners=False)
        for name in ("conv1a", "conv1b", None, "conv2a", "conv2b", None, "conv3a", "conv3b", None, "conv4a", "conv4b", "convDa"):
            if name is None:
                x = Fn.max_pool2d(x, 2, 2)
            else:
                x = Fn.relu(Fn.conv2d(x, torch.from_numpy(w[name + ".weight"]), torch.from_numpy(w[name + ".bias"]), padding=1))
        d = Fn.conv2d(x, torch.from_numpy(w["convDb.weight"]), torch.from_numpy(w["convDb.bias"]))
    D = d[0].reshape(256, -1).numpy().astype(np.float64)
    mean = D.mean(1)
    ev, U = np.linalg.eigh(np.cov(D))
    ev = np.maximum(ev, 0.0)
    A = (U * (1.0 / np.sqrt(ev + lam * ev.mean()))[None, :]) @ U.T
    W = w["convDb.weight"].reshape(256, 256).astype(np.float64)
    w["convDb.weight"] = (A @ W).reshape(256, 256, 1, 1).astype(np.float32)
    w["convDb.bias"] = (A @ (w["convDb.bias"].astype(np.float64) - mean)).astype(np.float32)


# --------------------------------------------------------------------------- fp16 activation range
FP16_MAX = 65504.0
ACT_TARGET = 2048.0      # where fold_activation_scales puts the calibration maximum of a layer it has to touch: 2^11, a factor 32 below the fp16 maximum

_TRUNK = ("conv1a", "conv1b", None, "conv2a", "conv2b", None, "conv3a", "conv3b", None, "conv4a", "conv4b")


def activation_maxima(w: Dict[str, np.ndarray], images=None, size: int = 512) -> Dict[str, float]:
    """max |activation| behind every ReLU conv of the detector (trunk, convPa / convDa, line.conv1) over calibration images, fp32 on the CPU (torch as plumbing).
    images: uint8 [h, w] arrays; default = three synthetic frames.  The network's 2-byte storage (cfg.precision 0 / 1) holds these values:
    fp16 overflows to inf above 65504 (the reference builds its stage-0 engine with kTF32 — an 8-bit exponent — and the rest with kFP16, src/plnet.cpp:205,216)."""
    import torch
    import torch.nn.functional as Fn

    from . import synth
    if images is None:
        images = [synth.gabor_image(480, 752, 4321 + i) for i in range(3)]
    out: Dict[str, float] = {}
    t = lambda k: torch.from_numpy(np.ascontiguousarray(w[k], dtype=np.float32))
    with torch.no_grad():
        for img in images:
            x = Fn.interpolate(torch.from_numpy(np.asarray(img, np.float32) / np.float32(255.0))[None, None], size=(size, size), mode="bilinear", align_corners=False)
            taps = {}
            for name in _TRUNK:
                if name is None:
                    x = Fn.max_pool2d(x, 2, 2)
                    continue
                x = Fn.relu(Fn.conv2d(x, t(name + ".weight"), t(name + ".bias"), padding=1))
                taps[name] = x
                out[name] = max(out.get(name, 0.0), float(x.abs().max()))
            for name, src in (("convPa", "conv4b"), ("convDa", "conv4b"), ("line.conv1", "conv3a")):
                if name + ".weight" in w:
                    y = Fn.relu(Fn.conv2d(taps[src], t(name + ".weight"), t(name + ".bias"), padding=1))
                    out[name] = max(out.get(name, 0.0), float(y.abs().max()))
    return out


def fold_activation_scales(w: Dict[str, np.ndarray], images=None, limit: float = ACT_TARGET, target: float = ACT_TARGET):
    """Keep every 2-byte activation of the detector inside the fp16 range WITHOUT changing the function the network computes: a ReLU network is positively
    homogeneous, so layer L may hand on c_L x its activations (c_L a power of two) if the next layer's weights are divided by c_L — power-of-two factors are exact in
    fp32 and fp16 alike, max-pool commutes with them, and the factors are undone where the fp32 heads take over:
        W_L' = W_L c_L / c_(L-1),  b_L' = b_L c_L         (trunk, convPa, convDa, line.conv1)
        convPb' = convPb / c_Pa,  convDb' = convDb / c_Da,  line.head' = line.head / c_line.conv1      (biases unchanged: logits, descriptors, line maps as before)
    c_L = 2^k is chosen from calibration maxima (activation_maxima): 1 while a layer's maximum is <= `limit` (a trained network: nothing is touched), else the power
    of two that puts it at or below `target`.  -> (new pack, report {layer: (calibration max, c_L)}).  tools/onnx_to_pack.py applies it to every detector pack."""
    mx = activation_maxima(w, images)
    c: Dict[str, float] = {}
    for name, m in mx.items():
        c[name] = 1.0 if (m <= limit or not np.isfinite(m) or m <= 0) else float(2.0 ** np.floor(np.log2(target / m)))
    return rescale_activations(w, c), {k: (mx[k], c[k]) for k in mx}


def rescale_activations(w: Dict[str, np.ndarray], c: Dict[str, float]) -> Dict[str, np.ndarray]:
    """The re-parameterisation behind fold_activation_scales: layer L hands on c[L] x its activations (powers of two; a layer absent from `c` keeps factor 1),
    undone at the fp32 heads — the same function, other intermediate magnitudes.  (tests/ use it the other way round: factors ABOVE 1 make a healthy pack overflow.)"""
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in w.items()}
    prev = {"conv1a": None, "conv1b": "conv1a", "conv2a": "conv1b", "conv2b": "conv2a", "conv3a": "conv2b", "conv3b": "conv3a", "conv4a": "conv3b", "conv4b": "conv4a",
            "convPa": "conv4b", "convDa": "conv4b", "line.conv1": "conv3a"}
    f = lambda k: float(c.get(k, 1.0)) if k else 1.0
    for name, p in prev.items():
        if name + ".weight" not in out:
            continue
        out[name + ".weight"] = (out[name + ".weight"] * np.float32(f(name) / f(p))).astype(np.float32)
        out[name + ".bias"] = (out[name + ".bias"] * np.float32(f(name))).astype(np.float32)
    for head, src in (("convPb", "convPa"), ("convDb", "convDa"), ("line.head", "line.conv1")):
        if head + ".weight" in out:
            out[head + ".weight"] = (out[head + ".weight"] / np.float32(f(src))).astype(np.float32)
    return out


_SP_CACHE: Dict[Tuple[int, bool], Dict[str, np.ndarray]] = {}


def synthetic_superpoint(seed: int = 1234, structured: bool = True) -> Dict[str, np.ndarray]:
    """Seeded SuperPoint weights.  `structured` (default): detector logits shaped like a trained detector's and a
    whitened descriptor head (`_whiten_descriptor_head`), so that stereo pairs produce matchable descriptors;
    structured=False is the plain He-uniform draw of round 1 (descriptors nearly collinear)."""
    key = (seed, structured)
    if key not in _SP_CACHE:
        w = synthetic(superpoint_spec(), seed)
        # Make the heat map look like a trained detector's: strong dustbin, peaky logits.
        rng = np.random.default_rng(seed + 1)
        w["convPb.weight"] = (w["convPb.weight"] * 6.0).astype(np.float32)
        b = rng.uniform(-0.5, 0.5, size=(65,)).astype(np.float32)
        b[64] = 4.0
        w["convPb.bias"] = b
        if structured:
            _whiten_descriptor_head(w, seed)
        _SP_CACHE[key] = w
    return {k: v.copy() for k, v in _SP_CACHE[key].items()}


def synthetic_plnet_s0(seed: int = 1234) -> Dict[str, np.ndarray]:
    """A PLNet stage-0 pack: the SuperPoint-VGG point branch (`synthetic_superpoint`) + the seeded line branch."""
    w = synthetic_superpoint(seed)
    w.update(synthetic(plnet_line_spec(), seed + 40))
    # junction logits: a clear "no junction" prior so that the probability map is peaky, like a trained head's
    w["line.head.bias"][128 + 5] = 2.0
    # Structured like the matchers' weights, so that the path DOWNSTREAM of the line branch carries lines (a plain draw gives the REAL
    # stage-1 head, tests/golden/plnet_s1.airfe, features of std 0.7 it has never seen: every one of the ~1100 candidate lines scores
    # < 0.5 and 1 line survives; the head is confident around features of std <= 0.15 — tools/plnet_s0_calibrate.py):
    #   * LOI / thin / aux feature channels scaled by 0.2: 92 % of the candidates score > 0.5, 36 % pass the reference's 0.75;
    #   * md1 / md2 (the two half-angles of the HAFM decoding) biased by +1: proposals half as long again, median final line 50 px,
    #     51 % pass the 50-px length threshold.  ~240 lines per synthetic frame survive both filters (EuRoC frames: 100-300).
    for sl in (slice(0, 128), slice(137, 145)):
        w["line.head.weight"][sl] *= np.float32(0.2)
        w["line.head.bias"][sl] *= np.float32(0.2)
    w["line.head.bias"][128 + 1] = 1.0
    w["line.head.bias"][128 + 2] = 1.0
    return w


def synthetic_lightglue(seed: int = 1234, n_layers: int = LG_LAYERS, structured: bool = True) -> Dict[str, np.ndarray]:
    """Seeded LightGlue weights.  Every tensor is a He-uniform draw (gain 0.6); `structured` (default) then shapes three
    things so that the network MATCHES instead of rejecting everything (a Kaiming `final_proj` gives log-assignments of
    -6 .. -20 on every pair: 0-11 matches out of 200 planted correspondences, which left filter_matches untested):
      * ffn.3 of every block is scaled by 0.06 — residual updates of ~0.3 per block, |x| grows 1 -> ~3 over 18 blocks, the
        transformer output carries two thirds of the final state (cos(x, descriptor) ~0.33), and activations stay in the
        range trained LightGlue layers keep them in (the plain draw lets |x| reach 37 with one common direction);
      * final_proj = 12 I + draw: similarity ~ 9 <x0, x1>, a confident but finite softmax (planted pairs sit 8-12 above
        the log-sum-exp of their row / column);
      * matchability: weight / 10, bias +4  (sigmoid ~0.98, like keypoints a trained head believes in).
    Oracle (fp32) on the planted-correspondence generator of tests/test_gpu_lightglue.py: 203 matches at N = 400,
    514 at N = 1024."""
    w = synthetic(lightglue_spec(n_layers), seed + 10, gain=0.6)
    rng = np.random.default_rng(seed + 11)
    # upstream init: normal(0, gamma^-2), gamma = 1
    w["posenc.Wr.weight"] = rng.normal(0.0, 1.0, size=(32, 2)).astype(np.float32)
    if structured:
        for k in w:
            if k.endswith("ffn.3.weight"):
                w[k] = (w[k] * np.float32(0.06)).astype(np.float32)
        a = f"log_assignment.{n_layers - 1}"
        w[a + ".final_proj.weight"] = (w[a + ".final_proj.weight"] + np.float32(12.0) * np.eye(LG_DIM, dtype=np.float32)).astype(np.float32)
        w[a + ".matchability.weight"] = (w[a + ".matchability.weight"] * np.float32(0.1)).astype(np.float32)
        w[a + ".matchability.bias"] = np.array([4.0], dtype=np.float32)
    return w


def synthetic_superglue(seed: int = 1234, n_layers: int = SG_LAYERS, structured: bool = True) -> Dict[str, np.ndarray]:
    """Seeded SuperGlue weights; `structured` as for LightGlue: mlp.3 of every layer and the keypoint encoder's last layer
    scaled down (the descriptor stays the dominant part of the state), final_proj = c I + draw, low dustbin score."""
    w = synthetic(superglue_spec(n_layers), seed + 20, gain=0.6)
    w["bin_score"] = np.array([1.0], dtype=np.float32)
    if structured:
        for k in w:
            if k.endswith("mlp.3.weight"):
                w[k] = (w[k] * np.float32(SG_RES_GAIN)).astype(np.float32)
        w["kenc.encoder.4.weight"] = (w["kenc.encoder.4.weight"] * np.float32(SG_KENC_GAIN)).astype(np.float32)
        w["final_proj.weight"] = (w["final_proj.weight"] + np.float32(SG_FINAL_DIAG) * np.eye(256, dtype=np.float32)).astype(np.float32)
    return w


def _fold_pair(w1: np.ndarray, b1: np.ndarray, wo: np.ndarray, bo: np.ndarray):
    """ffn.0 over cat(x, Wo a + bo)  ==  ffn.0' over cat(x, a):  W1' = [W1x | W1m Wo], b1' = b1 + W1m bo.  Sums in float64 in the order the loader uses
    (airslam_amd/csrc/airfe_load.hip make_ffn0_folded: j ascending), rounded once to fp32 — the same bits as the library's own fold."""
    w1m = w1[:, 256:].astype(np.float64)
    acc = np.zeros((512, 256), np.float64)
    bacc = b1.astype(np.float64).copy()
    wo64, bo64 = wo.astype(np.float64), bo.astype(np.float64)
    for j in range(256):
        acc += w1m[:, j:j + 1] * wo64[j:j + 1, :]
        bacc += w1m[:, j] * bo64[j]
    return np.concatenate([w1[:, :256], acc.astype(np.float32)], 1).astype(np.float32), bacc.astype(np.float32)


def fold_out_proj(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The pack with every attention out-projection (LightGlue self_attn.out_proj / cross_attn.to_out, SuperGlue attn.merge) multiplied into the message half of
    the ffn.0 / mlp.0 that follows it, and the out-projection itself replaced by the identity: the SAME function (two linear maps with nothing between them), and
    what `airfe_tuning::fold_out_proj` (the default) does inside the library when it packs the weights.  With the identity as out-projection the message a 2-byte
    context computes IS the attention output, so `fold_out_proj = 0` on this pack and `fold_out_proj = 1` on the original must give the same bits
    (tests/test_gpu_lightglue.py) — the test that pins the loader's fold."""
    o = dict(w)
    eye, zero = np.eye(256, dtype=np.float32), np.zeros(256, np.float32)
    for k in list(w):
        for out, f0 in ((".self_attn.out_proj", ".self_attn.ffn.0"), (".cross_attn.to_out", ".cross_attn.ffn.0"), (".attn.merge", ".mlp.0")):
            if k.endswith(out + ".weight"):
                pre = k[: -len(out + ".weight")]
                o[pre + f0 + ".weight"], o[pre + f0 + ".bias"] = _fold_pair(w[pre + f0 + ".weight"].reshape(512, 512), w[pre + f0 + ".bias"],
                                                                           w[k].reshape(256, 256), w[pre + out + ".bias"])
                if out == ".attn.merge":
                    # SuperGlue's attention output has the channel order c = d * 4 + h (MultiHeadedAttention's view(dim, heads)); the library keeps it head-major
                    # (k = h * 64 + d) and permutes merge's input columns when it packs.  For the identity form to hand mlp.0 the library's own operand — the same
                    # K order, hence the same bits — the stand-in for merge is the PERMUTATION that yields the head-major vector, and mlp.0's message
                    # columns follow it: msg[j] = a[hm(j)], W1'[:, 256 + j] = (W1m Wm)[:, hm(j)] with hm(j) = (j & 63) * 4 + (j >> 6).  Still the same function.
                    j = np.arange(256)
                    hm = (j & 63) * 4 + (j >> 6)
                    f = o[pre + f0 + ".weight"]
                    o[pre + f0 + ".weight"] = np.concatenate([f[:, :256], f[:, 256:][:, hm]], 1)
                    perm = np.zeros((256, 256), np.float32)
                    perm[j, hm] = 1.0
                    o[k] = perm.reshape(w[k].shape)
                else:
                    o[k] = eye.reshape(w[k].shape).copy()
                o[pre + f0 + ".weight"] = np.ascontiguousarray(o[pre + f0 + ".weight"].reshape(w[pre + f0 + ".weight"].shape))
                o[pre + out + ".bias"] = zero.copy()
    return o


def synthetic_plnet_s1(seed: int = 1234) -> Dict[str, np.ndarray]:
    w = synthetic(plnet_s1_spec(), seed + 30)
    w["sample_t"] = linspace_t()
    return w


def synthetic_vocabulary(seed: int = 1234, k: int = 10, L: int = 4, stop_fraction: float = 0.05) -> Dict[str, np.ndarray]:
    """Stand-in for voc/point_voc_L4.bin (absent from the reference checkout): a k-ary, L-level DBoW2 tree over 256-d unit descriptors
    in breadth-first order (children of a node are consecutive), as hierarchical k-means would leave it — every child is its parent
    plus a perturbation that shrinks with depth.  Leaves carry word ids 0.. and idf-like weights; a few are 'stopped' (weight 0)."""
    rng = np.random.default_rng(seed + 50)
    n = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((n, 256), np.float32)
    first = np.zeros(n, np.int32); nch = np.zeros(n, np.int32); word = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
    level_start, nxt, words = 0, 1, 0
    for lvl in range(L + 1):
        cnt = k ** lvl
        for i in range(level_start, level_start + cnt):
            if lvl < L:
                first[i], nch[i] = nxt, k
                ch = desc[i][None] + rng.normal(size=(k, 256)).astype(np.float32) * np.float32(0.9 / (lvl + 1))
                desc[nxt:nxt + k] = ch / np.linalg.norm(ch, axis=1, keepdims=True)
                nxt += k
            else:
                word[i] = words; words += 1
                weight[i] = 0.0 if rng.uniform() < stop_fraction else float(rng.uniform(0.5, 8.0))
        level_start += cnt
    return dict(desc=desc, first_child=first, n_children=nch, word_id=word, weight=weight, k=k, L=L)


# ---------------------------------------------------------------------------- packs
def save_pack(path: str, tensors: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())


def load_pack(path: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not an airfe weight pack")
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            (nd,) = struct.unpack("<I", f.read(4))
            dims = struct.unpack(f"<{nd}I", f.read(4 * nd)) if nd else ()
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(4 * n), dtype="<f4").reshape(dims).copy()
    return out


def check_spec(tensors: Dict[str, np.ndarray], spec: Spec) -> None:
    for name, shape in spec:
        if name not in tensors:
            raise KeyError(f"weight pack is missing tensor {name!r}")
        if tuple(tensors[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tensors[name].shape} != expected {shape}")
