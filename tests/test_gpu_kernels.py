"""Kernel-level parity on the GPU: the MFMA conv / GEMM kernels behind the C ABI vs fp32 references on inputs
pre-rounded to the storage type (so only accumulation order and the output rounding differ)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from gpu_common import context, diag, to_2byte

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", [1, 0], ids=["fp16", "bf16"])
@pytest.mark.parametrize("cin,cout,pool,B,H,W", [
    (64, 64, False, 1, 32, 32), (64, 64, True, 2, 32, 48), (64, 128, False, 1, 16, 32),
    (128, 128, True, 2, 16, 16), (128, 256, False, 1, 16, 32), (128, 128, False, 3, 64, 64)])
def test_conv3x3(cin, cout, pool, B, H, W, prec):
    ctx, _, _ = context("sp", precision=prec)
    rng = np.random.default_rng(cin * 7 + cout + int(pool))
    x = to_2byte(rng.normal(size=(B, cin, H, W)).astype(np.float32), prec)
    w = to_2byte((rng.normal(size=(cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), prec)
    b = rng.normal(size=(cout,)).astype(np.float32) * 0.1
    y = ctx.debug_conv3x3(x, w, b, pool)
    ref = Fn.relu(Fn.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1))
    if pool:
        ref = Fn.max_pool2d(ref, 2, 2)
    ref = ref.numpy()
    err = np.abs(y - ref)
    tol = 2.0 ** (-10 if prec else -7) * np.abs(ref) + 1e-3          # output stored in 2 bytes: half-ulp 2^-9 (bf16) / 2^-12 (fp16) relative
    bad = err > tol
    worst = np.unravel_index(np.argmax(err - tol), err.shape)
    diag(f"conv3x3_{cin}_{cout}_{int(pool)}_{B}x{H}x{W}_{'fp16' if prec else 'bf16'}", max_err=err.max(), n_bad=int(bad.sum()), total=bad.size,
         worst=list(map(int, worst)), y_at=y[worst], ref_at=ref[worst], mean_abs_ref=np.abs(ref).mean(),
         bad_by_channel=np.nonzero(bad.sum(axis=(0, 2, 3)))[0][:32], bad_by_row=np.nonzero(bad.sum(axis=(0, 1, 3)))[0][:32],
         bad_by_col=np.nonzero(bad.sum(axis=(0, 1, 2)))[0][:32])
    assert not bad.any()


# launch_gemm picks the kernel by row count: M <= 4096 gemm_small_kernel (no LDS), M >= 16000 and M % 256 == 0 gemm8_kernel
# (kernels_gemm8.hip), otherwise gemm_kernel.  "staged" moves the thresholds so that the same shapes run through the other two.
GEMM_POLICIES = {"default": {}, "staged": {"gemm_small_max_m": 0, "gemm8_min_m": 4096}}


@pytest.mark.parametrize("prec", [1, 0], ids=["fp16", "bf16"])
@pytest.mark.parametrize("policy", list(GEMM_POLICIES))
@pytest.mark.parametrize("K,N,M,relu", [(256, 256, 128, False), (256, 65, 200, False), (512, 512, 64, True),
                                        (512, 256, 300, False), (128, 128, 128, False), (256, 768, 1000, False),
                                        (256, 512, 4096, False), (512, 256, 4352, True), (256, 65, 4096, False),
                                        (128, 320, 8192, False), (256, 512, 4224, False), (256, 512, 16384, False),
                                        (512, 256, 16640, True)])
def test_gemm(K, N, M, relu, policy, prec):
    ctx, _, _ = context("sp", tuning=GEMM_POLICIES[policy], precision=prec)
    rng = np.random.default_rng(K + N + M)
    x = to_2byte(rng.normal(size=(M, K)).astype(np.float32), prec)
    w = to_2byte((rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32), prec)
    b = rng.normal(size=(N,)).astype(np.float32)
    y = ctx.debug_gemm(x, w, b, relu)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if relu:
        ref = np.maximum(ref, 0)
    err = np.abs(y - ref)
    bad = err > 2e-4 * (1 + np.abs(ref))
    worst = np.unravel_index(np.argmax(err), err.shape)
    diag(f"gemm_{K}_{N}_{M}_{policy}_{'fp16' if prec else 'bf16'}", max_err=err.max(), n_bad=int(bad.sum()), worst=list(map(int, worst)), y_at=y[worst],
         ref_at=ref[worst], bad_cols=np.nonzero(bad.sum(0))[0][:32], bad_rows=np.nonzero(bad.sum(1))[0][:32])
    assert not bad.any()
