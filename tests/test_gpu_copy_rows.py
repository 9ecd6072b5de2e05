"""airfe_copy_rows_dev (include/airfe.h): the valid rows of device buffers to device or pinned host memory in one launch — counts read on the device, clamped to the
capacity, 16-byte and 4-byte aligned runs, empty jobs; against numpy slicing."""
import numpy as np
import pytest

from gpu_common import context

pytestmark = pytest.mark.gpu


def test_copy_rows_to_pinned_and_device_memory():
    import torch
    ctx, _, _ = context("sp", max_batch=2, enc_chunk=2)
    rng = np.random.default_rng(3)
    B, cap = 7, 50
    src = torch.from_numpy(rng.normal(size=(B, cap, 259)).astype(np.float32)).cuda()
    lines = torch.from_numpy(rng.normal(size=(B, cap, 4))).cuda()                      # float64 rows of 32 bytes
    odd = torch.from_numpy(rng.integers(0, 1 << 30, size=(B, cap + 1), dtype=np.int32)).cuda()
    cnt = torch.tensor([0, 1, 49, 50, 77, -3, 13], dtype=torch.int32).cuda()          # 77 > cap: clamped; -3: nothing
    want_n = [0, 1, 49, 50, 50, 0, 13]
    h_feat = torch.full((B, cap, 259), -1.0).pin_memory()
    h_lines = torch.full((B, cap, 4), -1.0, dtype=torch.float64).pin_memory()
    d_odd = torch.full((B, cap), -1, dtype=torch.int32).cuda()
    h_cnt = torch.full((B,), -9, dtype=torch.int32).pin_memory()
    jobs = [(cnt, h_cnt, None, 4, B)]
    for b in range(B):
        jobs += [(src[b], h_feat[b], cnt[b:b + 1], 1036, cap), (lines[b], h_lines[b], cnt[b:b + 1], 32, cap),
                 (odd[b, 1:], d_odd[b], cnt[b:b + 1], 4, cap)]                          # source 4-byte aligned only: the dword path
    plan = ctx.copy_rows_plan(jobs)
    st = torch.cuda.Stream()
    for _ in range(12):                                                                # (more launches than the ring has slots)
        ctx.copy_rows_dev(plan, stream=st.cuda_stream)
    st.synchronize()
    np.testing.assert_array_equal(h_cnt.numpy(), cnt.cpu().numpy())
    s, l, o = src.cpu().numpy(), lines.cpu().numpy(), odd.cpu().numpy()
    for b in range(B):
        n = want_n[b]
        np.testing.assert_array_equal(h_feat[b, :n].numpy(), s[b, :n])
        assert (h_feat[b, n:].numpy() == -1).all()
        np.testing.assert_array_equal(h_lines[b, :n].numpy(), l[b, :n])
        assert (h_lines[b, n:].numpy() == -1).all()
        np.testing.assert_array_equal(d_odd[b, :n].cpu().numpy(), o[b, 1:1 + n])
        assert (d_odd[b, n:].cpu().numpy() == -1).all()
    from airslam_amd import api
    with pytest.raises(api.AirfeError, match="multiple of 4"):
        ctx.copy_rows_dev(ctx.copy_rows_plan([(src[0], h_feat[0], None, 6, 1)]), stream=st.cuda_stream)


def test_pack_rows_back_to_back_with_device_side_offsets():
    """airfe_pack_rows_dev: the same jobs packed into ONE device block, job j at offsets[j] (16-byte aligned), offsets[n] = bytes used — the exclusive scan of the
    counts is taken on the device (700 jobs: more than one per thread of the scan's workgroup)"""
    import torch
    ctx, _, _ = context("sp", max_batch=2, enc_chunk=2)
    rng = np.random.default_rng(4)
    n, cap = 700, 9
    src = torch.from_numpy(rng.integers(0, 1 << 30, size=(n, cap, 7), dtype=np.int32)).cuda()          # rows of 28 bytes
    cnt_np = rng.integers(-1, 12, size=n).astype(np.int32)
    cnt = torch.from_numpy(cnt_np).cuda()
    jobs = [(src[j], None, cnt[j:j + 1] if j % 5 else None, 28, cap) for j in range(n)]                # every fifth job: no count = its capacity
    plan = ctx.copy_rows_plan(jobs)
    packed = torch.full((n * ((cap * 28 + 15) // 16 * 16),), 255, dtype=torch.uint8).cuda()
    off = torch.zeros((n + 1,), dtype=torch.int64).cuda()
    st = torch.cuda.Stream()
    ctx.pack_rows_dev(plan, packed, off, stream=st.cuda_stream)
    st.synchronize()
    o, p, s = off.cpu().numpy(), packed.cpu().numpy(), src.cpu().numpy()
    pos = 0
    for j in range(n):
        rows = cap if j % 5 == 0 else int(np.clip(cnt_np[j], 0, cap))
        assert o[j] == pos and pos % 16 == 0
        np.testing.assert_array_equal(p[pos:pos + rows * 28].view(np.int32), s[j, :rows].reshape(-1))
        pos += (rows * 28 + 15) // 16 * 16
    assert o[n] == pos
