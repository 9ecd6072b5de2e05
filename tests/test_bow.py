"""BoW quantisation of the path's descriptors (Database::FrameToBow, src/bow/database.cc:57-89)."""
import numpy as np
import pytest

from airslam_amd import weights
from oracle import ref_post
from planted import features


def _voc_features(voc, n, seed):
    """descriptors near random leaves of the tree (what trained-vocabulary words look like to the descriptors that built them)"""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["n_children"] == 0)[0]
    f = features(n, seed)
    pick = rng.choice(leaves, size=n)
    d = voc["desc"][pick] + 0.15 * rng.normal(size=(n, 256)).astype(np.float32)
    f[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    return f


def test_oracle_descent_on_a_hand_built_tree():
    # root -> two children (e0, e1); child e0 -> two leaves (e0 + e2, e0 - e2); child e1 is a leaf
    e = np.eye(256, dtype=np.float32)
    voc = dict(desc=np.stack([0 * e[0], e[0], e[1], e[0] + e[2], e[0] - e[2]]), first_child=np.array([1, 3, 0, 0, 0], np.int32),
               n_children=np.array([2, 2, 0, 0, 0], np.int32), word_id=np.array([0, 0, 7, 3, 4], np.int32),
               weight=np.array([0, 0, 2.0, 0.0, 1.5]))
    d = np.stack([e[1], e[0] + 0.5 * e[2], e[0] - 0.5 * e[2], 0.5 * (e[0] + e[1])])
    words, w = ref_post.bow_transform(voc, d)
    # e1 -> word 7; towards +e2 -> leaf 3 whose weight is 0 -> UINT_MAX; towards -e2 -> word 4; the exact tie -> FIRST child (e0), then
    # again a tie between its leaves -> first leaf (stopped)
    assert words.tolist() == [7, 0xFFFFFFFF, 4, 0xFFFFFFFF] and w.tolist() == [2.0, 0.0, 1.5, 0.0]
    bow, wf = ref_post.frame_to_bow(words, w)
    assert bow == {4: 1.5 / 3.5, 7: 2.0 / 3.5} and wf == {4: [2], 7: [0]}


def test_synthetic_vocabulary_shape():
    voc = weights.synthetic_vocabulary(1234, k=10, L=4)
    assert voc["desc"].shape == (11111, 256) and int((voc["n_children"] == 0).sum()) == 10000
    assert sorted(voc["word_id"][voc["n_children"] == 0].tolist()) == list(range(10000))
    assert 0.02 < (voc["weight"][voc["n_children"] == 0] == 0).mean() < 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,n", [(10, 4, 400), (8, 3, 1024), (3, 2, 1)])
def test_bow_transform_vs_oracle(k, L, n):
    from airslam_amd import api
    from gpu_common import diag
    voc = weights.synthetic_vocabulary(1234, k=k, L=L)
    ctx = api.Context(superpoint=None, max_batch=2, max_keypoints=max(n, 16))
    ctx.bow_load(voc)
    f = _voc_features(voc, n, 5 * n + k)
    words, w = ctx.bow_transform(f)
    rw, rwt, margin = ref_post.bow_transform(voc, f[:, 3:], return_margin=True)
    diag(f"bow_{k}_{L}_{n}", n=n, near_ties=int((margin <= 1e-4).sum()), stopped=int((rw == 0xFFFFFFFF).sum()), distinct_words=len(set(rw.tolist())))
    # every feature, near-ties included: the kernel sums the distance in the order the reference's build does (kernels_ext.hip), the oracle is
    # pinned to that build (tests/test_ref_pin_cpu.py, family `bow`)
    np.testing.assert_array_equal(words, rw)
    np.testing.assert_array_equal(w, rwt)                           # the vocabulary's own doubles (WordValue is a double in the reference), no float round trip
    assert ref_post.frame_to_bow(words, w) == ref_post.frame_to_bow(rw, rwt)
    assert (rw == 0xFFFFFFFF).any() or n < 50                # stopped words are exercised
    ctx.close()
