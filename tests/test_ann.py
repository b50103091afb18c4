"""Few-shot kNN classifier: the oracle's vote against hand-checked cases on CPU; the GPU classifier against
the oracle (neighbour indices and predicted classes bit-exact, distances to fp32 rounding)."""
import numpy as np
import pytest
import torch


def test_oracle_weights_and_vote_follow_the_reference_rules():
    from oracle import ann as OA
    w = OA.get_weights(np.array([[1., 2., 4.], [0., 3., 0.]], np.float32))
    np.testing.assert_allclose(w, [[1., .5, .25], [1., 0., 1.]])                 # a zero distance -> indicator row
    G = np.array([[0., 0.], [1., 0.], [0., 1.], [5., 5.]], np.float32)
    y = np.array([3, 1, 1, 7])
    # query at the origin: exact hit on class 3 outvotes two class-1 neighbours
    assert OA.predict(G, y, (np.array([[0., 0.]], np.float32),), k=3)[0] == 3
    # slightly off: 1/0.1 for class 3 vs 1/0.9 + 1/1.0.. for class 1
    assert OA.predict(G, y, (np.array([[.1, 0.]], np.float32),), k=3)[0] == 3
    # equal weights: ties go to the smallest class id
    G2 = np.array([[1., 0.], [-1., 0.]], np.float32)
    assert OA.predict(G2, np.array([9, 4]), (np.array([[0., 0.]], np.float32),), k=2)[0] == 4


@pytest.mark.gpu
@pytest.mark.parametrize('metric', ['euclidean', 'cosine'])
def test_gpu_classifier_matches_oracle(metric):
    from multimodal_plankton_recognition_amd.ann import ANNClassifier
    from oracle import ann as OA
    rng = np.random.default_rng(0)
    n_cls, shots, D = 12, 9, 64
    centers = rng.normal(size=(n_cls, D)).astype(np.float32)
    y = np.repeat(np.arange(n_cls) * 3 + 1, shots)                            # class ids need not be 0..C-1
    G = (centers[np.repeat(np.arange(n_cls), shots)] + 0.8 * rng.normal(size=(n_cls * shots, D))).astype(np.float32)
    Xi = (centers[rng.integers(0, n_cls, 301)] + 0.9 * rng.normal(size=(301, D))).astype(np.float32)
    Xp = (Xi + 0.5 * rng.normal(size=Xi.shape)).astype(np.float32)
    Xi[:5] = G[[3, 17, 40, 77, 100]]                                           # exact hits: zero distances
    clf = ANNClassifier(G, y, n_neighbors=32, metric=metric, random_state=0)
    for k in (1, 5, 16):
        (idx, dist), = clf.kneighbors(Xi, k=k, epsilon=.3)
        ridx, rdist = OA.kneighbors(G, Xi, k, metric)
        assert np.array_equal(idx, ridx), k
        np.testing.assert_allclose(dist, rdist, rtol=2e-5, atol=2e-6)
        if metric == 'euclidean':
            assert (dist[:5, 0] == 0).all()
        for Xs in ((Xi,), (Xp,), (Xi, Xp)):
            assert np.array_equal(clf.predict(*Xs, k=k, epsilon=.3), OA.predict(G, y, Xs, k, metric)), (k, len(Xs))


@pytest.mark.gpu
def test_gpu_classifier_query_blocks_and_argument_checks():
    from multimodal_plankton_recognition_amd import ann as A
    rng = np.random.default_rng(1)
    G = rng.normal(size=(50, 32)).astype(np.float32)
    y = rng.integers(0, 5, 50)
    X = rng.normal(size=(700, 32)).astype(np.float32)
    clf = A.ANNClassifier(G, y)
    ref = clf.predict(X, k=7)
    old = A.QUERY_BLOCK
    A.QUERY_BLOCK = 256                      # several GEMM blocks, ragged last one
    try:
        assert np.array_equal(clf.predict(X, k=7), ref)
    finally:
        A.QUERY_BLOCK = old
    with pytest.raises(ValueError):
        clf.predict(X, k=51)
    with pytest.raises(NotImplementedError):
        A.ANNClassifier(G, y, metric='manhattan')
