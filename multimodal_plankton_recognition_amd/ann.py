"""Few-shot nearest-neighbour classifier on the gfx950 kernels -- drop-in for /root/reference/src/ann.py:6-34
(same class name, ``kneighbors(*X, k=...)`` / ``predict(*X, k=...)`` contract, numpy in / numpy out).

The reference builds an approximate NN-descent index "to mimic deterministic NN-search"; here the search is exact:
inner products on the exact-fp32 MFMA GEMM, k min-scans per query row, distances re-evaluated directly for the reported
neighbours, then the reference's vote (weights 1/d, zero distances -> indicator weights, weighted mode with ties to the
smallest class) in one kernel (csrc/knn.hip).  ``nndescent_args`` are accepted for compatibility; only ``metric``
('euclidean' | 'cosine') matters for an exact search.
"""
import numpy as np
import torch

from . import _native as N
from . import ops

F32 = torch.float32
_METRIC = {'euclidean': 0, 'cosine': 1}
QUERY_BLOCK = 8192            # queries per GEMM block: 8192 x gallery fp32 products (128 MB at 4096 gallery points)


def _dev(x, device):
    t = torch.as_tensor(np.ascontiguousarray(x)) if not torch.is_tensor(x) else x
    return t.to(device=device, dtype=F32).contiguous()


class ANNClassifier:

    def __init__(self, X, y, device=None, **nndescent_args):
        metric = nndescent_args.get('metric', 'euclidean')
        if metric not in _METRIC:
            raise NotImplementedError(f"ANNClassifier: metric '{metric}' (exact search supports euclidean and cosine)")
        self.metric = _METRIC[metric]
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.y_ = np.asarray(y).copy()
        self.G = _dev(X, self.device)
        self.labels = torch.as_tensor(self.y_.astype(np.int64)).to(self.device)
        self.gn = torch.empty(self.G.shape[0], dtype=F32, device=self.device)
        N.call('mpr_knn_sqnorm', self.G, self.gn, *self.G.shape)

    def _query(self, x, k):
        """-> (idx int64 [n, k], dist fp32 [n, k]) device tensors, ascending (distance, index)."""
        X = _dev(x, self.device)
        nq, D = X.shape
        ng = self.G.shape[0]
        if D != self.G.shape[1]:
            raise ValueError(f'query dimension {D} != gallery dimension {self.G.shape[1]}')
        k = int(k)
        if not 0 < k <= ng:
            raise ValueError(f'k={k} neighbours from a gallery of {ng}')
        idx = torch.empty(nq, k, dtype=torch.int64, device=self.device)
        dist = torch.empty(nq, k, dtype=F32, device=self.device)
        qn = torch.empty(nq, dtype=F32, device=self.device)
        N.call('mpr_knn_sqnorm', X, qn, nq, D)
        for s in range(0, nq, QUERY_BLOCK):
            e = min(nq, s + QUERY_BLOCK)
            dots = ops.gemm(X[s:e], self.G, trans_b=True)
            N.call('mpr_knn_select', dots, qn[s:e], self.gn, X[s:e], self.G, self.metric, e - s, ng, k, D, idx[s:e], dist[s:e])
        return idx, dist

    def kneighbors(self, *X, k=10, epsilon=0.1, **unused):
        """src/ann.py:15-16: one (indices, distances) pair per query modality (numpy)."""
        return tuple(tuple(t.cpu().numpy() for t in self._query(x, k)) for x in X)

    def predict(self, *X, k=10, epsilon=0.1, **unused):
        """src/ann.py:19-25: neighbours of every modality hstacked, inverse-distance weighted mode of their classes."""
        parts = [self._query(x, k) for x in X]
        idx = torch.cat([p[0] for p in parts], 1).contiguous()
        dist = torch.cat([p[1] for p in parts], 1).contiguous()
        pred = torch.empty(idx.shape[0], dtype=torch.int64, device=self.device)
        N.call('mpr_knn_vote', idx, dist, self.labels, idx.shape[0], idx.shape[1], pred)
        return pred.cpu().numpy().astype(int).ravel()
