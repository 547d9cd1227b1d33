"""Exact top-k retrieval with the fused HIP kernel behind the reference's ANN plug-in interface
(cornac/models/ann/recom_ann_base.py: `BaseANN.knn_query(query[B, d], k) -> (neighbors[B, k],
distances[B, k])`, "smaller distance is better", :91-98, 122-126; `rank` :100-149;
`recommend_batch` :182-235).

Any model that exposes `get_vector_measure / get_user_vectors / get_item_vectors` (the reference's
`ANNMixin`, cornac/models/recommender.py:51-81: BPR folds its item bias into an extra dimension,
recom_bpr.pyx:309-333; PMF, NMF, HPF, LightGCN, ... likewise) rides the same kernel: SURVEY.md
section 8f, rank 4.  Unlike Annoy / Faiss / HNSW / ScaNN the search is exact, so `rank` and
`recommend_batch` return exactly the head the base model's own `rank` would.
"""
import numpy as np

from .exception import ScoreException
from .models import RecommenderBase

MEASURE_L2 = "l2"
MEASURE_DOT = "dot"
MEASURE_COSINE = "cosine"


class HipExactANN(RecommenderBase):
    """Drop-in for a `BaseANN` subclass (AnnoyANN, FaissANN, HNSWLibANN, ScaNNANN) with exact search."""

    def __init__(self, model, name="HipExactANN", verbose=False, device=0):
        RecommenderBase.__init__(self, name=name, verbose=verbose, trainable=False)
        if not getattr(model, "_ann_supported", False) and not hasattr(model, "get_item_vectors"):
            raise ValueError(f"{getattr(model, 'name', model)} doesn't support ANN search")
        self.model = model
        self.hip_device = device
        self._hip = None
        if getattr(model, "is_fitted", False) and getattr(model, "train_set", None) is not None:
            RecommenderBase.fit(self, model.train_set, getattr(model, "val_set", None))

    def fit(self, train_set, val_set=None):
        RecommenderBase.fit(self, train_set, val_set)
        if not self.model.is_fitted:
            self.model.fit(train_set, val_set)
        self.build_index()
        return self

    def build_index(self):
        """recom_ann_base.py:78-89; cosine is served as a dot product of normalised vectors."""
        self.measure = self.model.get_vector_measure()
        if self.measure not in (MEASURE_DOT, MEASURE_COSINE):
            raise ValueError(f"HipExactANN serves '{MEASURE_DOT}' and '{MEASURE_COSINE}', not '{self.measure}'")
        self.user_vectors = np.array(self.model.get_user_vectors(), dtype=np.float32)
        self.item_vectors = np.array(self.model.get_item_vectors(), dtype=np.float32)
        if self.measure == MEASURE_COSINE:
            self.item_vectors /= np.maximum(np.linalg.norm(self.item_vectors, axis=1, keepdims=True), 1e-30)
        self.higher_is_better = True
        from .engine import HipScorer
        if self._hip is not None:
            self._hip.close()
        self._hip = HipScorer(self.hip_device)
        self._hip.set_factors(self.user_vectors, self.item_vectors)
        self._hip_query_is_users = True
        return self

    def _engine(self):
        if self._hip is None:
            self.build_index()
        return self._hip

    def _prepared_query(self, query):
        q = np.ascontiguousarray(query, np.float32)
        if self.measure == MEASURE_COSINE:
            q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-30)
        return q

    def knn_query(self, query, k):
        """(neighbors int64 [B, k], distances fp32 [B, k]) with distance = -score, ascending
        (recom_ann_base.py:122-126 turns them back into scores); k = -1 means every item."""
        import torch
        eng = self._engine()
        q = self._prepared_query(query)
        n_items = len(self.item_vectors)
        k = n_items if k == -1 else int(k)
        if k < 1 or k > n_items:
            raise ValueError(f"k={k} is invalid, there are {n_items} items in total.")
        eng.set_user_factors(q)
        self._hip_query_is_users = False
        eng.set_exclusion_csr(np.zeros(1, np.int64), np.zeros(0, np.int32))
        rows = np.arange(len(q), dtype=np.int32)
        if k <= 1024:                                   # fused kernel up to 32, partial ranking of dense rows beyond
            ids, scores = eng.rank_topk(rows, k)
        else:
            out_i, out_s = [], []
            step = max(1, (1 << 27) // n_items)
            for b in range(0, len(q), step):
                dense = eng.score_dense(rows[b:b + step])
                ranked, _ = eng.rank_full(rows[b:b + step], dense)
                ranked = ranked[:, :k]
                out_i.append(ranked)
                out_s.append(torch.gather(dense, 1, ranked.long()))
            ids, scores = torch.cat(out_i), torch.cat(out_s)
        return ids.cpu().numpy().astype(np.int64), -scores.cpu().numpy()

    def rank(self, user_idx, item_indices=None, k=-1, **kwargs):
        """The contract of BaseANN.rank (recom_ann_base.py:100-149): the k nearest items lead the ranking, every other
        item follows in index order with score -inf; `item_scores` is aligned with `item_indices` (or covers the known
        items).  One exact fused launch supplies the neighbours."""
        if user_idx is None or not 0 <= user_idx < len(self.user_vectors):
            raise ScoreException("Can't make score prediction for user %d" % user_idx)
        neigh, dist = self.knn_query(self.user_vectors[user_idx][None, :], k=k)
        head = neigh[0]
        n_total = self.total_items
        scores = np.full(n_total, -np.inf)
        scores[head] = -dist[0]
        is_head = np.zeros(n_total, bool)
        is_head[head] = True
        order = np.concatenate([head, np.flatnonzero(~is_head)])
        if item_indices is None:
            return order[: self.num_items], scores[: self.num_items]
        wanted = np.zeros(n_total, bool)
        wanted[np.asarray(item_indices)] = True
        return order[wanted[order]], scores[item_indices]

    def recommend(self, user_id, k=-1, remove_seen=False, train_set=None):
        return self.recommend_batch([user_id], k=k, remove_seen=remove_seen, train_set=train_set)[0]

    def recommend_batch(self, batch_users, k=-1, remove_seen=False, train_set=None):
        """The batched entry point of recom_ann_base.py:182-235 (same arguments and errors): one fused launch for the
        whole batch.  remove_seen drops the user's train ITEMS from the k neighbours (the list may shrink); the
        reference masks list POSITIONS there (:223-225), which is not reproduced."""
        rows = np.fromiter((self.uid_map.get(uid, -1) for uid in batch_users), np.int64, len(batch_users))
        if (rows < 0).any():
            raise ValueError(f"{batch_users} is unknown to the model.")
        if not -1 <= k <= self.total_items:
            raise ValueError(f"k={k} is invalid, there are {self.total_users} users in total.")
        if remove_seen and train_set is None:
            raise ValueError("train_set must be provided to remove seen items.")
        neigh, _ = self.knn_query(self.user_vectors[rows], k=k)
        raw = np.asarray(self.item_ids, dtype=object)
        out = []
        for u, row in zip(rows.tolist(), neigh):
            if remove_seen and u < train_set.csr_matrix.shape[0]:
                csr = train_set.csr_matrix
                row = row[~np.isin(row, csr.indices[csr.indptr[u]: csr.indptr[u + 1]])]
            out.append(raw[row].tolist())
        return out
