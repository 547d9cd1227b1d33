"""Reference-shaped model facades over the HIP engine.

`HipScoringMixin` re-implements `Recommender.score` / `Recommender.rank`
(cornac/models/recommender.py:480-498, 533-661) and adds the batched `rank_batch`, keeping the
reference's signatures, return types, exceptions and the three result caches
(`ranked_items`, `item_scores`, `item_scores_mapped_indices`, recommender.py:160-168, 655-659).
It can be mixed into the reference's own classes or used through the
stand-alone `BPR` / `MF` / `WMF` classes below, which take already-trained parameters
(`init_params=...`, like the reference's `trainable=False` usage; training is out of scope).

There is no CPU fallback: every score and every ordering comes from libifr.so.  The only host
work is bookkeeping (id maps, candidate arrays, the all-ties case of an unknown user, whose
"ranking" is the candidate list in ascending item index).
"""
import numpy as np

from .exception import ScoreException


class RecommenderBase:
    """The subset of cornac.models.Recommender (recommender.py:84-170, 362-464, 500-502, 708-756)
    that the hot path and its callers touch."""

    def __init__(self, name, trainable=True, verbose=False, **kwargs):
        self.name = name
        self.trainable = trainable
        self.verbose = verbose
        self.is_fitted = False
        self.num_users = None
        self.num_items = None
        self.uid_map = None
        self.iid_map = None
        self.max_rating = None
        self.min_rating = None
        self.global_mean = None
        self.article_pool = None
        for key, value in kwargs.items():
            setattr(self, key, value)
        self.ranked_items = {}
        self.item_scores = {}
        self.item_scores_mapped_indices = {}

    @property
    def total_users(self):
        return len(self.uid_map) if self.uid_map is not None else self.num_users

    @property
    def total_items(self):
        return len(self.iid_map) if self.iid_map is not None else self.num_items

    @property
    def user_ids(self):
        return list(self.uid_map.keys())

    @property
    def item_ids(self):
        return list(self.iid_map.keys())

    def fit(self, train_set, val_set=None):
        """recommender.py:362-402: bookkeeping only."""
        self.num_users = train_set.num_users
        self.num_items = train_set.num_items
        self.uid_map = train_set.uid_map
        self.iid_map = train_set.iid_map
        self.min_rating = train_set.min_rating
        self.max_rating = train_set.max_rating
        self.global_mean = train_set.global_mean
        self.train_set = train_set
        self.val_set = val_set
        self.is_fitted = True
        return self

    def knows_user(self, user_idx):
        return user_idx is not None and user_idx >= 0 and user_idx < self.num_users

    def knows_item(self, item_idx):
        return item_idx is not None and item_idx >= 0 and item_idx < self.num_items

    def is_unknown_user(self, user_idx):
        return not self.knows_user(user_idx)

    def is_unknown_item(self, item_idx):
        return not self.knows_item(item_idx)

    def transform(self, test_set):
        pass

    def default_score(self):
        return self.global_mean

    def save_recommendations(self, save_dir, pool_size=None):
        """recommender.py:244-274: the three caches as `recommendations.pkl`, `item_scores.pkl` and
        `item_scores_mapped_indices.pkl` under save_dir (plain dicts user index -> array).

        pool_size=None writes what the reference writes: every cached user's full ranking and full score
        vector (the lazy caches materialise every user for it).  pool_size=P writes the heads only --
        P ranked items, their scores and the matching index arrays -- which is all the re-ranking stage
        reads (rerankers/reranker.py:620-639) and stays small at any catalogue size
        (`recommenders_amd.eval.export_candidate_pools`)."""
        import os
        import pickle
        os.makedirs(save_dir, exist_ok=True)
        if pool_size is None or int(pool_size) < 0:
            parts = (dict(self.ranked_items.items()), dict(self.item_scores.items()),
                     dict(self.item_scores_mapped_indices.items()))
        else:
            from .eval import export_candidate_pools
            parts = export_candidate_pools(self, pool_size)
        names = ("recommendations.pkl", "item_scores.pkl", "item_scores_mapped_indices.pkl")
        for name, part in zip(names, parts):
            with open(os.path.join(save_dir, name), "wb") as f:
                pickle.dump(part, f)
        if self.verbose:
            print(f"Recommendations, item scores and their index maps saved to {save_dir}")
        return [os.path.join(save_dir, n) for n in names]

    def recommend(self, user_id, k=-1, remove_seen=False, train_set=None):
        """Top-k raw item ids for one raw user id: the interface (arguments, errors, returned list) of
        recommender.py:708-756, served by the batched engine.  The reference sorts the whole catalogue for every call
        (`rank` with k left at -1, :751) and cuts the head afterwards; here a head of up to 1024 items comes from ONE
        fused / partial-ranking launch over the user's row: `k + |seen|` items are fetched, the seen ones dropped on
        the host.  Longer lists, k = -1 and users the model cannot score go through `rank` like the reference's.
        The fast path does not fill `ranked_items` / `item_scores` (a k-item head is not the full ranking they hold)."""
        if user_id not in self.uid_map:
            raise ValueError(f"{user_id} is unknown to the model.")
        n_total = self.total_items
        if not -1 <= k <= n_total:
            raise ValueError(f"k={k} is invalid, there are {self.total_users} users in total.")
        u = self.uid_map[user_id]
        seen = np.zeros(0, np.int64)
        if remove_seen:
            if train_set is None:
                raise ValueError("train_set must be provided to remove seen items.")
            csr = train_set.csr_matrix
            if u < csr.shape[0]:
                seen = np.unique(csr.indices[csr.indptr[u]: csr.indptr[u + 1]])
        raw_ids = self.item_ids
        fetch = k + len(seen)
        if 0 < k and fetch <= min(1024, self.num_items) and hasattr(self, "rank_batch") and self._hip_user_row(u) >= 0:
            head = self.rank_batch([u], fetch)[0][0]
            head = head[head >= 0]
            if len(seen):
                head = head[~np.isin(head, seen)]
            if len(head) >= k:                      # (fewer: unknown items would follow -- the full path below)
                return [raw_ids[i] for i in head[:k].tolist()]
        cand = np.arange(n_total)
        if len(seen):
            cand = np.setdiff1d(cand, seen, assume_unique=True)
        order = self.rank(u, cand)[0]
        return [raw_ids[i] for i in (order if k == -1 else order[:k]).tolist()]


class HipScoringMixin:
    """score / rank / rank_batch on the HIP engine for dot-product models.

    A concrete class supplies `_hip_factor_params() -> (U, V, ubase, ibase)` (fp32 arrays:
    score(u, i) = (ibase[i] + ubase[u]) + <U[u], V[i]>) and `_hip_user_row(user_idx)`: the row of
    U to use, -1 for "the model raises ScoreException for this user", or the index of the
    all-zero row appended after U for "the user term is dropped" (MF, recom_mf.py:277-280)."""

    hip_device = 0
    score_dtype = np.float32

    # ---- engine life cycle
    def _hip_engine(self, rows=None):
        """The engine that holds the whole model, or -- rows=(r0, r1), a rank of a sharded evaluation -- one that holds
        only the user rows [r0, r1) of the user-side parameters (engine row i = user r0 + i, `eng.row0` = r0); the
        item side is replicated.  The last shard engine is kept."""
        if rows is not None:
            r0, r1 = int(rows[0]), int(rows[1])
            shard = getattr(self, "_hip_shard", None)
            if shard is not None and shard[0] == (r0, r1):
                return shard[1]
            if shard is not None:
                shard[1].close()
            from .engine import HipScorer
            eng = HipScorer(self.hip_device)
            try:
                self._hip_setup(eng, (r0, r1))
            except RuntimeError:
                # a model that was built on the device keeps its parameters in the full engine only: use that one
                # (row0 = 0; ranking_eval then indexes it with the users' own rows)
                eng.close()
                self._hip_shard = None
                return self._hip_engine()
            eng.row0 = r0
            self._hip_shard = ((r0, r1), eng)
            return eng
        eng = getattr(self, "_hip", None)
        if eng is None:
            from .engine import HipScorer
            eng = HipScorer(self.hip_device)
            self._hip_setup(eng)
            eng.row0 = 0
            self._hip = eng
        return eng

    def _hip_setup(self, eng, rows=None):
        """Upload the model to a fresh engine (dot-product models; item-item models override).  rows=(r0, r1): only
        those user rows."""
        U, V, ubase, ibase = self._hip_factor_params()
        # all-zero rows for every user index the model has no factors for (up to total_users)
        # plus one spare: users whose user term is dropped, or who cannot be scored, map there
        n_ext = max(int(self.total_users or 0), U.shape[0]) + 1
        r0, r1 = (0, n_ext) if rows is None else (int(rows[0]), int(rows[1]))
        n_loc = max(r1 - r0, 1)
        have = max(0, min(r1, U.shape[0]) - r0)                   # rows of the range the model has factors for
        Ue = np.zeros((n_loc, U.shape[1]), np.float32)
        if have:
            Ue[:have] = np.asarray(U[r0:r0 + have], np.float32)
        ube = None
        if ubase is not None:
            ube = np.zeros(n_loc, np.float32)
            hb = max(0, min(r1, len(ubase)) - r0)
            if hb:
                ube[:hb] = np.asarray(ubase[r0:r0 + hb], np.float32)
        if ube is None and ibase is not None:
            ube = np.zeros(n_loc, np.float32)
        eng.set_factors(Ue, np.ascontiguousarray(V, np.float32), ube,
                        None if ibase is None else np.ascontiguousarray(ibase, np.float32))
        if rows is None:
            self._hip_zero_row = n_ext - 1

    def hip_invalidate(self):
        """Drop the device copy (call after the parameters change)."""
        eng = getattr(self, "_hip", None)
        if eng is not None:
            eng.close()
        self._hip = None
        shard = getattr(self, "_hip_shard", None)
        if shard is not None:
            shard[1].close()
        self._hip_shard = None

    # ---- Recommender.score
    def score(self, user_idx, item_idx=None, **kwargs):
        row = self._hip_user_row(user_idx)
        if row < 0:
            raise ScoreException("Can't make score prediction for user %d" % user_idx)
        if item_idx is not None and self._hip_item_must_be_known and self.is_unknown_item(item_idx):
            raise ScoreException("Can't make score prediction for item %d" % item_idx)
        eng = self._hip_engine()
        dense = self._hip_score_row(eng, row)
        if item_idx is None:
            return self._hip_shape_scores(dense.cpu().numpy().astype(self.score_dtype, copy=False))
        return self.score_dtype(dense[int(item_idx)].item())

    def _hip_shape_scores(self, scores):
        return scores

    def _hip_score_row(self, eng, row):
        """The score row of `row` as the device tensor `score` returns values from."""
        return eng.score_dense(np.asarray([row], np.int32))[0]

    _hip_item_must_be_known = True

    # ---- Recommender.rank
    def rank(self, user_idx, item_indices=None, k=-1, **kwargs):
        """recommender.py:533-661 (the article_pool branch is the caller's: it only pre-selects
        item_indices).  Returns (item_rank int64, item_scores aligned with item_indices)."""
        import torch
        eng = self._hip_engine()
        total_items = self.total_items
        row = self._hip_user_row(user_idx)
        if row < 0:
            # ScoreException -> constant default_score (recommender.py:626-627): every candidate
            # ties, the canonical order is ascending item index
            all_scores = np.ones(total_items) * self.default_score()
            cand = np.arange(self.num_items) if item_indices is None else np.asarray(item_indices)
            item_scores = all_scores[cand]
            ranked = np.sort(cand, kind="stable")
        else:
            exact = self._hip_score_row(eng, row).contiguous()   # device [num_items]: fp32, or fp64 (precision="f64")
            if exact.numel() != total_items:
                # unknown items get the minimum known score (recommender.py:631-635)
                pad = torch.full((total_items - exact.numel(),), exact.min().item(), dtype=exact.dtype,
                                 device=exact.device)
                exact = torch.cat([exact, pad])
            # ranking keys: the scores themselves -- float64 sums are ordered as float64 (ifr_rank_items_f64), like the
            # reference's argsort of its float64 scores (recommender.py:643-652)
            cand = np.arange(self.num_items) if item_indices is None else np.asarray(item_indices)
            cand_dev = torch.as_tensor(cand.astype(np.int32), device=exact.device)
            ranked = eng.rank_items(exact, cand_dev).cpu().numpy().astype(cand.dtype if cand.dtype.kind == "i" else np.int64)
            item_scores = exact[cand_dev.long()].cpu().numpy().astype(self.score_dtype, copy=False)
        if k != -1 and k > len(cand):
            raise ValueError("kth(=%d) out of bounds (%d)" % (k, len(cand)))    # np.argpartition's error
        self.ranked_items[user_idx] = ranked
        self.item_scores[user_idx] = item_scores
        self.item_scores_mapped_indices[user_idx] = cand
        return ranked, item_scores

    # ---- batched entry point (the shape of BaseANN.knn_query / recommend_batch,
    # cornac/models/ann/recom_ann_base.py:91-98, 182-235)
    def rank_batch(self, user_indices, k, exclude_csr=None):
        """Top-k (ids int32 [n,k], scores fp32 [n,k]) for many users in one fused launch.
        exclude_csr: optional scipy CSR / (indptr, indices) of items to leave out per user."""
        eng = self._hip_engine()
        users = np.asarray(user_indices)
        rows = np.asarray([self._hip_user_row(int(u)) for u in users], np.int64)
        if (rows < 0).any():
            raise ScoreException("rank_batch: the model cannot score user %d" % int(users[rows < 0][0]))
        if exclude_csr is not None:
            ptr, idx = (exclude_csr.indptr, exclude_csr.indices) if hasattr(exclude_csr, "indptr") else exclude_csr
            if (rows != users).any():
                raise ValueError("rank_batch: exclusions need user rows that exist in the model")
            eng.set_exclusion_csr(np.asarray(ptr, np.int64), np.asarray(idx, np.int32))
        elif getattr(eng, "has_exclusions", False) or getattr(eng, "exclusion_source", None) is not None:
            # nothing is excluded unless a CSR is passed: drop what an earlier ranking_eval / export left installed
            eng.set_exclusion_csr(np.zeros(1, np.int64), np.zeros(0, np.int32))
        if getattr(eng, "catalogue_items", eng.n_items) != eng.n_items:
            eng.set_catalogue_items(eng.n_items)      # rank the model's own items, not an evaluation's widened catalogue
        ids, scores = eng.rank_topk(rows.astype(np.int32), int(k))
        return ids.cpu().numpy(), scores.cpu().numpy()


class BPR(HipScoringMixin, RecommenderBase):
    """cornac.models.BPR scoring (cornac/models/bpr/recom_bpr.pyx:272-297): s = Bi + V.U[u].
    init_params: {'U': [total_users,k], 'V': [total_items,k], 'Bi': [total_items]}."""

    def __init__(self, name="BPR", k=10, use_bias=True, trainable=False, verbose=False, init_params=None, seed=None,
                 **kwargs):
        RecommenderBase.__init__(self, name=name, trainable=trainable, verbose=verbose)
        if trainable:
            raise NotImplementedError("training is outside the hot path: fit the reference model and pass init_params")
        self.k, self.use_bias, self.seed = int(k), use_bias, seed
        self.init_params = {} if init_params is None else init_params
        self.u_factors = self.init_params.get("U")
        self.i_factors = self.init_params.get("V")
        self.i_biases = self.init_params.get("Bi")

    def _hip_factor_params(self):
        bi = self.i_biases if (self.i_biases is not None and self.use_bias) else np.zeros(len(self.i_factors), np.float32)
        return self.u_factors, self.i_factors, None, bi

    _hip_item_must_be_known = False     # BPR.score never raises (recom_bpr.pyx:290-297)
    _ann_supported = True

    def get_vector_measure(self):       # recom_bpr.pyx:299-307
        return "dot"

    def get_user_vectors(self):         # recom_bpr.pyx:309-320: [U, 1]
        u = np.asarray(self.u_factors, np.float32)
        return np.concatenate([u, np.ones([u.shape[0], 1], np.float32)], axis=1) if self.use_bias else u

    def get_item_vectors(self):         # recom_bpr.pyx:322-333: [V, Bi]
        v = np.asarray(self.i_factors, np.float32)
        return np.concatenate([v, np.asarray(self.i_biases, np.float32).reshape(-1, 1)], axis=1) if self.use_bias else v

    def _hip_user_row(self, user_idx):
        # factors are allocated for total_users (recom_bpr.pyx:146-151); the reference indexes u_factors[user_idx]
        # (:290): IndexError beyond the array, NumPy's wrap-around for a negative index inside it
        n = len(self.u_factors)
        if not -n <= int(user_idx) < n:
            raise IndexError(f"index {int(user_idx)} is out of bounds for axis 0 with size {n}")
        return int(user_idx) % n


class MF(HipScoringMixin, RecommenderBase):
    """cornac.models.MF scoring (cornac/models/mf/recom_mf.py:254-286):
    s = ((mu + Bi) + Bu[u]) + V.U[u]; an unknown user keeps only mu + Bi."""

    def __init__(self, name="MF", k=10, use_bias=True, trainable=False, verbose=False, init_params=None, seed=None,
                 **kwargs):
        RecommenderBase.__init__(self, name=name, trainable=trainable, verbose=verbose)
        if trainable:
            raise NotImplementedError("training is outside the hot path: fit the reference model and pass init_params")
        self.k, self.use_bias, self.seed = k, use_bias, seed
        self.init_params = {} if init_params is None else init_params
        self.u_factors = self.init_params.get("U")
        self.i_factors = self.init_params.get("V")
        self.u_biases = self.init_params.get("Bu")
        self.i_biases = self.init_params.get("Bi")
        self._mu_override = self.init_params.get("mu")

    def fit(self, train_set, val_set=None):
        RecommenderBase.fit(self, train_set, val_set)
        mu = self.global_mean if self._mu_override is None else self._mu_override
        self.global_mean = np.float32(mu if self.use_bias else 0.0)          # recom_mf.py:156
        return self

    def _hip_factor_params(self):
        nu, ni = self.u_factors.shape[0], self.i_factors.shape[0]
        bu = np.zeros(nu, np.float32) if self.u_biases is None else np.asarray(self.u_biases, np.float32)
        bi = np.zeros(ni, np.float32) if self.i_biases is None else np.asarray(self.i_biases, np.float32)
        ibase = (np.float32(self.global_mean) + bi).astype(np.float32)       # fl(mu + Bi), recom_mf.py:276
        return self.u_factors, self.i_factors, bu, ibase

    def _hip_user_row(self, user_idx):
        if self.knows_user(user_idx):
            return int(user_idx)
        self._hip_engine()
        return int(user_idx) if self.num_users <= user_idx < self._hip_zero_row else self._hip_zero_row


class WMF(HipScoringMixin, RecommenderBase):
    """cornac.models.WMF scoring (cornac/models/wmf/recom_wmf.py:213-239): s = V.U[u];
    unknown users / items raise ScoreException."""

    def __init__(self, name="WMF", k=200, trainable=False, verbose=False, init_params=None, seed=None, **kwargs):
        RecommenderBase.__init__(self, name=name, trainable=trainable, verbose=verbose)
        if trainable:
            raise NotImplementedError("training is outside the hot path: fit the reference model and pass init_params")
        self.k, self.seed = k, seed
        self.init_params = {} if init_params is None else init_params
        self.U = self.init_params.get("U")
        self.V = self.init_params.get("V")

    def _hip_factor_params(self):
        return self.U, self.V, None, None

    def _hip_user_row(self, user_idx):
        return int(user_idx) if self.knows_user(user_idx) else -1


class ItemItemScoringMixin(HipScoringMixin):
    """score / rank / rank_batch for item-item models: s_u = (x_u . M) * col_scale.

    A concrete class supplies `_hip_item_model() -> (M, histories, col_scale)`: M a dense
    [I, I] array / tensor or a scipy CSR matrix, histories a scipy CSR [users, I] whose rows hold
    x_u, col_scale an [I] array or None.  The reference computes these models in fp64
    ("Precision mix"); the HIP path scores in fp32 (dense M) or in 64-bit
    fixed point (CSR M) and returns float64 arrays like the reference."""

    score_dtype = np.float64

    def _hip_setup(self, eng, rows=None):
        import scipy.sparse as sp
        M, hist, col_scale = self._hip_item_model()
        if rows is not None:                                        # a shard: the histories of its users only
            hist = sp.csr_matrix(hist)
            r0, r1 = int(rows[0]), int(rows[1])
            hist = hist[min(r0, hist.shape[0]):min(r1, hist.shape[0])]
        if sp.issparse(M):
            M = sp.csr_matrix(M)
            M.sort_indices()
            eng.set_item_matrix_csr(M.indptr.astype(np.int64), M.indices.astype(np.int32),
                                    M.data.astype(np.float32), M.shape[0], col_scale)
        else:
            eng.set_item_matrix_dense(M, col_scale, precision=getattr(self, "precision", "f32"))
        self._hip_set_histories(eng, hist, shard=rows is not None)

    def _hip_set_histories(self, eng, hist, shard=False):
        import scipy.sparse as sp
        hist = sp.csr_matrix(hist)
        eng.set_history_csr(hist.indptr.astype(np.int64), hist.indices.astype(np.int32), hist.data.astype(np.float32))
        if not shard:
            self._hip_zero_row = max(int(self.total_users or 0), hist.shape[0])     # a user without history

    def _hip_user_row(self, user_idx):
        return int(user_idx) if self.knows_user(user_idx) else -1

    def _hip_score_row(self, eng, row):
        if eng.rows_are_f64():            # float64 dense matrix, or a CSR matrix (exact sums): the scores that are ranked
            return eng.score_dense_f64(np.asarray([row], np.int32))[0][: eng.n_items]
        return eng.score_dense(np.asarray([row], np.int32))[0]


class EASE(ItemItemScoringMixin, RecommenderBase):
    """cornac.models.EASE scoring (cornac/models/ease/recom_ease.py:99-126): s = X[u, :] . B with
    X = train_set.matrix (:75).  init_params: {'B': [I, I]} (an already fitted B, e.g. the reference's); without it
    `fit` builds B on the GPU with the reference's closed form (:78-95): Gram scatter + rocSOLVER Cholesky
    (ifr_fit_ease; B goes straight into the scoring engine, `self.B` is its float64 host copy up to
    `device_B_copy_max_items` items and None above).  There is no host build in this package (the NumPy restatement
    of the closed form is test infrastructure: the checker of tests/test_gpu_fit.py).
    precision="f64" keeps B in float64 on the device, sums every score in float64 in history order -- the
    reference's arithmetic -- and RANKS those float64 sums (64-bit keys); the default "f32" halves the bytes the
    HBM-bound scoring kernel reads."""

    device_B_copy_max_items = 16384          # 2 GiB of float64 on the host

    def __init__(self, name="EASEᴿ", lamb=500, posB=True, trainable=True, verbose=True, seed=None, B=None, U=None,
                 init_params=None, precision="f32", **kwargs):
        RecommenderBase.__init__(self, name=name, trainable=trainable, verbose=verbose)
        self.lamb, self.posB, self.seed = lamb, posB, seed
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        self.precision = precision
        self.B = B if B is not None else (init_params or {}).get("B")
        self.U = U

    def fit(self, train_set, val_set=None):
        import scipy.sparse as sp
        RecommenderBase.fit(self, train_set, val_set)
        self.U = train_set.matrix                                   # recom_ease.py:75
        self.hip_invalidate()
        if self.B is None:
            if not self.trainable:
                raise ValueError("EASE(trainable=False) needs init_params={'B': ...}")
            from .engine import HipScorer
            X = sp.csr_matrix(self.U)
            X.sum_duplicates()
            X.sort_indices()
            eng = HipScorer(self.hip_device)
            try:
                n_items = X.shape[1]
                self.B = eng.fit_ease(X.indptr.astype(np.int64), X.indices.astype(np.int32), X.data.astype(np.float32),
                                      n_items, self.lamb, self.posB, want_B=n_items <= self.device_B_copy_max_items,
                                      precision=self.precision)
                self._hip_set_histories(eng, X)
            except Exception:
                eng.close()
                raise
            eng.row0 = 0
            self._hip = eng                                     # B is already resident: no second upload
        return self

    def _hip_item_model(self):
        if self.B is None:
            raise RuntimeError("EASE: B was built on the device and not copied back (more than device_B_copy_max_items "
                               "items); it went with the engine that hip_invalidate() closed -- call fit() again")
        return self.B, self.U, None

    def _hip_shape_scores(self, scores):
        return scores.reshape(1, -1)        # the reference returns the (1, I) product of a sparse row (:124)


class RP3_Beta(ItemItemScoringMixin, RecommenderBase):
    """cornac.models.RP3_Beta scoring (cornac/models/rp3_beta/recom_rp3beta.py:99-131,
    graph_recommender.py:87-117): s = P3[u, items] / deg(item)^beta, evaluated in item-item form
    s = (P_ui[u, :] . W) * deg^-beta with W = P_iu . P_ui, so the (U+I)^2 cube P3 is never built.
    `fit` builds P_ui, W and the degrees from the binarised train matrix (rating > 0, :76) on the GPU
    (ifr_fit_rp3beta: counting-sort transpose + rocSPARSE SpGEMM pattern + ordered float64 sums; W stays on the
    device as the CSR item matrix).  init_params = {'P_ui': csr [U, I], 'W': csr [I, I], 'item_degrees': int [I]}
    installs arrays built elsewhere instead (item_matrix "csr" | "dense" | "auto": dense below 8192 items); there is
    no host build in this package (the SciPy restatement is test infrastructure, the checker of the tests)."""

    def __init__(self, beta=0.7, name="RP3_beta", trainable=True, verbose=False, item_matrix="auto", init_params=None,
                 precision="f32", **kwargs):
        RecommenderBase.__init__(self, name=name, trainable=trainable, verbose=verbose)
        self.beta = beta
        self.item_matrix = item_matrix
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        if precision == "f64" and (item_matrix != "dense" or not init_params):
            raise ValueError("RP3_Beta(precision='f64') needs item_matrix='dense' and init_params (a float64 W)")
        self.precision = precision
        ip = init_params or {}
        self.P_ui, self.W, self.item_degrees = ip.get("P_ui"), ip.get("W"), ip.get("item_degrees")
        if (self.W is None) != (self.P_ui is None) or (self.W is None) != (self.item_degrees is None):
            raise ValueError("RP3_Beta init_params needs all of 'P_ui', 'W' and 'item_degrees'")

    def fit(self, train_set, val_set=None):
        import scipy.sparse as sp
        RecommenderBase.fit(self, train_set)
        self.hip_invalidate()
        if self.W is not None:                                        # arrays built elsewhere: uploaded on first use
            return self
        if self.item_matrix == "dense":
            raise ValueError("RP3_Beta: the device build keeps W as a CSR matrix; pass init_params for a dense W")
        A = sp.csr_matrix(train_set.csr_matrix, dtype=np.float32, copy=True)    # never write into the data set
        A.data[:] = (A.data > 0).astype(np.float32)                  # recom_rp3beta.py:73-93
        A.eliminate_zeros()
        A.sum_duplicates()
        A.sort_indices()
        from .engine import HipScorer
        eng = HipScorer(self.hip_device)
        try:
            self.w_nnz = eng.fit_rp3beta(A.indptr.astype(np.int64), A.indices.astype(np.int32), A.shape[1], self.beta)
        except Exception:
            eng.close()
            raise
        self._hip_zero_row = max(int(self.total_users or 0), A.shape[0])
        eng.row0 = 0
        self._hip = eng
        return self

    def _hip_item_model(self):
        if self.W is None:
            raise RuntimeError("RP3_Beta: W was built on the device and went with the engine that hip_invalidate() "
                               "closed -- call fit() again")
        # the reference keeps the degrees in an int array, so its 0.0001 guard stores 0 and a
        # zero-degree item scores 0 * inf = nan (graph_recommender.py:107-115); same here
        with np.errstate(divide="ignore"):
            pen = (1.0 / (np.asarray(self.item_degrees).astype(np.float64) ** self.beta)).astype(np.float32)
        mode = self.item_matrix
        if mode == "auto":
            mode = "dense" if self.W.shape[0] <= 8192 else "csr"
        if self.precision == "f64":
            # float64 route: W stays float64 and carries 1/deg^beta (folded in float64: one rounding of 1e-16 per entry),
            # every sum is a float64 chain and the float64 sums are what is ranked.  The history weights 1/deg(u) stay
            # fp32: one common factor per user, which moves no order.  (The reference itself rounds P, P^2 and P^3 to
            # fp32, graph_recommender.py:36-38, 83-84: its own values carry ~1e-7 of relative rounding.)
            with np.errstate(divide="ignore", invalid="ignore"):
                pen64 = 1.0 / (np.asarray(self.item_degrees).astype(np.float64) ** self.beta)
                M = np.asarray(self.W.astype(np.float64).toarray()) * pen64[None, :]
            return M, self.P_ui.astype(np.float32), None
        M = self.W.astype(np.float32)
        if mode == "dense":
            M = M.toarray()
        return M, self.P_ui.astype(np.float32), pen

    def rank(self, user_idx, item_indices=None, k=-1, **kwargs):
        if self.is_unknown_user(user_idx):                            # recom_rp3beta.py:155-158: not caught
            raise ScoreException("Can't make score prediction for (user_id=%d)" % user_idx)
        return HipScoringMixin.rank(self, user_idx, item_indices, k, **kwargs)


def wrap_reference_model(model, device=0):
    """Give a fitted reference model (cornac.models.BPR / MF / WMF instance) the HIP scoring path
    without touching its class: returns a facade that shares its parameter arrays and fit
    bookkeeping.  INTEGRATION.md shows the subclassing alternative."""
    name = type(model).__name__
    if name == "BPR":
        out = BPR(name=model.name, k=model.k, use_bias=model.use_bias,
                  init_params={"U": model.u_factors, "V": model.i_factors, "Bi": model.i_biases})
    elif name == "MF":
        out = MF(name=model.name, k=model.k, use_bias=model.use_bias,
                 init_params={"U": model.u_factors, "V": model.i_factors, "Bu": model.u_biases, "Bi": model.i_biases,
                              "mu": model.global_mean})
    elif name == "WMF":
        out = WMF(name=model.name, k=model.k, init_params={"U": model.U, "V": model.V})
    elif name == "EASE":
        out = EASE(name=model.name, lamb=model.lamb, posB=model.posB, B=model.B, U=model.U, trainable=False)
    elif name == "RP3_Beta":
        out = RP3_Beta(beta=model.beta, name=model.name)
        out.fit(model.train_set)                 # item-item form built on the device from the same train matrix
    else:
        raise TypeError(f"no HIP scoring facade for {name}")
    for attr in ("num_users", "num_items", "uid_map", "iid_map", "min_rating", "max_rating", "global_mean",
                 "is_fitted"):
        setattr(out, attr, getattr(model, attr))
    out.hip_device = device
    return out
