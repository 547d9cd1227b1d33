"""ranking_eval / cache_rankings with the reference's signatures
(cornac/eval_methods/base_method.py:143-305), executed as a few batched HIP launches instead of
one Python iteration per user.

Host side (this file): turn the train / validation / test CSR matrices into the two CSRs the
kernels read (exclusions and truth, SURVEY.md Appendix B rules 1-3), pick the launch path from
the metric list, shard the test users across ranks when torch.distributed is initialised, and
assemble the reference's return value `(avg_results, user_results)`.
"""
import weakref
from collections.abc import Mapping

import numpy as np
import scipy.sparse as sp

from . import dist as ifr_dist
from .engine import MAX_DENSE_METRIC_POS, METRIC_COLUMNS, METRICS_PER_K, DeviceCsr
from .metrics import metric_kind

MAX_FUSED_K = 128         # fused score + top-K kernel of dot-product models (item-item models: 32); larger cut-offs,
MAX_FUSED_K_ITEM = 32     # k = -1, MRR and MAP take dense rows + ifr_eval_dense_metrics (positions of the positives by counting)


# --------------------------------------------------------------------------- CSR preparation
def _positive_csr(mat, thr, n_rows, n_cols, row0=0):
    """Rows [row0, row0 + n_rows) of `mat` restricted to rating >= thr and item < n_cols, as a boolean CSR with
    n_rows rows (rows beyond mat.shape[0] are empty: base_method.py:266-270)."""
    mat = sp.csr_matrix(mat)
    if row0 or n_rows < mat.shape[0]:
        mat = mat[row0:min(row0 + n_rows, mat.shape[0])]
    keep = mat.data >= thr
    coo_rows = np.repeat(np.arange(mat.shape[0]), np.diff(mat.indptr))[keep]
    cols = mat.indices[keep]
    ok = (cols < n_cols) & (coo_rows < n_rows)
    out = sp.csr_matrix((np.ones(int(ok.sum()), np.int8), (coo_rows[ok], cols[ok])), shape=(n_rows, n_cols))
    out.sum_duplicates()
    out.sort_indices()
    return out


def users_with_positives(test_mat, rating_threshold, n_rows):
    """has_pos[u]: user u has a test item rated >= threshold *before* clipping to the catalogue (base_method.py:257-259)."""
    test_mat = sp.csr_matrix(test_mat)
    keep = test_mat.data >= rating_threshold
    rows = np.repeat(np.arange(test_mat.shape[0]), np.diff(test_mat.indptr))[keep]
    has_pos = np.zeros(n_rows, bool)
    has_pos[rows[rows < n_rows]] = True
    return has_pos


def build_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, n_rows, row0=0):
    """-> (truth CSR, exclusion CSR, has_pos) for the users [row0, row0 + n_rows) per base_method.py:246-283 (row i of
    the results is user row0 + i: a rank of a sharded evaluation only builds its own users' rows).

    truth[u]     = test items rated >= threshold, clipped to [0, n_items)            (gt_pos)
    exclusion[u] = (train U validation items rated >= threshold) minus truth[u]     (not ranked)
    has_pos[u]   = user has a test item rated >= threshold *before* clipping (:257-259)."""
    has_pos = users_with_positives(test_mat, rating_threshold, row0 + n_rows)[row0:]
    truth = _positive_csr(test_mat, rating_threshold, n_rows, n_items, row0)
    excl = _positive_csr(train_mat, rating_threshold, n_rows, n_items, row0)
    if val_mat is not None:
        excl = excl + _positive_csr(val_mat, rating_threshold, n_rows, n_items, row0)
    excl = excl - excl.multiply(truth)              # test positives stay candidates (mask sum = 1)
    excl.eliminate_zeros()
    excl.sort_indices()
    return truth, excl, has_pos


# Experiment / hyper-parameter search evaluate many models on the same splits (base_method.py:1106-1116,
# hyperopt.py:151-159): the CSRs above depend only on the splits, so the last few are kept.
_EVAL_CSR_CACHE = []
_EVAL_CSR_CACHE_SIZE = 8
_FINGERPRINT_FULL_NNZ = 1 << 20


def _fingerprint(mat):
    """Cheap content fingerprint of a CSR matrix: shape, nnz and checksums of its three arrays -- over every entry up
    to 2^20 stored ratings, over about 2^16 evenly strided samples beyond (a repeated evaluation at 20 M ratings
    would otherwise spend more time summing ratings than ranking).  An in-place edit of a large matrix that misses
    every sample goes unnoticed: call `clear_eval_cache()` after editing a split in place."""
    if mat is None:
        return None
    m = sp.csr_matrix(mat)
    step = 1 if m.nnz <= _FINGERPRINT_FULL_NNZ else max(1, m.nnz >> 16)
    return (m.shape, m.nnz, float(m.data[::step].sum(dtype=np.float64)), int(m.indices[::max(97, step)].sum(dtype=np.int64)),
            int(m.indptr[::97].sum(dtype=np.int64)))


def clear_eval_cache():
    """Forget the memoised evaluation CSRs (after editing a split in place)."""
    del _EVAL_CSR_CACHE[:]


def _cached(kind, mats, key, build):
    for ent in _EVAL_CSR_CACHE:
        if ent[0] == kind and len(ent[1]) == len(mats) and all(a is b for a, b in zip(ent[1], mats)) and ent[2] == key:
            return ent[3]
    out = build()
    _EVAL_CSR_CACHE.insert(0, (kind, mats, key, out))
    del _EVAL_CSR_CACHE[_EVAL_CSR_CACHE_SIZE:]
    return out


def cached_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, n_rows, row0=0):
    """build_eval_csrs, memoised on the identity of the three matrices plus a cheap content fingerprint
    (shape, nnz, checksums), so that in-place edits are noticed."""
    key = (float(rating_threshold), int(n_items), int(n_rows), int(row0), _fingerprint(train_mat), _fingerprint(test_mat),
           _fingerprint(val_mat))
    return _cached("csrs", (train_mat, test_mat, val_mat), key,
                   lambda: build_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, n_rows, row0))


def cached_users_with_positives(test_mat, rating_threshold, n_rows):
    key = (float(rating_threshold), int(n_rows), _fingerprint(test_mat))
    return _cached("has_pos", (test_mat,), key, lambda: users_with_positives(test_mat, rating_threshold, n_rows))


# --------------------------------------------------------------------------- candidate sets built on the device
# SciPy builds the two CSRs above in 0.2 s at ML-20M size -- 27 x the fused kernel's pass -- and the 75 M interactions of
# a C3 shard never fit that budget.  Splits with at least this many stored ratings (or splits handed over as
# `DeviceCsr`) go through ifr_build_eval_csrs instead: the raw arrays are uploaded once (memoised like the host CSRs),
# one wave per user row filters them, and nothing but the list of evaluated users comes back to the host.
DEVICE_BUILD_MIN_NNZ = 1 << 22


def _split_nnz(mat):
    return 0 if mat is None else int(mat.nnz)


def _split_key(mat):
    if isinstance(mat, DeviceCsr):
        return ("dev", mat.shape, mat.nnz, int(mat.indptr.data_ptr()), int(mat.indices.data_ptr()), int(mat.data.data_ptr()))
    return _fingerprint(mat)


def device_users_with_positives(test_mat, rating_threshold):
    """(users int64 ascending on the host, n_rows) of a DeviceCsr test split: the users with a test item rated >=
    threshold (base_method.py:254-259), and 1 + the last user with any stored test entry."""
    import torch
    keep = test_mat.data >= rating_threshold
    cs = torch.zeros(test_mat.nnz + 1, dtype=torch.int64, device=keep.device)
    torch.cumsum(keep, 0, out=cs[1:])
    cnt = cs[test_mat.indptr[1:]] - cs[test_mat.indptr[:-1]]
    users = torch.nonzero(cnt > 0).flatten().cpu().numpy().astype(np.int64)
    stored = torch.nonzero(test_mat.indptr[1:] > test_mat.indptr[:-1]).flatten()
    n_rows = int(stored[-1].item()) + 1 if stored.numel() else 0
    return users, n_rows


class DeviceEvalCsrs:
    """The truth / exclusion CSRs of one (splits, threshold, catalogue, user range) built INSIDE an engine by
    ifr_build_eval_csrs.  The object is the identity the engines and the result caches compare (`exclusion_source`); the
    SciPy forms are read back from the engine only if a host consumer asks (per-user caches, unknown-item tails)."""

    def __init__(self, mats, thr, n_items, n_rows, row0):
        self.mats, self.thr, self.n_items, self.n_rows, self.row0 = mats, float(thr), int(n_items), int(n_rows), int(row0)
        self.stats = None
        self._host = {}
        self._uploads = {}

    def install(self, eng):
        """Make `eng` hold these CSRs (a no-op when it already does)."""
        if getattr(eng, "exclusion_source", None) is self and getattr(eng, "truth_source", None) is self:
            return
        key = str(eng.device)
        ups = self._uploads.get(key)
        if ups is None:                              # raw arrays: uploaded once per device, reused by every engine on it
            ups = [None if m is None else DeviceCsr.of(m, eng.device) for m in self.mats]
            self._uploads = {key: ups}
        st = eng.build_eval_csrs(ups[0], ups[2], ups[1], self.thr, self.n_items, self.row0, self.n_rows)
        self.stats = dict(truth_nnz=st["truth_nnz"], excl_nnz=st["excl_nnz"], max_truth_row=st["max_truth_row"])
        eng.exclusion_source = eng.truth_source = self
        self._eng = weakref.ref(eng)

    def host(self, which):
        """SciPy CSR ("truth" / "exclusion"), read back from the engine that holds the device copy."""
        m = self._host.get(which)
        if m is None:
            eng = self._eng() if getattr(self, "_eng", None) is not None else None
            if eng is None:
                if any(isinstance(x, DeviceCsr) for x in self.mats):
                    raise RuntimeError("the engine that built these evaluation CSRs is gone: run ranking_eval again")
                truth, excl, _ = build_eval_csrs(self.mats[0], self.mats[1], self.mats[2], self.thr, self.n_items, self.n_rows, self.row0)
                self._host.update(truth=truth, exclusion=excl)
                return self._host[which]
            self.install(eng)
            indptr, indices = eng.read_eval_csr(which)
            m = sp.csr_matrix((np.ones(len(indices), np.int8), indices, indptr), shape=(self.n_rows, self.n_items))
            self._host[which] = m
        return m


def cached_device_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, n_rows, row0=0):
    key = (float(rating_threshold), int(n_items), int(n_rows), int(row0), _split_key(train_mat), _split_key(test_mat),
           _split_key(val_mat))
    return _cached("dev_csrs", (train_mat, test_mat, val_mat), key,
                   lambda: DeviceEvalCsrs((train_mat, test_mat, val_mat), rating_threshold, n_items, n_rows, row0))


def _host_csr(x, which="exclusion"):
    return x.host(which) if isinstance(x, DeviceEvalCsrs) else x


class UserResults(Mapping):
    """`user_results[i]` of ranking_eval: user index -> metric value, like the reference's dict
    (base_method.py:236-239, 299), but backed by two arrays: building 138 K-entry dicts per metric cost more than the
    ranking itself.  Reads like a dict (`[]`, `in`, `len`, iteration, `.items()`, `.values()`, `==` with a dict) and
    pickles as a plain dict."""

    __slots__ = ("_users", "_vals", "_src")

    def __init__(self, users, values):
        """`values`: an array, or a callable returning one (a column still on the device: copied on the first read)."""
        users = np.asarray(users, np.int64)
        self._src = None
        if callable(values):
            if len(users) > 1 and not (users[1:] > users[:-1]).all():
                values = values()
            else:
                self._users, self._vals, self._src = users, None, values
                return
        values = np.asarray(values, np.float64)
        if len(users) > 1 and not (users[1:] > users[:-1]).all():
            order = np.argsort(users, kind="stable")
            users, values = users[order], values[order]
        self._users, self._vals = users, values

    @property
    def _values(self):
        if self._vals is None:
            self._vals = np.asarray(self._src(), np.float64)
            self._src = None
        return self._vals

    def __getitem__(self, u):
        i = int(np.searchsorted(self._users, u))
        if i >= len(self._users) or self._users[i] != u:
            raise KeyError(u)
        return float(self._values[i])

    def __contains__(self, u):
        i = int(np.searchsorted(self._users, u))
        return i < len(self._users) and self._users[i] == u

    def __iter__(self):
        return iter(self._users.tolist())

    def __len__(self):
        return len(self._users)

    def keys(self):
        return self._users.tolist()

    def values(self):
        return self._values.tolist()

    def items(self):
        return zip(self._users.tolist(), self._values.tolist())

    def arrays(self):
        """(users int64 ascending, values float64): the backing arrays, for vectorised consumers."""
        return self._users, self._values

    def to_dict(self):
        return dict(self.items())

    def __eq__(self, other):
        if isinstance(other, UserResults):
            return np.array_equal(self._users, other._users) and np.array_equal(self._values, other._values, equal_nan=True)
        if isinstance(other, Mapping):
            return self.to_dict() == dict(other)
        return NotImplemented

    def __repr__(self):
        return f"UserResults({len(self)} users)"

    def __reduce__(self):
        return (dict, (self.to_dict(),))


# --------------------------------------------------------------------------- lazy result caches
class LazyRankCache(Mapping):
    """Dict-like view that materialises `model.ranked_items[u]` / `item_scores[u]` /
    `item_scores_mapped_indices[u]` on first access.  The reference stores three full-length
    arrays per evaluated user (recommender.py:655-659); at 10M x 1M that is ~200 TB
    (SURVEY.md section 7), so the batched path keeps only the recipe."""

    def __init__(self, owner, field, users, explicit=None):
        self._owner, self._field = owner, field
        self._users = users                      # sorted int array of users with a lazy entry
        self._explicit = {} if explicit is None else explicit

    def _lazy_has(self, u):
        i = np.searchsorted(self._users, u)
        return i < len(self._users) and self._users[i] == u

    def __contains__(self, u):
        return u in self._explicit or self._lazy_has(u)

    def __getitem__(self, u):
        if u not in self._explicit:
            if not self._lazy_has(u):
                raise KeyError(u)
            self._owner._materialise_cached_user(int(u))
        return self._explicit[u]

    def __setitem__(self, u, value):
        self._explicit[u] = value

    def __iter__(self):
        seen = set(self._explicit)
        yield from self._explicit
        for u in self._users:
            if int(u) not in seen:
                yield int(u)

    def __len__(self):
        # lazy users + explicit entries that are not lazy users (no Python set of every user: 15 ms at 138 K users)
        n = len(self._users)
        if not self._explicit:
            return n
        if n == 0:
            return len(self._explicit)
        try:
            ek = np.fromiter(self._explicit.keys(), dtype=np.int64, count=len(self._explicit))
        except (TypeError, ValueError):
            return len(set(self._explicit) | set(int(u) for u in self._users))
        pos = np.minimum(np.searchsorted(self._users, ek), n - 1)
        return n + int((np.asarray(self._users)[pos] != ek).sum())

    def __reduce__(self):                         # pickles as the plain dict the reference writes
        return (dict, (dict(self.items()),))


class _CacheOwner:
    """Holds what is needed to rebuild a user's full ranking exactly as the batched evaluation saw it (same candidate
    set).  `excl` holds the exclusion rows of users [row0, row0 + excl.shape[0]).

    A first access to a lazy user materialises a whole BLOCK of lazy users in one pass -- dense score rows, one
    segmented sort, candidate lists and aligned scores by device-side compaction, three device-to-host copies --
    because callers that read the caches walk over many users (re-rankers, save_recommendations,
    static_rerank_evaluator.py:262-296); one launch + copy per user costs 250 us, the block path a few tens."""

    block_pairs = 1 << 27          # users x items of one materialisation block (at most `block_users` users)
    block_users = 2048

    def __init__(self, model, excl, n_items, users, row0=0):
        # excl_source: a SciPy CSR, or the DeviceEvalCsrs it is read back from on the first host use
        self.model, self.excl_source, self.n_items, self.users, self.row0 = model, excl, n_items, users, int(row0)

    @property
    def excl(self):
        return _host_csr(self.excl_source)

    def excl_row(self, u):
        r = int(u) - self.row0
        return self.excl[r].indices if 0 <= r < self.excl.shape[0] else np.zeros(0, np.int32)

    def excl_global(self):
        """The exclusion CSR with one row per user index from 0 (what an engine holding all users indexes)."""
        if self.row0 == 0:
            return self.excl
        g = getattr(self, "_excl_global", None)
        if g is None:
            g = sp.vstack([sp.csr_matrix((self.row0, self.excl.shape[1]), dtype=self.excl.dtype), self.excl]).tocsr()
            self._excl_global = g
        return g

    def _materialise_one(self, u):
        mask = np.ones(self.n_items, bool)
        mask[self.excl_row(u)] = False
        cand = np.nonzero(mask)[0]
        # model.rank refills the three caches itself (recommender.py:655-659)
        return self.model.rank(user_idx=u, item_indices=cand, k=-1)

    def _materialise_cached_user(self, u):
        m = self.model
        eng = m._hip_engine() if hasattr(m, "_hip_engine") else None
        caches = (getattr(m, "ranked_items", None), getattr(m, "item_scores", None), getattr(m, "item_scores_mapped_indices", None))
        if eng is None or not all(isinstance(c, LazyRankCache) for c in caches) or not hasattr(eng, "rank_full") \
                or getattr(getattr(eng, "device", None), "type", "cpu") != "cuda" or self.n_items != eng.n_items:
            return self._materialise_one(u)
        import torch
        done = caches[0]._explicit
        pos = int(np.searchsorted(self.users, u))
        b_max = max(1, min(self.block_users, self.block_pairs // max(1, self.n_items)))
        block = []
        for v in self.users[pos:pos + 4 * b_max].tolist():
            if v in done or not 0 <= v < eng.n_users or m._hip_user_row(v) != v:
                continue
            block.append(v)
            if len(block) == b_max:
                break
        if u not in block:                                 # a user the engine has no row of its own for
            return self._materialise_one(u)
        if isinstance(self.excl_source, DeviceEvalCsrs) and self.row0 == 0:
            self.excl_source.install(eng)                          # built on the device: nothing to upload
        else:
            excl = self.excl_global()
            if getattr(eng, "exclusion_source", None) is not excl:
                eng.set_exclusion_csr(excl.indptr.astype(np.int64), excl.indices.astype(np.int32))
                eng.exclusion_source = excl
        eng.set_catalogue_items(self.n_items)
        rows = torch.as_tensor(np.asarray(block, np.int32), device=eng.device)
        # [b, n_items] in the model's native score type: fp32, or the float64 sums of a precision="f64" model (ranked
        # and handed out unrounded, like the one-user path through model.rank)
        dense = eng.score_rows(rows) if hasattr(eng, "score_rows") else eng.score_dense(rows)
        ranked, n_cand = eng.rank_full(rows, dense)
        # candidates in ascending item index and the scores aligned with them (recommender.py:643): compaction of
        # the not-excluded columns, row by row
        keep = torch.ones(dense.shape, dtype=torch.bool, device=eng.device)
        # the block's exclusion entries straight from the CSR arrays (a SciPy row slice per user costs more than the
        # user's share of the launches)
        r = np.asarray(block, np.int64) - self.row0
        inside = (r >= 0) & (r < self.excl.shape[0])
        rc = np.where(inside, r, 0)
        starts = np.where(inside, self.excl.indptr[rc], 0).astype(np.int64)
        lens = np.where(inside, self.excl.indptr[rc + 1] - self.excl.indptr[rc], 0).astype(np.int64)
        total = int(lens.sum())
        if total:
            r_idx = np.repeat(np.arange(len(block)), lens)
            first = np.cumsum(lens) - lens                                   # position of every row's first entry
            src = np.arange(total) - np.repeat(first, lens) + np.repeat(starts, lens)
            c_idx = self.excl.indices[src].astype(np.int64)
            keep[torch.as_tensor(r_idx, device=eng.device), torch.as_tensor(c_idx, device=eng.device)] = False
        def to_host(t, slot):
            # through ONE page-locked staging buffer per owner and slot (pageable memory moves these ~100 KB-per-user
            # copies at a fifth of the link rate), then out into ordinary arrays: the cache entries must not keep
            # page-locked memory alive -- walking every user would pin tens of GB of host RAM
            t = t.contiguous()
            need = t.numel() * t.element_size()
            stage = self._staging.get(slot)
            if stage is None or stage.numel() < need:
                stage = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, pin_memory=True)
                self._staging[slot] = stage
            h = stage[:need].view(t.dtype).view(t.shape)
            h.copy_(t, non_blocking=True)
            return h
        if not hasattr(self, "_staging"):
            self._staging = {}
        sdt = getattr(m, "score_dtype", np.float32)
        picked = dense[keep]
        if (sdt == np.float64) != (picked.dtype == torch.float64):
            picked = picked.to(torch.float64 if sdt == np.float64 else torch.float32)
        cand_t = to_host(keep.nonzero()[:, 1], 0)                            # int64, row-major: per user ascending
        scores_t = to_host(picked, 1)
        ranked_t = to_host(ranked.to(torch.int64), 2)
        n_h = n_cand.cpu().numpy().astype(np.int64)
        torch.cuda.synchronize(eng.device)
        cand_h, scores_h, ranked_h = cand_t.numpy().copy(), scores_t.numpy().copy(), ranked_t.numpy()
        off = np.concatenate([[0], np.cumsum(n_h)])
        for i, v in enumerate(block):
            caches[0][v] = ranked_h[i, : n_h[i]].copy()
            caches[1][v] = scores_h[off[i]: off[i + 1]]
            caches[2][v] = cand_h[off[i]: off[i + 1]]
        return caches[0]._explicit[u], caches[1]._explicit[u]


def cache_rankings(model, user_idx, item_indices, k=-1):
    """base_method.py:143-182."""
    if not hasattr(model, "ranked_items"):
        model.ranked_items = {}
    if not hasattr(model, "item_scores"):
        model.item_scores = {}
    if user_idx in model.ranked_items and user_idx in model.item_scores:
        return model.ranked_items[user_idx], model.item_scores[user_idx]
    if not getattr(model, "is_fitted", False):
        raise RuntimeError("Model is not fitted. Please call `model.fit()` before ranking.")
    item_rank, item_scores = model.rank(user_idx=user_idx, item_indices=item_indices, k=k)
    model.ranked_items[user_idx] = item_rank
    model.item_scores[user_idx] = item_scores
    return item_rank, item_scores


def _free_device_bytes(eng):
    """Bytes the next allocations may take on the engine's device (None on a CPU stand-in)."""
    if hasattr(eng, "free_bytes"):
        return eng.free_bytes()
    dev = getattr(eng, "device", None)
    if dev is None or getattr(dev, "type", "cpu") != "cuda":
        return None
    import torch
    free, _ = torch.cuda.mem_get_info(dev)
    reusable = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)     # cached blocks torch can hand out again
    return int(free + reusable)


def _rows_that_fit(eng, want_rows, row_bytes):
    """Dense score rows per batch: what the caller's bound asks for, at most half of the free device memory.
    `row_bytes` = every byte one row of the chosen branch keeps on the device at a time (the fp32 row, and for the
    sorted-list branch its id row, two 64-bit sort keys per item and the sort's temporary storage)."""
    free = _free_device_bytes(eng)
    if free is None:
        return max(1, int(want_rows))
    return max(1, min(int(want_rows), (free // 2) // int(row_bytes)))


def export_candidate_pools(model, pool_size, batch_users=1 << 16, _engine=None):
    """Heads of the cached rankings for the re-ranking stage: the re-rankers keep only the first `pool_size`
    ranked items of a user (rerankers/reranker.py:620-639) and look their scores up through
    `item_scores_mapped_indices` (static_rerank_evaluator.py:262-296), so the hand-off needs
    pool_size ids + pool_size scores per user, not three catalogue-long arrays.

    Returns three plain dicts shaped like `model.ranked_items / item_scores / item_scores_mapped_indices`
    for every user the last `ranking_eval` calls cached, with the candidate sets those evaluations used:
    ranked_items[u] = the first pool_size items of the full ranking (canonical tie order),
    item_scores[u][j] = score of item_scores_mapped_indices[u][j], and the mapping is that same head.
    Lazy users are ranked in batches by the fused / select top-K kernels; users whose full ranking was
    already materialised, or who have fewer known candidates than pool_size, are cut from the full ranking."""
    import torch
    from .engine import MAX_SELECT_TOPK
    owner = getattr(model, "_hip_cache_owner", None)
    if owner is None:
        raise RuntimeError("export_candidate_pools: no cached evaluation on this model, call ranking_eval first")
    pool_size = int(pool_size)
    if pool_size <= 0 or pool_size > MAX_SELECT_TOPK:
        raise ValueError(f"pool_size must be in 1..{MAX_SELECT_TOPK} (use the full caches beyond that)")
    eng = model._hip_engine() if _engine is None else _engine
    ranked_out, scores_out, mapped_out = {}, {}, {}

    def from_full(u):
        ranked = np.asarray(model.ranked_items[u])                  # materialises a lazy user
        lut_items = np.asarray(model.item_scores_mapped_indices[u])
        lut_scores = np.asarray(model.item_scores[u])
        head = ranked[:pool_size]
        order = np.argsort(lut_items, kind="stable")
        pos = order[np.searchsorted(lut_items[order], head)]
        ranked_out[u], scores_out[u], mapped_out[u] = head, lut_scores[pos], head.copy()

    explicit = getattr(model.ranked_items, "_explicit", {})
    lazy = np.asarray([int(u) for u in owner.users if int(u) not in explicit], np.int64)
    for u in explicit:
        from_full(int(u))
    if len(lazy):
        eng.set_catalogue_items(owner.n_items)
        if isinstance(owner.excl_source, DeviceEvalCsrs) and owner.row0 == 0:
            owner.excl_source.install(eng)
        else:
            excl = owner.excl_global()
            eng.set_exclusion_csr(excl.indptr.astype(np.int64), excl.indices.astype(np.int32))
        k = min(pool_size, eng.n_items)
        for lo in range(0, len(lazy), int(batch_users)):
            part = lazy[lo:lo + int(batch_users)]
            ids, sc = eng.rank_topk(torch.as_tensor(part.astype(np.int32), device=eng.device), k)
            ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
            for r, u in enumerate(part.tolist()):
                if (ids[r] < 0).any() or k < pool_size:
                    from_full(u)                                   # short list: unknown items may follow the known ones
                else:
                    head = ids[r].astype(np.int64)
                    ranked_out[u], scores_out[u], mapped_out[u] = head, sc[r].copy(), head.copy()
    return ranked_out, scores_out, mapped_out


# --------------------------------------------------------------------------- metric plan
def plan_metrics(metrics, max_fused_k=MAX_FUSED_K):
    """-> (specs, ks, need_auc, need_list).  specs[i] = (kind, k) of metrics[i]; need_list: some metric needs the dense
    rows (a cut-off beyond the fused kernel's `max_fused_k`, k = -1, MRR, MAP)."""
    specs = []
    for mt in metrics:
        kind = metric_kind(mt)
        k = getattr(mt, "k", -1)
        if hasattr(k, "__len__"):
            raise ValueError("list-valued k must be expanded by the caller (BaseMethod.organize_metrics does)")
        specs.append((kind, int(k)))
    per_k = [(kind, k) for kind, k in specs if kind in METRIC_COLUMNS]
    need_list = any(kind in ("mrr", "map") for kind, _ in specs) or any(k <= 0 or k > max_fused_k for _, k in per_k)
    need_auc = any(kind == "auc" for kind, _ in specs)
    ks = sorted({k for _, k in per_k}, key=lambda v: (v <= 0, v))
    return specs, ks, need_auc, need_list


def _append_unknown_items(ids, users, excl, n_known, n_items, row0=0):
    """Top-K lists of users with fewer than K known candidates continue with the unknown items
    [n_known, n_items) that are not excluded, in ascending index (host bookkeeping: their order does
    not depend on any score)."""
    short = (ids < 0).any(dim=1).nonzero().flatten().cpu().numpy()
    if len(short) == 0:
        return ids
    host = ids[short].cpu().numpy()
    for r, row in zip(short, host):
        u = int(users[r]) - row0
        ex = excl[u].indices if 0 <= u < excl.shape[0] else np.zeros(0, np.int64)
        free = np.setdiff1d(np.arange(n_known, n_items), ex[ex >= n_known])
        k0 = int((row >= 0).sum())
        take = free[: len(row) - k0]
        row[k0:k0 + len(take)] = take
    ids = ids.clone()
    ids[short] = ids.new_tensor(host)
    return ids


# --------------------------------------------------------------------------- ranking_eval
def ranking_eval(model, metrics, train_set, test_set, val_set=None, rating_threshold=1.0, exclude_unknowns=True,
                 verbose=False, *, batch_pairs=None, distributed="auto", gather_user_results=True, _engine=None):
    """Evaluate `model` on ranking metrics: reference signature and return value
    (base_method.py:185-305); `user_results[i]` is a `UserResults` mapping (reads and pickles like the reference's
    dict).  Keyword-only extras: `batch_pairs` bounds users x items of a dense
    score batch (AUC / k = -1 paths; default 2^32 -- a 16 GB buffer -- and 2^28 where full sorted lists are
    built), `distributed` shards the test users over the ranks of an
    initialised torch.distributed group and all-reduces the metric sums (one RCCL/gloo
    all-reduce), `gather_user_results` also gathers the per-user values to every rank.

    Sharded runs (world > 1): a rank builds and uploads only ITS users' rows -- the exclusion / truth CSRs of the
    user range [r0, r1) it evaluates and an engine that holds only those rows of the user-side parameters
    (`model._hip_engine(rows=(r0, r1))`); item-side parameters are replicated.
    `_engine` (tests) is an engine, or a callable (r0, r1) -> engine over those user rows."""
    if len(metrics) == 0:
        return [], []
    import torch
    plan_metrics(metrics)                          # validates the metric list before anything else
    if not getattr(model, "is_fitted", False):
        raise RuntimeError("Model is not fitted. Please call `model.fit()` before ranking.")
    if _engine is None and not hasattr(model, "_hip_engine"):
        raise TypeError("ranking_eval needs a model with the HIP scoring path (recommenders_amd.models.HipScoringMixin); "
                        "there is no CPU fallback")

    n_items = train_set.num_items if exclude_unknowns else test_set.num_items
    # users with a stored test entry = np.unique(test_set.uir_tuple[0]) (base_method.py:254), read off the CSR rows
    test_mat = test_set.csr_matrix
    train_mat = train_set.csr_matrix
    val_mat = None if val_set is None else val_set.csr_matrix
    any_dev = any(isinstance(m, DeviceCsr) for m in (train_mat, val_mat, test_mat))
    if isinstance(test_mat, DeviceCsr):
        users, n_rows = _cached("dev_users", (test_mat,), (float(rating_threshold), _split_key(test_mat)),
                                lambda: device_users_with_positives(test_mat, rating_threshold))
    else:
        test_users = np.flatnonzero(np.diff(sp.csr_matrix(test_mat).indptr)).astype(np.int64)
        n_rows = int(test_users.max()) + 1 if len(test_users) else 0
        has_pos = cached_users_with_positives(test_mat, rating_threshold, n_rows)
        users = test_users[has_pos[test_users]]                      # base_method.py:258-259

    # ---- shard users over ranks: contiguous ranges of the evaluated users, and only their rows of everything
    world, rank = ifr_dist.world_and_rank(distributed)
    lo, hi = ifr_dist.shard_range(len(users), rank, world)
    my_users = users[lo:hi]
    sharded = world > 1 and (_engine is None or callable(_engine))
    if sharded:
        r0, r1 = (int(my_users[0]), int(my_users[-1]) + 1) if len(my_users) else (0, 0)
    else:
        r0, r1 = 0, n_rows
    if _engine is None:
        eng = model._hip_engine(rows=(r0, r1)) if sharded else model._hip_engine()
    else:
        eng = _engine(r0, r1) if callable(_engine) else _engine
    row0 = int(getattr(eng, "row0", 0)) if sharded else 0
    if sharded and row0 != r0:
        if row0 != 0:
            raise RuntimeError(f"the engine for user rows [{r0}, {r1}) starts at row {row0}")
        sharded, r0, r1 = False, 0, n_rows           # the model only has an engine over all users: index it with their own rows
    specs, ks, need_auc, need_list = plan_metrics(
        metrics, eng.max_fused_topk() if hasattr(eng, "max_fused_topk") else MAX_FUSED_K_ITEM)
    if n_items < eng.n_items:
        raise ValueError(f"ranking over {n_items} items but the model scores {eng.n_items}")
    # n_items > eng.n_items: exclude_unknowns=False with items the model has never seen.  They score the
    # user's minimum (recommender.py:631-635), i.e. they tie at the bottom and, having the largest
    # indices, rank after every known candidate.
    eng.set_catalogue_items(n_items)
    # reference quirk kept: users already in the model's caches are NOT re-ranked
    # (base_method.py:166-167), so their earlier candidate set (exclusion rows) stays in force
    prev = getattr(model, "_hip_cache_owner", None)
    prev_rows = prev is not None and prev.n_items == n_items and prev.row0 == r0 and len(prev.users)
    on_gpu = hasattr(eng, "build_eval_csrs") and getattr(getattr(eng, "device", None), "type", "cpu") == "cuda"
    big = sum(_split_nnz(m) for m in (train_mat, val_mat, test_mat)) >= DEVICE_BUILD_MIN_NNZ
    if any_dev and not on_gpu:
        raise TypeError("DeviceCsr splits need an engine on a GPU")
    device_csrs = None
    if on_gpu and (any_dev or big):
        device_csrs = cached_device_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, r1 - r0, r0)
        if prev_rows and prev.excl_source is not device_csrs and not any_dev:
            device_csrs = None                                       # rows of an earlier evaluation to carry over: host algebra
    if device_csrs is not None:
        # (with DeviceCsr splits the quirk above is dropped: an evaluation on new splits ranks every user again)
        device_csrs.install(eng)
        truth = excl = device_csrs
        max_truth_row = device_csrs.stats["max_truth_row"] if device_csrs.stats else int(np.diff(device_csrs.host("truth").indptr).max(initial=0))
    else:
        truth, excl, _ = cached_eval_csrs(train_mat, test_mat, val_mat, rating_threshold, n_items, r1 - r0, r0)
        if prev is not None and prev.excl_source is excl:
            pass                                                     # same splits as last time: the rows are identical
        elif prev_rows:
            n_loc = r1 - r0
            reuse = prev.users[(prev.users >= r0) & (prev.users < r1)] - r0
            if len(reuse):
                keep = np.ones(n_loc, bool)
                keep[reuse] = False
                pex = prev.excl
                old = pex[:n_loc] if pex.shape[0] >= n_loc else sp.vstack(
                    [pex, sp.csr_matrix((n_loc - pex.shape[0], n_items), dtype=pex.dtype)]).tocsr()
                excl = (sp.diags(keep.astype(np.int8)) @ excl + sp.diags((~keep).astype(np.int8)) @ old).tocsr()
                excl.sort_indices()
        # the memoised CSRs are uploaded once per engine: a repeated evaluation on the same splits finds them in place
        if getattr(eng, "exclusion_source", None) is not excl:
            eng.set_exclusion_csr(excl.indptr.astype(np.int64), excl.indices.astype(np.int32))
            eng.exclusion_source = excl
        if getattr(eng, "truth_source", None) is not truth:
            eng.set_truth_csr(truth.indptr.astype(np.int64), truth.indices.astype(np.int32))
            eng.truth_source = truth
        max_truth_row = int(np.diff(truth.indptr).max()) if truth.shape[0] else 0

    # Users the model does not know keep their own index: the engine holds an all-zero factor row
    # for each of them (HipScoringMixin._hip_engine), which reproduces both "the user term is
    # dropped" (MF, recom_mf.py:277-280) and "constant default score, every candidate ties"
    # (ScoreException, recommender.py:626-627: same order, same metrics).
    loc_users = my_users - row0                                  # the engine's (and the CSRs') row of every user
    if len(loc_users) and int(loc_users.max()) >= eng.n_users:
        raise ValueError(f"test user index {int(my_users.max())} beyond the {eng.n_users} rows the engine holds")
    n = len(my_users)
    values = np.full((n, len(specs)), np.nan)
    col_of = {}
    if n:
        dev_users = torch.as_tensor(loc_users.astype(np.int32), device=eng.device)
        # top-K cut-offs and AUC together: the fused kernel also writes the score rows AUC reads (one GEMM, not two)
        fused_dense = bool(not need_list and ks and need_auc and hasattr(eng, "can_rank_topk_dense")
                           and eng.can_rank_topk_dense(max(ks)))
        if not need_list and ks:
            kmax = max(ks)
            if fused_dense:
                ld = (n_items + 31) // 32 * 32
                bpi = eng.row_bytes_per_item() if hasattr(eng, "row_bytes_per_item") else 4     # 8: a float64 item matrix
                step = _rows_that_fit(eng, min(n, int((batch_pairs or 1 << 32) // ld)), bpi * ld)
                ids = torch.empty((n, kmax), dtype=torch.int32, device=eng.device)
                rows = torch.empty((step, ld), dtype=torch.float64 if bpi == 8 else torch.float32, device=eng.device)
                auc = np.empty(n)
                for b in range(0, n, step):
                    bu = dev_users[b:b + step]
                    _, _, dense = eng.rank_topk_dense(bu, kmax, rows, out_ids=ids[b:b + step])
                    auc[b:b + step] = eng.auc_dense(bu, dense).cpu().numpy()
                del rows
                col_of[("auc", -1)] = auc
            else:
                ids, _ = eng.rank_topk(dev_users, kmax)
            if n_items > eng.n_items:
                ids = _append_unknown_items(ids, my_users, _host_csr(excl), eng.n_items, n_items, row0)
            per_user, _, _ = eng.eval_topk_metrics(dev_users, ids, ks)       # stays on the device until somebody reads it
            for qi, k in enumerate(ks):
                for ci, name in enumerate(METRIC_COLUMNS):
                    col_of[(name, k)] = per_user[:, METRICS_PER_K * qi + ci]
        need_auc = need_auc and not fused_dense
        if need_list or need_auc:
            lst_ks = ks if ks else [-1]
            max_pos = max_truth_row
            dense_metrics = max_pos <= (eng.max_dense_metric_pos() if hasattr(eng, "max_dense_metric_pos") else MAX_DENSE_METRIC_POS)
            # rows only (counting metrics, AUC): 16 GB batches, large launches run 20 % faster per user than 1 GB
            # ones; the sorted-list fallback also holds 64-bit sort keys and ids per pair, so it keeps 2^28 pairs
            sorted_lists = need_list and not dense_metrics
            default_pairs = 1 << 28 if sorted_lists else 1 << 32
            # per row and item: 4 B of scores; the sorted-list branch adds 4 B of ranked ids, 16 B of sort keys (in and
            # out) and rocPRIM's temporary storage (below 4 B per item)
            bpi = eng.row_bytes_per_item() if hasattr(eng, "row_bytes_per_item") else 4     # 8: float64 rows (precision="f64")
            row_bytes = (bpi + ((20 + 2 * bpi) if sorted_lists else 0)) * n_items
            step = _rows_that_fit(eng, min(n, (batch_pairs or default_pairs) // max(1, n_items)), row_bytes)
            if sorted_lists:
                step = max(1, min(step, (2 ** 32 - 1) // max(1, n_items)))        # ifr_rank_full: 32-bit pair count
            per_list = np.empty((n, METRICS_PER_K * len(lst_ks) + 2)) if need_list else None
            auc = np.empty(n) if need_auc else None
            rows = eng.new_score_rows(step) if hasattr(eng, "new_score_rows") else None     # ONE buffer for every batch
            for b in range(0, n, step):
                bu = dev_users[b:b + step]
                score_rows = eng.score_rows if hasattr(eng, "score_rows") else eng.score_dense      # native score type
                dense = score_rows(bu) if rows is None else score_rows(bu, rows[: len(bu)])
                if need_auc:
                    auc[b:b + step] = eng.auc_dense(bu, dense).cpu().numpy()
                if need_list and dense_metrics:
                    # positions of the positives by counting: no ranking is built
                    pl, _, _ = eng.eval_dense_metrics(bu, dense, lst_ks)
                    per_list[b:b + step] = pl.cpu().numpy()
                elif need_list:
                    ranked, n_cand = eng.rank_full(bu, dense)
                    pl, _, _ = eng.eval_ranked_metrics(bu, ranked, n_cand, lst_ks)
                    per_list[b:b + step] = pl.cpu().numpy()
            del rows
            if need_list:
                for qi, k in enumerate(lst_ks):
                    for ci, name in enumerate(METRIC_COLUMNS):
                        col_of[(name, k)] = per_list[:, METRICS_PER_K * qi + ci]
                col_of[("mrr", -1)] = per_list[:, -2]
                col_of[("map", -1)] = per_list[:, -1]
            if need_auc:
                col_of[("auc", -1)] = auc
        cols = [col_of[(kind, k) if kind in METRIC_COLUMNS else (kind, -1)] for kind, k in specs]
        if all(torch.is_tensor(c) for c in cols):
            # every metric came out of the fused path: the sums are taken on the device and the per-user values stay
            # there until `user_results` is read (one D2H of n x len(metrics) doubles, off the critical path)
            dev_values = torch.stack(cols, dim=1)
            dev_sums = dev_values.sum(dim=0).cpu().numpy()
            values = None
        else:
            for i, c in enumerate(cols):
                values[:, i] = c.cpu().numpy() if torch.is_tensor(c) else c
            dev_values = None
        if any(kind == "mrr" for kind, _ in specs):
            i = [kind for kind, _ in specs].index("mrr")
            if np.isnan(values[:, i]).any():
                raise ValueError("No matched between ground-truth items and recommendations")   # ranking.py:215-218
    else:
        dev_values = None

    # ---- aggregate: sums + count, one all-reduce across ranks (base_method.py:302-303)
    if n and dev_values is not None:
        sums = np.concatenate([dev_sums, [float(n)]])
    else:
        sums = np.concatenate([values.sum(axis=0), [float(n)]]) if n else np.zeros(len(specs) + 1)
    sums = ifr_dist.all_reduce_sums(sums, world, getattr(eng, "device", None))
    if sums[-1] == 0:
        raise ZeroDivisionError("division by zero")                  # len(user_results[i]) == 0 in the reference
    avg_results = [float(sums[i] / sums[-1]) for i in range(len(specs))]
    if world > 1 and gather_user_results:
        if dev_values is not None:
            values = dev_values.cpu().numpy()
        all_users, all_values = ifr_dist.gather_rows(my_users, values, world)
        user_results = [UserResults(all_users, all_values[:, i]) for i in range(len(specs))]
    elif dev_values is not None:
        host = []                                                    # one copy shared by the metrics' columns

        def column(i):
            def read():
                if not host:
                    host.append(dev_values.cpu().numpy())
                return host[0][:, i]
            return read
        user_results = [UserResults(my_users, column(i)) for i in range(len(specs))]
    else:
        user_results = [UserResults(my_users, values[:, i]) for i in range(len(specs))]

    # ---- result caches: lazy, same candidate sets as this evaluation
    if hasattr(model, "rank"):
        old_users = prev.users if (prev is not None and prev.row0 == r0) else np.zeros(0, np.int64)
        same = len(old_users) == 0 or (len(old_users) == len(my_users) and np.array_equal(old_users, my_users))
        owner = _CacheOwner(model, excl, n_items, my_users if same else np.union1d(old_users, my_users), r0)
        model._hip_cache_owner = owner
        for field in ("ranked_items", "item_scores", "item_scores_mapped_indices"):
            cur = getattr(model, field, {})
            explicit = cur._explicit if isinstance(cur, LazyRankCache) else cur
            setattr(model, field, LazyRankCache(owner, field, owner.users, explicit))
    return avg_results, user_results
