"""HipScorer: the batched engine behind the reference-shaped facade.

A thin, allocation-owning wrapper over the C ABI (include/ifr.h).  PyTorch is used only
for device buffers and the current HIP stream; all compute happens in libifr.so.
"""
import ctypes

import numpy as np

from . import _lib


METRICS_PER_K = 6      # ndcg, recall, precision, hit ratio, f1, ncrr (include/ifr.h IFR_METRICS_PER_K)
METRIC_COLUMNS = ("ndcg", "recall", "precision", "hit", "f1", "ncrr")
MAX_FUSED_TOPK = 128   # ifr_rank_topk, dot-product models (include/ifr.h IFR_MAX_TOPK)
MAX_FUSED_TOPK_ITEM = 32   # ifr_rank_topk, dense item matrix (IFR_MAX_TOPK_ITEM)
MAX_FUSED_TOPK_ITEM_CSR = 128   # ... CSR item matrix (IFR_MAX_TOPK_ITEM_CSR)
MAX_TOPK_METRICS = 128 # ifr_eval_topk_metrics
MAX_SELECT_TOPK = 1024 # ifr_select_topk
MAX_DENSE_METRIC_POS = 4096   # ifr_eval_dense_metrics: positives per row


class IfrError(RuntimeError):
    """A non-zero status from libifr.so (status -> RuntimeError)."""


def _torch():
    import torch
    return torch


def _host_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class DeviceCsr:
    """A rating matrix in device memory: the three arrays of a canonical CSR matrix (columns ascending inside a row) as
    torch tensors -- indptr int64, indices int32, data float64 / float32 -- plus its shape.  What `Dataset.csr_matrix`
    is to the reference's evaluation loop (cornac/data/dataset.py:229-238), for splits too large to filter on the host."""

    __slots__ = ("indptr", "indices", "data", "shape")

    def __init__(self, indptr, indices, data, shape):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, (int(shape[0]), int(shape[1]))

    @property
    def nnz(self):
        return int(self.indices.numel())

    @classmethod
    def of(cls, mat, device):
        """`mat` itself when it already is a DeviceCsr, else an upload of the SciPy matrix (made canonical first)."""
        if isinstance(mat, cls):
            return mat
        import scipy.sparse as sp
        torch = _torch()
        m = sp.csr_matrix(mat)
        if not m.has_canonical_format:
            m = m.copy()
            m.sum_duplicates()
            m.sort_indices()
        data = m.data if m.data.dtype in (np.float32, np.float64) else m.data.astype(np.float64)
        return cls(torch.as_tensor(m.indptr.astype(np.int64, copy=False)).to(device),
                   torch.as_tensor(m.indices.astype(np.int32, copy=False)).to(device),
                   torch.as_tensor(np.ascontiguousarray(data)).to(device), m.shape)


class HipScorer:
    """Device-resident model + CSR data and the batched rank / metric entry points."""

    def __init__(self, device=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise IfrError("HipScorer needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self._lib = _lib.load()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        h = ctypes.c_void_p()
        rc = self._lib.ifr_create(ctypes.byref(h), self.device.index)
        if rc != 0:
            raise IfrError(f"ifr_create failed with status {rc}")
        self._h = h
        self.n_users = self.n_items = self.k = 0
        self.has_exclusions = False
        self.has_truth = False
        self.item_model = None          # None (dot-product factors) | "dense" | "csr"

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ifr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.ifr_last_error(self._h)
            raise IfrError(f"libifr status {rc}: {msg.decode() if msg else ''}")

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _any_ptr(self, x, dtype):
        """(pointer, keep-alive) for a NumPy array or torch tensor of the given dtype."""
        torch = _torch()
        if x is None:
            return ctypes.c_void_p(0), None
        if isinstance(x, torch.Tensor):
            t = x.contiguous()
            want = {np.float32: torch.float32, np.int32: torch.int32, np.int64: torch.int64}[dtype]
            if t.dtype != want:
                t = t.to(want)
            return ctypes.c_void_p(t.data_ptr()), t
        a = np.ascontiguousarray(x, dtype=dtype)
        return _host_ptr(a), a

    def _dev_i32(self, x):
        torch = _torch()
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(self.device)

    def _dev_users(self, x):
        """int32 device tensor of user rows, checked against [0, n_users): the reference indexes its parameter arrays
        with the user index and raises IndexError beyond them; the kernels only clamp.  The check of a device tensor
        costs one small reduction + sync, skipped when the same tensor OBJECT (identity, version) was checked last."""
        torch = _torch()
        if self.n_users <= 0:                     # no user side set: the library reports the missing state itself
            return self._dev_i32(x)
        if isinstance(x, torch.Tensor) and x.is_cuda:
            u = x.to(device=self.device, dtype=torch.int32).contiguous()
            # "the same tensor" = the same Python object at the same version: a storage address alone would also match a
            # new tensor that the caching allocator placed where a freed, already checked one lived
            last = getattr(self, "_users_checked", None)
            same = last is not None and last[0]() is x and last[1:] == (x._version, self.n_users)
            if u.numel() and not same:
                lo, hi = (int(v) for v in torch.aminmax(x))
                if lo < 0 or hi >= self.n_users:
                    raise IndexError(f"user index {lo if lo < 0 else hi} is out of bounds for {self.n_users} user rows")
                import weakref
                self._users_checked = (weakref.ref(x), x._version, self.n_users)
            return u
        a = np.asarray(x.numpy() if isinstance(x, torch.Tensor) else x)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= self.n_users):
            bad = int(a.min()) if int(a.min()) < 0 else int(a.max())
            raise IndexError(f"user index {bad} is out of bounds for {self.n_users} user rows")
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def set_option(self, name, value):
        self._check(self._lib.ifr_set_option(self._h, name.encode(), int(value)))

    # ------------------------------------------------------------------ data
    def set_factors(self, U, V, ubase=None, ibase=None):
        self.exclusion_source = self.truth_source = None    # the library drops CSRs of another catalogue size
        n_users, k = U.shape
        n_items, k2 = V.shape
        if k != k2:
            raise ValueError("U and V must have the same number of factors")
        pu, ku = self._any_ptr(U, np.float32)
        pv, kv = self._any_ptr(V, np.float32)
        pub, kub = self._any_ptr(ubase, np.float32)
        pib, kib = self._any_ptr(ibase, np.float32)
        self._check(self._lib.ifr_set_factors(self._h, pu, n_users, pv, n_items, k, pub, pib))
        self.n_users, self.n_items, self.k = int(n_users), int(n_items), int(k)
        self.catalogue_items = self.n_items
        self.item_model = None
        del ku, kv, kub, kib

    def set_user_factors(self, U, ubase=None):
        """Replace the user side (query vectors) only; the item factors stay on the device."""
        n_users, k = U.shape
        pu, ku = self._any_ptr(U, np.float32)
        pub, kub = self._any_ptr(ubase, np.float32)
        self._check(self._lib.ifr_set_user_factors(self._h, pu, n_users, k, pub))
        self.n_users = int(n_users)
        del ku, kub

    def set_item_matrix_dense(self, M, col_scale=None, precision="f32"):
        """Item-item model with a dense M [I, I] (EASE's B, or a dense RP3-beta W); NumPy or torch.  precision "f64"
        keeps M in float64 on the device: each score is a float64 sum in history order (the reference's X[u].dot(B) in
        float64, recom_ease.py:123-124), rounded once to fp32 for ranking; score_dense_f64 returns it unrounded."""
        self.exclusion_source = self.truth_source = None    # the library drops CSRs of another catalogue size
        n_items = int(M.shape[0])
        if M.shape[1] != n_items:
            raise ValueError("the item matrix must be square")
        torch = _torch()
        f64 = precision == "f64"
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        tdt, ndt = (torch.float64, np.float64) if f64 else (torch.float32, np.float32)
        if isinstance(M, torch.Tensor):
            M = M if M.dtype == tdt else M.to(tdt)
            ld = M.stride(0) if M.stride(1) == 1 else None
            if ld is None:
                M = M.contiguous()
                ld = M.stride(0)
            pm, km = ctypes.c_void_p(M.data_ptr()), M
        else:
            M = np.ascontiguousarray(M, ndt)
            ld = n_items
            pm, km = _host_ptr(M), M
        pc, kc = self._any_ptr(col_scale, np.float32)
        setter = self._lib.ifr_set_item_matrix_dense_f64 if f64 else self._lib.ifr_set_item_matrix_dense
        self._check(setter(self._h, pm, n_items, int(ld), pc))
        self.n_items, self.k, self.n_users = n_items, 0, 0
        self.catalogue_items = n_items
        self.item_model = "dense"
        self.item_precision = precision
        self.has_exclusions = False
        del km, kc

    def fit_ease(self, indptr, indices, values, n_items, lamb, posB=True, want_B=False, precision="f32"):
        """EASE closed form on the device (recom_ease.py:75-95) from X = train_set.matrix (CSR); B becomes the
        dense item matrix of this scorer.  want_B: also return B as a host float64 [I, I] array.  precision "f64"
        keeps B in float64 on the device and scores in float64 (the reference's arithmetic)."""
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        self.set_option("item_f64", 1 if precision == "f64" else 0)
        self.item_precision = precision
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        pv, kv = self._any_ptr(values, np.float32)
        n_items = int(n_items)
        B = np.empty((n_items, n_items), np.float64) if want_B else None
        self._check(self._lib.ifr_fit_ease(self._h, pp, pi, pv, len(indptr) - 1, n_items, len(indices), float(lamb),
                                           int(bool(posB)), _host_ptr(B) if want_B else None))
        self.n_items, self.k, self.n_users = n_items, 0, 0
        self.catalogue_items = n_items
        self.item_model = "dense"
        self.has_exclusions = False
        del kp, ki, kv
        return B

    def fit_rp3beta(self, indptr, indices, n_items, beta):
        """RP3-beta item-item model on the device from the binarised train pattern A (CSR): W = P_iu . P_ui becomes
        the CSR item matrix, P_ui the histories, 1/deg^beta the column scale.  Returns the number of entries of W."""
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        w_nnz = ctypes.c_int64(0)
        n_users = len(indptr) - 1
        self._check(self._lib.ifr_fit_rp3beta(self._h, pp, pi, n_users, int(n_items), len(indices), float(beta),
                                              ctypes.byref(w_nnz)))
        self.n_items, self.k, self.n_users = int(n_items), 0, 2 ** 31 - 1      # users past the last row: no history
        self.catalogue_items = self.n_items
        self.item_model = "csr"
        self.has_exclusions = False
        del kp, ki
        return int(w_nnz.value)

    def set_item_matrix_csr(self, indptr, indices, values, n_items, col_scale=None):
        """Item-item model with a sparse M (CSR rows, strictly ascending columns)."""
        self.exclusion_source = self.truth_source = None    # the library drops CSRs of another catalogue size
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        pv, kv = self._any_ptr(values, np.float32)
        pc, kc = self._any_ptr(col_scale, np.float32)
        self._check(self._lib.ifr_set_item_matrix_csr(self._h, pp, pi, pv, int(n_items), len(indices), pc))
        self.n_items, self.k, self.n_users = int(n_items), 0, 0
        self.catalogue_items = self.n_items
        self.item_model = "csr"
        self.has_exclusions = False
        del kp, ki, kv, kc

    def set_history_csr(self, indptr, indices, values):
        """Per-user histories (items, weights) of an item-item model."""
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        pv, kv = self._any_ptr(values, np.float32)
        n_rows = len(indptr) - 1
        self._check(self._lib.ifr_set_history_csr(self._h, pp, pi, pv, n_rows, len(indices)))
        # any user index is valid for an item-item model: users past the last row have no history
        self.n_users = 2 ** 31 - 1
        del kp, ki, kv

    def item_fixed_point_shift(self):
        sh = ctypes.c_int(0)
        self._check(self._lib.ifr_item_fixed_point_shift(self._h, ctypes.byref(sh)))
        return sh.value

    def item_csr_layout(self):
        """(window words per row and window of the wave-per-user kernel's table -- 0: not built --, share of longer segments)."""
        w, f = ctypes.c_int(0), ctypes.c_double(0.0)
        self._check(self._lib.ifr_item_csr_layout(self._h, ctypes.byref(w), ctypes.byref(f)))
        return w.value, f.value

    def set_exclusion_csr(self, indptr, indices):
        self.exclusion_source = None     # ranking_eval notes here which matrix the device copy was made from
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        n_rows = len(indptr) - 1
        self._check(self._lib.ifr_set_exclusion_csr(self._h, pp, pi, n_rows, len(indices)))
        self.has_exclusions = n_rows > 0
        del kp, ki

    def set_truth_csr(self, indptr, indices):
        self.truth_source = None
        pp, kp = self._any_ptr(indptr, np.int64)
        pi, ki = self._any_ptr(indices, np.int32)
        n_rows = len(indptr) - 1
        self._check(self._lib.ifr_set_truth_csr(self._h, pp, pi, n_rows, len(indices)))
        self.has_truth = n_rows > 0
        del kp, ki

    def build_eval_csrs(self, train, val, test, rating_threshold, n_items, row0, n_rows):
        """ranking_eval's candidate-set builder on the device (ifr_build_eval_csrs; base_method.py:246-283): from the raw
        train / validation (or None) / test rating matrices -- `DeviceCsr`, or SciPy CSR matrices that are uploaded --
        the truth and exclusion CSRs of the users [row0, row0 + n_rows) are built and installed in this engine.
        -> dict(has_pos: uint8 device tensor [n_rows], truth_nnz, excl_nnz, max_truth_row)."""
        from ._lib import CsrSplit
        torch = _torch()
        splits = [None if m is None else DeviceCsr.of(m, self.device) for m in (train, val, test)]
        f64 = any(sp is not None and sp.data.dtype == torch.float64 for sp in splits)
        keep, cs = [], []
        for sp in splits:
            if sp is None:
                cs.append(None)
                continue
            data = sp.data.to(torch.float64) if f64 else sp.data
            keep.append(data)
            cs.append(CsrSplit(sp.indptr.data_ptr(), sp.indices.data_ptr(), data.data_ptr(), sp.shape[0]))
        has_pos = torch.zeros(max(int(n_rows), 0), dtype=torch.uint8, device=self.device)
        tn, en, mx = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
        ref = lambda c: None if c is None else ctypes.byref(c)
        self.exclusion_source = self.truth_source = None
        self._check(self._lib.ifr_build_eval_csrs(self._h, ref(cs[0]), ref(cs[1]), ref(cs[2]), int(f64), float(rating_threshold),
                                                  int(n_items), int(row0), int(n_rows), has_pos.data_ptr(),
                                                  ctypes.byref(tn), ctypes.byref(en), ctypes.byref(mx)))
        self.has_exclusions = self.has_truth = n_rows > 0
        del keep
        return dict(has_pos=has_pos, truth_nnz=tn.value, excl_nnz=en.value, max_truth_row=mx.value)

    def read_eval_csr(self, which):
        """(indptr int64, indices int32) of the truth (`which` = "truth") or exclusion CSR this engine holds, on the host."""
        w = {"truth": 0, "exclusion": 1}[which]
        rows, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.ifr_read_eval_csr(self._h, w, None, None, ctypes.byref(rows), ctypes.byref(nnz)))
        indptr = np.zeros(rows.value + 1, np.int64)
        indices = np.zeros(nnz.value, np.int32)
        self._check(self._lib.ifr_read_eval_csr(self._h, w, indptr.ctypes.data, indices.ctypes.data, None, None))
        return indptr, indices

    # ------------------------------------------------------------------ compute
    def rank_topk(self, user_ids, K, out_ids=None, out_scores=None, dense_batch_pairs=1 << 28):
        """Top-K (ids int32 [n,K], scores fp32 [n,K]) for the listed users: the fused kernel up to K = 128 (item-item
        models: 32 with a dense item matrix, 128 with a CSR one), dense rows + partial ranking up to K = 1024 (`dense_batch_pairs` bounds users x items of one
        dense batch)."""
        torch = _torch()
        u = self._dev_users(user_ids)
        n = u.numel()
        if out_ids is None:
            out_ids = torch.empty((n, K), dtype=torch.int32, device=self.device)
        if out_scores is None:
            out_scores = torch.empty((n, K), dtype=torch.float32, device=self.device)
        if K <= self.max_fused_topk():
            self._check(self._lib.ifr_rank_topk(self._h, u.data_ptr(), n, int(K), out_ids.data_ptr(),
                                                out_scores.data_ptr(), self._stream()))
            return out_ids, out_scores
        # cut-offs beyond the fused kernel: dense rows in batches + partial ranking (ifr_select_topk)
        step = max(1, int(dense_batch_pairs // max(1, self.catalogue_items)))
        buf = None
        for b in range(0, n, step):
            ub = u[b:b + step]
            dense = self.score_rows(ub, None if buf is None else buf[: ub.numel()])
            buf = dense if buf is None else buf
            self.select_topk(ub, dense, K, out_ids[b:b + step], out_scores[b:b + step])
        return out_ids, out_scores

    def max_fused_topk(self):
        """Largest K ifr_rank_topk serves for the model that is set."""
        if self.item_model is None:
            return MAX_FUSED_TOPK
        return MAX_FUSED_TOPK_ITEM_CSR if self.item_model == "csr" else MAX_FUSED_TOPK_ITEM

    def wide_overflows(self):
        """Users the 16 < K <= 128 path ranked from dense rows so far because their candidate buffer overflowed."""
        c = ctypes.c_int64(0)
        self._check(self._lib.ifr_wide_overflows(self._h, ctypes.byref(c)))
        return c.value

    def wide_stats(self):
        """(users ranked by the 16 < K <= 128 path, candidates appended, candidates kept after the final bound)."""
        u, c, k = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.ifr_wide_stats(self._h, ctypes.byref(u), ctypes.byref(c), ctypes.byref(k)))
        return u.value, c.value, k.value

    NOALLOC_MAX_USERS = 65536     # ifr_rank_topk_noalloc (IFR_NOALLOC_MAX_USERS)
    NOALLOC_MAX_TOPK = 16         # IFR_NOALLOC_MAX_TOPK

    def rank_topk_noalloc(self, user_ids, K, out_ids, out_scores):
        """ifr_rank_topk_noalloc: rank_topk for callers that cannot allocate (HIP graph capture, serving loops).  The
        library call allocates nothing -- its workspace is a static device array -- so `user_ids` (int32 device tensor)
        and both outputs must be the caller's own preallocated device tensors.  Dot-product models, K <= 16,
        at most NOALLOC_MAX_USERS users per call; calls that overlap on one GPU race (include/ifr.h).  User rows
        are not range-checked here (the check would allocate); the kernel clamps them into [0, n_users)."""
        torch = _torch()
        for t, dt in ((user_ids, torch.int32), (out_ids, torch.int32), (out_scores, torch.float32)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dt and t.is_contiguous()):
                raise ValueError("rank_topk_noalloc: user_ids / out_ids / out_scores must be contiguous device tensors (int32, int32, float32)")
        n = user_ids.numel()
        if tuple(out_ids.shape) != (n, int(K)) or tuple(out_scores.shape) != (n, int(K)):
            raise ValueError("rank_topk_noalloc: outputs must be [n, K]")
        self._check(self._lib.ifr_rank_topk_noalloc(self._h, user_ids.data_ptr(), n, int(K), out_ids.data_ptr(),
                                                    out_scores.data_ptr(), self._stream()))
        return out_ids, out_scores

    MAX_TOPK_WITH_DENSE = 128     # ifr_rank_topk_dense (= IFR_MAX_TOPK)

    def can_rank_topk_dense(self, K):
        """True when rank_topk_dense serves this model and cut-off in one kernel: a dot-product model (K <= 128) or a
        dense item matrix, fp32 or float64 (K <= 32), and no unknown catalogue items to pad."""
        if max(getattr(self, "catalogue_items", 0), self.n_items) != self.n_items:
            return False
        if self.item_model is None:
            return 1 <= int(K) <= self.MAX_TOPK_WITH_DENSE
        return self.item_model == "dense" and 1 <= int(K) <= 32

    def rank_topk_dense(self, user_ids, K, dense=None, out_ids=None, out_scores=None):
        """rank_topk and the score rows in one launch (one scoring pass instead of two): returns (ids [n,K], scores
        [n,K], dense [n, n_items] view).  Rows are fp32 with n_items rounded up to 32 columns, or -- a float64 dense item
        matrix -- the float64 sums (what score_rows returns), columns rounded up to 4."""
        torch = _torch()
        u = self._dev_users(user_ids)
        n = u.numel()
        f64 = self.item_model == "dense" and self.rows_are_f64()
        ld = (self.n_items + 3) // 4 * 4 if f64 else (self.n_items + 31) // 32 * 32
        dt = torch.float64 if f64 else torch.float32
        if dense is None:
            dense = torch.empty((n, ld), dtype=dt, device=self.device)
        if dense.dim() != 2 or dense.shape[0] < n or dense.shape[1] < ld or dense.stride(1) != 1 or dense.dtype != dt:
            raise ValueError(f"rank_topk_dense: dense must be {dt} [>= {n}, >= {ld}] with unit column stride")
        if out_ids is None:
            out_ids = torch.empty((n, K), dtype=torch.int32, device=self.device)
        if out_scores is None:
            out_scores = torch.empty((n, K), dtype=torch.float32, device=self.device)
        fn = self._lib.ifr_rank_topk_dense_f64 if f64 else self._lib.ifr_rank_topk_dense
        self._check(fn(self._h, u.data_ptr(), n, int(K), out_ids.data_ptr(), out_scores.data_ptr(),
                       dense.data_ptr(), dense.stride(0), self._stream()))
        return out_ids, out_scores, dense[:n, : self.n_items]

    def select_topk(self, user_ids, scores, K, out_ids=None, out_scores=None):
        """The K <= 1024 best candidates of dense score rows in the order of rank_full (partial ranking)."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        n = u.numel()
        if out_ids is None:
            out_ids = torch.empty((n, K), dtype=torch.int32, device=self.device)
        if out_scores is None:
            out_scores = torch.empty((n, K), dtype=torch.float32, device=self.device)
        if not (out_ids.is_contiguous() and out_scores.is_contiguous()):
            raise ValueError("select_topk: output tensors must be contiguous")
        fn = self._lib.ifr_select_topk_f64 if scores.dtype == torch.float64 else self._lib.ifr_select_topk
        self._check(fn(self._h, u.data_ptr(), n, scores.data_ptr(), scores.stride(0), int(K),
                       out_ids.data_ptr(), out_scores.data_ptr(), self._stream()))
        return out_ids, out_scores

    def eval_topk_metrics(self, user_ids, topk_ids, ks):
        """(per_user fp64 [n, 6*len(ks)] = ndcg/recall/precision/hit/f1/ncrr per cut-off,
        evaluated int32 [n], sums fp64 [6*len(ks)+1] = column sums + evaluated count)."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        n = u.numel()
        ks_arr = (ctypes.c_int32 * len(ks))(*[int(k) for k in ks])
        ncol = METRICS_PER_K * len(ks)
        per_user = torch.empty((n, ncol), dtype=torch.float64, device=self.device)
        evaluated = torch.empty((n,), dtype=torch.int32, device=self.device)
        sums = torch.empty((ncol + 1,), dtype=torch.float64, device=self.device)
        self._check(self._lib.ifr_eval_topk_metrics(
            self._h, u.data_ptr(), n, topk_ids.data_ptr(), int(topk_ids.shape[1]), ks_arr, len(ks),
            per_user.data_ptr(), evaluated.data_ptr(), sums.data_ptr(), self._stream()))
        return per_user, evaluated, sums

    def eval_ranked_metrics(self, user_ids, ranked, n_ranked, ks):
        """Metrics from full ranked lists (ks may hold -1 = whole list).  per_user has
        6*len(ks) + 2 columns (the last two: reciprocal rank of the first hit, average precision)."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        n = u.numel()
        ks_arr = (ctypes.c_int32 * len(ks))(*[int(k) for k in ks])
        ncol = METRICS_PER_K * len(ks) + 2
        per_user = torch.empty((n, ncol), dtype=torch.float64, device=self.device)
        evaluated = torch.empty((n,), dtype=torch.int32, device=self.device)
        sums = torch.empty((ncol + 1,), dtype=torch.float64, device=self.device)
        self._check(self._lib.ifr_eval_ranked_metrics(
            self._h, u.data_ptr(), n, ranked.data_ptr(), ranked.stride(0), n_ranked.data_ptr(), ks_arr, len(ks),
            per_user.data_ptr(), evaluated.data_ptr(), sums.data_ptr(), self._stream()))
        return per_user, evaluated, sums

    def eval_dense_metrics(self, user_ids, scores, ks):
        """The columns of eval_ranked_metrics straight from dense score rows (positions of the positives by
        counting, no ranking is built); rows with at most MAX_DENSE_METRIC_POS positives."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        n = u.numel()
        ks_arr = (ctypes.c_int32 * len(ks))(*[int(k) for k in ks])
        ncol = METRICS_PER_K * len(ks) + 2
        per_user = torch.empty((n, ncol), dtype=torch.float64, device=self.device)
        evaluated = torch.empty((n,), dtype=torch.int32, device=self.device)
        sums = torch.empty((ncol + 1,), dtype=torch.float64, device=self.device)
        fn = self._lib.ifr_eval_dense_metrics_f64 if scores.dtype == torch.float64 else self._lib.ifr_eval_dense_metrics
        self._check(fn(self._h, u.data_ptr(), n, scores.data_ptr(), scores.stride(0), ks_arr, len(ks),
                       per_user.data_ptr(), evaluated.data_ptr(), sums.data_ptr(), self._stream()))
        return per_user, evaluated, sums

    def rank_items(self, scores_row, items):
        """Order `items` (int32 device tensor) by scores_row[item] descending, item ascending."""
        torch = _torch()
        it = self._dev_i32(items)
        out = torch.empty_like(it)
        fn = self._lib.ifr_rank_items_f64 if scores_row.dtype == torch.float64 else self._lib.ifr_rank_items
        self._check(fn(self._h, scores_row.data_ptr(), scores_row.numel(), it.data_ptr(), it.numel(), out.data_ptr(),
                       self._stream()))
        return out

    def set_catalogue_items(self, n_total):
        """Rank / evaluate over n_total >= n_items items: the extra ones are unknown to the model and
        score the user's minimum (recommender.py:631-635).  0 or n_items switches it off."""
        n_total = int(n_total)
        self._check(self._lib.ifr_set_option(self._h, b"catalogue_items", 0 if n_total == self.n_items else n_total))
        self.catalogue_items = max(n_total, self.n_items)

    def rows_are_f64(self):
        """True for a dense item matrix installed with precision="f64" and for every CSR item matrix (whose fixed-point
        sums are exact in float64): the native score rows are float64, and everything that ranks them orders the
        float64 values, like the reference does with its float64 scores (recom_ease.py:123-124,
        graph_recommender.py:104-117) -- and like the fused top-K kernels of these models."""
        return (self.item_model == "dense" and getattr(self, "item_precision", "f32") == "f64") or self.item_model == "csr"

    def max_dense_metric_pos(self):
        """Positives per row ifr_eval_dense_metrics[_f64] takes (float64 rows: 128-bit keys, half as many fit the LDS)."""
        return 2048 if self.rows_are_f64() else MAX_DENSE_METRIC_POS

    def row_bytes_per_item(self):
        return 8 if self.rows_are_f64() else 4

    def new_score_rows(self, n_rows):
        """An uninitialised [n_rows, catalogue] buffer for score_rows(out=...) in the model's native score type (fp32;
        float64 for precision="f64"): callers that loop over batches allocate it once (two live buffers would double
        the footprint the batch size was chosen for)."""
        torch = _torch()
        cols = max(getattr(self, "catalogue_items", 0), self.n_items)
        cols = (cols + 3) // 4 * 4                         # 16-byte aligned rows: the consumers read 4 scores at a time
        return torch.empty((int(n_rows), cols), dtype=torch.float64 if self.rows_are_f64() else torch.float32,
                           device=self.device)

    def score_rows(self, user_ids, out=None):
        """Score rows [n, catalogue] in the model's native score type: score_dense (fp32), or for precision="f64"
        the float64 sums (ifr_score_dense_f64), padded with the row minimum for unknown catalogue items.  auc_dense,
        eval_dense_metrics, rank_full, select_topk and rank_items take either and rank what they are given."""
        if not self.rows_are_f64():
            return self.score_dense(user_ids, out)
        torch = _torch()
        u = self._dev_users(user_ids)
        n = u.numel()
        cols = max(getattr(self, "catalogue_items", 0), self.n_items)
        if out is None:
            out = torch.empty((n, (cols + 3) // 4 * 4), dtype=torch.float64, device=self.device)
        self._check(self._lib.ifr_score_dense_f64(self._h, u.data_ptr(), n, out.data_ptr(), out.stride(0), self._stream()))
        if cols > self.n_items:
            self._check(self._lib.ifr_pad_scores_min_f64(self._h, out.data_ptr(), n, out.stride(0), self._stream()))
        return out[:n, :cols]

    def score_dense(self, user_ids, out=None):
        """Score rows [n, catalogue] (catalogue = n_items unless set_catalogue_items widened it)."""
        torch = _torch()
        u = self._dev_users(user_ids)
        n = u.numel()
        cols = max(getattr(self, "catalogue_items", 0), self.n_items)
        if out is None:
            out = torch.empty((n, cols), dtype=torch.float32, device=self.device)
        self._check(self._lib.ifr_score_dense(self._h, u.data_ptr(), n, out.data_ptr(), out.stride(0), self._stream()))
        if cols > self.n_items:
            self._check(self._lib.ifr_pad_scores_min(self._h, out.data_ptr(), n, out.stride(0), self._stream()))
        return out

    def score_dense_f64(self, user_ids, out=None):
        """float64 score rows [n, n_items] of a dense item matrix installed with precision="f64", or of a CSR item matrix
        (its exact fixed-point sums scaled back)."""
        torch = _torch()
        u = self._dev_users(user_ids)
        n = u.numel()
        if out is None:
            out = torch.empty((n, self.n_items), dtype=torch.float64, device=self.device)
        self._check(self._lib.ifr_score_dense_f64(self._h, u.data_ptr(), n, out.data_ptr(), out.stride(0), self._stream()))
        return out

    def auc_dense(self, user_ids, scores):
        """AUC per user from dense score rows (fp32, or float64 rows of a precision="f64" model)."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        out = torch.empty((u.numel(),), dtype=torch.float64, device=self.device)
        fn = self._lib.ifr_auc_dense_f64 if scores.dtype == torch.float64 else self._lib.ifr_auc_dense
        self._check(fn(self._h, u.data_ptr(), u.numel(), scores.data_ptr(), scores.stride(0), out.data_ptr(), self._stream()))
        return out

    def rank_full(self, user_ids, scores):
        """Full canonical ranking of dense rows: (ids int32 [n, ld] padded with -1, n_cand int32 [n])."""
        torch = _torch()
        u = self._dev_i32(user_ids)
        n = u.numel()
        ids = torch.empty((n, scores.stride(0)), dtype=torch.int32, device=self.device)
        n_cand = torch.empty((n,), dtype=torch.int32, device=self.device)
        fn = self._lib.ifr_rank_full_f64 if scores.dtype == torch.float64 else self._lib.ifr_rank_full
        self._check(fn(self._h, u.data_ptr(), n, scores.data_ptr(), scores.stride(0), ids.data_ptr(), n_cand.data_ptr(),
                       self._stream()))
        return ids, n_cand

    # ------------------------------------------------------------------ introspection
    def chain_order(self):
        perm = (ctypes.c_int32 * 256)()
        n = ctypes.c_int(0)
        self._check(self._lib.ifr_factor_chain_order(self._h, perm, ctypes.byref(n)))
        return np.asarray(perm[: n.value], dtype=np.int32)

    def profile_enable(self, on=True):
        self._check(self._lib.ifr_profile_enable(self._h, int(bool(on))))

    def profile_read(self):
        ms, launches, pairs = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        self._check(self._lib.ifr_profile_read(self._h, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(pairs)))
        return ms.value, launches.value, pairs.value
