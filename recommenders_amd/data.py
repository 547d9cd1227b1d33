"""Minimal host-side stand-in for the parts of cornac.data.Dataset the hot path reads
(cornac/data/dataset.py:229-238 csr_matrix, :260-362 build).  ranking_eval and the model
facades are duck-typed: a real cornac Dataset works unchanged; this class exists so the
package is usable (and testable) without the reference installed."""
from collections import OrderedDict

import numpy as np
import scipy.sparse as sp


class Dataset:
    def __init__(self, num_users, num_items, uid_map, iid_map, uir_tuple):
        self.num_users = int(num_users)
        self.num_items = int(num_items)
        self.uid_map = uid_map
        self.iid_map = iid_map
        self.uir_tuple = uir_tuple
        r = uir_tuple[2]
        self.num_ratings = len(r)
        self.max_rating = float(np.max(r)) if len(r) else 0.0
        self.min_rating = float(np.min(r)) if len(r) else 0.0
        self.global_mean = float(np.mean(r)) if len(r) else 0.0
        self._csr = None

    @classmethod
    def from_uir(cls, data, global_uid_map=None, global_iid_map=None, exclude_unknowns=False):
        """Build from (user_id, item_id, rating) tuples, extending the given global id maps in
        place like Dataset.build (dataset.py:300-340): unknown ids get the next free index, or
        the tuple is dropped when exclude_unknowns."""
        uid_map = OrderedDict() if global_uid_map is None else global_uid_map
        iid_map = OrderedDict() if global_iid_map is None else global_iid_map
        us, it, rs = [], [], []
        seen = set()
        for uid, iid, r in data:
            if exclude_unknowns and (uid not in uid_map or iid not in iid_map):
                continue
            if (uid, iid) in seen:
                continue
            seen.add((uid, iid))
            us.append(uid_map.setdefault(uid, len(uid_map)))
            it.append(iid_map.setdefault(iid, len(iid_map)))
            rs.append(float(r))
        uir = (np.asarray(us, dtype=np.int64), np.asarray(it, dtype=np.int64), np.asarray(rs, dtype=np.float64))
        nu = len(set(us)) if exclude_unknowns else len(uid_map)
        ni = len(set(it)) if exclude_unknowns else len(iid_map)
        ds = cls(len(uid_map), len(iid_map), uid_map, iid_map, uir)
        ds._seen_users, ds._seen_items = nu, ni
        return ds

    @classmethod
    def from_csr(cls, csr):
        csr = sp.csr_matrix(csr)
        coo = csr.tocoo()
        ds = cls(csr.shape[0], csr.shape[1], OrderedDict((i, i) for i in range(csr.shape[0])),
                 OrderedDict((i, i) for i in range(csr.shape[1])),
                 (coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data.astype(np.float64)))
        return ds

    @property
    def matrix(self):
        return self.csr_matrix

    @property
    def csr_matrix(self):
        if self._csr is None:
            u, i, r = self.uir_tuple
            m = sp.csr_matrix((r, (u, i)), shape=(self.num_users, self.num_items))
            m.sort_indices()
            self._csr = m
        return self._csr

    def reset(self):
        return self


class DeviceDataset:
    """A split whose rating matrix lives in device memory (`recommenders_amd.engine.DeviceCsr`): the three things
    ranking_eval reads of a Dataset -- num_users, num_items, csr_matrix (cornac/data/dataset.py:229-238) -- for splits
    too large for host-side sparse algebra (one GPU's shard of the 10 M x 1 M configuration: 75 M ratings)."""

    def __init__(self, csr):
        self.csr_matrix = csr
        self.num_users, self.num_items = csr.shape
        self.num_ratings = csr.nnz

    @property
    def matrix(self):
        return self.csr_matrix
