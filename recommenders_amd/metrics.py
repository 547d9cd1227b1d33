"""Ranking-metric descriptors with the reference's names, attributes and compute() signatures
(cornac/metrics/ranking.py).  ranking_eval only reads `.name` / `.k` / the class name, so the
reference's own metric objects can be passed instead.  compute() runs the HIP metric kernels on
a one-user batch (no CPU implementation lives in this package)."""
import numpy as np


class RankingMetric:
    """ranking.py:20-46."""

    def __init__(self, name=None, k=-1, higher_better=True):
        assert hasattr(k, "__len__") or k == -1 or k > 0
        self.type = "ranking"
        self.name = name
        self.k = k
        self.higher_better = higher_better

    def compute(self, **kwargs):
        raise NotImplementedError()


_SCORER = None


def _scorer():
    """The one engine handle the compute() methods share (a handle owns a HIP context entry and workspaces; every
    compute() call installs the state it needs, so nothing is carried from one call to the next)."""
    global _SCORER
    if _SCORER is None or _SCORER._h is None:
        from .engine import HipScorer
        _SCORER = HipScorer(0)
    return _SCORER


def _single_user_list_metrics(gt_pos, pd_rank, k):
    """Six per-cut-off values + (mrr, ap) for one ranked list through ifr_eval_ranked_metrics."""
    import torch
    sc = _scorer()
    gt = np.unique(np.asarray(gt_pos, dtype=np.int32))
    sc.set_truth_csr(np.asarray([0, len(gt)], np.int64), gt)
    ranked = torch.as_tensor(np.asarray(pd_rank, dtype=np.int32).reshape(1, -1), device=sc.device)
    n_ranked = torch.tensor([ranked.shape[1]], dtype=torch.int32, device=sc.device)
    per_user, _, _ = sc.eval_ranked_metrics(np.zeros(1, np.int32), ranked, n_ranked, [k])
    return per_user[0].cpu().numpy()


class _AtK(RankingMetric):
    _col = 0
    _fmt = "{}"

    def __init__(self, k=-1):
        RankingMetric.__init__(self, name=self._fmt.format(k), k=k)

    def compute(self, gt_pos, pd_rank, **kwargs):
        return float(_single_user_list_metrics(gt_pos, pd_rank, self.k)[self._col])


class NDCG(_AtK):
    """ranking.py:49-123."""
    _col, _fmt = 0, "NDCG@{}"


class Recall(_AtK):
    """ranking.py:352-386."""
    _col, _fmt = 1, "Recall@{}"


class Precision(_AtK):
    """ranking.py:315-349."""
    _col, _fmt = 2, "Precision@{}"


class HitRatio(_AtK):
    """ranking.py:277-312."""
    _col, _fmt = 3, "HitRatio@{}"


class FMeasure(_AtK):
    """ranking.py:389-431."""
    _col, _fmt = 4, "F1@{}"


class NCRR(_AtK):
    """ranking.py:126-180."""
    _col, _fmt = 5, "NCRR@{}"


class MRR(RankingMetric):
    """ranking.py:183-223."""

    def __init__(self):
        RankingMetric.__init__(self, name="MRR")

    def compute(self, gt_pos, pd_rank, **kwargs):
        v = float(_single_user_list_metrics(gt_pos, pd_rank, -1)[6])
        if np.isnan(v):
            raise ValueError("No matched between ground-truth items and recommendations")
        return v


class MAP(RankingMetric):
    """ranking.py:488-527 (average precision of one user; exact on tie-free scores)."""

    def __init__(self):
        RankingMetric.__init__(self, name="MAP")

    def compute(self, item_indices, pd_scores, gt_pos, **kwargs):
        import torch
        sc = _scorer()
        items = np.asarray(item_indices, dtype=np.int32)
        row = torch.full((int(items.max()) + 1,), float("-inf"), dtype=torch.float32, device=sc.device)
        row[torch.as_tensor(items.astype(np.int64), device=sc.device)] = torch.as_tensor(
            np.asarray(pd_scores, dtype=np.float32), device=sc.device)
        ranked = sc.rank_items(row, items).cpu().numpy()
        return float(_single_user_list_metrics(gt_pos, ranked, -1)[7])


class AUC(RankingMetric):
    """ranking.py:434-485."""

    def __init__(self):
        RankingMetric.__init__(self, name="AUC")

    def compute(self, item_indices, pd_scores, gt_pos, gt_neg=None, **kwargs):
        import torch
        sc = _scorer()
        items = np.asarray(item_indices, dtype=np.int64)
        n_items = int(items.max()) + 1
        scores = np.zeros(n_items, np.float32)
        scores[items] = np.asarray(pd_scores, dtype=np.float32)
        pos = np.unique(np.asarray(gt_pos, dtype=np.int32))
        pos = pos[np.isin(pos, items)]
        neg = np.setdiff1d(items, pos) if gt_neg is None else np.intersect1d(items, np.asarray(gt_neg))
        excl = np.setdiff1d(np.arange(n_items), np.union1d(pos, neg)).astype(np.int32)
        sc.set_factors(np.zeros((1, 1), np.float32), np.zeros((n_items, 1), np.float32))   # sizes the catalogue
        sc.set_truth_csr(np.asarray([0, len(pos)], np.int64), pos)
        sc.set_exclusion_csr(np.asarray([0, len(excl)], np.int64), excl)
        row = torch.as_tensor(scores.reshape(1, -1), device=sc.device)
        return float(sc.auc_dense(np.zeros(1, np.int32), row)[0].item())


# (kind, needs) for a metric object, by class name so that the reference's classes work too
_KINDS = {"NDCG": "ndcg", "Recall": "recall", "Precision": "precision", "HitRatio": "hit", "FMeasure": "f1",
          "NCRR": "ncrr", "AUC": "auc", "MRR": "mrr", "MAP": "map"}


def metric_kind(mt):
    for cls in type(mt).__mro__:
        if cls.__name__ in _KINDS:
            return _KINDS[cls.__name__]
    raise TypeError(f"unsupported ranking metric {type(mt).__name__}: the HIP path covers {sorted(_KINDS)}")
