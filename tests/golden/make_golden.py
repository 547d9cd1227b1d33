"""Generate golden fixtures by running the REFERENCE itself (build container only).

Usage (in the container that has /root/reference; never on the GPU box):

    python tests/golden/make_golden.py [--scratch /tmp/oracle]

What it does
  1. copies /root/reference/cornac into a scratch directory OUTSIDE the repo and
     builds the five Cython extensions the hot path needs;
  2. pre-seeds ``sys.modules`` with light stub packages so the reference's
     sub-modules import without its heavy ``__init__`` files (seaborn / tensorflow
     / powerlaw are absent here);
  3. fits the reference's BPR / MF / EASE / RP3_Beta (and WMF with
     ``trainable=False``) on seeded synthetic data and records the inputs and the
     reference's outputs of ``model.score``, ``model.rank`` and ``ranking_eval``
     as small ``.npz`` files next to this script.

Only arrays are written: no reference source, bytecode or binary enters the repo.
"""
import argparse
import importlib
import os
import shutil
import subprocess
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

SETUP_MIN = r'''
from setuptools import Extension, setup
from Cython.Build import cythonize
import numpy as np
ca = ["-Wno-unused-function", "-Wno-maybe-uninitialized", "-O3", "-ffast-math", "-fopenmp", "-std=c++11"]
la = ["-fopenmp", "-std=c++11"]
exts = [
    Extension("cornac.utils.fast_dot", ["cornac/utils/fast_dot.pyx"], language="c++", extra_compile_args=ca, extra_link_args=la),
    Extension("cornac.utils.fast_sparse_funcs", ["cornac/utils/fast_sparse_funcs.pyx"], include_dirs=[np.get_include()], language="c++"),
    Extension("cornac.utils.fast_dict", ["cornac/utils/fast_dict.pyx"], include_dirs=[np.get_include()], language="c++"),
    Extension("cornac.models.mf.backend_cpu", ["cornac/models/mf/backend_cpu.pyx"], include_dirs=[np.get_include()], language="c++", extra_compile_args=ca, extra_link_args=la),
    Extension("cornac.models.bpr.recom_bpr", ["cornac/models/bpr/recom_bpr.pyx"], include_dirs=[np.get_include(), "cornac/utils/external"], language="c++", extra_compile_args=ca, extra_link_args=la),
]
setup(name="oracle_min", ext_modules=cythonize(exts, language_level=3), script_args=["build_ext", "--inplace"])
'''


def prepare_scratch(scratch):
    """Scratch copy + extension build (flags mirror reference setup.py:78-104)."""
    if not os.path.isdir(os.path.join(scratch, "cornac")):
        os.makedirs(scratch, exist_ok=True)
        shutil.copytree(os.path.join(REF, "cornac"), os.path.join(scratch, "cornac"))
    import glob
    if not glob.glob(os.path.join(scratch, "cornac", "utils", "fast_dot*.so")):
        with open(os.path.join(scratch, "setup_min.py"), "w") as f:
            f.write(SETUP_MIN)
        subprocess.check_call([sys.executable, "setup_min.py"], cwd=scratch)


def import_reference(scratch):
    sys.path.insert(0, scratch)
    sys.modules["powerlaw"] = types.ModuleType("powerlaw")

    def stub(name):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(scratch, *name.split("."))]
        m.__package__ = name
        sys.modules[name] = m
        return m

    stub("cornac")
    u = stub("cornac.utils")
    mo = stub("cornac.models")
    for sub in ["bpr", "mf", "wmf", "ease", "rp3_beta"]:
        stub("cornac.models." + sub)
    common = importlib.import_module("cornac.utils.common")
    for n in ["get_rng", "validate_format", "estimate_batches", "normalize"]:
        setattr(u, n, getattr(common, n))
    u.common = common
    u.fast_dot = importlib.import_module("cornac.utils.fast_dot").fast_dot
    rec = importlib.import_module("cornac.models.recommender")
    for n in ["Recommender", "NextBasketRecommender", "NextItemRecommender"]:
        setattr(mo, n, getattr(rec, n))
    ns = types.SimpleNamespace()
    ns.BPR = importlib.import_module("cornac.models.bpr.recom_bpr").BPR
    ns.MF = importlib.import_module("cornac.models.mf.recom_mf").MF
    ns.WMF = importlib.import_module("cornac.models.wmf.recom_wmf").WMF
    ns.EASE = importlib.import_module("cornac.models.ease.recom_ease").EASE
    ns.RP3 = importlib.import_module("cornac.models.rp3_beta.recom_rp3beta").RP3_Beta
    importlib.import_module("cornac.eval_methods")
    ns.ranking_eval = importlib.import_module("cornac.eval_methods.base_method").ranking_eval
    ns.RatioSplit = importlib.import_module("cornac.eval_methods").RatioSplit
    ns.BaseMethod = importlib.import_module("cornac.eval_methods").BaseMethod
    ns.Dataset = importlib.import_module("cornac.data").Dataset
    met = importlib.import_module("cornac.metrics")
    ns.NDCG, ns.Recall, ns.Precision, ns.AUC = met.NDCG, met.Recall, met.Precision, met.AUC
    ns.MRR, ns.MAP, ns.NCRR, ns.HitRatio, ns.FMeasure = met.MRR, met.MAP, met.NCRR, met.HitRatio, met.FMeasure
    ns.fast_dot = u.fast_dot
    return ns


# ----------------------------------------------------------------------------
def synth_uir(rng, n_users, n_items, nnz, zipf=1.2, ratings=(1, 2, 3, 4, 5)):
    u = rng.integers(0, n_users, nnz)
    i = (rng.zipf(zipf, nnz) - 1) % n_items
    r = rng.choice(np.asarray(ratings, dtype=np.float64), nnz)
    seen = {}
    for a, b, c in zip(u, i, r):
        seen[(int(a), int(b))] = float(c)
    return [(str(a), str(b), c) for (a, b), c in seen.items()]


def csr_pack(prefix, csr, out):
    out[prefix + "_indptr"] = csr.indptr.astype(np.int64)
    out[prefix + "_indices"] = csr.indices.astype(np.int32)
    out[prefix + "_data"] = csr.data.astype(np.float32)
    out[prefix + "_shape"] = np.asarray(csr.shape, np.int64)


def metric_list(ns, spec):
    out = []
    for name, k in spec:
        if name == "ndcg":
            out.append(ns.NDCG(k=k))
        elif name == "recall":
            out.append(ns.Recall(k=k))
        elif name == "precision":
            out.append(ns.Precision(k=k))
        elif name == "auc":
            out.append(ns.AUC())
        elif name == "mrr":
            out.append(ns.MRR())
        elif name == "map":
            out.append(ns.MAP())
        elif name == "ncrr":
            out.append(ns.NCRR(k=k))
        elif name == "hit":
            out.append(ns.HitRatio(k=k))
        elif name == "f1":
            out.append(ns.FMeasure(k=k))
    return out


def run_eval(ns, model, spec, split, head=20, n_score_users=6, head_len=None):
    """reference ranking_eval + per-user heads of model.ranked_items."""
    head = head if head_len is None else head_len
    mets = metric_list(ns, spec)
    avg, per = ns.ranking_eval(model, mets, split.train_set, split.test_set,
                               val_set=split.val_set, rating_threshold=split.rating_threshold,
                               exclude_unknowns=split.exclude_unknowns)
    users = np.asarray(sorted(per[0].keys()), np.int32)
    vals = np.asarray([[per[m][int(u)] for m in range(len(mets))] for u in users], np.float64)
    ranked = np.full((len(users), head), -1, np.int32)
    rscore = np.full((len(users), head), -np.inf, np.float64)
    n_cand = np.zeros(len(users), np.int32)
    for r, u in enumerate(users):
        ri = model.ranked_items[int(u)]
        cand = model.item_scores_mapped_indices[int(u)]
        sc = model.item_scores[int(u)]
        lut = dict(zip(cand.tolist(), np.asarray(sc, np.float64).tolist()))
        h = min(head, len(ri))
        ranked[r, :h] = ri[:h]
        rscore[r, :h] = [lut[int(x)] for x in ri[:h]]
        n_cand[r] = len(ri)
    score_users = users[:: max(1, len(users) // n_score_users)][:n_score_users]
    score_vecs = np.stack([np.asarray(model.score(int(u)), np.float64).ravel() for u in score_users])
    return dict(avg=np.asarray(avg, np.float64), users=users, per_user=vals, ranked_head=ranked,
                ranked_head_scores=rscore, n_cand=n_cand, score_users=score_users.astype(np.int32),
                score_vecs=score_vecs)


def pack_split(split, out):
    csr_pack("train", split.train_set.csr_matrix, out)
    csr_pack("test", split.test_set.csr_matrix, out)
    if split.val_set is not None:
        csr_pack("val", split.val_set.csr_matrix, out)
    out["rating_threshold"] = np.float64(split.rating_threshold)
    out["exclude_unknowns"] = np.int32(split.exclude_unknowns)
    out["num_users"] = np.int32(split.train_set.num_users)
    out["num_items"] = np.int32(split.train_set.num_items)
    out["total_users"] = np.int32(split.total_users)
    out["total_items"] = np.int32(split.total_items)
    out["global_mean"] = np.float64(split.train_set.global_mean)


def _packed_test(split):
    d = {}
    csr_pack("test", split.test_set.csr_matrix, d)
    return d


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def add_result(out, tag, res):
    for k, v in res.items():
        out[f"{tag}__{k}"] = v



# ----------------------------------------------------------------------------
# Gap-guaranteed fixtures ("Tie semantics" / "fp32 accumulation order").
# The reference's order among (near-)equal scores is implementation-defined (BLAS summation order, NumPy's unstable
# argsort), so bit-exact ids and 1e-6 metrics are only meaningful on users whose decisive scores are separated by more
# than any rounding can move them.  These helpers keep exactly those users in the TEST split (the train / validation
# splits and the fitted model are untouched), so that a parity test can assert on EVERY evaluated user, no filter.
GAP_REL = 1e-4


def user_gap_ok(cand, scores, ranked, pos, head, n_known, positions, rel=GAP_REL):
    """True when, for this user of the reference run,
      * the top-(head + 1) candidate scores are pairwise separated by more than rel * scale, and
      * every test positive is separated by more than rel * scale from every other candidate --
        exact ties are tolerated only where both sides tie in EVERY implementation (two items unknown to the model,
        which share the padded minimum, recommender.py:631-635, or two scores that are exactly 0.0: items an item-item
        model cannot reach) and only when `positions` is False (AUC counts strict wins, ranking.py:485, so such ties
        count 0 everywhere; metrics that read the POSITION of a positive would see the unspecified tie order).
    scale = max(|s_a|, |s_b|, 0.1 * max|s| of the user): fp32 dot products carry errors relative to sum |u_f v_f|,
    not to the (possibly cancelling) score."""
    sc = np.asarray(scores, np.float64)
    cand = np.asarray(cand)
    if not np.isfinite(sc).all():
        return False
    smax = np.abs(sc).max() if len(sc) else 0.0
    lut = dict(zip(cand.tolist(), sc.tolist()))
    hs = np.asarray([lut[int(i)] for i in ranked[: head + 1]])
    if len(hs) > 1:
        tol = rel * np.maximum(np.maximum(np.abs(hs[:-1]), np.abs(hs[1:])), 0.1 * smax)
        if not ((hs[:-1] - hs[1:]) > tol).all():
            return False
    where = np.flatnonzero(np.isin(cand, pos))
    for j in where:
        d = np.abs(sc - sc[j])
        tol = rel * np.maximum(np.maximum(np.abs(sc), abs(sc[j])), 0.1 * smax)
        close = d <= tol
        close[j] = False
        if not positions:
            structural = (sc == sc[j]) & (((cand >= n_known) & (cand[j] >= n_known)) | ((sc == 0.0) & (sc[j] == 0.0)))
            close &= ~structural
        if close.any():
            return False
    return True


def gap_ok_users(model, split, head, positions):
    """Evaluated users of the reference run whose cached ranking passes user_gap_ok."""
    test_mat = split.test_set.csr_matrix
    thr = split.rating_threshold
    ok = []
    for u in sorted(model.ranked_items.keys()):
        row = test_mat.getrow(u)
        pos = row.indices[row.data >= thr]
        if user_gap_ok(model.item_scores_mapped_indices[u], model.item_scores[u], model.ranked_items[u], pos, head,
                       model.num_items, positions):
            ok.append(u)
    return np.asarray(ok, np.int64)


def clear_caches(model):
    model.ranked_items, model.item_scores, model.item_scores_mapped_indices = {}, {}, {}


def with_test_users(ns, split, keep_users):
    """A copy of `split` whose test set only holds the interactions of `keep_users` (same id maps and shapes)."""
    import copy
    ts = split.test_set
    u, i, r = ts.uir_tuple
    keep = np.isin(u, keep_users)
    out = copy.copy(split)
    out.test_set = ns.Dataset(ts.num_users, ts.num_items, ts.uid_map, ts.iid_map, (u[keep], i[keep], r[keep]))
    return out


def gap_filtered_eval(ns, models, spec, split, head, positions, **kw):
    """Reference evaluation of every model on the test users that pass the gap rule for ALL of `models`
    -> (filtered split, {name: run_eval result})."""
    keep = None
    for m in models.values():
        clear_caches(m)
        ns.ranking_eval(m, metric_list(ns, spec), split.train_set, split.test_set, val_set=split.val_set,
                        rating_threshold=split.rating_threshold, exclude_unknowns=split.exclude_unknowns)
        ok = gap_ok_users(m, split, head, positions)
        print(f"    {type(m).__name__}: {len(ok)} of {len(m.ranked_items)} evaluated users pass the gap rule")
        keep = ok if keep is None else np.intersect1d(keep, ok)
    fsplit = with_test_users(ns, split, keep)
    res = {}
    for name, m in models.items():
        clear_caches(m)
        res[name] = run_eval(ns, m, spec, fsplit, **kw)
        assert np.array_equal(res[name]["users"], keep), "the filtered evaluation covers other users than the gap rule kept"
    return fsplit, res


# ----------------------------------------------------------------------------
def make_toy(ns):
    """300x500 toy, ratings 1-5, threshold 3, test 0.2 + val 0.1; all five models."""
    rng = np.random.default_rng(123)
    data = synth_uir(rng, 300, 500, 12000)
    rs = ns.RatioSplit(data=data, test_size=0.2, val_size=0.1, rating_threshold=3.0, seed=123,
                       exclude_unknowns=True, verbose=False)
    spec = [("ndcg", 10), ("recall", 10), ("precision", 10), ("auc", -1), ("ndcg", 5), ("recall", 20)]
    out = {"metric_names": np.asarray([f"{n}@{k}" for n, k in spec])}
    pack_split(rs, out)
    nu, ni = rs.train_set.num_users, rs.train_set.num_items

    bpr = ns.BPR(k=50, max_iter=30, seed=123)
    bpr.fit(rs.train_set)
    out["bpr_U"], out["bpr_V"], out["bpr_Bi"] = bpr.u_factors, bpr.i_factors, bpr.i_biases
    add_result(out, "bpr", run_eval(ns, bpr, spec, rs))

    mf = ns.MF(k=10, max_iter=10, seed=123, use_bias=True)
    mf.fit(rs.train_set)
    out["mf_U"], out["mf_V"], out["mf_Bu"], out["mf_Bi"] = mf.u_factors, mf.i_factors, mf.u_biases, mf.i_biases
    out["mf_mu"] = np.float32(mf.global_mean)
    add_result(out, "mf", run_eval(ns, mf, spec, rs))

    Uw = rng.normal(0, 0.1, (nu, 128)).astype(np.float32)
    Vw = rng.normal(0, 0.1, (ni, 128)).astype(np.float32)
    wmf = ns.WMF(k=128, trainable=False, init_params={"U": Uw, "V": Vw}, verbose=False)
    wmf.fit(rs.train_set)
    out["wmf_U"], out["wmf_V"] = Uw, Vw
    add_result(out, "wmf", run_eval(ns, wmf, spec, rs))

    ease = ns.EASE(lamb=50, verbose=False)
    ease.fit(rs.train_set)
    out["ease_B"] = ease.B.astype(np.float64)
    add_result(out, "ease", run_eval(ns, ease, spec, rs))

    rp3 = ns.RP3(beta=0.7)
    rp3.fit(rs.train_set)
    out["rp3_beta"] = np.float64(0.7)
    add_result(out, "rp3", run_eval(ns, rp3, spec, rs))
    save("toy_all_models.npz", out)


def make_gap(ns):
    """The north-star bar on ALL evaluated users: 600 x 700 split, ratings 1-5, threshold 3, test 0.2 +
    val 0.1, all five models; for every model its own test split = the users whose reference run passes the gap rule
    above at head = 20 (so top-20 ids, NDCG/Recall/Precision at 5/10/20 and AUC are decided by gaps > 1e-4 relative)."""
    rng = np.random.default_rng(2025)
    data = synth_uir(rng, 600, 700, 30000)
    rs = ns.RatioSplit(data=data, test_size=0.2, val_size=0.1, rating_threshold=3.0, seed=2025,
                       exclude_unknowns=True, verbose=False)
    spec = [("ndcg", 10), ("recall", 10), ("precision", 10), ("auc", -1), ("ndcg", 5), ("recall", 20), ("ndcg", 20)]
    out = {"metric_names": np.asarray([f"{n}@{k}" for n, k in spec]), "gap_rel": np.float64(GAP_REL), "gap_head": np.int32(20)}
    pack_split(rs, out)
    nu, ni = rs.train_set.num_users, rs.train_set.num_items
    models = {}
    models["bpr"] = ns.BPR(k=50, max_iter=30, seed=123)
    models["mf"] = ns.MF(k=10, max_iter=100, learning_rate=0.02, seed=123, use_bias=True)   # 10 epochs leave the scores within 1.5 of the mean: few gaps
    Uw = rng.normal(0, 0.1, (nu, 128)).astype(np.float32)
    Vw = rng.normal(0, 0.1, (ni, 128)).astype(np.float32)
    models["wmf"] = ns.WMF(k=128, trainable=False, init_params={"U": Uw, "V": Vw}, verbose=False)
    models["ease"] = ns.EASE(lamb=50, verbose=False)
    models["rp3"] = ns.RP3(beta=0.7)
    for m in models.values():
        m.fit(rs.train_set)
    out["bpr_U"], out["bpr_V"], out["bpr_Bi"] = models["bpr"].u_factors, models["bpr"].i_factors, models["bpr"].i_biases
    mf = models["mf"]
    out["mf_U"], out["mf_V"], out["mf_Bu"], out["mf_Bi"] = mf.u_factors, mf.i_factors, mf.u_biases, mf.i_biases
    out["mf_mu"] = np.float32(mf.global_mean)
    out["wmf_U"], out["wmf_V"] = Uw, Vw
    out["ease_B"] = models["ease"].B.astype(np.float64)
    out["rp3_beta"] = np.float64(0.7)
    for name, m in models.items():
        fsplit, res = gap_filtered_eval(ns, {name: m}, spec, rs, head=20, positions=False, head_len=21, n_score_users=4)
        csr_pack(f"{name}__test", fsplit.test_set.csr_matrix, out)
        add_result(out, name, res[name])
        print(f"  gap fixture, {name}: {len(res[name]['users'])} users keep the gap rule")
    save("gap_all_models.npz", out)


def make_c1(ns):
    """BASELINE.json configs[0]: BPR k=50, MovieLens-100K *shape* (943 x 1682, ~100K
    ratings; the real file needs network), RatioSplit 0.2, threshold 4.0, seed 123
    (docs/source/user/quickstart.rst:269-277), NDCG@10 / Recall@10."""
    rng = np.random.default_rng(100)
    data = synth_uir(rng, 943, 1682, 125000, zipf=1.15)
    rs = ns.RatioSplit(data=data, test_size=0.2, rating_threshold=4.0, seed=123,
                       exclude_unknowns=True, verbose=False)
    spec = [("ndcg", 10), ("recall", 10)]
    out = {"metric_names": np.asarray([f"{n}@{k}" for n, k in spec])}
    pack_split(rs, out)
    bpr = ns.BPR(k=50, max_iter=50, learning_rate=0.01, lambda_reg=0.001, seed=123)
    bpr.fit(rs.train_set)
    out["bpr_U"], out["bpr_V"], out["bpr_Bi"] = bpr.u_factors, bpr.i_factors, bpr.i_biases
    # the test split keeps the users whose reference top-11 scores are separated (gap rule above; no AUC here, so the
    # positives outside the head decide nothing)
    clear_caches(bpr)
    ns.ranking_eval(bpr, metric_list(ns, spec), rs.train_set, rs.test_set, rating_threshold=rs.rating_threshold)
    keep = [u for u in sorted(bpr.ranked_items) if user_gap_ok(bpr.item_scores_mapped_indices[u], bpr.item_scores[u],
                                                              bpr.ranked_items[u], [], 11, bpr.num_items, False)]
    print(f"  c1: {len(keep)} of {len(bpr.ranked_items)} users keep the gap rule")
    rs = with_test_users(ns, rs, np.asarray(keep))
    clear_caches(bpr)
    out.update({k: v for k, v in _packed_test(rs).items()})
    out["gap_rel"] = np.float64(GAP_REL)
    add_result(out, "bpr", run_eval(ns, bpr, spec, rs, head=12, n_score_users=2))
    save("c1_bpr_ml100k_shape.npz", out)


def make_unknowns(ns):
    """exclude_unknowns=False with unknown users/items in the test split (MF min-padding
    path recommender.py:631-635, BPR total_users allocation recom_bpr.pyx:146-151)."""
    rng = np.random.default_rng(7)
    data = synth_uir(rng, 200, 380, 5000)
    rng.shuffle(data)
    cut = int(0.8 * len(data))
    train = [d for d in data[:cut] if int(d[0]) < 185 and int(d[1]) < 335]
    test = data[cut:]
    bm = ns.BaseMethod.from_splits(train_data=train, test_data=test, rating_threshold=3.0,
                                   exclude_unknowns=False, verbose=False)
    spec = [("ndcg", 10), ("recall", 10), ("precision", 10), ("auc", -1)]
    out = {"metric_names": np.asarray([f"{n}@{k}" for n, k in spec])}
    pack_split(bm, out)
    mf = ns.MF(k=10, max_iter=100, learning_rate=0.02, seed=5, use_bias=True)
    mf.fit(bm.train_set)
    out["mf_U"], out["mf_V"], out["mf_Bu"], out["mf_Bi"] = mf.u_factors, mf.i_factors, mf.u_biases, mf.i_biases
    out["mf_mu"] = np.float32(mf.global_mean)
    bpr = ns.BPR(k=16, max_iter=20, seed=5)
    bpr.fit(bm.train_set)
    out["bpr_U"], out["bpr_V"], out["bpr_Bi"] = bpr.u_factors, bpr.i_factors, bpr.i_biases
    # one test split for both models: the users that pass the gap rule under BOTH (head 19 = the heads the tests read)
    fsplit, res = gap_filtered_eval(ns, {"mf": mf, "bpr": bpr}, spec, bm, head=19, positions=False)
    out.update(_packed_test(fsplit))
    out["gap_rel"] = np.float64(GAP_REL)
    for name in ("mf", "bpr"):
        add_result(out, name, res[name])
    print(f"  unknowns: {len(res['mf']['users'])} users keep the gap rule")
    save("unknowns_mf_bpr.npz", out)


def make_list_metrics(ns):
    """Cut-offs beyond the fused top-K (k = -1, 50, 100, 200) and the whole-list metrics
    (MRR, MAP, NCRR, HitRatio, FMeasure): the reference's own per-user values for BPR and
    EASE on a 260 x 420 split (ranking.py:49-530 through base_method.py:185-310)."""
    rng = np.random.default_rng(321)
    data = synth_uir(rng, 260, 420, 9000)
    rs = ns.RatioSplit(data=data, test_size=0.25, rating_threshold=3.0, seed=321,
                       exclude_unknowns=True, verbose=False)
    spec = [("ndcg", -1), ("ndcg", 50), ("ndcg", 100), ("recall", 50), ("recall", 200), ("precision", 50),
            ("precision", -1), ("mrr", -1), ("map", -1), ("ncrr", -1), ("ncrr", 50), ("hit", 50), ("f1", 50),
            ("f1", -1), ("recall", 10), ("ndcg", 10)]
    out = {"metric_names": np.asarray([f"{n}@{k}" for n, k in spec])}
    pack_split(rs, out)
    bpr = ns.BPR(k=24, max_iter=30, seed=321)
    bpr.fit(rs.train_set)
    out["bpr_U"], out["bpr_V"], out["bpr_Bi"] = bpr.u_factors, bpr.i_factors, bpr.i_biases
    ease = ns.EASE(lamb=30, verbose=False)
    ease.fit(rs.train_set)
    out["ease_B"] = ease.B.astype(np.float64)
    # these metrics read the POSITION of every positive: the users kept are those whose positives are separated from
    # every other candidate under both models (no tie of any kind: positions = True)
    fsplit, res = gap_filtered_eval(ns, {"bpr": bpr, "ease": ease}, spec, rs, head=8, positions=True, head_len=8, n_score_users=2)
    out.update(_packed_test(fsplit))
    out["gap_rel"] = np.float64(GAP_REL)
    for name in ("bpr", "ease"):
        add_result(out, name, res[name])
    print(f"  list metrics: {len(res['bpr']['users'])} users keep the gap rule")
    save("list_metrics_bpr_ease.npz", out)


def make_fastdot(ns):
    """Outputs of the reference's compiled fast_dot on random fp32/fp64 inputs."""
    rng = np.random.default_rng(11)
    out = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        vec = rng.normal(size=37).astype(dt)
        mat = rng.normal(size=(91, 37)).astype(dt)
        init = rng.normal(size=91).astype(dt)
        res = init.copy()
        ns.fast_dot(vec, mat, res)
        out.update({f"{tag}_vec": vec, f"{tag}_mat": mat, f"{tag}_init": init, f"{tag}_out": res})
    save("fast_dot.npz", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/oracle")
    ap.add_argument("--only", default="", help="comma-separated subset of: fastdot,toy,gap,c1,unknowns,list "
                    "(BPR's threaded trainer is not run-to-run reproducible, so regenerate only what changed)")
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    prepare_scratch(args.scratch)
    os.chdir(args.scratch)
    ns = import_reference(args.scratch)
    makers = {"fastdot": make_fastdot, "toy": make_toy, "gap": make_gap, "c1": make_c1, "unknowns": make_unknowns,
              "list": make_list_metrics}
    for name in (args.only.split(",") if args.only else list(makers)):
        makers[name](ns)


if __name__ == "__main__":
    main()
