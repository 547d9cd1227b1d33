import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_oracle import _csr
from recommenders_amd.data import Dataset
from recommenders_amd.models import RP3_Beta
d = np.load("tests/golden/toy_all_models.npz")
train = Dataset.from_csr(_csr(d, "train"))
for form in ("dense", "csr"):
    m = RP3_Beta(beta=0.7, item_matrix=form).fit(train)
    eng = m._hip_engine()
    users = d["rp3__users"][:5].astype(np.int32)
    ids, sc = eng.rank_topk(users, 20)
    rows = eng.score_dense(users).cpu().numpy()
    print(form, "ids", ids.cpu().numpy()[:2], "\nsc", sc.cpu().numpy()[:2])
    print("argsort", np.argsort(-rows[:2], axis=1, kind="stable")[:, :20])
    print("ref head", d["rp3__ranked_head"][:2])
    print("nan count", np.isnan(rows).sum(), "max", np.nanmax(rows))
from recommenders_amd import ranking_eval, metrics as M
test = Dataset.from_csr(_csr(d, "test")); val = Dataset.from_csr(_csr(d, "val"))
for form in ("dense",):
    m = RP3_Beta(beta=0.7, item_matrix=form).fit(train)
    avg, per = ranking_eval(m, [M.NDCG(k=10), M.Recall(k=10)], train, test, val_set=val, rating_threshold=3.0)
    print("avg", avg, "ref", d["rp3__avg"][:2])
    eng = m._hip_engine()
    users = d["rp3__users"][:3].astype(np.int32)
    ids, sc = eng.rank_topk(users, 20)
    print(ids.cpu().numpy(), sc.cpu().numpy())
    print("ref", d["rp3__ranked_head"][:3], d["rp3__ranked_head_scores"][:3])
    u = int(users[0])
    print([per[0][int(x)] for x in users], d["rp3__per_user"][:3, 0])
