"""ifr_rank_topk_noalloc (include/ifr.h): the entry point that allocates nothing, against ifr_rank_topk.

Collected last on purpose (the round's end runs pytest -x).  Two calls in a row through the C ABI, the second with a
smaller input: the static workspace keeps the first call's contents, so stale partial lists or bound words would show
up as wrong ids in the second call.  Bar: bit-exact ids and scores (integer / index work).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _scorer(seed, n_users, n_items, k, bias):
    from recommenders_amd import HipScorer, synth
    rng = np.random.default_rng(seed)
    U, V, ub, ib = synth.factor_model(rng, n_users, n_items, k, bias)
    tr_ptr, tr_idx, _, _ = synth.interactions(rng, n_users, n_items, 40, lo=0)
    sc = HipScorer(0)
    sc.set_factors(U, V, ub, ib)
    sc.set_exclusion_csr(tr_ptr, tr_idx)
    return sc


@pytest.mark.parametrize("k,bias,K", [(128, False, 10), (50, True, 16)])
def test_noalloc_matches_rank_topk_on_two_calls_in_a_row(torch, k, bias, K):
    sc = _scorer(77 + K, 5000, 6007, k, bias)
    dev = sc.device
    rng = np.random.default_rng(1)
    first = torch.as_tensor(rng.permutation(5000)[:3001].astype(np.int32), device=dev)      # ragged: 3001 = 23 * 128 + 57
    second = torch.as_tensor(rng.permutation(5000)[:700].astype(np.int32), device=dev)      # smaller, other users
    outs = [(torch.empty((u.numel(), K), dtype=torch.int32, device=dev),
             torch.empty((u.numel(), K), dtype=torch.float32, device=dev)) for u in (first, second)]
    torch.cuda.synchronize(dev)
    # the no-alloc calls come FIRST on this scorer: rank_topk has not grown any workspace yet, so a regression that fell
    # back to the growing workspace would show as a drop of the device's free memory
    free_before, _ = torch.cuda.mem_get_info(dev)
    sc.rank_topk_noalloc(first, K, *outs[0])
    sc.rank_topk_noalloc(second, K, *outs[1])
    torch.cuda.synchronize(dev)
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after == free_before, "the no-alloc entry point changed the device's free memory"
    expect = [tuple(t.clone() for t in sc.rank_topk(u, K)) for u in (first, second)]
    for (ids, scores), (e_ids, e_scores) in zip(outs, expect):
        np.testing.assert_array_equal(ids.cpu().numpy(), e_ids.cpu().numpy())
        np.testing.assert_array_equal(scores.cpu().numpy(), e_scores.cpu().numpy())


@pytest.mark.parametrize("n", [1, 128, 700])
def test_noalloc_small_batches_over_a_large_catalogue(torch, n):
    """The serving-loop case: a small batch leaves room in the static workspace for hundreds of item
    chunks; the launch plan must still stay within what the merge kernel reads (64 chunks).  200 003 items."""
    from recommenders_amd import HipScorer, synth
    rng = np.random.default_rng(9)
    U, V, ub, ib = synth.factor_model(rng, 900, 200003, 64, True)
    sc = HipScorer(0)
    sc.set_factors(U, V, ub, ib)
    dev = sc.device
    users = torch.as_tensor(rng.permutation(900)[:n].astype(np.int32), device=dev)
    ids = torch.empty((n, 10), dtype=torch.int32, device=dev)
    scores = torch.empty((n, 10), dtype=torch.float32, device=dev)
    sc.rank_topk_noalloc(users, 10, ids, scores)
    sc.rank_topk_noalloc(users, 10, ids, scores)                       # again over the stale workspace
    e_ids, e_scores = sc.rank_topk(users, 10)
    np.testing.assert_array_equal(ids.cpu().numpy(), e_ids.cpu().numpy())
    np.testing.assert_array_equal(scores.cpu().numpy(), e_scores.cpu().numpy())


def test_noalloc_rejects_what_does_not_fit(torch):
    from recommenders_amd.engine import IfrError
    sc = _scorer(5, 300, 500, 32, False)
    dev = sc.device
    n = sc.NOALLOC_MAX_USERS + 1
    users = torch.zeros((n,), dtype=torch.int32, device=dev)
    ids = torch.empty((n, 10), dtype=torch.int32, device=dev)
    scores = torch.empty((n, 10), dtype=torch.float32, device=dev)
    with pytest.raises(IfrError, match="IFR_NOALLOC_MAX_USERS"):
        sc.rank_topk_noalloc(users, 10, ids, scores)
    with pytest.raises(IfrError, match=r"K must be in \[1, 16\]"):
        sc.rank_topk_noalloc(users[:10].contiguous(), 20, torch.empty((10, 20), dtype=torch.int32, device=dev),
                             torch.empty((10, 20), dtype=torch.float32, device=dev))
    # the largest batch that is accepted still runs, with the longest lists
    n = sc.NOALLOC_MAX_USERS
    users = torch.arange(n, dtype=torch.int32, device=dev) % 300
    ids = torch.empty((n, 16), dtype=torch.int32, device=dev)
    scores = torch.empty((n, 16), dtype=torch.float32, device=dev)
    sc.rank_topk_noalloc(users, 16, ids, scores)
    e_ids, e_scores = sc.rank_topk(users[:300], 16)
    np.testing.assert_array_equal(ids[:300].cpu().numpy(), e_ids.cpu().numpy())
    np.testing.assert_array_equal(ids[300:600].cpu().numpy(), e_ids.cpu().numpy())
    np.testing.assert_array_equal(scores[:300].cpu().numpy(), e_scores.cpu().numpy())
