"""GPU tests of the item-item kernels at the catalogue sizes of BASELINE.json's configs[3] and configs[4] (the fused
kernel's configs[2] shape has its own: test_gpu_parity.py::test_target_shape_one_million_items_bit_exact):
    C4  EASE shape   : 50 000 items, dense B generated on the device (10 GB); the checker gets the rows the histories read
    C5  RP3-beta shape: 125 000 items, CSR W with 256 neighbours per row = 31 windows of 4 096 columns per user
Ids and scores bit-exact against oracle/ifr_oracle.c orc_ranking_eval_item, metrics to 1e-12."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _histories(rng, lengths, pool, n_items, n_test=5):
    """Ragged histories drawn from `pool` (ascending inside a row) + a few held-out items per user."""
    rows, truth = [], []
    for n in lengths:
        r = np.sort(rng.choice(pool, size=min(n, len(pool)), replace=False)).astype(np.int32)
        rows.append(r)
        t = np.setdiff1d(rng.integers(0, n_items, n_test), r).astype(np.int32)
        truth.append(np.sort(t))
    ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    tptr = np.cumsum([0] + [len(t) for t in truth]).astype(np.int64)
    return ptr, np.concatenate(rows).astype(np.int32), tptr, np.concatenate(truth).astype(np.int32)


def test_c4_shape_dense_50000_items_bit_exact(torch):
    from recommenders_amd import HipScorer, synth
    n_items, K = 50_000, 10
    dev = torch.device("cuda:0")
    B = synth.ease_like_matrix(n_items, 11, dev)                                 # [50 000, 50 000] fp32 on the device
    rng = np.random.default_rng(5)
    pool = np.unique(np.r_[rng.choice(n_items, 600, replace=False), [0, 1, 255, 256, 4095, 4096, 25_000, n_items - 2, n_items - 1]])
    lengths = [int(v) for v in rng.integers(5, 60, 300)]
    lengths[3], lengths[9], lengths[200] = 0, 300, 1                             # no history, a long one, a single row
    h_ptr, h_idx, t_ptr, t_idx = _histories(rng, lengths, pool, n_items)
    h_val = rng.integers(1, 6, len(h_idx)).astype(np.float32)
    col_scale = (1.0 / rng.integers(1, 400, n_items) ** 0.5).astype(np.float32)
    sc = HipScorer(0)
    sc.set_item_matrix_dense(B, col_scale)
    sc.set_history_csr(h_ptr, h_idx, h_val)
    sc.set_exclusion_csr(h_ptr, h_idx)
    sc.set_truth_csr(t_ptr, t_idx)
    users = np.r_[np.arange(len(lengths)), [len(lengths) + 3]].astype(np.int32)[::-1].copy()      # + a user without any row
    ids, scores = sc.rank_topk(users, K)
    per_user, _, _ = sc.eval_topk_metrics(users, ids, [K])
    # the checker reads only the rows of B that some history holds: [R, 50 000] on the host, histories renumbered to them
    rows, h_local = np.unique(h_idx, return_inverse=True)
    M_sub = B[torch.as_tensor(rows.astype(np.int64), device=dev)].cpu().numpy()
    o_ids, o_sc, o_met, _ = orc.c_ranking_eval_item(M_sub, None, 0, n_items, col_scale, h_ptr, h_local.astype(np.int32), h_val, users,
                                                    h_ptr, h_idx, t_ptr, t_idx, [K], False, K, n_threads=8)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    np.testing.assert_allclose(per_user.cpu().numpy()[:, :3], o_met[:, :3], atol=1e-12)
    # dense rows of a few users (what AUC / the list metrics read) against the checker's score vector
    few = users[-4:]
    dense = sc.score_dense(few).cpu().numpy()
    for r, u in enumerate(few):
        a, b = h_ptr[u], h_ptr[u + 1]
        ref = orc.c_score_item_dense(M_sub, h_local[a:b].astype(np.int32), h_val[a:b], col_scale)
        np.testing.assert_array_equal(dense[r, :n_items], ref)
    sc.close()
    del B


@pytest.mark.parametrize("cand_cap,wave", [(0, 1), (0, 2), (0, 0), (32, 0)])
def test_c5_shape_csr_125000_items_256_neighbours_bit_exact(torch, cand_cap, wave):
    """31 windows per user; a user with 400 history rows (staging rounds, ~3 300 entries per window, an exclusion row
    beyond the LDS copy), one with 200 rows (one lane per row, segments longer than the prefetch), a single-row user, a
    user without history and one past the last row; with the candidate buffer cut to max(32, K) entries the first windows
    overflow it and are ranked by the exact argmax rounds."""
    from recommenders_amd import HipScorer, synth
    n_items, nnz = 125_000, 256
    rng = np.random.default_rng(21)
    m_ptr, m_idx, m_val = synth.sparse_item_matrix(rng, n_items, nnz)
    assert (n_items + 4095) // 4096 == 31
    lengths = [int(v) for v in rng.integers(5, 80, 260)]
    lengths[7], lengths[11], lengths[13], lengths[17] = 0, 400, 200, 1
    h_ptr, h_idx, t_ptr, t_idx = _histories(rng, lengths, np.arange(n_items), n_items)
    cnt = np.diff(h_ptr)
    h_val = np.repeat((1.0 / np.maximum(cnt, 1)).astype(np.float32), cnt)
    col_scale = (1.0 / rng.integers(1, 3000, n_items) ** 0.7).astype(np.float32)
    sc = HipScorer(0)
    sc.set_item_matrix_csr(m_ptr, m_idx, m_val, n_items, col_scale)
    sc.set_history_csr(h_ptr, h_idx, h_val)
    sc.set_exclusion_csr(h_ptr, h_idx)
    sc.set_truth_csr(t_ptr, t_idx)
    sc.set_option("csr_wave", wave)             # 1 / 2: one wave per user (histories <= 256 rows; 2 = whatever the matrix looks like), the rest handed over
    if cand_cap:
        sc.set_option("csr_cand_cap", cand_cap)
    users = np.r_[np.arange(len(lengths)), [len(lengths) + 9]].astype(np.int32)
    shift = sc.item_fixed_point_shift()
    assert shift == orc.item_fixed_point_shift(m_val, h_ptr, h_val, m_idx, col_scale)
    for K in (10, 32, 100):
        ids, scores = sc.rank_topk(users, K)
        per_user, _, _ = sc.eval_topk_metrics(users, ids, [K])
        o_ids, o_sc, o_met, _ = orc.c_ranking_eval_item(None, (m_ptr, m_idx, m_val), shift, n_items, col_scale, h_ptr, h_idx, h_val,
                                                        users, h_ptr, h_idx, t_ptr, t_idx, [K], False, K, n_threads=8)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
        np.testing.assert_allclose(per_user.cpu().numpy()[:, :3], o_met[:, :3], atol=1e-12)
    sc.close()


@pytest.mark.parametrize("nnz,words", [(500, 2), (1000, 3)])
def test_c5_range_more_neighbours_two_and_three_window_words(torch, nnz, words):
    """SURVEY.md names 100 ... 1000 neighbours per row for C5.  With 500 / 1000 evenly spread neighbours a (row,
    1024-column window) segment holds up to 6 / 9 entries: the wave-per-user kernel's table takes two / three 16-byte
    words per segment (and a lane three / two history rows).  Histories of 1 ... 400 rows: one to three rows per lane
    and the hand-over to the workgroup kernel."""
    from recommenders_amd import HipScorer, synth
    n_items = 125_000
    rng = np.random.default_rng(33 + words)
    m_ptr, m_idx, m_val = synth.sparse_item_matrix(rng, n_items, nnz)
    lengths = [int(v) for v in rng.integers(3, 70, 120)] + [1, 64, 65, 100, 128, 129, 160, 192, 193, 300, 400, 0]
    h_ptr, h_idx, t_ptr, t_idx = _histories(rng, lengths, np.arange(n_items), n_items)
    cnt = np.diff(h_ptr)
    h_val = np.repeat((1.0 / np.maximum(cnt, 1)).astype(np.float32), cnt)
    col_scale = (1.0 / rng.integers(1, 3000, n_items) ** 0.7).astype(np.float32)
    sc = HipScorer(0)
    sc.set_item_matrix_csr(m_ptr, m_idx, m_val, n_items, col_scale)
    got_words, long_share = sc.item_csr_layout()
    assert got_words == words and long_share <= 0.003
    sc.set_history_csr(h_ptr, h_idx, h_val)
    sc.set_exclusion_csr(h_ptr, h_idx)
    sc.set_truth_csr(t_ptr, t_idx)
    users = np.arange(len(lengths), dtype=np.int32)
    shift = sc.item_fixed_point_shift()
    for K in (10, 32):
        o_ids, o_sc, _, _ = orc.c_ranking_eval_item(None, (m_ptr, m_idx, m_val), shift, n_items, col_scale, h_ptr, h_idx, h_val,
                                                    users, h_ptr, h_idx, t_ptr, t_idx, [K], False, K, n_threads=8)
        for wave in (1, 0):
            sc.set_option("csr_wave", wave)
            ids, scores = sc.rank_topk(users, K)
            np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
            np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    sc.close()
