"""hostio.py: the page-locked destination pool behind the literal numpy / CPU-tensor returns (reid/rerank.py:127, reid/evaluators.py:37-39)."""
import gc

import numpy as np
import pytest
import torch


def test_pool_never_hands_out_memory_that_is_still_referenced():
    """a block is re-used only when no tensor, view or numpy array refers to its storage (CPU: the pool falls back to pageable blocks,
    the ownership rule is the same)"""
    from ssg_amd import hostio
    hostio.pool_clear()
    a = hostio.pinned_empty((64, 64), torch.float64)
    na, p = a.numpy(), a.data_ptr()
    na[:] = 7.0
    assert hostio.pinned_empty((64, 64), torch.float64).data_ptr() != p        # `a` is alive
    del a; gc.collect()
    assert hostio.pinned_empty((64, 64), torch.float64).data_ptr() != p        # its numpy view is alive
    row = na[3:5]
    del na; gc.collect()
    assert hostio.pinned_empty((64, 64), torch.float64).data_ptr() != p        # a slice of the view is alive
    assert float(row.min()) == 7.0
    del row; gc.collect()
    assert hostio.pinned_empty((64, 64), torch.float64).data_ptr() == p        # nobody left: the pages are re-used
    assert hostio.pinned_empty((8, 3), torch.float16).shape == (8, 3)
    hostio.pool_clear()


@pytest.mark.gpu
def test_numpy_returns_through_the_pool_equal_plain_copies_and_do_not_alias():
    """re_ranking's numpy return (rerank.py:127) now lands in pooled page-locked memory, the float64 matrix chunk by chunk: the same
    bits as Tensor.cpu() of the materialised matrices, arrays of an earlier call stay intact while a later call runs, and the pages
    come back once the arrays are dropped."""
    import os
    from conftest import clustered
    from ssg_amd import hostio, rerank
    dev = torch.device("cuda", 0)
    N, Ns, d = 1500, 700, 128
    tgt, src = clustered(N, d, 11), clustered(Ns, d, 12, intra=0.7)
    hostio.pool_clear()
    os.environ["SSG_D2H_CHUNK_MB"] = "2"                 # 174 rows per chunk: nine chunks, both device buffers in use
    try:
        e1, f1 = rerank.re_ranking(src, tgt, lambda_value=0.3)
    finally:
        del os.environ["SSG_D2H_CHUNK_MB"]
    h = rerank.re_ranking_device(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), lambda_value=0.3)
    assert np.array_equal(e1.view(np.uint16), h.euclid.cpu().numpy().view(np.uint16))
    assert np.array_equal(np.asarray(f1), h.final_dist().cpu().numpy())
    assert not f1.flags.writeable and f1.valid_handle() is not None and e1.flags.writeable
    keep_e, keep_f = e1.copy(), np.array(f1)
    tgt2 = clustered(N, d, 13)
    e2, f2 = rerank.re_ranking(src, tgt2, lambda_value=0.3)                  # a second call of the same size while e1 / f1 are alive
    assert np.array_equal(e1, keep_e) and np.array_equal(np.asarray(f1), keep_f), "an earlier result was overwritten"
    assert not np.array_equal(np.asarray(f2), keep_f)
    p_f2 = np.asarray(f2).ctypes.data
    del e1, f1, e2, f2; gc.collect()
    st = hostio.pool_stats()
    assert st["blocks"] >= 4 and st["free_blocks"] == st["blocks"], st
    e3, f3 = rerank.re_ranking(src, tgt, lambda_value=0.3)                   # third call: no new pages
    assert hostio.pool_stats()["blocks"] == st["blocks"]
    assert np.array_equal(np.asarray(f3), keep_f) and np.array_equal(e3, keep_e)
    # no_rerank: the half matrix alone
    e4, none = rerank.re_ranking(src, tgt, no_rerank=True)
    assert none is None and e4.dtype == np.float16 and e4.shape == (N, N)
    del e3, f3, e4; gc.collect()
    hostio.pool_clear()
    assert p_f2 != 0
