"""The page-locked list-buffer pool of the Python front (orthosfm_amd/pipeline.py) without a device: checkout,
hand-back, reuse of a fitting buffer, freeing of a too small idle one.  hipHostMalloc is replaced by numpy here."""
import numpy as np


def test_list_buffer_pool_checks_out_and_frees(monkeypatch):
    from orthosfm_amd import capi, pipeline as P
    made, freed = [], []
    monkeypatch.setattr(capi, "pinned_rows", lambda rows: (made.append(rows), np.zeros((rows, 2), np.int32))[1])
    monkeypatch.setattr(capi, "pinned_free", lambda arr: freed.append(int(arr.shape[0])) or True)
    pool = P._ListBufferPool()
    a = pool.acquire(100)
    b = pool.acquire(100)
    assert a is not b and made == [100, 100]            # two jobs at once never share a buffer
    pool.release(a)
    assert pool.acquire(50) is a                          # an idle buffer that fits is taken again
    pool.release(a); pool.release(b)
    c = pool.acquire(500)                                 # too small: one idle buffer is freed, a larger one made
    assert freed == [100] and made[-1] == 500 and c.shape[0] == 500
    assert pool.idle_rows() == [100]
