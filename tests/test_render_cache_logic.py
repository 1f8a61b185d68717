"""RenderCache bookkeeping without a device: what is a hit, what is a miss, what is left alone (the GPU side -- that a hit
renders the same bits -- is tests/test_gpu_rerender.py)."""
import gc
import weakref

import torch


def _entry(D, sig, tensors):
    return D._CacheEntry(object(), sig, tuple(weakref.ref(t) for t in tensors), object(), 123)


def test_hit_needs_the_same_signature_and_the_same_tensor_objects():
    import diff_gaussian_rasterization as D
    cache = D.RenderCache(max_entries=2)
    a, b = torch.zeros(4, 3), torch.zeros(4)
    e = _entry(D, ("sig", 1), (a, b))
    cache._store("k", e)
    assert cache._lookup("k", ("sig", 1), (a, b)) == (e, True) and cache.hits == 1
    assert cache._lookup("k", ("sig", 2), (a, b)) == (None, True)                 # another signature
    assert cache._lookup("other", ("sig", 1), (a, b)) == (None, True)             # another key
    a2 = torch.zeros(4, 3)                                                         # an equal tensor is not THE tensor
    assert cache._lookup("k", ("sig", 1), (a2, b)) == (None, True)
    assert cache.misses == 3
    del a
    gc.collect()                                                                   # a dead tensor can never match again
    a3 = torch.zeros(4, 3)
    assert cache._lookup("k", ("sig", 1), (a3, b)) == (None, True)


def test_signature_follows_versions_shapes_and_settings():
    import diff_gaussian_rasterization as D
    x = torch.zeros(5, 3)
    cam = [torch.eye(4), torch.eye(4), torch.zeros(3)]
    rs = D.GaussianRasterizationSettings(64, 96, 0.5, 0.4, torch.zeros(3), 1.0, cam[0], cam[1], 3, cam[2], False, False)
    s0 = D._cache_sig((x, None), rs)
    assert s0 == D._cache_sig((x, None), rs)
    x.add_(1.0)                                                                    # in place: the version moves
    assert D._cache_sig((x, None), rs) != s0
    s1 = D._cache_sig((x, None), rs)
    cam[0].mul_(2.0)                                                               # the camera edited in place
    assert D._cache_sig((x, None), rs) != s1
    s2 = D._cache_sig((x, None), rs)
    assert D._cache_sig((x, None), rs._replace(scale_modifier=1.5)) != s2
    assert D._cache_sig((x, None), rs._replace(sh_degree=2)) != s2
    assert D._cache_sig((x, None), rs._replace(image_width=97)) != s2
    assert D._cache_sig((x, None), rs._replace(bg=torch.ones(3))) == s2            # the background is read per render
    assert D._cache_sig((x, None), rs, extra=("pair", True)) != s2


def test_an_entry_waiting_for_its_backward_is_bypassed_and_lru_evicts():
    import diff_gaussian_rasterization as D
    cache = D.RenderCache(max_entries=2)
    t = torch.zeros(2)
    e = _entry(D, "s", (t,))
    cache._store("k", e)
    token = D._RenderToken()
    e.token = weakref.ref(token)
    assert e.busy()
    assert cache._lookup("k", "s", (t,)) == (None, False) and cache.bypassed == 1   # neither used nor replaced
    token.done = True                                                               # its backward ran
    assert not e.busy() and cache._lookup("k", "s", (t,)) == (e, True)
    e.token = weakref.ref(D._RenderToken())                                         # the graph died without a backward
    gc.collect()
    assert not e.busy()
    cache._store("k2", _entry(D, "s", (t,)))
    cache._lookup("k", "s", (t,))                                                   # k is now the most recently used
    cache._store("k3", _entry(D, "s", (t,)))
    assert list(cache.entries) == ["k", "k3"]                                       # k2 went
