"""GPU parity for the scan-to-map association: correspondence indices and float32 squared distances must be
BIT-EXACT against the brute-force oracle (ties by ascending map index)."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _check(api, ctx, oracle, m, q, pose, thr, exact_all=False):
    mp = api.Map(ctx, m, thr if np.isfinite(thr) else 4.0); sc = api.Scan(ctx, q)
    api.knn3(mp, sc, pose, thr)
    idx, d2, valid = sc.download()
    i0, d0, v0 = oracle.knn3(m, q, pose, thr, method=0, threads=oracle.max_threads())
    assert np.array_equal(valid, v0)
    sel = np.ones(len(v0), bool) if exact_all else v0.astype(bool)
    assert np.array_equal(idx[sel], i0[sel])
    assert np.array_equal(d2[sel].view(np.uint32), d0[sel].view(np.uint32))     # bit-exact float32
    mp.close(); sc.close()
    return v0


def test_knn_random_cloud_with_ties(ctx, oracle):
    from lvio_fusion_amd import api
    rng = np.random.default_rng(4)
    M, Q = 20000, 5000
    m = np.zeros((M, 4), np.float32); m[:, :3] = rng.uniform(-10, 10, (M, 3))
    m[1000:2000, :3] = m[0:1000, :3]              # exact duplicates => d2 ties resolved by index
    q = np.zeros((Q, 8), np.float32); q[:, :3] = rng.uniform(-12, 12, (Q, 3))   # pcl::PointXYZI stride (8 floats)
    pose = np.array([0.01, -0.02, 0.3, 0.95, 0.5, -0.25, 0.1]); pose[:4] /= np.linalg.norm(pose[:4])
    v = _check(api, ctx, oracle, m, q, pose, 1.0)
    assert 0 < v.sum() < Q
    _check(api, ctx, oracle, m, q[:777], pose, np.float32(np.inf), exact_all=True)   # exact 3-NN for every point


def test_knn_config3_subset(ctx, oracle):
    from lvio_fusion_amd import api
    c = syn.config3_icp()
    rng = np.random.default_rng(0)
    qs = c["query"][np.sort(rng.choice(c["query"].shape[0], 4000, replace=False))]
    for thr in (c["thr_ground"], c["thr_surf"]):
        v = _check(api, ctx, oracle, c["map"], qs, c["pose0"], thr)
        assert v.mean() > 0.5


def test_knn_edge_cases(ctx, oracle):
    from lvio_fusion_amd import api
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    q = np.zeros((5, 4), np.float32); q[:, 0] = np.arange(5)
    # fewer than 3 map points: never valid (the reference would index out of range)
    m2 = np.zeros((2, 4), np.float32)
    _check(api, ctx, oracle, m2, q, pose, 4.0)
    # empty scan / empty map
    mp = api.Map(ctx, np.zeros((0, 4), np.float32), 4.0); sc = api.Scan(ctx, q)
    api.knn3(mp, sc, pose, 4.0)
    assert not sc.download()[2].any()
    mp.close(); sc.close()
    mp = api.Map(ctx, m2, 4.0); sc = api.Scan(ctx, np.zeros((0, 4), np.float32))
    api.knn3(mp, sc, pose, 4.0)
    assert sc.download()[0].shape == (0, 3)
    mp.close(); sc.close()
    # all map points identical (degenerate bounding box), queries far outside the grid
    m3 = np.tile(np.array([[1.0, 2.0, 3.0, 0.0]], np.float32), (10, 1))
    q3 = np.array([[1, 2, 3, 0], [100, 100, 100, 0], [-50, 2, 3, 0], [1.5, 2, 3, 0]], np.float32)
    _check(api, ctx, oracle, m3, q3, pose, 4.0)
    _check(api, ctx, oracle, m3, q3, pose, np.float32(np.inf), exact_all=True)


@pytest.mark.gpu
def test_maps_created_in_one_batch_equal_maps_created_one_by_one():
    """lvf_map_create_batch shares the host waits of the index builds between the maps (the old-frame maps of a set of loop-closure
    candidates: relocator.cpp:196-206 -> mapping.cpp:251-262); every map must be the one lvf_map_create builds: same pyramid, same
    association (indices and d2 bit for bit).  Mixed sizes and gates, an empty cloud, a one-point cloud."""
    import ctypes as C
    from lvio_fusion_amd import api, _lib
    rng = np.random.default_rng(77)
    ctx = api.Context(0)
    clouds = [rng.uniform(-20, 20, (5000, 3)).astype(np.float32) * np.array([1, 1, 0.05], np.float32),      # a flat sheet (dense cells)
              rng.normal(0, 6, (1200, 4)).astype(np.float32)[:, :3].copy(),
              np.zeros((0, 3), np.float32),
              np.array([[1.0, 2.0, 3.0]], np.float32),
              rng.uniform(-3, 3, (30000, 3)).astype(np.float32)]
    thr = [4.0, 1.0, 1.0, 4.0, 0.25]
    batch = api.Map.create_batch(ctx, clouds, thr)
    assert len(batch) == len(clouds) and api.Map.create_batch(ctx, [], 1.0) == []
    pose = np.array([0.02, -0.01, 0.03, 1.0, 0.1, -0.2, 0.05]); pose[:4] /= np.linalg.norm(pose[:4])
    q = rng.uniform(-8, 8, (4000, 3)).astype(np.float32)

    def pyramid(m):
        stats = np.zeros((len(q), 6), np.int32); lv = np.zeros((8, 4), np.float32); nl = C.c_int()
        sc = api.Scan(ctx, q)
        api._chk(ctx.L.lvf_knn3_debug_stats2(m.h, sc.h, pose.ctypes.data_as(_lib.c_double_p), 1.0, stats.ctypes.data_as(_lib.c_int_p), 6, lv.ctypes.data_as(_lib.c_float_p), C.byref(nl)))
        sc.close()
        return lv[:nl.value].copy()

    for cloud, t, mb in zip(clouds, thr, batch):
        ms = api.Map(ctx, cloud, t)
        assert np.array_equal(pyramid(ms), pyramid(mb))
        sa, sb = api.Scan(ctx, q), api.Scan(ctx, q)
        api.knn3(ms, sa, pose, t); api.knn3(mb, sb, pose, t)
        ia, da, va = sa.download(); ib, db, vb = sb.download()
        assert np.array_equal(ia, ib) and np.array_equal(da, db) and np.array_equal(va, vb)
        for h in (sa, sb, ms, mb):
            h.close()
    ctx.close()


@pytest.mark.gpu
def test_map_batch_rejects_a_bad_cloud_as_a_whole():
    """One cloud with an infinite coordinate (an unbounded grid) fails the whole lvf_map_create_batch call (no map is returned), like lvf_map_create fails for it alone;
    the context stays usable."""
    from lvio_fusion_amd import api
    rng = np.random.default_rng(5)
    ctx = api.Context(0)
    good = rng.uniform(-5, 5, (2000, 3)).astype(np.float32)
    bad = good.copy(); bad[17, 1] = np.inf
    with pytest.raises(api.LvfError):
        api.Map(ctx, bad, 1.0)
    with pytest.raises(api.LvfError):
        api.Map.create_batch(ctx, [good, bad, good], 1.0)
    with pytest.raises(api.LvfError):
        api.Map.create_batch(ctx, [good], [-1.0])
    maps = api.Map.create_batch(ctx, [good, good[:10]], [1.0, 4.0])          # the context still works
    sc = api.Scan(ctx, good[:100])
    api.knn3(maps[0], sc, np.array([0, 0, 0, 1.0, 0, 0, 0]), 1.0)
    idx, d2, valid = sc.download()
    assert np.array_equal(idx[:, 0], np.arange(100)) and np.all(d2[:, 0] == 0.0)      # every query is a map point
    for h in maps + [sc]:
        h.close()
    ctx.close()
