"""Independent checks of the oracle's map-cloud maintenance restatement (oracle/cloud.h) — the reference's PCL calls cannot
run here, so the declared semantics are cross-checked against numpy / scipy formulations."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from lvio_fusion_amd import synthetic as syn


@pytest.fixture(scope="module")
def scene():
    c = syn.config3_icp(seed=321, n_query=6000, n_az=400)
    q = c["query"].copy()
    q[:, 3] = np.random.default_rng(0).uniform(0, 255, len(q)).astype(np.float32)     # intensity
    return c, q


def test_transform_matches_double_precision(oracle, scene):
    c, q = scene
    out = oracle.cloud_transform(q, c["pose0"])
    ref = syn.se3_apply(c["pose0"], q[:, :3].astype(np.float64))
    assert np.abs(out[:, :3] - ref).max() < 2e-5                   # float32 arithmetic at ~30 m range
    assert np.array_equal(out[:, 3], q[:, 3])
    # the per-point float routine used by the association is the same function
    one = oracle.se3_apply_f32(c["pose0"].astype(np.float32), q[7, :3])
    assert np.array_equal(one, out[7, :3])


def test_voxel_filter_groups_like_numpy(oracle, scene):
    _, q = scene
    leaf = np.float32(0.4)
    out = oracle.voxel_filter(q, leaf)
    inv = np.float32(1.0) / leaf
    ijk = np.floor(q[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(q[:, :3].min(0) * inv).astype(np.int64)
    div = ijk.max(0) + 1
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq, inverse = np.unique(idx, return_inverse=True)
    assert out.shape == (len(uniq), 4)
    ref = np.zeros((len(uniq), 4)); np.add.at(ref, inverse, q.astype(np.float64)); ref /= np.bincount(inverse)[:, None]
    assert np.allclose(out, ref, rtol=2e-6, atol=1e-5)            # float accumulation vs double
    # idempotent up to voxel re-centring: filtering again never increases the count
    assert len(oracle.voxel_filter(out, leaf)) <= len(out)


def test_radius_outlier_matches_kdtree_count(oracle, scene):
    _, q = scene
    sub = q[:3000]
    keep = oracle.radius_outlier_keep(sub, 0.8, 4)
    tree = cKDTree(sub[:, :3].astype(np.float64))
    cnt = np.array([len(v) for v in tree.query_ball_point(sub[:, :3].astype(np.float64), 0.8)])
    d = np.abs(cnt - 5)                                            # only counts at the decision boundary may differ (float vs double, < vs <=)
    agree = keep.astype(bool) == (cnt > 4)
    assert agree[d > 1].all() and agree.mean() > 0.995
    assert 0 < keep.sum() < len(keep)


def test_segment_plane_recovers_the_ground(oracle, scene):
    c, q = scene
    g = q[c["query_ground"]]
    rng = np.random.default_rng(3)
    clutter = g[:300].copy(); clutter[:, 2] += rng.uniform(0.3, 2.0, 300).astype(np.float32)
    pts = np.concatenate([g, clutter]).astype(np.float32)
    mask, co, it = oracle.segment_plane(pts, 0.02 * 2.5, 100, seed=12345)
    assert 1 <= it <= 100
    assert abs(np.linalg.norm(co[:3]) - 1) < 1e-6 and co[2] > 0.99          # the scene's ground is z = const in the body frame
    assert mask[:len(g)].mean() > 0.9 and mask[len(g):].sum() == 0
    d = np.abs(pts[:, :3].astype(np.float64) @ co[:3] + co[3])
    assert (d[mask > 0] < 0.05 + 1e-6).all()
    # same seed, same answer; another seed may pick other samples but finds the same plane
    m2, co2, _ = oracle.segment_plane(pts, 0.05, 100, seed=12345)
    assert np.array_equal(mask, m2)
    m3, co3, _ = oracle.segment_plane(pts, 0.05, 100, seed=7)
    assert np.abs(co3 - co).max() < 5e-3


def test_lidar_extract_restatement_is_sane(oracle):
    """oracle/extract.h on a synthetic revolution: the flat ground ends up in the ground cloud, walls/boxes in surf, sparse clutter
    is rejected as outlier segments, and every stage's bookkeeping is self-consistent."""
    scan = syn.raw_scan(seed=3)
    r = oracle.lidar_extract(scan, syn.lidar_extrinsic())
    assert np.isnan(scan[:, 0]).sum() > 0 and r["n_filtered"] < len(scan)
    d2 = (scan[:, :3] ** 2).sum(1)
    assert r["n_filtered"] == int(((d2 > 25) & (d2 < 900)).sum())
    occ = r["range_mat"] < 1e30
    assert 0.2 < occ.mean() < 0.9 and (r["label_mat"][~occ] == -1).all()
    gm = r["ground_mat"] == 1
    assert gm[11:20].mean() > 0.5 and gm[61:].sum() == 0                    # rings that meet the road inside the 5-30 m gate; rows above ground_rows never do
    assert (r["label_mat"][gm] == -1).all()
    raw_g = r["ground_raw"]
    assert np.median(np.abs(raw_g[:, 2] + 1.73)) < 0.05                     # mostly the road (flat box tops qualify as "ground" too)
    ring = np.round(raw_g[:, 3])
    assert ring.min() >= 0 and ring.max() < 64 and np.abs(raw_g[:, 3] - ring).max() < 0.16   # intensity = ring + cycle_time * rel_time, rel_time in about [-0.25, 1.25]
    assert len(r["surf"]) > 100 and len(r["ground"]) > 100
    # the voxel / outlier / plane tail only ever removes points
    assert len(r["surf"]) < len(r["surf_raw"]) and len(r["ground"]) < len(r["ground_raw"])


def test_cr_atan2f_is_the_correctly_rounded_float(oracle):
    """oracle/cr_math.h (and its device twin csrc/cr_math.hpp): atan2 of float arguments rounded ONCE to float — checked against
    mpmath at 50 digits, including the rounding decision (nearest of the neighbouring floats)."""
    import mpmath as mp
    mp.mp.dps = 50
    rng = np.random.default_rng(1)
    n = 4000
    y = (rng.normal(0, 1, n) * 10.0 ** rng.integers(-4, 4, n)).astype(np.float32)
    x = (rng.normal(0, 1, n) * 10.0 ** rng.integers(-4, 4, n)).astype(np.float32)
    y[:10] = [0, 0, 1, -1, 1, -1, 1e-30, 5, -7.5, 3]; x[:10] = [1, -1, 0, 0, 1, -1, 1, -1e-30, -2, 3]
    got = oracle.cr_atan2f(y, x)
    for i in range(n):
        v = mp.atan2(mp.mpf(float(y[i])), mp.mpf(float(x[i])))
        f = np.float32(float(v))
        cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
        best = min(cands, key=lambda c: abs(mp.mpf(float(c)) - v))
        assert got[i] == best, (y[i], x[i], got[i], best)
    assert oracle.cr_atan2f(np.float32([-0.0]), np.float32([-2.0]))[0] == np.float32(-np.pi)       # signed zero: the libm convention
    assert np.isnan(oracle.cr_atan2f(np.float32([np.nan]), np.float32([1.0]))[0])
