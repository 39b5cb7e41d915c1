"""The control code around the hot path on the MI355X against the REFERENCE'S OWN TEXT: tests/mapping_replay.py's compositions of
Mapping::Optimize / Relocate, PoseGraph::Optimize and Relocator::UpdateNewSubmap, with every step taken through the C-ABI (lvf_cloud_transform = MergeScan,
lvf_map_create + lvf_scan_match = BuildMapFrame's cloud + the ground / surf sub-problems, lvf_forward_update, lvf_problem_solve over the pose priors,
lvf_relocate_rotation_solve), compared with tests/golden/ref_v5.npz — what src/mapping.cpp, src/pose_graph.cpp and src/relocator.cpp, compiled unmodified
(oracle/ref_driver_mapping.cpp), left behind on the same cases.  Tolerance 1e-6 relative (north_star); scores and block counts exactly."""
import os

import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests import mapping_replay as mr

pytestmark = pytest.mark.gpu

R5 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v5.npz"))


def close(a, b, tol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.all(np.abs(a - b) <= tol * (1e-3 + np.abs(b)))


class GpuBackend:
    def __init__(self, api, ctx):
        self.api, self.ctx = api, ctx
        self.se3_mul, self.se3_inv = syn.se3_mul, syn.se3_inv          # host-side pose algebra of the caller (SE3d operator* / inverse)
        self.pose_graph_target = api.relative_rpyxyz

    def transform(self, cloud, pose):
        c = self.api.Cloud(self.ctx, cloud)
        w = c.transform(pose)
        out = w.download()
        c.close(); w.close()
        return out

    def scan_match(self, mg, ms, qg, qs, map_pose, frame_pose, outer, prior_w):
        api = self.api
        opt = api.scan_match_options(mr.RES, outer_iterations=outer, prior_weight=prior_w)
        mpg = api.Map(self.ctx, mg, opt.thr_ground) if len(mg) else None
        scg = api.Scan(self.ctx, qg) if mpg is not None else None
        mps = api.Map(self.ctx, ms, opt.thr_surf) if len(ms) else None
        scs = api.Scan(self.ctx, qs) if mps is not None else None
        res = api.scan_match(mpg, scg, mps, scs, map_pose, frame_pose, opt)
        for h in (mpg, scg, mps, scs):
            if h is not None:
                h.close()
        return np.array(res.pose[:]), res.score_ground, res.score_surf

    def forward_update(self, T, poses, vw):
        return self.api.forward_update(self.ctx, T, poses, vw)

    def pose_graph_solve(self, Pc, const, pr):
        api = self.api
        st = api.State(self.ctx, len(Pc), 0); st.set(api.POSES, Pc)
        b = api.pose_prior_batch(self.ctx, pr["kf_a"], pr["kf_b"], pr["target"], pr["weight"], pr["v"])
        prob = api.Problem(self.ctx, st, None, None, None, None); prob.set_pose_priors(b)
        for k in np.flatnonzero(const):
            prob.set_pose_constant(int(k), True)
        prob.solve(api.default_solver_options())
        out = st.get(api.POSES).reshape(-1, 7).copy()
        prob.close(); b.close(); st.close()
        return out

    def rotation_solve(self, relocated, unrelocated):
        q, _ = self.api.relocate_rotation_solve(self.ctx, relocated, unrelocated, [0, 0, 0, 1.0])
        return np.asarray(q, np.float64)

    def environment_solve(self, c):
        """the two-keyframe window [last, cur] of Environment::Optimize through the flat C-ABI: lvf_problem_set_pose_constant / _set_vbb_constant"""
        api = self.api
        n = len(c["inv_depth"])
        cfg = dict(imu=[dict(samples=c["samples"], acc0=c["acc0"], gyr0=c["gyr0"], ba=c["ba3"][1], bg=c["bg3"][1], kf_i=0, kf_j=1)])
        pre = api.preintegrate_or_none(self.ctx, cfg)
        st = api.State(self.ctx, 2, 0)
        for field, val in ((api.POSES, c["pose3"][1:]), (api.VEL, c["vel3"][1:]), (api.BA, c["ba3"][1:]), (api.BG, c["bg3"][1:]), (api.W_VISUAL, np.full(2, c["w_visual"]))):
            st.set(field, val)
        pw = mr.world_points(_HostAlgebra, c)
        bpo = api.pose_only_batch(self.ctx, c["cam0"], c["left_ob"], np.ones(n, np.int32), np.arange(n, dtype=np.int32), pw)
        bimu = api.imu_batch(self.ctx, pre, [0], [1])
        prob = api.Problem(self.ctx, st, None, None, bpo, bimu)
        prob.set_pose_constant(0, True)
        for k in (0, 1):
            prob.set_vbb_constant(k, True, True, True)
        prob.solve(api.default_solver_options())
        out = st.get(api.POSES).reshape(-1, 7)[1].copy()
        prob.close(); bpo.close(); bimu.close(); st.close()
        return out


class _HostAlgebra:
    se3_apply = staticmethod(syn.se3_apply)


@pytest.fixture(scope="module")
def backend():
    from lvio_fusion_amd import api
    ctx = api.Context(0)
    yield GpuBackend(api, ctx)
    ctx.close()


def test_mapping_optimize_chain_equals_the_reference_text(backend):
    c = mr.optimize_case()
    P = mr.mapping_optimize(backend, c)
    assert close(P, R5["optimize_pose"]), np.abs(P - R5["optimize_pose"]).max(axis=1)


@pytest.mark.parametrize("name,kw", mr.RELOCATE_CASES)
def test_mapping_relocate_equals_the_reference_text(backend, name, kw):
    c = mr.relocate_case(**kw)
    score, rel, map_pose, counts = mr.mapping_relocate(backend, c)
    assert score == int(R5[name + "_score"])
    assert close(rel, R5[name + "_relative_o_c"]) and tuple(R5[name + "_map_counts"]) == counts


def test_relocate_through_the_batched_entry_point_equals_the_reference_text(backend):
    """lvf_scan_match_batch (what a rank runs for its share of the loop-closure candidates) on the relocate cases at once"""
    api, ctx = backend.api, backend.ctx
    names = mr.RELOCATE_CASES
    opt = api.scan_match_options(mr.RES, outer_iterations=4, prior_weight=0.0)
    jobs, handles, cases = [], [], []
    for name, kw in names:
        c = mr.relocate_case(**kw)
        idx = [0, 1, 2]
        mg = np.concatenate([backend.transform(c["ground"][j], c["pose"][j]) for j in idx]); ms = np.concatenate([backend.transform(c["surf"][j], c["pose"][j]) for j in idx])
        hs = [api.Map(ctx, mg, opt.thr_ground), api.Scan(ctx, c["cur_ground"]), api.Map(ctx, ms, opt.thr_surf), api.Scan(ctx, c["cur_surf"])]
        handles += hs
        jobs.append(dict(map_ground=hs[0], scan_ground=hs[1], map_surf=hs[2], scan_surf=hs[3], map_pose=c["pose"][0], frame_pose=syn.se3_mul(c["pose"][1], c["rel_in"]),
                         last_pose=c["pose"][1]))
        cases.append(name)
    res, best = api.scan_match_batch(ctx, jobs, opt, 20)
    for name, r in zip(cases, res):
        assert r.score == int(R5[name + "_score"])
        assert close(np.array(r.relative_o_c[:]), R5[name + "_relative_o_c"])
    # Relocator::CorrectLoop's arg-max over `loop_closure->score = score - 20`, `>=` (relocator.cpp:196-204): the LAST of the best-scoring candidates
    scores = [int(R5[n + "_score"]) for n in cases]
    assert best == max(i for i, v in enumerate(scores) if v == max(scores)) and max(scores) - 20 > 0
    for h in handles:
        h.close()


def test_pose_graph_optimize_equals_the_reference_text(backend):
    c = mr.pose_graph_case()
    P, vw = mr.pose_graph_optimize(backend, c)
    assert close(P, R5["pose_graph_pose"]), np.abs(P - R5["pose_graph_pose"]).max(axis=1)
    assert close(vw, R5["pose_graph_vw"])


def test_update_new_submap_equals_the_reference_text(backend):
    c = mr.submap_case()
    P = mr.update_new_submap(backend, c)
    assert close(P, R5["submap_pose"]), np.abs(P - R5["submap_pose"]).max(axis=1)


def test_environment_optimize_equals_the_reference_text(backend):
    c = mr.environment_case()
    P = mr.environment_optimize(backend, c)
    assert close(P, R5["environment_pose"]), np.abs(P - R5["environment_pose"])
