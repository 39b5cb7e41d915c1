"""The reference's CONTROL code around the hot path, pinned: Mapping::Optimize / Relocate (mapping.cpp:114-191,251-300),
PoseGraph::BuildProblem / Optimize / ForwardUpdate (pose_graph.cpp:163-252), Relocator::UpdateNewSubmap (relocator.cpp:247-282) run from the reference's
own text (oracle/_ref, oracle/ref_driver_mapping.cpp) on tests/mapping_replay.py's cases;
  * live (where /root/reference exists) they reproduce tests/golden/ref_v5.npz, and
  * the ORACLE compositions (oracle/icp.h, oracle/loop.h, oracle/lm.h put together by hand) equal the fixture to 1e-9 — scores and block counts exactly.
What stays DECLARED: the LM loop itself (Ceres is absent; oracle/ref_shim/ceres/solve_shim.h states it once more, generically) and PCL's RANSAC
(SegmentGround is a pass-through in the stand-in)."""
import os

import numpy as np
import pytest

from tests import mapping_replay as mr

R5 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v5.npz"))
RELOCATE = mr.RELOCATE_CASES


def close(a, b, tol=1e-9):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.all(np.abs(a - b) <= tol * (1.0 + np.abs(b)))


def test_fixture_is_what_the_reference_text_produces_now():
    from oracle import pyref
    if not pyref.can_build():
        pytest.skip("needs /root/reference")
    from tests.golden import make_ref_golden_mapping as gen
    live = gen.generate()
    assert set(live) == set(R5.files)
    for k, v in live.items():
        assert np.array_equal(np.asarray(v), R5[k]), k


def test_mapping_optimize_composition_equals_the_reference(oracle):
    c = mr.optimize_case()
    P = mr.mapping_optimize(mr.OracleBackend(oracle), c)
    assert close(P, R5["optimize_pose"])
    moved = np.abs(R5["optimize_pose"] - c["pose"]).max(axis=1)
    assert np.all(moved[:3] == 0) and np.all(moved[3:] > 1e-3), "the case must exercise the chain: every active keyframe moves"
    assert np.array_equal(R5["optimize_world_counts"], np.array([[len(g), len(s)] for g, s in zip(c["ground"], c["surf"])]))


@pytest.mark.parametrize("name,kw", RELOCATE)
def test_mapping_relocate_composition_equals_the_reference(oracle, name, kw):
    c = mr.relocate_case(**kw)
    score, rel, map_pose, counts = mr.mapping_relocate(mr.OracleBackend(oracle), c)
    assert score == int(R5[name + "_score"]), "int(score_ground + score_surf), mapping.cpp:279-299"
    assert close(rel, R5[name + "_relative_o_c"]) and np.array_equal(map_pose, R5[name + "_map_pose"]) and tuple(R5[name + "_map_counts"]) == counts


def test_relocate_scores_spread():
    s = [int(R5[n + "_score"]) for n, _ in RELOCATE]
    assert s[0] == 49 and s[0] > s[1] > s[2] and s[3] == 49, s          # saturated (20 + 30 - cost terms), partial overlap, poor overlap, the large case saturated again


def test_pose_graph_composition_equals_the_reference(oracle):
    c = mr.pose_graph_case()
    P, vw = mr.pose_graph_optimize(mr.OracleBackend(oracle), c)
    assert close(P, R5["pose_graph_pose"]) and close(vw, R5["pose_graph_vw"])
    # BuildProblem: per section PoseGraphError + RError, one closing PoseGraphError; AddParameterBlock for old, start and every section (pose_graph.cpp:172-198)
    ns = len(c["section_A"])
    assert tuple(R5["pose_graph_counts"]) == (2 * ns + 1, ns + 2, ns + 2)
    got = R5["pose_graph_pose"]
    assert np.array_equal(got[0], c["pose"][0]) and np.array_equal(got[1], c["pose"][1]), "the old frame is constant and nothing before the first section moves"
    assert np.array_equal(got[-1], c["start_after"])
    assert np.all(np.abs(got[2:-1] - c["pose"][2:-1]).max(axis=1) > 1e-2), "sections and the keyframes between them take up the correction"


def test_update_new_submap_composition_equals_the_reference(oracle):
    c = mr.submap_case()
    P = mr.update_new_submap(mr.OracleBackend(oracle), c)
    assert close(P, R5["submap_pose"])
    assert np.all(np.abs(R5["submap_pose"] - c["pose"]).max(axis=1) > 0.1)


def test_environment_optimize_composition_equals_the_reference(oracle):
    """Environment::Optimize (environment.cpp:18-115), the third adapt::Solve call site: one free pose, PoseOnly blocks, one ImuError with seven constant blocks"""
    c = mr.environment_case()
    P = mr.environment_optimize(mr.OracleBackend(oracle), c)
    assert close(P, R5["environment_pose"])
    assert np.abs(R5["environment_pose"][4:] - c["pose_true"][4:]).max() < 0.1 * np.abs(c["pose3"][2][4:] - c["pose_true"][4:]).max(), "the solve pulls the pose to the truth"
