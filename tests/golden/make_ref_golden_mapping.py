"""Generates tests/golden/ref_v5.npz: what the REFERENCE's own control code leaves behind on tests/mapping_replay.py's cases —
Mapping::Optimize / Mapping::Relocate (src/mapping.cpp), PoseGraph::BuildProblem / Optimize (src/pose_graph.cpp), Relocator::UpdateNewSubmap
(src/relocator.cpp) and Environment::Optimize (src/environment.cpp) compiled UNMODIFIED into oracle/_ref/liblvf_ref.so (oracle/ref_driver_mapping.cpp; third-party headers: the stand-ins of
oracle/ref_shim; ceres::Solve: the declared LM loop of oracle/ref_shim/ceres/solve_shim.h).  Needs /root/reference (build container).
    python tests/golden/make_ref_golden_mapping.py
The fixture is DATA (poses, scores, counts); the cases themselves are rebuilt from tests/mapping_replay.py wherever the fixture is used."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyref                      # noqa: E402
from tests import mapping_replay as mr        # noqa: E402


def generate():
    out = {}
    c = mr.optimize_case()
    r = pyref.mapping_optimize(c["time"], c["pose"], c["ground"], c["surf"], c["first_active"], c["w_ground"], c["w_surf"], c["w_visual"], c["n_features_left"])
    out["optimize_pose"] = r["pose"]; out["optimize_world_counts"] = r["world_counts"]
    for name, kw in mr.RELOCATE_CASES:
        c = mr.relocate_case(**kw)
        r = pyref.mapping_relocate(c["time"], c["pose"], c["ground"], c["surf"], c["old_index"], c["cur_ground"], c["cur_surf"], c["cur_pose"], c["rel_in"],
                                   c["w_ground"], c["w_surf"], c["w_visual"])
        out[name + "_score"] = np.array(r["score"]); out[name + "_relative_o_c"] = r["relative_o_c"]; out[name + "_map_pose"] = r["map_pose"]
        out[name + "_map_counts"] = r["map_counts"]
    c = mr.pose_graph_case()
    r = pyref.pose_graph_optimize(c["time"], c["pose"], c["vw"], c["section_A"], c["submap_A"], c["submap_B"], c["start_after"])
    out["pose_graph_pose"] = r["pose"]; out["pose_graph_vw"] = r["vw"]; out["pose_graph_counts"] = np.array(r["counts"])
    c = mr.submap_case()
    out["submap_pose"] = pyref.update_new_submap(c["time"], c["pose"], c["old_pose"], c["relative_o_c"], c["best"])
    c = mr.environment_case()
    out["environment_pose"] = pyref.environment_optimize(c["cam0"], c["cam1"], c["baseline"], c["pose3"], c["vel3"], c["ba3"], c["bg3"], c["w_visual"], c["samples"], c["acc0"],
                                                         c["gyr0"], c["noise4"], c["inv_depth"], c["right_ob"], c["left_ob"])
    return out


if __name__ == "__main__":
    if not pyref.can_build():
        sys.exit("needs /root/reference")
    d = generate()
    path = os.path.join(ROOT, "tests", "golden", "ref_v5.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, {k: v.shape for k, v in d.items()})
