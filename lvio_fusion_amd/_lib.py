"""ctypes loader for the C-ABI library (include/lvf.h -> lvio_fusion_amd/liblvf_hip.so).

The library is the product; this module only declares signatures.  It never falls back to a CPU
implementation: if the .so is missing it raises, and if no gfx950 device is usable lvf_ctx_create
fails with LVF_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LVF_LIB_PATH") or os.path.join(HERE, "liblvf_hip.so")      # (LVF_LIB_PATH: another build of the same library, for A/B runs on one box)
HEADER = os.path.join(os.path.dirname(HERE), "include", "lvf.h")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("extrinsic", C.c_double * 7)]


class IcpOptions(C.Structure):
    _fields_ = [("mode", C.c_int), ("thr", C.c_float), ("weight", C.c_double), ("huber_a", C.c_double),
                ("prior_weight", C.c_double), ("max_num_iterations", C.c_int)]


class IcpSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_residual_blocks", C.c_int),
                ("num_iterations", C.c_int), ("num_successful_steps", C.c_int)]


class ScanMatchOptions(C.Structure):
    _fields_ = [("thr_ground", C.c_float), ("thr_surf", C.c_float), ("weight_ground", C.c_double), ("weight_surf", C.c_double),
                ("huber_surf", C.c_double), ("prior_weight", C.c_double), ("outer_iterations", C.c_int), ("max_num_iterations", C.c_int)]


class ScanMatchResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("relative_o_c", C.c_double * 7), ("score_ground", C.c_double), ("score_surf", C.c_double),
                ("score", C.c_int), ("ground", IcpSummary), ("surf", IcpSummary)]


class ScanMatchJob(C.Structure):
    _fields_ = [("map_ground", C.c_void_p), ("scan_ground", C.c_void_p), ("map_surf", C.c_void_p), ("scan_surf", C.c_void_p),
                ("map_pose", C.c_double * 7), ("frame_pose", C.c_double * 7), ("last_pose", C.c_double * 7), ("has_last_pose", C.c_int)]


class WindowOptions(C.Structure):
    _fields_ = [("baseline", C.c_double), ("weak_visual_threshold", C.c_int), ("prior_weight", C.c_double), ("prior_v", C.c_double),
                ("device_assembly", C.c_int)]


class LidarParams(C.Structure):
    _fields_ = [("num_scans", C.c_int), ("horizon_scan", C.c_int), ("ang_res_y", C.c_float), ("ang_bottom", C.c_float), ("ground_rows", C.c_int),
                ("cycle_time", C.c_double), ("min_range", C.c_float), ("max_range", C.c_float), ("resolution", C.c_float), ("ransac_seed", C.c_uint64)]


class LidarExtractDebug(C.Structure):
    _fields_ = [("n_filtered", C.c_int), ("n_segmented", C.c_int), ("n_ground_raw", C.c_int), ("n_surf_raw", C.c_int),
                ("label_mat", C.POINTER(C.c_int32)), ("ground_mat", C.POINTER(C.c_int8)), ("range_mat", c_float_p), ("ground_raw", c_float_p),
                ("surf_raw", c_float_p)]


class SolverOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("max_solver_time_in_seconds", C.c_double), ("huber_a", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("min_relative_decrease", C.c_double)]


class SolverSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_residual_blocks", C.c_int), ("termination", C.c_int),
                ("num_unsuccessful_steps", C.c_int), ("termination_reason", C.c_int), ("hand_over_retries", C.c_int)]

    WHY = ("none", "gradient_tolerance", "parameter_tolerance", "function_tolerance", "min_trust_region_radius", "max_num_iterations",
           "consecutive_invalid_steps", "max_solver_time", "hand_over_retry_failed")          # LVF_WHY_* (include/lvf.h)

    @property
    def why(self):
        return self.WHY[self.termination_reason]


def build(force=False):
    """Compile every HIP translation unit for gfx950 into liblvf_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".hpp"))]
    srcs.append(HEADER)
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(s) for s in srcs):
        return SO_PATH
    subprocess.check_call(["bash", os.path.join(HERE, "csrc", "build.sh")])
    return SO_PATH


_lib = None

_VP = C.c_void_p
_SIGS = {
    "lvf_last_error": (C.c_char_p, []),
    "lvf_version": (C.c_char_p, []),
    "lvf_host_alloc": (C.c_void_p, [C.c_size_t]),
    "lvf_host_free": (None, [C.c_void_p, C.c_size_t]),
    "lvf_ctx_create": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "lvf_ctx_destroy": (C.c_int, [_VP]),
    "lvf_ctx_synchronize": (C.c_int, [_VP]),
    "lvf_ctx_stream": (_VP, [_VP]),
    "lvf_timer_begin": (C.c_int, [_VP]),
    "lvf_timer_end": (C.c_int, [_VP]),
    "lvf_timer_elapsed_ms": (C.c_int, [_VP, c_float_p]),
    "lvf_box_calibration": (C.c_int, [_VP, c_double_p]),
    "lvf_state_create": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "lvf_state_destroy": (C.c_int, [_VP]),
    "lvf_state_set": (C.c_int, [_VP, C.c_int, c_double_p]),
    "lvf_state_get": (C.c_int, [_VP, C.c_int, c_double_p]),
    "lvf_state_copy": (C.c_int, [_VP, _VP]),
    "lvf_state_create_from": (C.c_int, [_VP, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(_VP)]),
    "lvf_state_set_all": (C.c_int, [_VP, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "lvf_two_frame_set_shape": (C.c_int, [_VP, C.c_int, c_int_p]),
    "lvf_pose_only_create": (C.c_int, [_VP, C.POINTER(Camera), C.c_int, c_double_p, c_int_p, c_int_p, C.c_int, c_double_p, C.POINTER(_VP)]),
    "lvf_two_frame_create": (C.c_int, [_VP, C.POINTER(Camera), C.POINTER(Camera), C.c_int, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p, C.POINTER(_VP)]),
    "lvf_two_camera_create": (C.c_int, [_VP, C.POINTER(Camera), C.POINTER(Camera), C.c_int, c_double_p, c_double_p, c_int_p, c_int_p, C.POINTER(_VP)]),
    "lvf_two_camera_set_block_weights": (C.c_int, [_VP, c_double_p]),
    "lvf_batch_local_columns": (C.c_int, [_VP]),
    "lvf_batch_evaluate_local": (C.c_int, [_VP, _VP, C.c_double, c_double_p, c_double_p]),
    "lvf_problem_gradient": (C.c_int, [_VP, C.POINTER(SolverOptions), c_double_p, c_double_p]),
    "lvf_problem_stage_count": (C.c_int, []),
    "lvf_problem_stage_name": (C.c_char_p, [C.c_int]),
    "lvf_problem_stage_times": (C.c_int, [_VP, C.POINTER(SolverOptions), C.c_double, C.c_int, c_double_p, C.POINTER(C.c_int)]),
    "lvf_problem_stage_times2": (C.c_int, [_VP, C.POINTER(SolverOptions), C.c_double, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int)]),
    "lvf_relocate_r_evaluate": (C.c_int, [_VP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "lvf_relocate_rotation_solve": (C.c_int, [_VP, C.c_int, c_double_p, c_double_p, c_double_p, C.POINTER(SolverOptions), C.POINTER(SolverSummary)]),
    "lvf_forward_update": (C.c_int, [_VP, c_double_p, C.c_int, c_double_p, c_double_p]),
    "lvf_state_forward_update": (C.c_int, [_VP, c_double_p, C.c_int]),
    "lvf_window_reject_outliers": (C.c_int, [_VP, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "lvf_comm_get_unique_id": (C.c_int, [_VP]),
    "lvf_comm_create": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.POINTER(_VP)]),
    "lvf_comm_destroy": (C.c_int, [_VP]),
    "lvf_comm_world_size": (C.c_int, [_VP]),
    "lvf_comm_rank": (C.c_int, [_VP]),
    "lvf_comm_allgather": (C.c_int, [_VP, c_double_p, C.c_int, c_double_p]),
    "lvf_problem_batch_create": (C.c_int, [_VP, C.POINTER(_VP), C.c_int, C.POINTER(_VP)]),
    "lvf_problem_batch_destroy": (C.c_int, [_VP]),
    "lvf_problem_batch_size": (C.c_int, [_VP]),
    "lvf_problem_debug_force_handover_timeout": (C.c_int, [_VP, C.c_int]),
    "lvf_problem_debug_history": (C.c_int, [_VP, c_double_p]),
    "lvf_problem_batch_uses_tables": (C.c_int, [_VP, C.POINTER(SolverOptions)]),
    "lvf_problem_batch_lm_iteration": (C.c_int, [_VP, C.POINTER(SolverOptions), c_double_p, c_double_p, c_double_p, c_double_p, c_int_p]),
    "lvf_problem_batch_solve": (C.c_int, [_VP, C.POINTER(SolverOptions), C.POINTER(SolverSummary)]),
    "lvf_imu_create": (C.c_int, [_VP, C.c_int, c_double_p, c_int_p, c_int_p, C.POINTER(_VP)]),
    "lvf_lidar_plane_create": (C.c_int, [_VP, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double, C.POINTER(_VP)]),
    "lvf_pose_prior_create": (C.c_int, [_VP, C.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p, C.POINTER(_VP)]),
    "lvf_relative_rpyxyz": (C.c_int, [c_double_p, c_double_p, c_double_p]),
    "lvf_problem_set_pose_priors": (C.c_int, [_VP, _VP]),
    "lvf_batch_destroy": (C.c_int, [_VP]),
    "lvf_batch_size": (C.c_int, [_VP]),
    "lvf_batch_num_param_blocks": (C.c_int, [_VP]),
    "lvf_batch_evaluate": (C.c_int, [_VP, _VP, c_double_p, C.c_int]),
    "lvf_batch_download_residuals": (C.c_int, [_VP, c_double_p]),
    "lvf_batch_download_jacobian": (C.c_int, [_VP, C.c_int, c_double_p]),
    "lvf_batch_download_normals": (C.c_int, [_VP, c_double_p]),
    "lvf_batch_residuals_dev": (_VP, [_VP]),
    "lvf_batch_jacobian_dev": (_VP, [_VP, C.c_int]),
    "lvf_preintegrate": (C.c_int, [_VP, C.c_int, c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "lvf_map_create": (C.c_int, [_VP, c_float_p, C.c_int, C.c_int, C.c_float, C.POINTER(_VP)]),
    "lvf_map_create_batch": (C.c_int, [_VP, C.c_int, C.POINTER(c_float_p), c_int_p, C.c_int, c_float_p, C.POINTER(_VP)]),
    "lvf_map_destroy": (C.c_int, [_VP]),
    "lvf_scan_create": (C.c_int, [_VP, c_float_p, C.c_int, C.c_int, C.POINTER(_VP)]),
    "lvf_scan_destroy": (C.c_int, [_VP]),
    "lvf_knn3": (C.c_int, [_VP, _VP, c_double_p, C.c_float]),
    "lvf_scan_download": (C.c_int, [_VP, c_int_p, c_float_p, c_u8_p]),
    "lvf_knn3_debug_stats": (C.c_int, [_VP, _VP, c_double_p, C.c_float, c_int_p, c_float_p, C.POINTER(C.c_int)]),
    "lvf_debug_sort_pairs_u32": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP]),
    "lvf_knn3_debug_stats2": (C.c_int, [_VP, _VP, c_double_p, C.c_float, c_int_p, C.c_int, c_float_p, C.POINTER(C.c_int)]),
    "lvf_icp_solve": (C.c_int, [_VP, _VP, c_double_p, c_double_p, c_double_p, C.POINTER(IcpOptions), C.POINTER(IcpSummary)]),
    "lvf_cloud_create": (C.c_int, [_VP, c_float_p, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "lvf_cloud_destroy": (C.c_int, [_VP]),
    "lvf_cloud_size": (C.c_int, [_VP]),
    "lvf_cloud_download": (C.c_int, [_VP, c_float_p]),
    "lvf_cloud_transform": (C.c_int, [_VP, c_double_p, C.POINTER(_VP)]),
    "lvf_cloud_concat": (C.c_int, [_VP, C.POINTER(_VP), C.c_int, C.POINTER(_VP)]),
    "lvf_cloud_align_scan": (C.c_int, [_VP, C.c_double, _VP, C.c_double, C.c_double, C.c_double, C.POINTER(_VP), C.POINTER(C.c_int)]),
    "lvf_cloud_voxel_filter": (C.c_int, [_VP, C.c_float, C.POINTER(_VP)]),
    "lvf_cloud_radius_outlier_filter": (C.c_int, [_VP, C.c_float, C.c_int, C.POINTER(_VP)]),
    "lvf_cloud_segment_plane": (C.c_int, [_VP, C.c_float, C.c_int, C.c_uint64, C.POINTER(_VP), c_double_p, C.POINTER(C.c_int)]),
    "lvf_map_create_from_cloud": (C.c_int, [_VP, C.c_float, C.POINTER(_VP)]),
    "lvf_map_create_batch_from_clouds": (C.c_int, [_VP, C.c_int, C.POINTER(_VP), c_float_p, C.POINTER(_VP)]),
    "lvf_scan_create_from_cloud": (C.c_int, [_VP, C.POINTER(_VP)]),
    "lvf_lidar_params_default": (None, [C.POINTER(LidarParams)]),
    "lvf_debug_extract_host_counts": (C.c_int, [C.c_int]),
    "lvf_lidar_extract": (C.c_int, [_VP, c_float_p, C.c_int, C.c_int, C.POINTER(LidarParams), c_double_p, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(LidarExtractDebug)]),
    "lvf_scan_match_options_default": (None, [C.POINTER(ScanMatchOptions), C.c_double]),
    "lvf_scan_match": (C.c_int, [_VP, _VP, _VP, _VP, c_double_p, c_double_p, c_double_p, C.POINTER(ScanMatchOptions), C.POINTER(ScanMatchResult)]),
    "lvf_scan_match_batch": (C.c_int, [_VP, C.POINTER(ScanMatchJob), C.c_int, C.POINTER(ScanMatchOptions), C.c_int, C.POINTER(ScanMatchResult), c_int_p]),
    "lvf_lidar_solve": (C.c_int, [_VP, c_double_p, C.POINTER(IcpOptions), C.POINTER(IcpSummary)]),
    "lvf_prior3_evaluate": (C.c_int, [_VP, C.c_int, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p]),
    "lvf_window_options_default": (None, [C.POINTER(WindowOptions)]),
    "lvf_window_create": (C.c_int, [_VP, C.POINTER(Camera), C.POINTER(Camera), C.POINTER(WindowOptions), C.POINTER(_VP)]),
    "lvf_window_destroy": (C.c_int, [_VP]),
    "lvf_window_add_keyframe": (C.c_int, [_VP, C.c_int64, c_double_p, C.c_double]),
    "lvf_window_set_imu": (C.c_int, [_VP, C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p]),
    "lvf_window_add_landmark": (C.c_int, [_VP, C.c_int64, C.c_int64, c_double_p, c_double_p, C.c_double]),
    "lvf_window_add_observation": (C.c_int, [_VP, C.c_int64, C.c_int64, c_double_p]),
    "lvf_window_remove_observation": (C.c_int, [_VP, C.c_int64, C.c_int64]),
    "lvf_window_slide": (C.c_int, [_VP, C.c_int64]),
    "lvf_window_solve": (C.c_int, [_VP, C.POINTER(SolverOptions), C.POINTER(SolverSummary)]),
    "lvf_window_set_pose": (C.c_int, [_VP, C.c_int64, c_double_p]),
    "lvf_window_get_pose": (C.c_int, [_VP, C.c_int64, c_double_p]),
    "lvf_window_get_imu": (C.c_int, [_VP, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "lvf_window_get_inv_depth": (C.c_int, [_VP, C.c_int64, c_double_p]),
    "lvf_window_counts": (C.c_int, [_VP, c_int_p]),
    "lvf_window_debug_blocks": (C.c_int, [_VP, C.c_int, C.c_int, _VP, c_double_p, C.POINTER(C.c_int)]),
    "lvf_solver_options_default": (None, [C.POINTER(SolverOptions)]),
    "lvf_problem_create": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "lvf_problem_destroy": (C.c_int, [_VP]),
    "lvf_problem_set_pose_constant": (C.c_int, [_VP, C.c_int, C.c_int]),
    "lvf_problem_set_vbb_constant": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lvf_problem_cost": (C.c_int, [_VP, C.POINTER(SolverOptions), c_double_p]),
    "lvf_problem_lm_iteration": (C.c_int, [_VP, C.POINTER(SolverOptions), c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
    "lvf_problem_solve": (C.c_int, [_VP, C.POINTER(SolverOptions), C.POINTER(SolverSummary)]),
    "lvf_problem_reduced_dim": (C.c_int, [_VP]),
    "lvf_problem_download_reduced": (C.c_int, [_VP, c_double_p, c_double_p]),
}


def declared_symbols():
    """Every function name declared in include/lvf.h (used by the CPU export test)."""
    import re
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lvf_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
        _lib = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib
