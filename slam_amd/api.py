"""ctypes binding of the C-ABI in include/slam_mi355x.h -- plumbing for tests,
bench.py and __graft_entry__.py.  The product is the shared library (HIP
kernels + C++ host code); this file only marshals numpy arrays and device
pointers across that ABI.  There is no CPU path: if the library is missing or
no HIP device is usable, calls raise SlamError.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SLAM_AMD_MEASURE=1 loads the measurement build (python -m slam_amd.build --measure) for the tools/ scripts
LIB_PATH = os.path.join(HERE, "lib", "libslam_mi355x_measure.so" if os.environ.get("SLAM_AMD_MEASURE") == "1"
                        else "libslam_mi355x.so")
RCCL_LIB_PATH = os.path.join(HERE, "lib", "libslam_mi355x_rccl.so")

SLAM_OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_TOO_FEW_MODEL, E_TOO_FEW_SCENE, E_NOMEM, E_UNSUPPORTED, E_TIMEOUT, E_COMM = \
    -1, -2, -3, -4, -5, -6, -7, -8, -9
ICP_P2P, ICP_P2L = 0, 1
RAYCAST_TILED, RAYCAST_GLOBAL, RAYCAST_TILED_MERGE = 0, 1, 2


class SlamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slam_mi355x error %d: %s" % (code, msg))
        self.code = code


class IcpParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("min_delta", C.c_double), ("mode", C.c_int),
                ("normals_k", C.c_int), ("lanes_per_point", C.c_int), ("cell_size", C.c_double),
                ("force_global", C.c_int), ("build_on_host", C.c_int), ("first_iterations", C.c_int),
                ("far_div", C.c_int), ("split_launch", C.c_int), ("spread_scans", C.c_int), ("pair_scans", C.c_int),
                ("spread_wait_us", C.c_int), ("wave_tiles", C.c_int), ("list_min_halo", C.c_double), ("spread_tile", C.c_int)]


class IcpResult(C.Structure):
    _fields_ = [("iters", C.c_int), ("n_corr", C.c_int), ("delta", C.c_double)]


class GridParams(C.Structure):
    _fields_ = [("max_range", C.c_double), ("occupancy_increment", C.c_double),
                ("occupancy_decrement", C.c_double), ("min_cluster_points", C.c_int),
                ("rolling", C.c_int), ("raycast_impl", C.c_int), ("raycast_seg_items", C.c_int),
                ("raycast_wg_per_cu", C.c_int), ("raycast_max_workgroups", C.c_int)]


class MapperParams(C.Structure):
    _fields_ = [("grid_size_x", C.c_int), ("grid_size_y", C.c_int), ("resolution", C.c_double), ("grid", GridParams),
                ("icp", IcpParams), ("indist", C.c_double), ("max_scans", C.c_int), ("max_points", C.c_int),
                ("window_chunks", C.c_int), ("rebuild_every", C.c_int), ("target_points", C.c_int),
                ("keep_prior", C.c_int), ("merge_every", C.c_int), ("pipelined", C.c_int), ("strict_window", C.c_int),
                ("slots", C.c_int), ("thin_res", C.c_double), ("background_rebuild", C.c_int), ("registration_streams", C.c_int)]


class GsegParams(C.Structure):
    _fields_ = [("rmax", C.c_double), ("num_seedpoints", C.c_int), ("gp_lengthparameter", C.c_double),
                ("gp_covariancescale", C.c_double), ("gp_modelnoise", C.c_double),
                ("gp_groundmodelconfidence", C.c_double), ("gp_grounddataconfidence", C.c_double),
                ("gp_groundthreshold", C.c_double), ("robotheight", C.c_double),
                ("seeding_maxrange", C.c_double), ("seeding_maxheight", C.c_double)]


GSEG_DROPPED, GSEG_GROUND, GSEG_OBSTACLE, GSEG_OVERHEAD = 0, 1, 2, 3

RESULT_DTYPE = np.dtype([("iters", np.int32), ("n_corr", np.int32), ("delta", np.float64)])

_vp = C.c_void_p
_lib = None

# every symbol include/slam_mi355x.h declares (tests check the .so exports them all)
EXPORTS = [
    "slam_last_error", "slam_version", "slam_device_count", "slam_set_device", "slam_device_info",
    "slam_malloc", "slam_free", "slam_memset", "slam_memcpy_h2d", "slam_memcpy_d2h",
    "slam_memcpy_d2d", "slam_host_is_pinned", "slam_host_alloc", "slam_host_free", "slam_memcpy_h2d_async",
    "slam_memcpy_d2h_async", "slam_stream_wait_event", "slam_graph_begin_capture", "slam_graph_end_capture",
    "slam_graph_launch", "slam_graph_destroy",
    "slam_stream_create", "slam_stream_create_with_priority", "slam_stream_create_reserving_cus", "slam_stream_destroy", "slam_stream_synchronize",
    "slam_device_synchronize", "slam_event_create", "slam_event_destroy", "slam_event_record",
    "slam_event_synchronize", "slam_event_query", "slam_event_elapsed_ms",
    "slam_icp_default_params", "slam_icp_create", "slam_icp_create_dev", "slam_icp_destroy",
    "slam_icp_build_info", "slam_icp_index_blob",
    "slam_icp_set_max_iterations", "slam_icp_set_min_delta", "slam_icp_set_subsampling_step",
    "slam_icp_fit", "slam_icp_fit_batch_dev", "slam_icp_fit_batch_from_dev", "slam_icp_nearest_dev", "slam_icp_get_edge_weight",
    "slam_icp_get_normals",
    "slam_icp_index_info", "slam_icp_list_info",
    "slam_grid_default_params", "slam_grid_create", "slam_grid_destroy", "slam_grid_clear", "slam_grid_reset_counts",
    "slam_grid_set_min_cluster_points", "slam_grid_set_max_range", "slam_grid_set_pose",
    "slam_grid_get_pose", "slam_grid_add_endpoints", "slam_grid_add_endpoints_dev",
    "slam_grid_raycast", "slam_grid_raycast_dev", "slam_grid_raycast_scans_dev", "slam_grid_reserve",
    "slam_grid_finalize", "slam_grid_finalize_reset", "slam_grid_add_scan_inorder", "slam_grid_add_scan_inorder_dev", "slam_grid_transform_cloud_dev", "slam_grid_read_counts",
    "slam_grid_read_occupancy", "slam_grid_read_num_pts", "slam_grid_total_updates",
    "slam_grid_info", "slam_grid_window_cell", "slam_grid_counts_dev", "slam_grid_mark_rows", "slam_grid_raycast_stats", "slam_grid_dirty_rows", "slam_grid_dirty_rows_dev",
    "slam_grid_enable_accumulator", "slam_grid_fold",
    "slam_gseg_default_params", "slam_gseg_create", "slam_gseg_destroy", "slam_gseg_reserve",
    "slam_gseg_segment", "slam_gseg_segment_dev", "slam_gseg_split_dev", "slam_gseg_read_model",
    "slam_gseg_classify_ga_dev", "slam_gseg_classify_ga_counted_dev", "slam_gseg_classify_ga_extent_dev",
    "slam_ccicp_create", "slam_ccicp_destroy", "slam_ccicp_voxel_downsample_dev", "slam_ccicp_split_dev",
    "slam_ccicp_height_dev", "slam_ccicp_bin_order_dev", "slam_ccicp_select_dev", "slam_ccicp_scene_dev",
    "slam_ccicp_height_pose_dev", "slam_ccicp_split_box_dev", "slam_ccicp_height_rpy_pose_dev", "slam_ccicp_height_rpy_pose_mirror_dev", "slam_ccicp_scene_cloud_dev", "slam_ccicp_pack_scans_dev",
    "slam_mapper_default_params", "slam_mapper_create", "slam_mapper_destroy", "slam_mapper_next_slot", "slam_mapper_slots",
    "slam_mapper_chunk_buffers", "slam_mapper_push", "slam_mapper_wait", "slam_mapper_finish", "slam_mapper_grid",
    "slam_mapper_target", "slam_mapper_stats", "slam_mapper_set_merge",
]


def lib():
    """Loads slam_amd/lib/libslam_mi355x.so (built by slam_amd.build).  Loud if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlamError(E_NO_DEVICE, "HIP library not built: %s is missing "
                        "(run `python -m slam_amd.build`); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.slam_last_error.restype = C.c_char_p
    L.slam_version.restype = C.c_char_p
    L.slam_icp_destroy.restype = None
    L.slam_ccicp_destroy.restype = None
    L.slam_ccicp_destroy.argtypes = [C.c_void_p]
    L.slam_ccicp_voxel_downsample_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                                  C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.slam_ccicp_split_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slam_ccicp_select_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.slam_ccicp_bin_order_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    L.slam_ccicp_height_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.slam_ccicp_scene_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slam_ccicp_height_pose_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_double, C.c_void_p, C.c_void_p]
    L.slam_ccicp_split_box_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.slam_ccicp_height_rpy_pose_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.slam_ccicp_scene_cloud_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.slam_ccicp_pack_scans_dev.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slam_gseg_classify_ga_counted_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slam_gseg_classify_ga_extent_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slam_ccicp_height_rpy_pose_mirror_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                        C.c_void_p]
    L.slam_grid_destroy.restype = None
    L.slam_icp_default_params.restype = None
    L.slam_grid_default_params.restype = None
    L.slam_malloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    L.slam_free.argtypes = [_vp]
    L.slam_memset.argtypes = [_vp, C.c_int, C.c_size_t, _vp]
    L.slam_memcpy_h2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.slam_memcpy_d2h.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.slam_memcpy_d2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.slam_host_alloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    L.slam_host_free.argtypes = [_vp]
    L.slam_memcpy_h2d_async.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.slam_memcpy_d2h_async.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.slam_stream_wait_event.argtypes = [_vp, _vp]
    L.slam_graph_begin_capture.argtypes = [_vp]
    L.slam_graph_end_capture.argtypes = [_vp, C.POINTER(_vp)]
    L.slam_graph_launch.argtypes = [_vp, _vp]
    L.slam_graph_destroy.argtypes = [_vp]
    L.slam_stream_create.argtypes = [C.POINTER(_vp)]
    L.slam_stream_create_with_priority.argtypes = [C.POINTER(_vp), C.c_int]
    L.slam_stream_create_reserving_cus.argtypes = [C.POINTER(_vp), C.c_int]
    L.slam_stream_destroy.argtypes = [_vp]
    L.slam_stream_synchronize.argtypes = [_vp]
    L.slam_event_create.argtypes = [C.POINTER(_vp)]
    L.slam_event_destroy.argtypes = [_vp]
    L.slam_event_record.argtypes = [_vp, _vp]
    L.slam_event_synchronize.argtypes = [_vp]
    L.slam_event_query.argtypes = [_vp, C.POINTER(C.c_int)]
    L.slam_event_elapsed_ms.argtypes = [_vp, _vp, C.POINTER(C.c_float)]
    L.slam_device_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.slam_icp_create.argtypes = [_vp, C.c_int, _vp, C.c_int, C.POINTER(IcpParams), C.POINTER(_vp)]
    L.slam_icp_create_dev.argtypes = [_vp, C.c_int, _vp, C.c_int, C.POINTER(IcpParams), C.POINTER(_vp)]
    L.slam_icp_build_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.slam_icp_index_blob.argtypes = [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.slam_icp_destroy.argtypes = [_vp]
    L.slam_icp_set_max_iterations.argtypes = [_vp, C.c_int]
    L.slam_icp_set_min_delta.argtypes = [_vp, C.c_double]
    L.slam_icp_set_subsampling_step.argtypes = [_vp, C.c_int]
    L.slam_icp_fit.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_double,
                               C.POINTER(IcpResult)]
    L.slam_icp_fit_batch_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_double, _vp,
                                         _vp, _vp]
    L.slam_icp_fit_batch_from_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_double, _vp,
                                              _vp, _vp]
    L.slam_icp_nearest_dev.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp]
    L.slam_icp_get_edge_weight.argtypes = [_vp, _vp]
    L.slam_icp_get_normals.argtypes = [_vp, _vp]
    L.slam_icp_list_info.argtypes = [_vp] + [C.c_void_p] * 6
    L.slam_icp_index_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_double), C.POINTER(C.c_int),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.slam_grid_create.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(GridParams),
                                   C.POINTER(_vp)]
    L.slam_grid_destroy.argtypes = [_vp]
    L.slam_grid_clear.argtypes = [_vp, _vp]
    L.slam_grid_reset_counts.argtypes = [_vp, _vp]
    L.slam_grid_set_min_cluster_points.argtypes = [_vp, C.c_int]
    L.slam_grid_set_max_range.argtypes = [_vp, C.c_double]
    L.slam_grid_set_pose.argtypes = [_vp, C.c_double, C.c_double, _vp]
    L.slam_grid_get_pose.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.slam_grid_add_endpoints.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int]
    L.slam_grid_add_endpoints_dev.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp]
    L.slam_grid_raycast.argtypes = [_vp, _vp, _vp, C.c_int]
    L.slam_grid_raycast_dev.argtypes = [_vp, _vp, _vp, C.c_int, _vp]
    L.slam_grid_raycast_scans_dev.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
    L.slam_grid_reserve.argtypes = [_vp, C.c_int]
    L.slam_grid_finalize.argtypes = [_vp, _vp]
    L.slam_grid_finalize_reset.argtypes = [_vp, _vp]
    L.slam_grid_add_scan_inorder.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int]
    L.slam_grid_add_scan_inorder_dev.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp]
    L.slam_host_is_pinned.argtypes = [_vp]
    L.slam_grid_transform_cloud_dev.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp, _vp]
    L.slam_grid_read_counts.argtypes = [_vp, _vp, _vp]
    L.slam_grid_read_occupancy.argtypes = [_vp, _vp]
    L.slam_grid_read_num_pts.argtypes = [_vp, _vp]
    L.slam_grid_total_updates.argtypes = [_vp, C.POINTER(C.c_uint64)]
    L.slam_grid_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.slam_grid_window_cell.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.slam_grid_counts_dev.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]
    L.slam_grid_mark_rows.argtypes = [_vp, C.c_int, C.c_int, _vp]
    L.slam_grid_dirty_rows.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.slam_grid_dirty_rows_dev.argtypes = [_vp, C.POINTER(_vp)]
    L.slam_grid_enable_accumulator.argtypes = [_vp]
    L.slam_grid_fold.argtypes = [_vp, C.c_int, C.c_int, _vp]
    L.slam_gseg_default_params.restype = None
    L.slam_gseg_default_params.argtypes = [C.POINTER(GsegParams)]
    L.slam_gseg_create.argtypes = [C.POINTER(GsegParams), C.POINTER(_vp)]
    L.slam_gseg_destroy.restype = None
    L.slam_gseg_destroy.argtypes = [_vp]
    L.slam_gseg_reserve.argtypes = [_vp, C.c_int]
    L.slam_gseg_segment.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp]
    L.slam_gseg_segment_dev.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, _vp]
    L.slam_gseg_split_dev.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.slam_gseg_read_model.argtypes = [_vp, _vp, _vp, _vp]
    L.slam_gseg_classify_ga_dev.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, _vp]
    L.slam_grid_raycast_stats.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.slam_mapper_default_params.restype = None
    L.slam_mapper_default_params.argtypes = [C.POINTER(MapperParams)]
    L.slam_mapper_create.argtypes = [C.POINTER(MapperParams), _vp, C.c_int, _vp, C.c_int, C.POINTER(_vp)]
    L.slam_mapper_destroy.restype = None
    L.slam_mapper_destroy.argtypes = [_vp]
    L.slam_mapper_next_slot.argtypes = [_vp, C.POINTER(C.c_int)]
    L.slam_mapper_slots.argtypes = [_vp, C.POINTER(C.c_int)]
    L.slam_mapper_chunk_buffers.argtypes = [_vp, C.c_int] + [C.POINTER(_vp)] * 5
    L.slam_mapper_push.argtypes = [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int)]
    L.slam_mapper_wait.argtypes = [_vp, C.c_int, _vp, _vp]
    L.slam_mapper_finish.argtypes = [_vp]
    L.slam_mapper_grid.argtypes = [_vp, C.POINTER(_vp)]
    L.slam_mapper_target.argtypes = [_vp, C.POINTER(_vp)]
    L.slam_mapper_stats.argtypes = [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]
    L.slam_mapper_set_merge.argtypes = [_vp, _vp, _vp, _vp]
    _lib = L
    return L


def check(rc):
    if rc != SLAM_OK:
        raise SlamError(rc, lib().slam_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().slam_device_count(C.byref(n)))
    return n.value


def set_device(i):
    check(lib().slam_set_device(int(i)))


def device_info():
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(), C.c_size_t()
    check(lib().slam_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return name.value.decode(), cu.value, mem.value


def synchronize():
    check(lib().slam_device_synchronize())


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


class DeviceArray:
    """A typed block of HBM owned through slam_malloc / slam_free."""

    def __init__(self, shape, dtype):
        self.shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _vp()
        check(lib().slam_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a, stream=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        check(lib().slam_memcpy_h2d(self.ptr, _ptr(a), self.nbytes, stream))

    def download(self, stream=None):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().slam_memcpy_d2h(_ptr(out), self.ptr, self.nbytes, stream))
        return out

    def upload_async(self, pinned, stream):
        assert pinned.nbytes == self.nbytes
        check(lib().slam_memcpy_h2d_async(self.ptr, pinned.ptr, self.nbytes, _sp(stream)))

    def download_async(self, pinned, stream):
        assert pinned.nbytes == self.nbytes
        check(lib().slam_memcpy_d2h_async(pinned.ptr, self.ptr, self.nbytes, _sp(stream)))

    def copy_from(self, other, stream=None):
        assert other.nbytes == self.nbytes
        check(lib().slam_memcpy_d2d(self.ptr, other.ptr, self.nbytes, _sp(stream)))

    def zero(self, stream=None):
        check(lib().slam_memset(self.ptr, 0, self.nbytes, _sp(stream)))

    def view(self, first, shape):
        """`shape` elements of this block from element `first` on: shares the memory, does not own it."""
        v = object.__new__(DeviceArray)
        v.shape = tuple(shape)
        v.dtype = self.dtype
        v.nbytes = int(np.prod(v.shape)) * self.dtype.itemsize
        assert first >= 0 and first * self.dtype.itemsize + v.nbytes <= self.nbytes
        v.ptr = self.ptr + first * self.dtype.itemsize
        v.owner = self   # keeps the block alive
        return v

    def free(self):
        if getattr(self, "ptr", None) and getattr(self, "owner", None) is None:
            lib().slam_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A numpy view over pinned host memory (slam_host_alloc), for asynchronous copies."""

    def __init__(self, shape, dtype):
        self.shape = shape if isinstance(shape, tuple) else (int(shape),)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _vp()
        check(lib().slam_host_alloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().slam_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self, priority=None, reserve_cus_per_xcd=0, private_queue=False):
        """reserve_cus_per_xcd > 0: a stream whose kernels leave that many CUs of every XCD alone; private_queue: a stream with a
        hardware queue of its own (slam_stream_create_reserving_cus with 0; such streams have no priority of their own)."""
        p = _vp()
        if reserve_cus_per_xcd or private_queue:
            check(lib().slam_stream_create_reserving_cus(C.byref(p), int(reserve_cus_per_xcd)))
        elif priority is None:
            check(lib().slam_stream_create(C.byref(p)))
        else:
            check(lib().slam_stream_create_with_priority(C.byref(p), int(priority)))
        self.ptr = p.value

    def synchronize(self):
        check(lib().slam_stream_synchronize(self.ptr))

    def wait_event(self, ev):
        check(lib().slam_stream_wait_event(self.ptr, ev.ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().slam_stream_destroy(self.ptr)
            self.ptr = None


class Graph:
    """hipGraph of the library calls issued on `stream` inside the with-block (record once, replay)."""

    def __init__(self, stream):
        self.stream, self.ptr = stream, None

    def __enter__(self):
        check(lib().slam_graph_begin_capture(self.stream.ptr))
        return self

    def __exit__(self, *exc):
        p = _vp()
        rc = lib().slam_graph_end_capture(self.stream.ptr, C.byref(p))
        if exc[0] is None:
            check(rc)
            self.ptr = p.value
        return False

    def launch(self, stream=None):
        check(lib().slam_graph_launch(self.ptr, (stream or self.stream).ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().slam_graph_destroy(self.ptr)
            self.ptr = None


class Event:
    def __init__(self):
        p = _vp()
        check(lib().slam_event_create(C.byref(p)))
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().slam_event_record(self.ptr, stream.ptr if isinstance(stream, Stream) else stream))

    def synchronize(self):
        check(lib().slam_event_synchronize(self.ptr))

    def query(self):
        """True when the work recorded before the event has finished (no wait)."""
        d = C.c_int()
        check(lib().slam_event_query(self.ptr, C.byref(d)))
        return bool(d.value)

    def elapsed_ms(self, later):
        ms = C.c_float()
        check(lib().slam_event_elapsed_ms(self.ptr, later.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().slam_event_destroy(self.ptr)
            self.ptr = None


def _sp(stream):
    return stream.ptr if isinstance(stream, Stream) else stream


def icp_default_params(**kw):
    p = IcpParams()
    lib().slam_icp_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Icp:
    """IcpPointToPoint-shaped handle (icpPointToPoint.h:26-40) over the C-ABI."""

    def __init__(self, m_ga, m_nga, params=None, **kw):
        self.m_ga = np.ascontiguousarray(m_ga, dtype=np.float64).reshape(-1, 2)
        self.m_nga = np.ascontiguousarray(m_nga, dtype=np.float64).reshape(-1, 2)
        self.params = params or icp_default_params(**kw)
        h = _vp()
        check(lib().slam_icp_create(_ptr(self.m_ga), len(self.m_ga), _ptr(self.m_nga),
                                    len(self.m_nga), C.byref(self.params), C.byref(h)))
        self.h = h.value

    @classmethod
    def from_device(cls, d_ga, n_ga, d_nga, n_nga, params=None, **kw):
        """slam_icp_create_dev: the model arrays (f64 xy) are DeviceArrays / device pointers."""
        self = object.__new__(cls)
        self.m_ga = self.m_nga = None
        self.n_model = (int(n_ga), int(n_nga))
        self.params = params or icp_default_params(**kw)
        h = _vp()
        check(lib().slam_icp_create_dev(getattr(d_ga, "ptr", d_ga), int(n_ga), getattr(d_nga, "ptr", d_nga), int(n_nga),
                                        C.byref(self.params), C.byref(h)))
        self.h = h.value
        return self

    def build_info(self):
        on, ms = C.c_int(0), (C.c_double * 4)()
        check(lib().slam_icp_build_info(self.h, C.byref(on), ms))
        return bool(on.value), list(ms)

    def index_blob(self, which):
        """The cell index (which = 0) or the halo lists (1) as they lie in HBM, as bytes."""
        n = C.c_size_t(0)
        check(lib().slam_icp_index_blob(self.h, int(which), None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.uint8)
        if n.value:
            check(lib().slam_icp_index_blob(self.h, int(which), _ptr(buf), n.value, None))
        return buf

    def set_max_iterations(self, v):
        check(lib().slam_icp_set_max_iterations(self.h, int(v)))

    def set_min_delta(self, v):
        check(lib().slam_icp_set_min_delta(self.h, float(v)))

    def index_info(self):
        nx, ny, lanes, in_lds = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        cell, lds = C.c_double(), C.c_size_t()
        check(lib().slam_icp_index_info(self.h, C.byref(nx), C.byref(ny), C.byref(cell),
                                        C.byref(in_lds), C.byref(lds), C.byref(lanes)))
        two, first = C.c_int(), C.c_int()
        pitch, halo, cert, lb = C.c_double(), C.c_double(), C.c_double(), C.c_size_t()
        check(lib().slam_icp_list_info(self.h, C.byref(two), C.byref(first), C.byref(pitch), C.byref(halo),
                                       C.byref(cert), C.byref(lb)))
        return dict(nx=nx.value, ny=ny.value, cell=cell.value, in_lds=bool(in_lds.value),
                    lds_bytes=lds.value, lanes_per_point=lanes.value, two_forms=bool(two.value),
                    first_iterations=first.value, list_pitch=pitch.value, list_halo=halo.value,
                    list_certified_radius=cert.value, list_bytes=lb.value)

    def fit(self, t_ga, t_nga, R, t, indist=5.0):
        """Icp::fit (icp.cpp:80-114), host arrays; returns (R, t, IcpResult)."""
        t_ga = np.ascontiguousarray(t_ga, dtype=np.float64).reshape(-1, 2)
        t_nga = np.ascontiguousarray(t_nga, dtype=np.float64).reshape(-1, 2)
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(4).copy()
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(2).copy()
        res = IcpResult()
        check(lib().slam_icp_fit(self.h, _ptr(t_ga), len(t_ga), _ptr(t_nga), len(t_nga),
                                 _ptr(R), _ptr(t), float(indist), C.byref(res)))
        return R.reshape(2, 2), t, res

    def fit_batch_dev(self, d_pts, d_off, d_nga, n_scans, d_R, d_t, indist=5.0, d_result=None,
                      d_trace=None, stream=None):
        check(lib().slam_icp_fit_batch_dev(
            self.h, d_pts.ptr, d_off.ptr, d_nga.ptr, int(n_scans), d_R.ptr, d_t.ptr, float(indist),
            d_result.ptr if d_result is not None else None,
            d_trace.ptr if d_trace is not None else None, _sp(stream)))

    def fit_batch_from_dev(self, d_pts, d_off, d_nga, n_scans, d_R0, d_t0, d_R, d_t, indist=5.0, d_result=None,
                           d_trace=None, stream=None):
        """Initial poses read from d_R0 / d_t0, registered poses written to d_R / d_t."""
        check(lib().slam_icp_fit_batch_from_dev(
            self.h, d_pts.ptr, d_off.ptr, d_nga.ptr, int(n_scans), d_R0.ptr, d_t0.ptr, d_R.ptr, d_t.ptr, float(indist),
            d_result.ptr if d_result is not None else None,
            d_trace.ptr if d_trace is not None else None, _sp(stream)))

    def fit_batch(self, batch, indist=5.0, trace=False):
        """Host convenience: uploads a synth.ScanBatch, runs, downloads.
        Returns (R[S,4], t[S,2], result[S], trace[S,max_iter,8] or None)."""
        S = batch.n_scans
        d_pts = DeviceArray.from_host(batch.pts, np.float64)
        d_off = DeviceArray.from_host(batch.scan_off, np.int32)
        d_nga = DeviceArray.from_host(batch.scan_nga, np.int32)
        d_R = DeviceArray.from_host(batch.R, np.float64)
        d_t = DeviceArray.from_host(batch.t, np.float64)
        d_res = DeviceArray((S,), RESULT_DTYPE)
        d_res.zero()
        d_tr = None
        if trace:
            d_tr = DeviceArray((S, max(self.params.max_iter, 1), 8), np.float64)
            d_tr.zero()
        self.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, indist, d_res, d_tr)
        synchronize()
        return d_R.download(), d_t.download(), d_res.download(), (d_tr.download() if trace else None)

    def edge_weight(self):
        """IcpPointToPoint::getEdgeWeight (icpPointToPoint.cpp:233-316) of the last fit()."""
        out = np.zeros(9)
        check(lib().slam_icp_get_edge_weight(self.h, _ptr(out)))
        return out.reshape(3, 3)

    def normals(self):
        n = sum(self.n_model) if self.m_ga is None else len(self.m_ga) + len(self.m_nga)
        out = np.zeros((n, 2))
        check(lib().slam_icp_get_normals(self.h, _ptr(out)))
        return out

    def nearest(self, cls, q_xy):
        """KDTree::n_nearest(q, 1) for every row of q_xy (f32): (dis[n], idx[n])."""
        q = np.ascontiguousarray(q_xy, dtype=np.float32).reshape(-1, 2)
        d_q = DeviceArray.from_host(q)
        d_d = DeviceArray((len(q),), np.float32)
        d_i = DeviceArray((len(q),), np.int32)
        check(lib().slam_icp_nearest_dev(self.h, int(cls), d_q.ptr, len(q), d_d.ptr, d_i.ptr, None))
        synchronize()
        return d_d.download(), d_i.download()

    def close(self):
        if getattr(self, "h", None):
            lib().slam_icp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def grid_default_params(**kw):
    p = GridParams()
    lib().slam_grid_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Grid:
    """MLS-in-occupancy-mode-shaped handle (mls.h:104-242) over the C-ABI."""

    def __init__(self, size_x, size_y, resolution, params=None, **kw):
        self.size_x, self.size_y, self.resolution = int(size_x), int(size_y), float(resolution)
        self.params = params or grid_default_params(**kw)
        h = _vp()
        check(lib().slam_grid_create(self.size_x, self.size_y, self.resolution,
                                     C.byref(self.params), C.byref(h)))
        self.h = h.value
        self.cells = self.size_x * self.size_y

    def clear(self, stream=None):
        check(lib().slam_grid_clear(self.h, _sp(stream)))

    def reset_counts(self, stream=None):
        check(lib().slam_grid_reset_counts(self.h, _sp(stream)))

    def set_pose(self, x, y, stream=None):
        check(lib().slam_grid_set_pose(self.h, float(x), float(y), _sp(stream)))

    def get_pose(self):
        x, y = C.c_double(), C.c_double()
        check(lib().slam_grid_get_pose(self.h, C.byref(x), C.byref(y)))
        return x.value, y.value

    def set_min_cluster_points(self, v):
        check(lib().slam_grid_set_min_cluster_points(self.h, int(v)))

    def set_max_range(self, v):
        check(lib().slam_grid_set_max_range(self.h, float(v)))

    @staticmethod
    def _pts(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 1:
            a = a.reshape(-1, 2)
        return a

    def add_endpoints(self, obs, gnd):
        obs, gnd = self._pts(obs), self._pts(gnd)
        stride = obs.shape[1] if obs.size else (gnd.shape[1] if gnd.size else 2)
        check(lib().slam_grid_add_endpoints(self.h, _ptr(obs), len(obs), _ptr(gnd), len(gnd), stride))

    def add_scan_inorder(self, obs, gnd):
        obs, gnd = self._pts(obs), self._pts(gnd)
        stride = obs.shape[1] if obs.size else (gnd.shape[1] if gnd.size else 2)
        check(lib().slam_grid_add_scan_inorder(self.h, _ptr(obs), len(obs), _ptr(gnd), len(gnd),
                                               stride))

    def raycast(self, origin_xy, end_xy):
        o, e = self._pts(origin_xy), self._pts(end_xy)
        assert o.shape == e.shape and o.shape[1] == 2
        check(lib().slam_grid_raycast(self.h, _ptr(o), _ptr(e), len(e)))

    def raycast_dev(self, d_origin, d_end, n, stream=None):
        check(lib().slam_grid_raycast_dev(self.h, d_origin.ptr, d_end.ptr, int(n), _sp(stream)))

    def raycast_scans_dev(self, d_pts, d_off, n_scans, n_points, d_R, d_t, stream=None):
        check(lib().slam_grid_raycast_scans_dev(self.h, d_pts.ptr, d_off.ptr, int(n_scans),
                                                int(n_points), d_R.ptr, d_t.ptr, _sp(stream)))

    def reserve(self, max_beams):
        check(lib().slam_grid_reserve(self.h, int(max_beams)))

    def finalize(self, stream=None):
        check(lib().slam_grid_finalize(self.h, _sp(stream)))

    def finalize_reset(self, stream=None):
        check(lib().slam_grid_finalize_reset(self.h, _sp(stream)))

    def read_counts(self):
        hits = np.empty(self.cells, dtype=np.int32)
        misses = np.empty(self.cells, dtype=np.int32)
        check(lib().slam_grid_read_counts(self.h, _ptr(hits), _ptr(misses)))
        return hits, misses

    def read_occupancy(self):
        occ = np.empty(self.cells, dtype=np.int8)
        check(lib().slam_grid_read_occupancy(self.h, _ptr(occ)))
        return occ

    def read_num_pts(self):
        v = np.empty(self.cells, dtype=np.float64)
        check(lib().slam_grid_read_num_pts(self.h, _ptr(v)))
        return v

    def total_updates(self):
        n = C.c_uint64()
        check(lib().slam_grid_total_updates(self.h, C.byref(n)))
        return n.value

    def info(self):
        sx, sy, ox, oy, res = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(lib().slam_grid_info(self.h, C.byref(sx), C.byref(sy), C.byref(res), C.byref(ox),
                                   C.byref(oy)))
        return dict(size_x=sx.value, size_y=sy.value, resolution=res.value, origin_x=ox.value,
                    origin_y=oy.value)

    def window_cell(self):
        x, y = C.c_int(), C.c_int()
        check(lib().slam_grid_window_cell(self.h, C.byref(x), C.byref(y)))
        return x.value, y.value

    def raycast_stats(self):
        t, i, s = C.c_int(), C.c_int(), C.c_int()
        check(lib().slam_grid_raycast_stats(self.h, C.byref(t), C.byref(i), C.byref(s)))
        return dict(tiles=t.value, items=i.value, tile_write_backs=s.value)

    def dirty_rows(self):
        lo, hi = C.c_int(), C.c_int()
        check(lib().slam_grid_dirty_rows(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def enable_accumulator(self):
        check(lib().slam_grid_enable_accumulator(self.h))

    def fold(self, row_lo, row_hi, stream=None):
        check(lib().slam_grid_fold(self.h, int(row_lo), int(row_hi), _sp(stream)))

    def mark_rows(self, lo, hi, stream=None):
        """storage rows lo..hi of the planes were written through counts_dev()'s pointer"""
        check(lib().slam_grid_mark_rows(self.h, int(lo), int(hi), _sp(stream)))

    def counts_dev(self):
        p, n = _vp(), C.c_size_t()
        check(lib().slam_grid_counts_dev(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        if getattr(self, "h", None):
            lib().slam_grid_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mapper:
    """slam_mapper_t: the streaming form of the path (BASELINE config 5).  push() copies a chunk of a ScanBatch into
    the next slot's pinned buffers and enqueues it; wait() returns its registered poses."""

    def __init__(self, m_ga, m_nga, grid=None, icp=None, **kw):
        p = MapperParams()
        lib().slam_mapper_default_params(C.byref(p))
        for k, v in (grid or {}).items():
            setattr(p.grid, k, v)
        for k, v in (icp or {}).items():
            setattr(p.icp, k, v)
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        m_ga = np.ascontiguousarray(m_ga, dtype=np.float64).reshape(-1, 2)
        m_nga = np.ascontiguousarray(m_nga, dtype=np.float64).reshape(-1, 2)
        h = _vp()
        check(lib().slam_mapper_create(C.byref(p), _ptr(m_ga), len(m_ga), _ptr(m_nga), len(m_nga), C.byref(h)))
        self.h = h.value
        self._views = {}
        ns = C.c_int()
        check(lib().slam_mapper_slots(self.h, C.byref(ns)))
        self.n_slots = ns.value
        g = _vp()
        check(lib().slam_mapper_grid(self.h, C.byref(g)))
        self.grid = object.__new__(Grid)
        self.grid.h, self.grid.size_x, self.grid.size_y = g.value, p.grid_size_x, p.grid_size_y
        self.grid.resolution, self.grid.cells, self.grid.params = p.resolution, p.grid_size_x * p.grid_size_y, p.grid
        self.grid.close = lambda: None      # owned by the mapper

    def _slot_views(self, slot):
        if slot not in self._views:
            ptrs = [_vp() for _ in range(5)]
            check(lib().slam_mapper_chunk_buffers(self.h, slot, *[C.byref(x) for x in ptrs]))
            ns, npts = self.params.max_scans, self.params.max_points

            def view(ptr, count, dtype):
                buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr.value)
                return np.frombuffer(buf, dtype=dtype, count=count)
            self._views[slot] = (view(ptrs[0], 2 * npts, np.float64), view(ptrs[1], ns + 1, np.int32),
                                 view(ptrs[2], ns, np.int32), view(ptrs[3], 4 * ns, np.float64),
                                 view(ptrs[4], 2 * ns, np.float64))
        return self._views[slot]

    def push(self, batch, window_xy=(0.0, 0.0)):
        """batch: a synth.ScanBatch (scan_off from 0).  Returns the slot."""
        slot = C.c_int()
        check(lib().slam_mapper_next_slot(self.h, C.byref(slot)))
        pts, off, nga, R, t = self._slot_views(slot.value)
        S, P = batch.n_scans, batch.n_points
        pts[:2 * P] = batch.pts.reshape(-1)
        off[:S + 1] = batch.scan_off
        nga[:S] = batch.scan_nga
        R[:4 * S] = batch.R.reshape(-1)
        t[:2 * S] = batch.t.reshape(-1)
        out = C.c_int()
        check(lib().slam_mapper_push(self.h, S, P, float(window_xy[0]), float(window_xy[1]), C.byref(out)))
        self._n = getattr(self, "_n", {})
        self._n[out.value] = S
        return out.value

    def wait(self, slot):
        S = self._n.get(slot, 0)
        R, t = np.zeros((S, 4)), np.zeros((S, 2))
        check(lib().slam_mapper_wait(self.h, int(slot), _ptr(R), _ptr(t)))
        return R, t

    def finish(self):
        check(lib().slam_mapper_finish(self.h))

    def stats(self):
        c, m, r = C.c_long(), C.c_long(), C.c_long()
        ms, rows = C.c_double(), (C.c_int * 2)()
        check(lib().slam_mapper_stats(self.h, C.byref(c), C.byref(m), C.byref(r), C.byref(ms), rows))
        return dict(chunks=c.value, merges=m.value, rebuilds=r.value, rebuild_ms=ms.value, last_merge_rows=(rows[0], rows[1]))

    def target_index_info(self):
        h = _vp()
        check(lib().slam_mapper_target(self.h, C.byref(h)))
        icp = object.__new__(Icp)
        icp.h = h.value
        info = icp.index_info()
        info["built_on_device"], info["build_host_ms"] = icp.build_info()
        icp.h = None
        return info

    def use_comm(self, comm):
        check(rccl_lib().slam_mapper_use_comm(self.h, comm.h))
        self._comm = comm

    def close(self):
        if getattr(self, "h", None):
            lib().slam_mapper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroundSegmentation:
    """groundSegmentation-shaped handle (groundSegmentation.h:67-128) over the C-ABI."""

    def __init__(self, **kw):
        self.params = GsegParams()
        lib().slam_gseg_default_params(C.byref(self.params))
        for k, v in kw.items():
            setattr(self.params, k, v)
        h = _vp()
        check(lib().slam_gseg_create(C.byref(self.params), C.byref(h)))
        self.h = h.value

    def segment(self, xyz):
        """setupGroundSegmentation + segmentGround: one GSEG_* label per point."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = xyz.shape
        labels = np.zeros(max(n, 1), dtype=np.uint8)
        check(lib().slam_gseg_segment(self.h, _ptr(xyz), n, stride, _ptr(labels)))
        return labels[:n]

    def segment_dev(self, d_xyz, n, stride, d_labels, stream=None):
        check(lib().slam_gseg_segment_dev(self.h, d_xyz.ptr, int(n), int(stride), d_labels.ptr, _sp(stream)))

    def split_dev(self, d_xyz, n, stride, d_labels, d_ground, d_obstacle, d_counts, stream=None):
        check(lib().slam_gseg_split_dev(self.h, d_xyz.ptr, int(n), int(stride), d_labels.ptr, d_ground.ptr,
                                        d_obstacle.ptr, d_counts.ptr, _sp(stream)))

    def classify_ga(self, obstacle_xyz):
        """CCICP::classifyPoints over an obstacle cloud (host arrays): flags 1 GA / 0 NGA / 255 dropped."""
        xyz = np.ascontiguousarray(obstacle_xyz, dtype=np.float32)
        n, stride = xyz.shape
        if n == 0:
            return np.zeros(0, np.uint8)
        d_xyz = DeviceArray.from_host(xyz)
        d_f = DeviceArray((n,), np.uint8)
        check(lib().slam_gseg_classify_ga_dev(self.h, d_xyz.ptr, n, stride, d_f.ptr, None))
        synchronize()
        return d_f.download()

    def read_model(self):
        state = np.zeros(72 * 200, dtype=np.uint8)
        value = np.zeros(72 * 200)
        iters = np.zeros(72, dtype=np.int32)
        check(lib().slam_gseg_read_model(self.h, _ptr(state), _ptr(value), _ptr(iters)))
        return state, value, iters

    def close(self):
        if getattr(self, "h", None):
            lib().slam_gseg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ccicp:
    """The CCICP facade steps either side of the ICP (icpTools.cpp:222-381, 611-634) over the C-ABI."""
    ICP_MAX_PTS = 20000  # icpTools.h:21

    def __init__(self):
        h = _vp()
        check(lib().slam_ccicp_create(C.byref(h)))
        self.h = h.value

    def voxel_downsample(self, xyz, flags=None, leaf=(0.5, 0.5, 2.0)):
        """pcl::VoxelGrid as setSceneCloud uses it; returns [n_out, 4] = centroid x,y,z, ground_adj."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = xyz.shape
        if n == 0:
            return np.zeros((0, 4), np.float32)
        d_xyz = DeviceArray.from_host(xyz)
        d_flag = DeviceArray.from_host(np.ascontiguousarray(flags, np.uint8)) if flags is not None else None
        d_out = DeviceArray((n, 4), np.float32)
        n_out = C.c_int(0)
        check(lib().slam_ccicp_voxel_downsample_dev(self.h, d_xyz.ptr, d_flag.ptr if d_flag else None, n, stride,
                                                    leaf[0], leaf[1], leaf[2], d_out.ptr, n, C.byref(n_out), None))
        return d_out.download()[:n_out.value]

    def bin_order(self, xyz, flags):
        """The cloud in classifyPoints order with its flags: [n_kept, 4]."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = xyz.shape
        if n == 0:
            return np.zeros((0, 4), np.float32)
        d_xyz = DeviceArray.from_host(xyz)
        d_flag = DeviceArray.from_host(np.ascontiguousarray(flags, np.uint8))
        d_out = DeviceArray((n, 4), np.float32)
        n_out = C.c_int(0)
        check(lib().slam_ccicp_bin_order_dev(self.h, d_xyz.ptr, d_flag.ptr, n, stride, d_out.ptr, C.byref(n_out), None))
        return d_out.download()[:n_out.value]

    def split(self, xyzg, pose_xy=None, crop_dist=75.0, cap=ICP_MAX_PTS):
        """doICPMatch marshalling: optional crop around pose_xy, then (ga_xy, nga_xy) f64 with the cap."""
        xyzg = np.ascontiguousarray(xyzg, dtype=np.float32)
        n, stride = xyzg.shape
        if n == 0:
            return np.zeros((0, 2)), np.zeros((0, 2))
        d_in = DeviceArray.from_host(xyzg)
        d_ga = DeviceArray((cap, 2), np.float64)
        d_nga = DeviceArray((cap, 2), np.float64)
        counts = (C.c_int * 2)()
        cx, cy = pose_xy if pose_xy is not None else (0.0, 0.0)
        check(lib().slam_ccicp_split_dev(self.h, d_in.ptr, n, stride, 1 if pose_xy is not None else 0, cx, cy,
                                         crop_dist, cap, d_ga.ptr, d_nga.ptr, counts, None))
        return d_ga.download()[:counts[0]], d_nga.download()[:counts[1]]

    def height(self, ground_xyz, pose7):
        """doHeightInterpolate: (z, n_corr, nn_idx[4])."""
        g = np.ascontiguousarray(ground_xyz, dtype=np.float32)
        n, stride = g.shape if g.ndim == 2 else (0, 3)
        pose = (C.c_double * 7)(*pose7)
        z = C.c_double(0.0)
        nc = C.c_int(0)
        idx = (C.c_int * 4)()
        d_g = DeviceArray.from_host(g) if n else None
        check(lib().slam_ccicp_height_dev(self.h, d_g.ptr if d_g else None, n, stride, pose, C.byref(z), C.byref(nc),
                                          idx, None))
        return z.value, nc.value, list(idx)

    def close(self):
        if getattr(self, "h", None):
            lib().slam_ccicp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ RCCL merge
_rccl = None
RCCL_EXPORTS = ["slam_comm_unique_id", "slam_comm_create", "slam_comm_create_host", "slam_comm_adopt", "slam_comm_destroy",
                "slam_comm_info", "slam_comm_get_stats", "slam_comm_stats_reset", "slam_grid_allreduce", "slam_grid_allreduce_rows", "slam_grid_merge_begin",
                "slam_grid_merge_finish", "slam_grid_merge_async", "slam_comm_ticket_wait", "slam_comm_drain", "slam_comm_set_timeout",
                "slam_comm_check", "slam_mapper_use_comm"]


COMM_SUM, COMM_MIN = 0, 1
MERGE_THEN_NOTHING, MERGE_THEN_FINALIZE_RESET, MERGE_THEN_FOLD_FINALIZE = 0, 1, 2
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, C.POINTER(C.c_int32), C.c_size_t, C.c_int)


def rccl_lib():
    """Loads slam_amd/lib/libslam_mi355x_rccl.so (include/slam_mi355x_rccl.h)."""
    global _rccl
    if _rccl is not None:
        return _rccl
    lib()  # the core library first: the RCCL object links against it
    if not os.path.exists(RCCL_LIB_PATH):
        raise SlamError(E_UNSUPPORTED, "RCCL merge library not built: %s is missing" % RCCL_LIB_PATH)
    R = C.CDLL(RCCL_LIB_PATH, mode=C.RTLD_GLOBAL)
    R.slam_comm_unique_id.argtypes = [C.c_char_p]
    R.slam_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]
    R.slam_comm_adopt.argtypes = [_vp, C.POINTER(_vp)]
    R.slam_comm_create_host.argtypes = [C.c_int, C.c_int, HOST_ALLREDUCE_FN, _vp, C.POINTER(_vp)]
    R.slam_comm_destroy.argtypes = [_vp]
    R.slam_comm_destroy.restype = None
    R.slam_comm_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.slam_comm_get_stats.argtypes = [_vp, C.POINTER(CommStats)]
    R.slam_comm_stats_reset.argtypes = [_vp]
    R.slam_grid_allreduce.argtypes = [_vp, _vp, _vp]
    R.slam_grid_allreduce_rows.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp]
    R.slam_grid_merge_begin.argtypes = [_vp, _vp, _vp]
    R.slam_grid_merge_finish.argtypes = [_vp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.slam_grid_merge_async.argtypes = [_vp, _vp, _vp, C.c_int, _vp, C.POINTER(C.c_ulonglong)]
    R.slam_comm_ticket_wait.argtypes = [_vp, C.c_ulonglong, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.slam_comm_drain.argtypes = [_vp]
    R.slam_comm_set_timeout.argtypes = [_vp, C.c_double]
    R.slam_comm_check.argtypes = [_vp]
    R.slam_mapper_use_comm.argtypes = [_vp, _vp]
    _rccl = R
    return R


class CommStats(C.Structure):
    _fields_ = [("rank", C.c_int), ("n_ranks", C.c_int), ("transport", C.c_int), ("rccl_version", C.c_int),
                ("merges", C.c_longlong), ("rows", C.c_longlong), ("bytes", C.c_longlong), ("wait_ms", C.c_double),
                ("allreduce_ms", C.c_double), ("timed", C.c_longlong), ("helper_wait_ms", C.c_double), ("async_merges", C.c_longlong)]


class Comm:
    """One RCCL communicator per process/GPU.  `id_bytes` comes from Comm.unique_id()
    on rank 0 and is handed to the other ranks out of band (e.g. a torch.distributed
    broadcast or a file)."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(rccl_lib().slam_comm_unique_id(buf))
        return buf.raw

    def __init__(self, id_bytes, rank, n_ranks):
        h = _vp()
        check(rccl_lib().slam_comm_create(id_bytes, int(rank), int(n_ranks), C.byref(h)))
        self.h = h.value

    @classmethod
    def host(cls, rank, n_ranks, allreduce):
        """slam_comm_create_host: a communicator over a host transport.  allreduce(array, op) reduces a numpy int32
        array in place over all ranks (op = COMM_SUM / COMM_MIN) -- e.g. a gloo all_reduce of torch.from_numpy(array)."""
        self = object.__new__(cls)

        def thunk(_ctx, buf, count, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)), op)
                return 0
            except Exception as ex:   # an exception must not unwind through the C frames
                import sys
                print("host all-reduce failed: %r" % (ex,), file=sys.stderr)
                return 1
        self._thunk = HOST_ALLREDUCE_FN(thunk)      # kept alive as long as the communicator
        h = _vp()
        check(rccl_lib().slam_comm_create_host(int(rank), int(n_ranks), self._thunk, None, C.byref(h)))
        self.h = h.value
        return self

    def info(self):
        r, n = C.c_int(), C.c_int()
        check(rccl_lib().slam_comm_info(self.h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def stats(self):
        """slam_comm_get_stats as a dict (waits for the timed all-reduces)."""
        st = CommStats()
        check(rccl_lib().slam_comm_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in CommStats._fields_}

    def stats_reset(self):
        check(rccl_lib().slam_comm_stats_reset(self.h))

    def allreduce_grid(self, grid, stream=None):
        check(rccl_lib().slam_grid_allreduce(grid.h, self.h, _sp(stream)))

    def allreduce_rows(self, grid, row_lo, row_hi, stream=None):
        check(rccl_lib().slam_grid_allreduce_rows(grid.h, self.h, int(row_lo), int(row_hi), _sp(stream)))

    def merge_begin(self, grid, stream=None):
        """Unites the ranks' device-tracked dirty rows (8-byte all-reduce) and sends them to the host; returns at once."""
        check(rccl_lib().slam_grid_merge_begin(grid.h, self.h, _sp(stream)))

    def merge_finish(self, grid, stream=None):
        """Waits for the united range, enqueues the all-reduce of those rows; returns (row_lo, row_hi)."""
        lo, hi = C.c_int(), C.c_int()
        check(rccl_lib().slam_grid_merge_finish(grid.h, self.h, _sp(stream), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def merge_async(self, grid, stream=None, then=MERGE_THEN_FINALIZE_RESET, done=None):
        """slam_grid_merge_async: the whole merge (+ what follows it on the stream) from the communicator's helper thread;
        returns a ticket at once.  Nothing else goes to `stream` / `grid` before ticket_wait(ticket)."""
        t = C.c_ulonglong()
        check(rccl_lib().slam_grid_merge_async(grid.h, self.h, _sp(stream), int(then), done.ptr if done is not None else None, C.byref(t)))
        return t.value

    def ticket_wait(self, ticket):
        """The status and united row range (row_lo, row_hi) of a merge posted with merge_async (waits for the helper thread
        to have enqueued it; raises SlamError with E_TIMEOUT / E_COMM when a rank is lost)."""
        lo, hi = C.c_int(), C.c_int()
        check(rccl_lib().slam_comm_ticket_wait(self.h, int(ticket), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def drain(self):
        check(rccl_lib().slam_comm_drain(self.h))

    def set_timeout(self, seconds):
        check(rccl_lib().slam_comm_set_timeout(self.h, float(seconds)))

    def check(self):
        check(rccl_lib().slam_comm_check(self.h))

    def close(self):
        if getattr(self, "h", None):
            rccl_lib().slam_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
