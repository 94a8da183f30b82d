"""ctypes binding of librsgpu.so (include/rsgpu.h) for the Python harness.

PyTorch is plumbing only here: device memory (torch tensors) and the stream.
Every call goes through the C-ABI; a missing library or a missing GPU raises —
there is no CPU fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RS_STAMPS=1 selects the instrumented build (tools/*_stamps.py); the product library otherwise
# RS_LIB=<file name in this directory> selects an A/B build (build.py, RS_VARIANT)
LIB_PATH = os.path.join(_HERE, os.environ.get("RS_LIB") or ("librsgpu_stamps.so" if os.environ.get("RS_STAMPS") else "librsgpu.so"))

EXPORTS = [
    "rs_abi_version", "rs_context_create", "rs_context_destroy", "rs_context_set_stream", "rs_context_wait_for", "rs_context_fork",
    "rs_context_synchronize", "rs_context_set_int", "rs_stage_begin", "rs_stage_alloc", "rs_stage_upload", "rs_stage_download", "rs_stage_sync", "rs_last_error", "rs_hamming_knn2", "rs_match_descriptors",
    "rs_kdtree_build", "rs_kdtree_pack", "rs_reproj_match", "rs_reproj_match_sharded", "rs_map_create", "rs_map_destroy", "rs_frame_create", "rs_frame_destroy",
    "rs_map_add_keyframe", "rs_map_set_keyframe_pose", "rs_map_add_point", "rs_map_set_position", "rs_map_remove_point",
    "rs_map_add_observation", "rs_map_remove_observation", "rs_map_counts", "rs_map_get_positions", "rs_map_match", "rs_map_pose_graph", "rs_pose_graph", "rs_pose_relative", "rs_transform_points", "rs_map_bundle_adjust", "rs_triangulate", "rs_triangulate_host", "rs_triangulate_matches", "rs_triangulate_matches_batch", "rs_triangulate_tracks", "rs_parallax_requirements", "rs_point_errors", "rs_ba_default_options",
    "rs_bundle_adjust", "rs_bundle_adjust_batch", "rs_ba_get_trace", "rs_ba_get_stats", "rs_ba_get_cameras", "rs_reanchor_points", "rs_reanchor_points_host_poses", "rs_refine_pose", "rs_bundle_adjust_inertial", "rs_refine_pose_inertial", "rs_pack_pose", "rs_unpack_pose", "rs_pack_poses", "rs_unpack_poses", "rs_build_local_window",
    "rs_comm_get_unique_id", "rs_comm_init_rank", "rs_comm_destroy", "rs_comm_init_local", "rs_comm_count", "rs_prof_begin", "rs_prof_end", "rs_prof_counters", "rs_prof_empty_launch",
]

_lib = None


class RsError(RuntimeError):
    pass


def load():
    """Loads librsgpu.so.  torch is imported first so that the HIP runtime the
    library binds to (soname libamdhip64.so.7) is the one torch already mapped."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                          "(hipcc, gfx950); there is no CPU fallback")
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.rs_last_error.restype = C.c_char_p
        _lib.rs_last_error.argtypes = [C.c_void_p]
    return _lib


class FrameView(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("width", C.c_int), ("height", C.c_int), ("n_keypoints", C.c_int),
                ("d_keypoints", C.c_void_p), ("d_descriptors", C.c_void_p), ("d_kp_matched", C.c_void_p),
                ("d_kd_node_kp", C.c_void_p), ("d_kd_left", C.c_void_p), ("d_kd_right", C.c_void_p),
                ("kd_root", C.c_int), ("d_kd_packed", C.c_void_p)]


class MapView(C.Structure):
    _fields_ = [("n_points", C.c_int), ("d_positions", C.c_void_p), ("d_eligible", C.c_void_p),
                ("d_obs_ptr", C.c_void_p), ("d_obs_kf", C.c_void_p), ("d_obs_desc", C.c_void_p),
                ("d_kf_centers", C.c_void_p), ("d_desc_pool", C.c_void_p)]


class BaOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("huber_delta", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("max_num_consecutive_invalid_steps", C.c_int),
                ("jacobi_scaling", C.c_int)]


class BaSummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("iterations", C.c_int), ("successful_steps", C.c_int),
                ("usable", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BaIteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("candidate_cost", C.c_double), ("model_cost_change", C.c_double),
                ("radius", C.c_double), ("step_norm", C.c_double), ("x_norm", C.c_double),
                ("outcome", C.c_int), ("reserved0", C.c_int), ("reserved1", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class ImuFactor(C.Structure):
    """rs_imu_factor"""
    _fields_ = [("cam_i", C.c_int), ("cam_j", C.c_int), ("duration", C.c_double), ("rotation", C.c_double * 9),
                ("velocity", C.c_double * 3), ("position", C.c_double * 3), ("covariance", C.c_double * 81),
                ("bias_gyro", C.c_double * 3), ("bias_accel", C.c_double * 3), ("bias_jacobian", C.c_double * 54),
                ("gyro_bias_sigma", C.c_double), ("accel_bias_sigma", C.c_double)]


def imu_factor_array(imu):
    """synth.make_imu(...) dict -> ctypes array of rs_imu_factor."""
    n = len(imu["cam_i"])
    arr = (ImuFactor * max(n, 1))()
    for f in range(n):
        a = arr[f]
        a.cam_i, a.cam_j, a.duration = int(imu["cam_i"][f]), int(imu["cam_j"][f]), float(imu["duration"][f])
        for name in ("rotation", "velocity", "position", "covariance", "bias_gyro", "bias_accel", "bias_jacobian"):
            getattr(a, name)[:] = list(np.asarray(imu[name][f], np.float64).ravel())
        a.gyro_bias_sigma, a.accel_bias_sigma = float(imu["gyro_bias_sigma"]), float(imu["accel_bias_sigma"])
    return arr, n


class BaProblem(C.Structure):
    """rs_ba_problem"""
    _fields_ = [("n_cameras", C.c_int), ("n_points", C.c_int), ("n_obs", C.c_int), ("d_cameras", C.c_void_p),
                ("h_cam_free", C.c_void_p), ("d_points", C.c_void_p), ("d_obs_ptr", C.c_void_p), ("d_obs_cam", C.c_void_p),
                ("d_obs_uv", C.c_void_p), ("intrinsics", C.c_float * 4)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int), ("total_ms", C.c_double)]


def default_options():
    o = BaOptions()
    load().rs_ba_default_options(C.byref(o))
    return o


def _dp(t):
    """device pointer of a torch tensor (or None).  Checked and boxed once per tensor object (cached on it: a
    benchmark pass makes ~60 of these conversions, at ~1 us each they rival the kernels they feed); tensors handed to
    the library must not be resized afterwards."""
    if t is None:
        return None
    p = t.__dict__.get("_rs_ptr")
    if p is None:
        assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA/HIP tensors"
        p = C.c_void_p(t.data_ptr())
        t._rs_ptr = p
    return p


# ---- host-only helpers (no GPU needed) ---------------------------------------
class PoseGraphEdge(C.Structure):
    """rs_pose_graph_edge"""
    _fields_ = [("from_", C.c_int32), ("to", C.c_int32), ("relative", C.c_double * 16)]


def pose_graph_edges(loops):
    """[(from, to, relative 4x4), ...] -> ctypes array of rs_pose_graph_edge"""
    arr = (PoseGraphEdge * max(len(loops), 1))()
    for i, (a, b, rel) in enumerate(loops):
        arr[i].from_, arr[i].to = int(a), int(b)
        arr[i].relative[:] = list(np.asarray(rel, np.float64).reshape(16))
    return arr


def pose_graph(poses, loops, four_dof=False, gravity=(0.0, 0.0, 0.0), options=None):
    """optimization::pose_graph (host function).  Returns (poses' [n,4,4] f32, velocity rotations [n,3,3] f32, summary, trace)."""
    P = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    out, rot = np.zeros_like(P), np.zeros((len(P), 9), np.float32)
    g = (C.c_double * 3)(*[float(v) for v in gravity])
    s = BaSummary()
    cap = 256
    buf = (BaIteration * cap)()
    cnt = C.c_int(0)
    L = load()
    L.rs_pose_graph.restype = C.c_int
    rc = L.rs_pose_graph(len(P), P.ctypes.data_as(C.c_void_p), pose_graph_edges(loops), len(loops), int(bool(four_dof)), g,
                         None if options is None else C.byref(options), out.ctypes.data_as(C.c_void_p),
                         rot.ctypes.data_as(C.c_void_p), C.byref(s), buf, cap, C.byref(cnt))
    if rc:
        raise RuntimeError(f"rs_pose_graph failed with {rc}")
    return out.reshape(-1, 4, 4), rot.reshape(-1, 3, 3), s.as_dict(), [buf[i].as_dict() for i in range(min(cnt.value, cap))]


def pose_relative(pose_from, pose_to):
    rel = np.zeros(16)
    a = np.ascontiguousarray(pose_from, np.float32).reshape(16)
    b = np.ascontiguousarray(pose_to, np.float32).reshape(16)
    load().rs_pose_relative(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), rel.ctypes.data_as(C.c_void_p))
    return rel.reshape(4, 4)


def pack_pose(pose):
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    cam = np.zeros(6)
    load().rs_pack_pose(p.ctypes.data_as(C.c_void_p), cam.ctypes.data_as(C.c_void_p))
    return cam


def unpack_pose(cam):
    c = np.ascontiguousarray(cam, np.float64)
    p = np.zeros(16, np.float32)
    load().rs_unpack_pose(c.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))
    return p.reshape(4, 4)


_HOST_PTR = {}


def _hp(a):
    """Boxed address of a host array, cached per array object (ndarray.ctypes costs ~2 us a time; the per-frame calls of a
    pass hand the same few arrays over and over).  The cache keeps the array alive; arrays must not be resized."""
    e = _HOST_PTR.get(id(a))
    if e is None or e[0] is not a:
        if len(_HOST_PTR) > 256:
            _HOST_PTR.clear()
        e = _HOST_PTR[id(a)] = (a, C.c_void_p(a.ctypes.data))
    return e[1]


def unpack_poses(cams, mask, out):
    """In place: out [n][16] f32 rows of the frames selected by mask [n] u8 (None = all) are rewritten."""
    c = np.ascontiguousarray(cams, np.float64)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == 16 * len(c)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    load().rs_unpack_poses(_hp(c), len(c), None if m is None else _hp(m), _hp(out))
    return out


def parallax_requirements(poses, kf_pose, min_parallax_cosine=0.999848, rotation_parallax_factor=0.20):
    """rs_parallax_requirements (host, the host's libm): required parallax cosine per first-sighting pose [n] f32."""
    P = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    out = np.zeros(len(P), np.float32)
    rc = load().rs_parallax_requirements(P.ctypes.data_as(C.c_void_p), len(P), int(kf_pose), C.c_float(min_parallax_cosine),
                                         C.c_float(rotation_parallax_factor), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RsError(f"rs_parallax_requirements -> {rc}")
    return out


def kdtree_build(keypoints):
    kp = np.ascontiguousarray(keypoints, np.float32)
    n = len(kp)
    node_kp = np.zeros(max(n, 1), np.int32)
    left = np.zeros(max(n, 1), np.int32)
    right = np.zeros(max(n, 1), np.int32)
    root = np.zeros(1, np.int32)
    rc = load().rs_kdtree_build(kp.ctypes.data_as(C.c_void_p), n, node_kp.ctypes.data_as(C.c_void_p),
                                left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p),
                                root.ctypes.data_as(C.c_void_p))
    if rc:
        raise RsError(f"rs_kdtree_build -> {rc}")
    return node_kp[:n], left[:n], right[:n], int(root[0])


def build_local_window(n_kf, new_frame, window, fix_oldest, frame_ptr, frame_pt, pt_ptr, pt_obs):
    a = [np.ascontiguousarray(x, np.int32) for x in (frame_ptr, frame_pt, pt_ptr, pt_obs)]
    of = np.zeros(n_kf + 1, np.int32)
    oo = np.zeros(n_kf + 1, np.uint8)
    cnt = np.zeros(1, np.int32)
    rc = load().rs_build_local_window(n_kf, new_frame, window, int(fix_oldest),
                                      *[x.ctypes.data_as(C.c_void_p) for x in a],
                                      of.ctypes.data_as(C.c_void_p), oo.ctypes.data_as(C.c_void_p),
                                      cnt.ctypes.data_as(C.c_void_p))
    if rc:
        raise RsError(f"rs_build_local_window -> {rc}")
    n = int(cnt[0])
    return of[:n].copy(), oo[:n].copy()


# ---- GPU context -------------------------------------------------------------
class Context:
    def __init__(self, device=0):
        import torch
        self.lib = load()
        self.torch = torch
        if not torch.cuda.is_available():
            raise RsError("no GPU visible: librsgpu has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        rc = self.lib.rs_context_create(int(device), C.byref(self.h))
        if rc:
            raise RsError(f"rs_context_create -> {rc}")
        self.use_stream(torch.cuda.current_stream(self.device))

    def use_stream(self, stream):
        self._check(self.lib.rs_context_set_stream(self.h, C.c_void_p(stream.cuda_stream if stream is not None else None)), "set_stream")

    def wait_for(self, *others):
        """Everything enqueued on this context from now on waits for what is enqueued so far on `others` (no host wait)."""
        arr = self.__dict__.setdefault("_wait_arrays", {}).get(others)
        if arr is None:
            arr = self._wait_arrays[others] = (C.c_void_p * len(others))(*[o.h.value for o in others])
        self._check(self.lib.rs_context_wait_for(self.h, arr, len(others)), "rs_context_wait_for")

    def fork(self, *others):
        """Everything enqueued on `others` from now on waits for what is enqueued so far on this context (one event)."""
        arr = self.__dict__.setdefault("_wait_arrays", {}).get(others)
        if arr is None:
            arr = self._wait_arrays[others] = (C.c_void_p * len(others))(*[o.h.value for o in others])
        self._check(self.lib.rs_context_fork(self.h, arr, len(others)), "rs_context_fork")

    def set_int(self, name, value):
        self._check(self.lib.rs_context_set_int(self.h, name.encode(), int(value)), "rs_context_set_int")

    def close(self):
        if self.h:
            self.lib.rs_context_destroy(self.h)
            self.h = C.c_void_p()

    def _check(self, rc, what):
        if rc:
            msg = self.lib.rs_last_error(self.h)
            raise RsError(f"{what} -> status {rc}: {msg.decode() if msg else ''}")

    def dev(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype)))
        return t.to(self.device)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    # -- a4
    def hamming_knn2(self, d_query, d_train, nq, nt, batch=1):
        t = self.torch
        outs = [self.empty((batch, max(nq, 1)), t.int32) for _ in range(4)]
        self._check(self.lib.rs_hamming_knn2(self.h, _dp(d_query), nq, _dp(d_train), nt, batch,
                                             *[_dp(o) for o in outs]), "rs_hamming_knn2")
        return outs

    def match_descriptors(self, d_query, d_train, nq, nt, batch=1, max_distance=64, raw=False, out=None):
        t = self.torch
        if out is None:
            out = dict(mq=self.empty((batch, max(nq, 1)), t.int32), mt=self.empty((batch, max(nq, 1)), t.int32),
                       cnt=self.empty((batch,), t.int32))
            if raw:
                out["raw"] = [self.empty((batch, max(nq, 1)), t.int32) for _ in range(4)]
        rawp = [_dp(o) for o in out["raw"]] if "raw" in out else [None] * 4
        self._check(self.lib.rs_match_descriptors(self.h, _dp(d_query), nq, _dp(d_train), nt, batch,
                                                  int(max_distance), _dp(out["mq"]), _dp(out["mt"]),
                                                  _dp(out["cnt"]), *rawp), "rs_match_descriptors")
        return out

    # -- a2/a3
    def make_frame_view(self, frame, pack=False):
        """frame: dict of numpy arrays (synth.make_match_scene); returns (FrameView, keepalive)."""
        fv = FrameView()
        keep = {}
        fv.pose[:] = list(np.asarray(frame["pose"], np.float32).reshape(16))
        fv.fx, fv.fy, fv.cx, fv.cy = [float(v) for v in frame["K"]]
        fv.width, fv.height = int(frame["width"]), int(frame["height"])
        fv.n_keypoints = len(frame["keypoints"])
        for name, key, dt in (("d_keypoints", "keypoints", np.float32), ("d_descriptors", "descriptors", np.uint8),
                              ("d_kp_matched", "kp_matched", np.uint8), ("d_kd_node_kp", "kd_node_kp", np.int32),
                              ("d_kd_left", "kd_left", np.int32), ("d_kd_right", "kd_right", np.int32)):
            keep[name] = self.dev(frame[key], dt)
            setattr(fv, name, keep[name].data_ptr())
        fv.kd_root = int(frame["kd_root"])
        fv.d_kd_packed = None
        if pack and fv.n_keypoints > 0:        # once per frame: shared by the match calls of that frame
            keep["d_kd_packed"] = self.empty((5 * fv.n_keypoints,), self.torch.int32)
            self._check(self.lib.rs_kdtree_pack(self.h, C.byref(fv), _dp(keep["d_kd_packed"])), "rs_kdtree_pack")
            fv.d_kd_packed = keep["d_kd_packed"].data_ptr()
        return fv, keep

    def make_map_view(self, mp):
        mv = MapView()
        keep = {}
        mv.n_points = len(mp["positions"])
        for name, key, dt in (("d_positions", "positions", np.float32), ("d_eligible", "eligible", np.uint8),
                              ("d_obs_ptr", "obs_ptr", np.int32), ("d_obs_kf", "obs_kf", np.int32),
                              ("d_obs_desc", "obs_desc", np.int32), ("d_kf_centers", "kf_centers", np.float32),
                              ("d_desc_pool", "desc_pool", np.uint8)):
            keep[name] = self.dev(mp[key], dt)
            setattr(mv, name, keep[name].data_ptr())
        return mv, keep

    def reproj_match_sharded(self, fv, mv, point_base, replace=0, max_distance=64):
        """rs_reproj_match_sharded: mv is this rank's shard, point_base its first point's map order."""
        t = self.torch
        N, P = fv.n_keypoints, mv.n_points
        out = dict(point_kp=self.empty((max(P, 1),), t.int32), point_dist=self.empty((max(P, 1),), t.int32),
                   prop_point=self.empty((max(N, 1),), t.int32), prop_dist=self.empty((max(N, 1),), t.int32),
                   match_kp=self.empty((max(N, 1),), t.int32), match_point=self.empty((max(N, 1),), t.int32),
                   count=self.empty((1,), t.int32))
        self._check(self.lib.rs_reproj_match_sharded(self.h, C.byref(fv), C.byref(mv), int(point_base), int(replace), int(max_distance),
                                                     _dp(out["point_kp"]), _dp(out["point_dist"]), _dp(out["prop_point"]),
                                                     _dp(out["prop_dist"]), _dp(out["match_kp"]), _dp(out["match_point"]),
                                                     _dp(out["count"])), "rs_reproj_match_sharded")
        return out

    def reproj_match(self, fv, mv, replace=0, max_distance=64, out=None):
        t = self.torch
        N, P = fv.n_keypoints, mv.n_points
        if out is None:
            out = dict(point_kp=self.empty((max(P, 1),), t.int32), point_dist=self.empty((max(P, 1),), t.int32),
                       prop_point=self.empty((max(N, 1),), t.int32), prop_dist=self.empty((max(N, 1),), t.int32),
                       match_kp=self.empty((max(N, 1),), t.int32), match_point=self.empty((max(N, 1),), t.int32),
                       count=self.empty((1,), t.int32))
        self._check(self.lib.rs_reproj_match(self.h, C.byref(fv), C.byref(mv), int(replace), int(max_distance),
                                             _dp(out["point_kp"]), _dp(out["point_dist"]), _dp(out["prop_point"]),
                                             _dp(out["prop_dist"]), _dp(out["match_kp"]), _dp(out["match_point"]),
                                             _dp(out["count"])), "rs_reproj_match")
        return out

    # -- a5-a7
    def triangulate(self, d_uv1, d_uv2, n, d_poses, n_poses, K, d_idx1=None, d_idx2=None,
                    min_parallax_cosine=0.9999, max_reproj=2.0, out=None):
        t = self.torch
        if out is None:
            out = dict(xyz=self.empty((max(n, 1), 3), t.float32), keep=self.empty((max(n, 1),), t.uint8),
                       out_index=self.empty((max(n, 1),), t.int32), out_xyz=self.empty((max(n, 1), 3), t.float32),
                       count=self.empty((1,), t.int32))
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        self._check(self.lib.rs_triangulate(self.h, _dp(d_uv1), _dp(d_uv2), int(n), _dp(d_poses), int(n_poses),
                                            _dp(d_idx1), _dp(d_idx2), Kc, C.c_float(min_parallax_cosine),
                                            C.c_float(max_reproj), _dp(out["xyz"]), _dp(out["keep"]),
                                            _dp(out["out_index"]), _dp(out["out_xyz"]), _dp(out["count"])),
                    "rs_triangulate")
        return out

    def triangulate_host(self, uv1, uv2, pose1, pose2, K, min_parallax_cosine=0.9999, max_reproj=2.0):
        """rs_triangulate_host: numpy in, (match_index [m], xyz [m][3]) out; one launch for n <= 256."""
        a = np.ascontiguousarray(uv1, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(uv2, np.float32).reshape(-1, 2)
        n = len(a)
        p1 = np.ascontiguousarray(pose1, np.float32).reshape(16)
        p2 = np.ascontiguousarray(pose2, np.float32).reshape(16)
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        idx, xyz = np.zeros(max(n, 1), np.int32), np.zeros((max(n, 1), 3), np.float32)
        cnt = C.c_int(0)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)      # noqa: E731
        self._check(self.lib.rs_triangulate_host(self.h, vp(a), vp(b), n, vp(p1), vp(p2), Kc, C.c_float(min_parallax_cosine),
                                                 C.c_float(max_reproj), vp(idx), vp(xyz), C.byref(cnt)), "rs_triangulate_host")
        return idx[:cnt.value].copy(), xyz[:cnt.value].copy()

    def triangulate_matches(self, d_kp1, d_kp2, d_mt, d_mq, d_cnt, max_matches, d_poses, K,
                            min_parallax_cosine=0.9999, max_reproj=2.0, out=None):
        t = self.torch
        n = max_matches
        if out is None:
            out = dict(xyz=self.empty((max(n, 1), 3), t.float32), keep=self.empty((max(n, 1),), t.uint8),
                       out_index=self.empty((max(n, 1),), t.int32), out_xyz=self.empty((max(n, 1), 3), t.float32),
                       count=self.empty((1,), t.int32))
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        self._check(self.lib.rs_triangulate_matches(self.h, _dp(d_kp1), _dp(d_kp2), _dp(d_mt), _dp(d_mq), _dp(d_cnt),
                                                    int(n), _dp(d_poses), Kc, C.c_float(min_parallax_cosine),
                                                    C.c_float(max_reproj), _dp(out["xyz"]), _dp(out["keep"]),
                                                    _dp(out["out_index"]), _dp(out["out_xyz"]), _dp(out["count"])),
                    "rs_triangulate_matches")
        return out

    def triangulate_matches_batch(self, d_kp1, d_kp2, d_mt, d_mq, d_cnt, d_poses, K,
                                  min_parallax_cosine=0.9999, max_reproj=2.0, out=None):
        """cfg 4: d_kp1 [B][n1][2], d_kp2 [B][n2][2], d_mt / d_mq [B][stride], d_cnt [B], d_poses [B][2][16]."""
        t = self.torch
        B, n1, n2, stride = int(d_kp1.shape[0]), int(d_kp1.shape[1]), int(d_kp2.shape[1]), int(d_mt.shape[1])
        if out is None:
            out = dict(xyz=self.empty((B, stride, 3), t.float32), keep=self.empty((B, stride), t.uint8),
                       out_index=self.empty((B, stride), t.int32), out_xyz=self.empty((B, stride, 3), t.float32),
                       count=self.empty((B,), t.int32))
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        self._check(self.lib.rs_triangulate_matches_batch(
            self.h, B, _dp(d_kp1), n1, _dp(d_kp2), n2, _dp(d_mt), _dp(d_mq), _dp(d_cnt), stride, _dp(d_poses), Kc,
            C.c_float(min_parallax_cosine), C.c_float(max_reproj), _dp(out["xyz"]), _dp(out["keep"]),
            _dp(out["out_index"]), _dp(out["out_xyz"]), _dp(out["count"])), "rs_triangulate_matches_batch")
        return out

    # -- §8(f) rank 1: Mapper::triangulate_tracks body
    def triangulate_tracks(self, d_track_uv, d_sight_ptr, d_sight_pose, d_sight_uv, d_poses, kf_pose, K, d_skip=None,
                           any_parallax_cosine=1.0, max_reproj=4.0, min_parallax_cosine=0.999848,
                           rotation_parallax_factor=0.20, min_new_points=100, out=None, d_required=None):
        t = self.torch
        n = int(d_track_uv.shape[0])
        m = max(n, 1)
        if out is None:
            out = dict(status=self.empty((m,), t.uint8), xyz=self.empty((m, 3), t.float32),
                       parallax_cos=self.empty((m,), t.float32), required_cos=self.empty((m,), t.float32),
                       accepted=self.empty((m,), t.int32), inconsistent=self.empty((m,), t.int32),
                       counts=self.empty((3,), t.int32))
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        self._check(self.lib.rs_triangulate_tracks(
            self.h, n, _dp(d_track_uv), None if d_skip is None else _dp(d_skip), _dp(d_sight_ptr), _dp(d_sight_pose),
            _dp(d_sight_uv), _dp(d_poses), int(d_poses.shape[0]), int(kf_pose), Kc, C.c_float(any_parallax_cosine),
            C.c_float(max_reproj), C.c_float(min_parallax_cosine), C.c_float(rotation_parallax_factor),
            int(min_new_points), _dp(out["status"]), _dp(out["xyz"]), _dp(out["parallax_cos"]),
            _dp(out["required_cos"]), _dp(out["accepted"]), _dp(out["inconsistent"]), _dp(out["counts"]),
            None if d_required is None else _dp(d_required)),
            "rs_triangulate_tracks")
        return out

    # -- §8(f) rank 3: Mapper::cull_points / Slam::reprojection_error arithmetic
    def point_errors(self, d_positions, d_obs_ptr, d_obs_pose, d_obs_uv, d_poses, K, max_mean_error=3.0, out=None):
        t = self.torch
        n = int(d_positions.shape[0])
        m = max(n, 1)
        if out is None:
            out = dict(mean_err=self.empty((m,), t.float32), cull=self.empty((m,), t.uint8),
                       cull_idx=self.empty((m,), t.int32), cull_count=self.empty((1,), t.int32),
                       sums=self.empty((2,), t.float64))
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        self._check(self.lib.rs_point_errors(self.h, n, _dp(d_positions), _dp(d_obs_ptr), _dp(d_obs_pose), _dp(d_obs_uv),
                                             _dp(d_poses), int(d_poses.shape[0]), Kc, C.c_float(max_mean_error),
                                             _dp(out["mean_err"]), _dp(out["cull"]), _dp(out["cull_idx"]),
                                             _dp(out["cull_count"]), _dp(out["sums"])), "rs_point_errors")
        return out

    # -- a9-a13
    def bundle_adjust(self, d_cams, cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv, K, options=None):
        cam_free = np.ascontiguousarray(cam_free, np.uint8)
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        s = BaSummary()
        self._check(self.lib.rs_bundle_adjust(self.h, int(d_cams.shape[0]), int(d_points.shape[0]),
                                              int(d_obs_cam.shape[0]), _dp(d_cams),
                                              cam_free.ctypes.data_as(C.c_void_p), _dp(d_points), _dp(d_obs_ptr),
                                              _dp(d_obs_cam), _dp(d_obs_uv), Kc,
                                              None if options is None else C.byref(options), C.byref(s)),
                    "rs_bundle_adjust")
        return s.as_dict()

    def bundle_adjust_inertial(self, d_cams, cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv, K, imu, options=None):
        """rs_bundle_adjust_inertial; imu = synth.make_imu dict.  Returns (summary, velocity [C][3], bias [C][6])."""
        cam_free = np.ascontiguousarray(cam_free, np.uint8)
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        vel = np.array(imu["cam_velocity"], np.float64, order="C")
        bias = np.array(imu["cam_bias"], np.float64, order="C")
        g = np.ascontiguousarray(imu["gravity"], np.float64)
        arr, nf = imu_factor_array(imu)
        s = BaSummary()
        self._check(self.lib.rs_bundle_adjust_inertial(
            self.h, int(d_cams.shape[0]), int(d_points.shape[0]), int(d_obs_cam.shape[0]), _dp(d_cams),
            cam_free.ctypes.data_as(C.c_void_p), _dp(d_points), _dp(d_obs_ptr), _dp(d_obs_cam), _dp(d_obs_uv), Kc,
            vel.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), arr, nf, g.ctypes.data_as(C.c_void_p),
            None if options is None else C.byref(options), C.byref(s)), "rs_bundle_adjust_inertial")
        return s.as_dict(), vel, bias

    def refine_pose_inertial(self, cam, d_points, d_uv, K, prior=None, delta=None, options=None):
        """rs_refine_pose_inertial; prior = (predicted 3x3, sigma) or delta = dict(imu=<one-factor dict>, prev_pose,
        prev_velocity, prev_bias, velocity).  Returns cam, velocity, summary."""
        cam = np.array(cam, np.float64, order="C")
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        s = BaSummary()
        vel = np.zeros(3)
        vp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)      # noqa: E731
        kind, pred, sigma, keep, args = 0, None, 0.0, [], [None, None, None, None, None]
        if prior is not None:
            kind, sigma = 1, float(prior[1])
            keep.append(np.ascontiguousarray(prior[0], np.float64))
            pred = keep[-1].ctypes.data_as(C.c_void_p)
        if delta is not None:
            kind = 2
            farr, _ = imu_factor_array(delta["imu"])
            keep += [np.ascontiguousarray(delta[k], np.float64) for k in ("prev_pose", "prev_velocity", "prev_bias")]
            keep.append(np.ascontiguousarray(delta["imu"]["gravity"], np.float64))
            vel = np.array(delta["velocity"], np.float64)
            args = [keep[-4].ctypes.data_as(C.c_void_p), keep[-3].ctypes.data_as(C.c_void_p), keep[-2].ctypes.data_as(C.c_void_p),
                    farr, keep[-1].ctypes.data_as(C.c_void_p)]
        del vp
        self._check(self.lib.rs_refine_pose_inertial(
            self.h, cam.ctypes.data_as(C.c_void_p), _dp(d_points), _dp(d_uv), int(d_points.shape[0]), Kc, kind, pred,
            C.c_double(sigma), args[0], args[1], args[2], args[3], args[4], vel.ctypes.data_as(C.c_void_p),
            None if options is None else C.byref(options), C.byref(s)), "rs_refine_pose_inertial")
        return cam, vel, s.as_dict()

    def bundle_adjust_batch(self, problems, options=None):
        """rs_bundle_adjust_batch; problems = list of (d_cams, cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv, K).
        Returns the list of summaries."""
        n = len(problems)
        arr = (BaProblem * max(n, 1))()
        keep = []
        for i, (dc, free, dp, optr, ocam, ouv, K) in enumerate(problems):
            f = np.ascontiguousarray(free, np.uint8)
            keep.append(f)
            a = arr[i]
            a.n_cameras, a.n_points, a.n_obs = int(dc.shape[0]), int(dp.shape[0]), int(ocam.shape[0])
            a.d_cameras, a.h_cam_free, a.d_points = dc.data_ptr(), f.ctypes.data, dp.data_ptr()
            a.d_obs_ptr, a.d_obs_cam, a.d_obs_uv = optr.data_ptr(), ocam.data_ptr(), ouv.data_ptr()
            a.intrinsics[:] = [float(v) for v in K]
        out = (BaSummary * max(n, 1))()
        self._check(self.lib.rs_bundle_adjust_batch(self.h, n, arr, None if options is None else C.byref(options), out),
                    "rs_bundle_adjust_batch")
        return [out[i].as_dict() for i in range(n)]

    def ba_trace(self):
        """Per-iteration record of the last bundle_adjust on this context (list of dicts)."""
        cap = 1024
        buf = (BaIteration * cap)()
        n = C.c_int(0)
        self._check(self.lib.rs_ba_get_trace(self.h, buf, cap, C.byref(n)), "rs_ba_get_trace")
        return [buf[i].as_dict() for i in range(min(n.value, cap))]

    def ba_cameras(self, out):
        """Cameras after the last bundle_adjust from the pinned mirror (no device read-back); out [C][6] f64."""
        assert out.dtype == np.float64 and out.flags.c_contiguous
        self._check(self.lib.rs_ba_get_cameras(self.h, _hp(out), int(out.shape[0])), "rs_ba_get_cameras")
        return out

    def ba_stats(self):
        buf = (C.c_int * 8)()
        self._check(self.lib.rs_ba_get_stats(self.h, buf), "rs_ba_get_stats")
        return dict(rounds=buf[0], fresh_rounds=buf[1], set_evaluations=buf[2], rounds_enqueued=buf[3], handoff_retries=buf[4])

    def reanchor_points(self, d_point_idx, d_frame_idx, d_before, d_after, d_positions):
        """Mapper::bundle_adjust's tail (src/Mapper.cpp:380-393); d_positions [P][3] f32 is updated in place."""
        n = int(d_frame_idx.shape[0])
        self._check(self.lib.rs_reanchor_points(self.h, n, None if d_point_idx is None else _dp(d_point_idx),
                                                _dp(d_frame_idx), _dp(d_before), _dp(d_after), int(d_before.shape[0]),
                                                _dp(d_positions)), "rs_reanchor_points")

    def reanchor_points_host_poses(self, d_point_idx, d_frame_idx, h_before, h_after, d_positions):
        """rs_reanchor_points with the poses in host arrays ([n_frames][16] f32, C order), as the reference holds them."""
        assert h_before.dtype == np.float32 and h_after.dtype == np.float32 and h_before.flags.c_contiguous and h_after.flags.c_contiguous
        n = int(d_frame_idx.shape[0])
        self._check(self.lib.rs_reanchor_points_host_poses(self.h, n, None if d_point_idx is None else _dp(d_point_idx),
                                                           _dp(d_frame_idx), _hp(h_before), _hp(h_after),
                                                           int(h_before.shape[0]), _dp(d_positions)), "rs_reanchor_points_host_poses")

    def transform_points(self, d_obs_ptr, d_obs_kf, d_before, d_after, d_positions):
        """transform_points of the pose graph (src/Optimization.cpp:512-536); d_positions [P][3] f32 is updated in place."""
        self._check(self.lib.rs_transform_points(self.h, int(d_positions.shape[0]), _dp(d_obs_ptr), _dp(d_obs_kf), _dp(d_before),
                                                 _dp(d_after), int(d_before.shape[0]), _dp(d_positions)), "rs_transform_points")

    def refine_pose(self, cam, d_points, d_uv, K, options=None):
        cam = np.array(cam, np.float64, order="C")
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        s = BaSummary()
        self._check(self.lib.rs_refine_pose(self.h, cam.ctypes.data_as(C.c_void_p), _dp(d_points), _dp(d_uv),
                                            int(d_points.shape[0]), Kc,
                                            None if options is None else C.byref(options), C.byref(s)),
                    "rs_refine_pose")
        return cam, s.as_dict()

    # -- multi-GPU
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = load().rs_comm_get_unique_id(buf)
        if rc:
            raise RsError(f"rs_comm_get_unique_id -> {rc}")
        return bytes(buf)

    def comm_init(self, uid, n_ranks, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self.lib.rs_comm_init_rank(self.h, buf, int(n_ranks), int(rank)), "rs_comm_init_rank")

    @staticmethod
    def comm_init_local(contexts):
        """In-process group: contexts[i] becomes rank i (see rs_comm_init_local)."""
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        rc = load().rs_comm_init_local(arr, len(contexts))
        if rc:
            raise RsError(f"rs_comm_init_local -> {rc}")

    def comm_destroy(self):
        self._check(self.lib.rs_comm_destroy(self.h), "rs_comm_destroy")

    def comm_count(self):
        """(ranks as the communicator itself reports them, kind: 0 none / 1 RCCL / 2 in-process group)"""
        n, k = C.c_int(0), C.c_int(0)
        self._check(self.lib.rs_comm_count(self.h, C.byref(n), C.byref(k)), "rs_comm_count")
        return n.value, k.value

    # -- profiling
    def prof_begin(self):
        self._check(self.lib.rs_prof_begin(self.h), "rs_prof_begin")

    def prof_end(self):
        ent = (ProfEntry * 32)()
        n = C.c_int(0)
        self._check(self.lib.rs_prof_end(self.h, ent, C.byref(n)), "rs_prof_end")
        return {ent[i].name.decode(): (ent[i].launches, ent[i].total_ms) for i in range(n.value)}

    def prof_counters(self, n=16):
        buf = (C.c_uint64 * n)()
        self._check(self.lib.rs_prof_counters(self.h, buf, n), "rs_prof_counters")
        return list(buf)

    def empty_launch_us(self, n=2000):
        us = C.c_double(0.0)
        self._check(self.lib.rs_prof_empty_launch(self.h, int(n), C.byref(us)), "rs_prof_empty_launch")
        return us.value

    def synchronize(self):
        self._check(self.lib.rs_context_synchronize(self.h), "rs_context_synchronize")


# ---- §8(f) rank 4: the resident map (rs_map / rs_frame) -----------------------------------------------------
class ResidentFrame:
    def __init__(self, ctx, keypoints, descriptors):
        self.ctx = ctx
        kp = np.ascontiguousarray(keypoints, np.float32)
        de = np.ascontiguousarray(descriptors, np.uint8)
        self.n = len(kp)
        self.h = C.c_void_p()
        ctx._check(ctx.lib.rs_frame_create(ctx.h, kp.ctypes.data_as(C.c_void_p), de.ctypes.data_as(C.c_void_p), self.n, C.byref(self.h)), "rs_frame_create")

    def close(self):
        if self.h:
            self.ctx.lib.rs_frame_destroy(self.h)
            self.h = C.c_void_p()


class ResidentMap:
    """Thin wrapper of rs_map: every method is one C-ABI call."""

    def __init__(self, ctx):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        ctx._check(self.lib.rs_map_create(ctx.h, C.byref(self.h)), "rs_map_create")

    def close(self):
        if self.h:
            self.lib.rs_map_destroy(self.h)
            self.h = C.c_void_p()

    def _f3(self, v):
        return (C.c_float * 3)(*[float(x) for x in v])

    def add_keyframe(self, frame, pose):
        out = C.c_int(-1)
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.ctx._check(self.lib.rs_map_add_keyframe(self.h, frame.h, p.ctypes.data_as(C.c_void_p), C.byref(out)), "rs_map_add_keyframe")
        return out.value

    def set_keyframe_pose(self, kf, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.ctx._check(self.lib.rs_map_set_keyframe_pose(self.h, int(kf), p.ctypes.data_as(C.c_void_p)), "rs_map_set_keyframe_pose")

    def add_point(self, xyz):
        out = C.c_int(-1)
        self.ctx._check(self.lib.rs_map_add_point(self.h, self._f3(xyz), C.byref(out)), "rs_map_add_point")
        return out.value

    def set_position(self, point, xyz):
        self.ctx._check(self.lib.rs_map_set_position(self.h, int(point), self._f3(xyz)), "rs_map_set_position")

    def remove_point(self, point):
        self.ctx._check(self.lib.rs_map_remove_point(self.h, int(point)), "rs_map_remove_point")

    def add_observation(self, point, kf, keypoint):
        self.ctx._check(self.lib.rs_map_add_observation(self.h, int(point), int(kf), int(keypoint)), "rs_map_add_observation")

    def remove_observation(self, point, kf):
        self.ctx._check(self.lib.rs_map_remove_observation(self.h, int(point), int(kf)), "rs_map_remove_observation")

    def counts(self):
        buf = (C.c_int * 4)()
        self.ctx._check(self.lib.rs_map_counts(self.h, buf), "rs_map_counts")
        return dict(slots=buf[0], alive=buf[1], observations=buf[2], key_frames=buf[3])

    def positions(self):
        n = self.counts()["slots"]
        out = np.zeros((max(n, 1), 3), np.float32)
        self.ctx._check(self.lib.rs_map_get_positions(self.h, 0, n, out.ctypes.data_as(C.c_void_p)), "rs_map_get_positions")
        return out[:n]

    def match(self, frame, pose, K, width, height, kp_matched=None, matched_points=(), required_observer=-1, only_points=None,
              replace=0, max_distance=64):
        n = frame.n
        mk, mp = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        cnt = C.c_int(0)
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        km = None if kp_matched is None else np.ascontiguousarray(kp_matched, np.uint8)
        mpts = np.ascontiguousarray(matched_points, np.int32)
        only = None if only_points is None else np.ascontiguousarray(only_points, np.int32)
        self.ctx._check(self.lib.rs_map_match(
            self.ctx.h, self.h, frame.h, p.ctypes.data_as(C.c_void_p), Kc, int(width), int(height),
            None if km is None else km.ctypes.data_as(C.c_void_p), mpts.ctypes.data_as(C.c_void_p), len(mpts), int(required_observer),
            None if only is None else only.ctypes.data_as(C.c_void_p), -1 if only is None else len(only), int(replace), int(max_distance),
            mk.ctypes.data_as(C.c_void_p), mp.ctypes.data_as(C.c_void_p), C.byref(cnt)), "rs_map_match")
        return mk[:cnt.value].copy(), mp[:cnt.value].copy()

    def pose_graph(self, loops, four_dof=False, gravity=(0.0, 0.0, 0.0), options=None):
        n = self.counts()["key_frames"]
        poses, rot = np.zeros((max(n, 1), 16), np.float32), np.zeros((max(n, 1), 9), np.float32)
        g = (C.c_double * 3)(*[float(v) for v in gravity])
        s = BaSummary()
        self.ctx._check(self.lib.rs_map_pose_graph(self.ctx.h, self.h, pose_graph_edges(loops), len(loops), int(bool(four_dof)), g,
                                                   None if options is None else C.byref(options), poses.ctypes.data_as(C.c_void_p),
                                                   rot.ctypes.data_as(C.c_void_p), C.byref(s)), "rs_map_pose_graph")
        return s.as_dict(), poses[:n].reshape(-1, 4, 4), rot[:n].reshape(-1, 3, 3)

    def bundle_adjust(self, kfs, free, K, options=None):
        kfs = np.ascontiguousarray(kfs, np.int32)
        free = np.ascontiguousarray(free, np.uint8)
        Kc = (C.c_float * 4)(*[float(v) for v in K])
        s = BaSummary()
        cap = self.counts()["slots"]
        poses = np.zeros((len(kfs), 16), np.float32)
        pts, xyz = np.zeros(max(cap, 1), np.int32), np.zeros((max(cap, 1), 3), np.float32)
        n = C.c_int(0)
        self.ctx._check(self.lib.rs_map_bundle_adjust(
            self.ctx.h, self.h, kfs.ctypes.data_as(C.c_void_p), free.ctypes.data_as(C.c_void_p), len(kfs), Kc,
            None if options is None else C.byref(options), C.byref(s), poses.ctypes.data_as(C.c_void_p),
            pts.ctypes.data_as(C.c_void_p), xyz.ctypes.data_as(C.c_void_p), cap, C.byref(n)), "rs_map_bundle_adjust")
        return s.as_dict(), poses, pts[:n.value].copy(), xyz[:n.value].copy()
