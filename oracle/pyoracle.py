"""ctypes bindings of oracle/liboracle.so — the CPU restatement of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/rs_oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package.  PARITY UNPINNED: the reference ships no fixtures for this path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def use_all_cores(on=True):
    """bench.py only: switch to / from the OpenMP build (liboracle_omp.so), for the all-cores CPU baseline."""
    global _LIB
    so = os.path.join(_HERE, "liboracle_omp.so" if on else "liboracle.so")
    if on:
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_omp.so"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(so):
        build()
    _LIB = C.CDLL(so)
    return _LIB


def cpu_model():
    """(model name, logical CPUs, a short tag that names baseline builds after the host CPU)."""
    import hashlib
    model, flags = "unknown", ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = line.split(":", 1)[1].strip()
    except OSError:
        pass
    tag = hashlib.sha1((model + flags).encode()).hexdigest()[:10]
    return model, os.cpu_count() or 1, tag


def use_baseline(kind):
    """bench.py only: switch the binding to a CPU-BASELINE build of the same sources (never used by tests):
    "fast"  = gcc -O3 -march=native, serial;  "fast_omp" = the same + OpenMP over queries / points / landmark
    blocks; None = back to the checker (liboracle.so, -O2, portable).  -march=native code is built on the
    machine that runs it: the file name carries a hash of the host CPU's model and flags."""
    global _LIB
    if kind is None:
        _LIB = None
        return lib()
    assert kind in ("fast", "fast_omp")
    so = os.path.join(_HERE, f"liboracle_{kind}_{cpu_model()[2]}.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", kind, "FAST_OUT=" + os.path.basename(so)], stdout=subprocess.DEVNULL)
    _LIB = C.CDLL(so)
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class FrameView(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("width", C.c_int), ("height", C.c_int), ("n_keypoints", C.c_int),
                ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("kp_matched", C.c_void_p),
                ("kd_node_kp", C.c_void_p), ("kd_left", C.c_void_p), ("kd_right", C.c_void_p),
                ("kd_root", C.c_int)]


class MapView(C.Structure):
    _fields_ = [("n_points", C.c_int), ("positions", C.c_void_p), ("eligible", C.c_void_p),
                ("obs_ptr", C.c_void_p), ("obs_kf", C.c_void_p), ("obs_desc", C.c_void_p),
                ("kf_centers", C.c_void_p), ("desc_pool", C.c_void_p)]


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


def default_options():
    o = BaOptions()
    lib().orc_ba_default_options(C.byref(o))
    return o


# ---------------------------------------------------------------- matching
def hamming_knn2(query, train):
    q = np.ascontiguousarray(query, np.uint8)
    t = np.ascontiguousarray(train, np.uint8)
    nq, nt = len(q), len(t)
    out = [np.full(nq, -7, np.int32) for _ in range(4)]
    rc = lib().orc_hamming_knn2(_p(q, u8p), nq, _p(t, u8p), nt, *[_p(o, i32p) for o in out])
    assert rc == 0
    return out


def match_descriptors(query, train, max_distance=64):
    q = np.ascontiguousarray(query, np.uint8)
    t = np.ascontiguousarray(train, np.uint8)
    nq, nt = len(q), len(t)
    mq = np.zeros(max(nq, 1), np.int32)
    mt = np.zeros(max(nq, 1), np.int32)
    cnt = np.zeros(1, np.int32)
    rc = lib().orc_match_descriptors(_p(q, u8p), nq, _p(t, u8p), nt, int(max_distance),
                                     _p(mq, i32p), _p(mt, i32p), _p(cnt, i32p))
    assert rc == 0
    n = int(cnt[0])
    return mq[:n].copy(), mt[:n].copy()


def kdtree_build(keypoints):
    kp = np.ascontiguousarray(keypoints, np.float32)
    n = len(kp)
    node_kp = np.zeros(max(n, 1), np.int32)
    left = np.zeros(max(n, 1), np.int32)
    right = np.zeros(max(n, 1), np.int32)
    root = np.zeros(1, np.int32)
    rc = lib().orc_kdtree_build(_p(kp, f32p), n, _p(node_kp, i32p), _p(left, i32p), _p(right, i32p),
                                _p(root, i32p))
    assert rc == 0
    return node_kp[:n], left[:n], right[:n], int(root[0])


def kdtree_radius(keypoints, tree, x, y, radius):
    kp = np.ascontiguousarray(keypoints, np.float32)
    node_kp, left, right, root = tree
    out = np.zeros(max(len(kp), 1), np.int32)
    L = lib()
    L.orc_kdtree_radius.argtypes = [f32p, i32p, i32p, i32p, C.c_int, C.c_float, C.c_float, C.c_float, i32p, C.c_int]
    n = L.orc_kdtree_radius(_p(kp, f32p), _p(node_kp, i32p), _p(left, i32p), _p(right, i32p), root,
                            float(x), float(y), float(radius), _p(out, i32p), len(out))
    return out[:n].copy()


def reproj_match(frame, mp, replace=0, max_distance=64):
    """frame/mp are dicts of numpy arrays (see synth.make_match_scene)."""
    fv = FrameView()
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data

    fv.pose[:] = list(np.asarray(frame["pose"], np.float32).reshape(16))
    fv.fx, fv.fy, fv.cx, fv.cy = [float(v) for v in frame["K"]]
    fv.width, fv.height = int(frame["width"]), int(frame["height"])
    N = len(frame["keypoints"])
    fv.n_keypoints = N
    fv.keypoints = ptr(frame["keypoints"], np.float32)
    fv.descriptors = ptr(frame["descriptors"], np.uint8)
    fv.kp_matched = ptr(frame["kp_matched"], np.uint8)
    fv.kd_node_kp = ptr(frame["kd_node_kp"], np.int32)
    fv.kd_left = ptr(frame["kd_left"], np.int32)
    fv.kd_right = ptr(frame["kd_right"], np.int32)
    fv.kd_root = int(frame["kd_root"])
    mv = MapView()
    P = len(mp["positions"])
    mv.n_points = P
    mv.positions = ptr(mp["positions"], np.float32)
    mv.eligible = ptr(mp["eligible"], np.uint8)
    mv.obs_ptr = ptr(mp["obs_ptr"], np.int32)
    mv.obs_kf = ptr(mp["obs_kf"], np.int32)
    mv.obs_desc = ptr(mp["obs_desc"], np.int32)
    mv.kf_centers = ptr(mp["kf_centers"], np.float32)
    mv.desc_pool = ptr(mp["desc_pool"], np.uint8)
    point_kp = np.zeros(max(P, 1), np.int32)
    point_dist = np.zeros(max(P, 1), np.int32)
    prop_point = np.zeros(max(N, 1), np.int32)
    prop_dist = np.zeros(max(N, 1), np.int32)
    mkp = np.zeros(max(N, 1), np.int32)
    mpt = np.zeros(max(N, 1), np.int32)
    cnt = np.zeros(1, np.int32)
    rc = lib().orc_reproj_match(C.byref(fv), C.byref(mv), int(replace), int(max_distance),
                                _p(point_kp, i32p), _p(point_dist, i32p), _p(prop_point, i32p),
                                _p(prop_dist, i32p), _p(mkp, i32p), _p(mpt, i32p), _p(cnt, i32p))
    assert rc == 0
    n = int(cnt[0])
    return dict(point_kp=point_kp[:P], point_dist=point_dist[:P], prop_point=prop_point[:N],
                prop_dist=prop_dist[:N], match_kp=mkp[:n].copy(), match_point=mpt[:n].copy())


# ----------------------------------------------------------- triangulation
def triangulate(uv1, uv2, poses, K, idx1=None, idx2=None, min_parallax_cosine=0.9999, max_reproj=2.0):
    uv1 = np.ascontiguousarray(uv1, np.float32)
    uv2 = np.ascontiguousarray(uv2, np.float32)
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    Kc = (C.c_float * 4)(*[float(v) for v in K])
    n = len(uv1)
    xyz = np.zeros((max(n, 1), 3), np.float32)
    keep = np.zeros(max(n, 1), np.uint8)
    oi = np.zeros(max(n, 1), np.int32)
    ox = np.zeros((max(n, 1), 3), np.float32)
    cnt = np.zeros(1, np.int32)
    i1 = None if idx1 is None else np.ascontiguousarray(idx1, np.int32)
    i2 = None if idx2 is None else np.ascontiguousarray(idx2, np.int32)
    L = lib()
    L.orc_triangulate.argtypes = [f32p, f32p, C.c_int, f32p, C.c_int, i32p, i32p, C.c_float * 4,
                                  C.c_float, C.c_float, f32p, u8p, i32p, f32p, i32p]
    rc = L.orc_triangulate(_p(uv1, f32p), _p(uv2, f32p), n, _p(poses, f32p), len(poses),
                           _p(i1, i32p), _p(i2, i32p), Kc, float(min_parallax_cosine), float(max_reproj),
                           _p(xyz, f32p), _p(keep, u8p), _p(oi, i32p), _p(ox, f32p), _p(cnt, i32p))
    assert rc == 0
    m = int(cnt[0])
    return dict(xyz=xyz[:n], keep=keep[:n], out_index=oi[:m].copy(), out_xyz=ox[:m].copy())


def triangulate_tracks(track_uv, sight_ptr, sight_pose, sight_uv, poses, kf_pose, K, skip=None,
                       any_parallax_cosine=1.0, max_reproj=4.0, min_parallax_cosine=0.999848,
                       rotation_parallax_factor=0.20, min_new_points=100):
    track_uv = np.ascontiguousarray(track_uv, np.float32).reshape(-1, 2)
    sight_ptr = np.ascontiguousarray(sight_ptr, np.int32)
    sight_pose = np.ascontiguousarray(sight_pose, np.int32)
    sight_uv = np.ascontiguousarray(sight_uv, np.float32).reshape(-1, 2)
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    n = len(track_uv)
    m = max(n, 1)
    Kc = (C.c_float * 4)(*[float(v) for v in K])
    status = np.zeros(m, np.uint8)
    xyz = np.zeros((m, 3), np.float32)
    pc = np.zeros(m, np.float32)
    rc_ = np.zeros(m, np.float32)
    acc = np.zeros(m, np.int32)
    inc = np.zeros(m, np.int32)
    cnt = np.zeros(3, np.int32)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    L = lib()
    L.orc_triangulate_tracks.argtypes = [C.c_int, f32p, u8p, i32p, i32p, f32p, f32p, C.c_int, C.c_int, C.c_float * 4,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, u8p, f32p, f32p, f32p,
                                         i32p, i32p, i32p, i32p, i32p]
    rc = L.orc_triangulate_tracks(n, _p(track_uv, f32p), _p(sk, u8p), _p(sight_ptr, i32p), _p(sight_pose, i32p),
                                  _p(sight_uv, f32p), _p(poses, f32p), len(poses), int(kf_pose), Kc,
                                  float(any_parallax_cosine), float(max_reproj), float(min_parallax_cosine),
                                  float(rotation_parallax_factor), int(min_new_points), _p(status, u8p), _p(xyz, f32p),
                                  _p(pc, f32p), _p(rc_, f32p), _p(acc, i32p),
                                  cnt[0:1].ctypes.data_as(i32p), cnt[1:2].ctypes.data_as(i32p), _p(inc, i32p),
                                  cnt[2:3].ctypes.data_as(i32p))
    assert rc == 0
    return dict(status=status[:n], xyz=xyz[:n], parallax_cos=pc[:n], required_cos=rc_[:n],
                accepted=acc[:int(cnt[0])].copy(), n_topped_up=int(cnt[1]), inconsistent=inc[:int(cnt[2])].copy())


def point_errors(positions, obs_ptr, obs_pose, obs_uv, poses, K, max_mean_error=3.0):
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    obs_ptr = np.ascontiguousarray(obs_ptr, np.int32)
    obs_pose = np.ascontiguousarray(obs_pose, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 2)
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    n = len(positions)
    m = max(n, 1)
    Kc = (C.c_float * 4)(*[float(v) for v in K])
    mean = np.zeros(m, np.float32)
    cull = np.zeros(m, np.uint8)
    idx = np.zeros(m, np.int32)
    cnt = np.zeros(1, np.int32)
    sums = np.zeros(2, np.float64)
    L = lib()
    L.orc_point_errors.argtypes = [C.c_int, f32p, i32p, i32p, f32p, f32p, C.c_int, C.c_float * 4, C.c_float, f32p, u8p,
                                   i32p, i32p, C.POINTER(C.c_double)]
    rc = L.orc_point_errors(n, _p(positions, f32p), _p(obs_ptr, i32p), _p(obs_pose, i32p), _p(obs_uv, f32p),
                            _p(poses, f32p), len(poses), Kc, float(max_mean_error), _p(mean, f32p), _p(cull, u8p),
                            _p(idx, i32p), _p(cnt, i32p), sums.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return dict(mean_err=mean[:n], cull=cull[:n], cull_idx=idx[:int(cnt[0])].copy(), err_sum=float(sums[0]),
                n_obs=int(sums[1]))


def reanchor_points(point_idx, frame_idx, before, after, positions):
    pos = np.array(positions, np.float32, order="C")
    fi = np.ascontiguousarray(frame_idx, np.int32)
    pi = None if point_idx is None else np.ascontiguousarray(point_idx, np.int32)
    rc = lib().orc_reanchor_points(len(fi), _p(pi, i32p), _p(fi, i32p), _p(np.ascontiguousarray(before, np.float32), f32p),
                                   _p(np.ascontiguousarray(after, np.float32), f32p), _p(pos, f32p))
    assert rc == 0
    return pos


def null_vector4(A):
    A = np.ascontiguousarray(A, np.float64).reshape(16)
    v = np.zeros(4)
    s = np.zeros(4)
    lib().orc_null_vector4(_p(A, f64p), _p(v, f64p), _p(s, f64p))
    return v, s


def pack_pose(pose):
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    cam = np.zeros(6)
    lib().orc_pack_pose(_p(p, f32p), _p(cam, f64p))
    return cam


def unpack_pose(cam):
    c = np.ascontiguousarray(cam, np.float64)
    p = np.zeros(16, np.float32)
    lib().orc_unpack_pose(_p(c, f64p), _p(p, f32p))
    return p.reshape(4, 4)


# --------------------------------------------------------------------- BA
def reprojection(cam, pt, uv, K):
    cam = np.ascontiguousarray(cam, np.float64)
    pt = np.ascontiguousarray(pt, np.float64)
    uv = np.ascontiguousarray(uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    r = np.zeros(2)
    jc = np.zeros(12)
    jp = np.zeros(6)
    lib().orc_reprojection(_p(cam, f64p), _p(pt, f64p), _p(uv, f32p), _p(Kc, f32p), _p(r, f64p),
                           _p(jc, f64p), _p(jp, f64p))
    return r, jc.reshape(2, 6), jp.reshape(2, 3)


class BaIteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("candidate_cost", C.c_double), ("model_cost_change", C.c_double),
                ("radius", C.c_double), ("step_norm", C.c_double), ("x_norm", C.c_double),
                ("outcome", C.c_int), ("pad", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


def bundle_adjust_trace(*args, **kw):
    """bundle_adjust + the per-iteration record of the trust-region loop (list of dicts)."""
    cap = 1024
    buf = (BaIteration * cap)()
    cnt = C.c_int(0)
    L = lib()
    L.orc_ba_set_trace(buf, cap, C.byref(cnt))
    try:
        out = bundle_adjust(*args, **kw)
    finally:
        L.orc_ba_set_trace(None, 0, None)
    return out + ([buf[i].as_dict() for i in range(cnt.value)],)


def bundle_adjust(cams, cam_free, points, obs_ptr, obs_cam, obs_uv, K, options=None):
    cams = np.array(cams, np.float64, order="C")
    points = np.array(points, np.float64, order="C")
    cam_free = np.ascontiguousarray(cam_free, np.uint8)
    obs_ptr = np.ascontiguousarray(obs_ptr, np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    s = BaSummary()
    rc = lib().orc_bundle_adjust(len(cams), len(points), len(obs_cam), _p(cams, f64p), _p(cam_free, u8p),
                                 _p(points, f64p), _p(obs_ptr, i32p), _p(obs_cam, i32p), _p(obs_uv, f32p),
                                 _p(Kc, f32p), None if options is None else C.byref(options), C.byref(s))
    assert rc == 0
    return cams, points, s.as_dict()


class ImuFactor(C.Structure):
    """orc_imu_factor / rs_imu_factor (identical layout)."""
    _fields_ = [("cam_i", C.c_int), ("cam_j", C.c_int), ("duration", C.c_double), ("rotation", C.c_double * 9),
                ("velocity", C.c_double * 3), ("position", C.c_double * 3), ("covariance", C.c_double * 81),
                ("bias_gyro", C.c_double * 3), ("bias_accel", C.c_double * 3), ("bias_jacobian", C.c_double * 54),
                ("gyro_bias_sigma", C.c_double), ("accel_bias_sigma", C.c_double)]


def imu_factor_array(imu, cls=ImuFactor):
    """synth.make_imu(...) dict -> ctypes array of factor structs."""
    n = len(imu["cam_i"])
    arr = (cls * max(n, 1))()
    for f in range(n):
        a = arr[f]
        a.cam_i, a.cam_j, a.duration = int(imu["cam_i"][f]), int(imu["cam_j"][f]), float(imu["duration"][f])
        for name in ("rotation", "velocity", "position", "covariance", "bias_gyro", "bias_accel", "bias_jacobian"):
            v = np.asarray(imu[name][f], np.float64).ravel()
            getattr(a, name)[:] = list(v)
        a.gyro_bias_sigma, a.accel_bias_sigma = float(imu["gyro_bias_sigma"]), float(imu["accel_bias_sigma"])
    return arr, n


def imu_preintegration(imu, f, pose_i, vel_i, bias_i, pose_j, vel_j):
    arr, _ = imu_factor_array(imu)
    r = np.zeros(9); J = np.zeros((9, 24))
    g = np.ascontiguousarray(imu["gravity"], np.float64)
    args = [np.ascontiguousarray(x, np.float64) for x in (pose_i, vel_i, bias_i, pose_j, vel_j)]
    lib().orc_imu_preintegration(C.byref(arr[f]), _p(g, f64p), *[_p(x, f64p) for x in args], _p(r, f64p), _p(J, f64p))
    return r, J


def imu_bias_walk(imu, f, bias_i, bias_j):
    arr, _ = imu_factor_array(imu)
    r = np.zeros(6); J = np.zeros((6, 12))
    lib().orc_imu_bias_walk(C.byref(arr[f]), _p(np.ascontiguousarray(bias_i, np.float64), f64p),
                            _p(np.ascontiguousarray(bias_j, np.float64), f64p), _p(r, f64p), _p(J, f64p))
    return r, J


def rotation_prior(predicted, sigma, pose):
    r = np.zeros(3); J = np.zeros((3, 6))
    lib().orc_rotation_prior(_p(np.ascontiguousarray(predicted, np.float64), f64p), C.c_double(sigma),
                             _p(np.ascontiguousarray(pose, np.float64), f64p), _p(r, f64p), _p(J, f64p))
    return r, J


def imu_whitener(cov):
    W = np.zeros((9, 9))
    lib().orc_imu_whitener(_p(np.ascontiguousarray(cov, np.float64), f64p), _p(W, f64p))
    return W


def bundle_adjust_inertial(cams, cam_free, points, obs_ptr, obs_cam, obs_uv, K, imu, options=None, trace=False):
    """Returns cams, points, velocity, bias, summary (+ trace list when asked)."""
    cams = np.array(cams, np.float64, order="C")
    points = np.array(points, np.float64, order="C")
    vel = np.array(imu["cam_velocity"], np.float64, order="C")
    bias = np.array(imu["cam_bias"], np.float64, order="C")
    cam_free = np.ascontiguousarray(cam_free, np.uint8)
    obs_ptr = np.ascontiguousarray(obs_ptr, np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    g = np.ascontiguousarray(imu["gravity"], np.float64)
    arr, nf = imu_factor_array(imu)
    s = BaSummary()
    L = lib()
    cap = 1024
    buf = (BaIteration * cap)()
    cnt = C.c_int(0)
    if trace:
        L.orc_ba_set_trace(buf, cap, C.byref(cnt))
    try:
        rc = L.orc_bundle_adjust_inertial(len(cams), len(points), len(obs_cam), _p(cams, f64p), _p(cam_free, u8p), _p(points, f64p),
                                          _p(obs_ptr, i32p), _p(obs_cam, i32p), _p(obs_uv, f32p), _p(Kc, f32p), _p(vel, f64p),
                                          _p(bias, f64p), arr, nf, _p(g, f64p), None if options is None else C.byref(options),
                                          C.byref(s))
    finally:
        if trace:
            L.orc_ba_set_trace(None, 0, None)
    assert rc == 0
    out = (cams, points, vel, bias, s.as_dict())
    return out + ([buf[i].as_dict() for i in range(cnt.value)],) if trace else out


def refine_pose_inertial(cam, points, uv, K, prior=None, delta=None, options=None):
    """prior = (predicted 3x3, sigma)  or  delta = dict(imu=<one-factor make_imu dict>, prev_pose, prev_velocity, prev_bias,
    velocity).  Returns cam, velocity, summary."""
    cam = np.array(cam, np.float64, order="C")
    points = np.ascontiguousarray(points, np.float64)
    uv = np.ascontiguousarray(uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    s = BaSummary()
    z3, z6, z9 = np.zeros(3), np.zeros(6), np.zeros(9)
    vel = np.zeros(3)
    kind, pred, sigma, pp, pv, pb, g, farr = 0, z9, 0.0, z6, z3, z6, z3, None
    if prior is not None:
        kind, pred, sigma = 1, np.ascontiguousarray(prior[0], np.float64), float(prior[1])
    if delta is not None:
        kind = 2
        farr, _ = imu_factor_array(delta["imu"])
        pp, pv, pb = [np.ascontiguousarray(delta[k], np.float64) for k in ("prev_pose", "prev_velocity", "prev_bias")]
        g = np.ascontiguousarray(delta["imu"]["gravity"], np.float64)
        vel = np.array(delta["velocity"], np.float64)
    rc = lib().orc_refine_pose_inertial(_p(cam, f64p), _p(points, f64p), _p(uv, f32p), len(points), _p(Kc, f32p), kind,
                                        _p(pred, f64p), C.c_double(sigma), _p(pp, f64p), _p(pv, f64p), _p(pb, f64p),
                                        farr, _p(g, f64p), _p(vel, f64p), None if options is None else C.byref(options),
                                        C.byref(s))
    assert rc == 0
    return cam, vel, s.as_dict()


def ba_linearize(cams, points, obs_ptr, obs_cam, obs_uv, K, huber_delta=5.991 ** 0.5):
    cams = np.ascontiguousarray(cams, np.float64)
    points = np.ascontiguousarray(points, np.float64)
    obs_ptr = np.ascontiguousarray(obs_ptr, np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    Cn, Pn = len(cams), len(points)
    U = np.zeros((Cn, 6, 6)); gc = np.zeros((Cn, 6)); V = np.zeros((Pn, 3, 3)); gp = np.zeros((Pn, 3))
    cost = np.zeros(1)
    L = lib()
    L.orc_ba_linearize.argtypes = [C.c_int, C.c_int, f64p, f64p, i32p, i32p, f32p, f32p, C.c_double,
                                   f64p, f64p, f64p, f64p, f64p]
    rc = L.orc_ba_linearize(Cn, Pn, _p(cams, f64p), _p(points, f64p), _p(obs_ptr, i32p), _p(obs_cam, i32p),
                            _p(obs_uv, f32p), _p(Kc, f32p), float(huber_delta), _p(U, f64p), _p(gc, f64p),
                            _p(V, f64p), _p(gp, f64p), _p(cost, f64p))
    assert rc == 0
    return U, gc, V, gp, float(cost[0])


def refine_pose(cam, points, uv, K, options=None):
    cam = np.array(cam, np.float64, order="C")
    points = np.ascontiguousarray(points, np.float64)
    uv = np.ascontiguousarray(uv, np.float32)
    Kc = np.ascontiguousarray(K, np.float32)
    s = BaSummary()
    rc = lib().orc_refine_pose(_p(cam, f64p), _p(points, f64p), _p(uv, f32p), len(points), _p(Kc, f32p),
                               None if options is None else C.byref(options), C.byref(s))
    assert rc == 0
    return cam, s.as_dict()


def build_local_window(n_kf, new_frame, window, fix_oldest, frame_ptr, frame_pt, pt_ptr, pt_obs):
    frame_ptr = np.ascontiguousarray(frame_ptr, np.int32)
    frame_pt = np.ascontiguousarray(frame_pt, np.int32)
    pt_ptr = np.ascontiguousarray(pt_ptr, np.int32)
    pt_obs = np.ascontiguousarray(pt_obs, np.int32)
    of = np.zeros(n_kf + 1, np.int32)
    oo = np.zeros(n_kf + 1, np.uint8)
    cnt = np.zeros(1, np.int32)
    rc = lib().orc_build_local_window(n_kf, new_frame, window, int(fix_oldest), _p(frame_ptr, i32p),
                                      _p(frame_pt, i32p), _p(pt_ptr, i32p), _p(pt_obs, i32p), _p(of, i32p),
                                      _p(oo, u8p), _p(cnt, i32p))
    assert rc == 0
    n = int(cnt[0])
    return of[:n].copy(), oo[:n].copy()


class PgEdge(C.Structure):
    """orc_pg_edge / rs_pose_graph_edge (identical layout)."""
    _fields_ = [("from_", C.c_int32), ("to", C.c_int32), ("relative", C.c_double * 16)]


def pg_edge_array(loops, cls=PgEdge):
    """[(from, to, relative 4x4), ...] -> ctypes array."""
    arr = (cls * max(len(loops), 1))()
    for i, (a, b, rel) in enumerate(loops):
        arr[i].from_ = int(a)
        arr[i].to = int(b)
        arr[i].relative[:] = list(np.asarray(rel, np.float64).reshape(16))
    return arr


def pose_graph(poses, loops, four_dof=False, gravity=(0.0, 0.0, 0.0), options=None, trace=False):
    """optimization::pose_graph.  Returns (poses' [n,4,4] f32, summary[, trace])."""
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    out = np.zeros_like(poses)
    g = np.ascontiguousarray(gravity, np.float64)
    arr = pg_edge_array(loops)
    s = BaSummary()
    L = lib()
    cap = 256
    buf = (BaIteration * cap)()
    cnt = C.c_int(0)
    if trace:
        L.orc_ba_set_trace(buf, cap, C.byref(cnt))
    try:
        rc = L.orc_pose_graph(len(poses), _p(poses, f32p), arr, len(loops), int(bool(four_dof)), _p(g, f64p),
                              None if options is None else C.byref(options), _p(out, f32p), C.byref(s))
    finally:
        if trace:
            L.orc_ba_set_trace(None, 0, None)
    assert rc == 0
    res = (out.reshape(-1, 4, 4), s.as_dict())
    return res + ([buf[i].as_dict() for i in range(cnt.value)],) if trace else res


def pose_graph_edge(four_dof, x_from, x_to, R0_from, R0_to, up, relative, loop):
    r = np.zeros(6)
    J = np.zeros((6, 12))
    a = [np.ascontiguousarray(v, np.float64) for v in (x_from, x_to, R0_from, R0_to, up, relative)]
    lib().orc_pose_graph_edge(int(bool(four_dof)), *[_p(v, f64p) for v in a], int(bool(loop)), _p(r, f64p), _p(J, f64p))
    return r, J


def transform_points(obs_ptr, obs_kf, before, after, positions):
    pos = np.array(positions, np.float32, order="C")
    op = np.ascontiguousarray(obs_ptr, np.int32)
    ok = np.ascontiguousarray(obs_kf, np.int32)
    rc = lib().orc_transform_points(len(op) - 1, _p(op, i32p), _p(ok, i32p), _p(np.ascontiguousarray(before, np.float32), f32p),
                                    _p(np.ascontiguousarray(after, np.float32), f32p), _p(pos, f32p))
    assert rc == 0
    return pos


def pose_relative(pose_from, pose_to):
    rel = np.zeros(16)
    lib().orc_pose_relative(_p(np.ascontiguousarray(pose_from, np.float32).reshape(16), f32p),
                            _p(np.ascontiguousarray(pose_to, np.float32).reshape(16), f32p), _p(rel, f64p))
    return rel.reshape(4, 4)
