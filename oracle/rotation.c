/*
 * oracle/rotation.c — CPU restatement of pack_pose / unpack_pose
 * (reference src/Optimization.cpp:100-112,144-159) and of the ceres/rotation.h
 * functions they call.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * Third-party semantics restated (Ceres 2.x rotation.h, not in /root/reference):
 *   RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis,
 *   AngleAxisToRotationMatrix (Rodrigues; first-order branch when
 *   theta^2 <= DBL_EPSILON), AngleAxisRotatePoint.  The reference instantiates
 *   the matrix conversions with T = float (Eigen::Matrix3f, :100-112).
 * Poses here are ROW-major 4x4 (R(i,j) = pose[4*i+j]).
 */
#include <math.h>

#include "rs_oracle.h"

#define DBL_EPS 2.220446049250313e-16

static void rotation_to_quaternion_f(const float* T, float q[4])
{
#define R(i, j) T[4 * (i) + (j)]
    const float trace = R(0, 0) + R(1, 1) + R(2, 2);
    if (trace >= 0.0f) {
        float t = sqrtf(trace + 1.0f);
        q[0] = 0.5f * t;
        t = 0.5f / t;
        q[1] = (R(2, 1) - R(1, 2)) * t;
        q[2] = (R(0, 2) - R(2, 0)) * t;
        q[3] = (R(1, 0) - R(0, 1)) * t;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3;
        const int k = (j + 1) % 3;
        float t = sqrtf(R(i, i) - R(j, j) - R(k, k) + 1.0f);
        q[i + 1] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R(k, j) - R(j, k)) * t;
        q[j + 1] = (R(j, i) + R(i, j)) * t;
        q[k + 1] = (R(k, i) + R(i, k)) * t;
    }
#undef R
}

static void quaternion_to_angle_axis_f(const float q[4], float aa[3])
{
    const float q1 = q[1], q2 = q[2], q3 = q[3];
    const float sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
    if (sin_squared_theta > 0.0f) {
        const float sin_theta = sqrtf(sin_squared_theta);
        const float cos_theta = q[0];
        const float two_theta = 2.0f * ((cos_theta < 0.0f) ? atan2f(-sin_theta, -cos_theta)
                                                           : atan2f(sin_theta, cos_theta));
        const float k = two_theta / sin_theta;
        aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
    } else {
        aa[0] = q1 * 2.0f; aa[1] = q2 * 2.0f; aa[2] = q3 * 2.0f;
    }
}

static void angle_axis_to_rotation_f(const float aa[3], float R[9] /* row-major */)
{
    const float theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > (float)DBL_EPS) {
        const float theta = sqrtf(theta2);
        const float wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
        const float c = cosf(theta), s = sinf(theta);
        R[0] = c + wx * wx * (1.0f - c);
        R[3] = wz * s + wx * wy * (1.0f - c);
        R[6] = -wy * s + wx * wz * (1.0f - c);
        R[1] = wx * wy * (1.0f - c) - wz * s;
        R[4] = c + wy * wy * (1.0f - c);
        R[7] = wx * s + wy * wz * (1.0f - c);
        R[2] = wy * s + wx * wz * (1.0f - c);
        R[5] = -wx * s + wy * wz * (1.0f - c);
        R[8] = c + wz * wz * (1.0f - c);
    } else {
        R[0] = 1.0f; R[3] = aa[2]; R[6] = -aa[1];
        R[1] = -aa[2]; R[4] = 1.0f; R[7] = aa[0];
        R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0f;
    }
}

/* src/Optimization.cpp:144-149; camera_center src/Frame.cpp:39-42 */
void orc_pack_pose(const float pose[16], double camera[6])
{
    float q[4], aa[3];
    rotation_to_quaternion_f(pose, q);
    quaternion_to_angle_axis_f(q, aa);
    for (int i = 0; i < 3; i++) {
        float c = (-pose[0 * 4 + i] * pose[3] + -pose[1 * 4 + i] * pose[7]) + -pose[2 * 4 + i] * pose[11];
        camera[i] = (double)aa[i];
        camera[3 + i] = (double)c;
    }
}

/* src/Optimization.cpp:151-159 */
void orc_unpack_pose(const double camera[6], float pose[16])
{
    float aa[3] = {(float)camera[0], (float)camera[1], (float)camera[2]};
    float c[3] = {(float)camera[3], (float)camera[4], (float)camera[5]};
    float R[9];
    angle_axis_to_rotation_f(aa, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) pose[4 * i + j] = R[3 * i + j];
        pose[4 * i + 3] = (-R[3 * i] * c[0] + -R[3 * i + 1] * c[1]) + -R[3 * i + 2] * c[2];
    }
    pose[12] = 0.0f; pose[13] = 0.0f; pose[14] = 0.0f; pose[15] = 1.0f;
}

/* ceres::AngleAxisRotatePoint<double> (used by ReprojectionError, src/Optimization.cpp:46) */
void orc_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3])
{
    const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > DBL_EPS) {
        const double theta = sqrt(theta2);
        const double c = cos(theta), s = sin(theta), ti = 1.0 / theta;
        const double w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
        const double wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                               w[0] * pt[1] - w[1] * pt[0]};
        const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - c);
        out[0] = pt[0] * c + wxp[0] * s + w[0] * tmp;
        out[1] = pt[1] * c + wxp[1] * s + w[1] * tmp;
        out[2] = pt[2] * c + wxp[2] * s + w[2] * tmp;
    } else {
        const double wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                               aa[0] * pt[1] - aa[1] * pt[0]};
        out[0] = pt[0] + wxp[0]; out[1] = pt[1] + wxp[1]; out[2] = pt[2] + wxp[2];
    }
}
