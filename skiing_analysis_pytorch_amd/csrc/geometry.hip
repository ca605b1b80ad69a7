// Geometry post-processing of the VGGT outputs, on device so the dense maps never leave HBM:
//   pose encoding -> extrinsics / intrinsics   (vggt/vggt/utils/pose_enc.py:62-124, rotation.py:14-44)
//   depth map -> world points                  (vggt/vggt/utils/geometry.py:15-117)
//   DLT triangulation of 2D joints over V views (vggt/triangulate.py:13-34, generalised from 2 to V
//                                               views: two rows of A per view)
#include <algorithm>

#include "common.h"

namespace skimi {

// [R, 9] = (T[3], quat xyzw[4], fov_h, fov_w) -> E [R,3,4] = [R(q) | T], K [R,3,3]
__global__ void pose_to_cameras_kernel(const float* __restrict__ pose, float* __restrict__ E, float* __restrict__ K,
                                       long rows, float H, float W) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = pose + r * 9;
    const float i = p[3], j = p[4], k = p[5], w = p[6];
    const float two_s = 2.0f / (i * i + j * j + k * k + w * w);   // rotation.py:27
    float* e = E + r * 12;
    e[0] = 1 - two_s * (j * j + k * k); e[1] = two_s * (i * j - k * w); e[2] = two_s * (i * k + j * w); e[3] = p[0];
    e[4] = two_s * (i * j + k * w); e[5] = 1 - two_s * (i * i + k * k); e[6] = two_s * (j * k - i * w); e[7] = p[1];
    e[8] = two_s * (i * k - j * w); e[9] = two_s * (j * k + i * w); e[10] = 1 - two_s * (i * i + j * j); e[11] = p[2];
    if (K) {
        float* q = K + r * 9;
        const float fy = (H / 2.0f) / tanf(p[7] / 2.0f);   // pose_enc.py:112-113
        const float fx = (W / 2.0f) / tanf(p[8] / 2.0f);
        q[0] = fx; q[1] = 0; q[2] = W / 2; q[3] = 0; q[4] = fy; q[5] = H / 2; q[6] = 0; q[7] = 0; q[8] = 1;
    }
}

// world = R^T (cam - t), cam = ((u-cx) d / fx, (v-cy) d / fy, d)   (geometry.py:47-117)
__global__ __launch_bounds__(256) void unproject_kernel(const float* __restrict__ depth, const float* __restrict__ E,
                                                        const float* __restrict__ K, float* __restrict__ out, int F, int H,
                                                        int W) {
    const long total = (long)F * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int u = (int)(i % W);
        const int v = (int)((i / W) % H);
        const long f = i / ((long)W * H);
        const float* e = E + f * 12;
        const float* k = K + f * 9;
        const float d = depth[i];
        const float x = ((float)u - k[2]) * d / k[0];
        const float y = ((float)v - k[5]) * d / k[4];
        const float z = d;
        // cam-to-world = [R^T | -R^T t] (closed_form_inverse_se3); points . R_c2w^T + t_c2w
        const float tx = -(e[0] * e[3] + e[4] * e[7] + e[8] * e[11]);
        const float ty = -(e[1] * e[3] + e[5] * e[7] + e[9] * e[11]);
        const float tz = -(e[2] * e[3] + e[6] * e[7] + e[10] * e[11]);
        out[i * 3 + 0] = x * e[0] + y * e[4] + z * e[8] + tx;
        out[i * 3 + 1] = x * e[1] + y * e[5] + z * e[9] + ty;
        out[i * 3 + 2] = x * e[2] + y * e[6] + z * e[10] + tz;
    }
}

// One thread per (time step, joint): A = rows {u P[2] - P[0], v P[1]...} over V views, the
// solution is the right-singular vector of A for the smallest singular value = eigenvector of
// A^T A (4x4, symmetric) for the smallest eigenvalue; cyclic Jacobi in double precision.
__global__ void triangulate_dlt_kernel(const float* __restrict__ Kc, const float* __restrict__ Rc,
                                       const float* __restrict__ tc, const float* __restrict__ kp, float* __restrict__ X,
                                       long T, int V, int J) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * J) return;
    const long t = idx / J;
    const int j = (int)(idx - t * J);
    double M[4][4] = {{0}};
    for (int v = 0; v < V; ++v) {
        const float* K = Kc + (t * V + v) * 9;
        const float* R = Rc + (t * V + v) * 9;
        const float* tt = tc + (t * V + v) * 3;
        double P[3][4];   // P = K [R | t]  (triangulate.py:13-16)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 4; ++b) {
                double s = 0;
                for (int c = 0; c < 3; ++c) s += (double)K[a * 3 + c] * (b < 3 ? (double)R[c * 3 + b] : (double)tt[c]);
                P[a][b] = s;
            }
        const double u = kp[((t * V + v) * J + j) * 2], w = kp[((t * V + v) * J + j) * 2 + 1];
        double r0[4], r1[4];
        for (int b = 0; b < 4; ++b) {
            r0[b] = u * P[2][b] - P[0][b];
            r1[b] = w * P[2][b] - P[1][b];
        }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) M[a][b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    double Q[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) off += M[a][b] * M[a][b];
        double diag = 0;
        for (int a = 0; a < 4; ++a) diag += M[a][a] * M[a][a];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (M[p][q] == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
                const double tn = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tn * tn + 1.0), sn = tn * cs;
                for (int k = 0; k < 4; ++k) {   // rotate columns p, q
                    const double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = cs * mkp - sn * mkq;
                    M[k][q] = sn * mkp + cs * mkq;
                }
                for (int k = 0; k < 4; ++k) {   // rotate rows p, q
                    const double mpk = M[p][k], mqk = M[q][k];
                    M[p][k] = cs * mpk - sn * mqk;
                    M[q][k] = sn * mpk + cs * mqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double qkp = Q[k][p], qkq = Q[k][q];
                    Q[k][p] = cs * qkp - sn * qkq;
                    Q[k][q] = sn * qkp + cs * qkq;
                }
            }
    }
    int best = 0;
    for (int a = 1; a < 4; ++a)
        if (M[a][a] < M[best][best]) best = a;
    const double wv = Q[3][best];
    float* o = X + idx * 3;
    o[0] = (float)(Q[0][best] / wv);   // (X / X[3])[:3]  (triangulate.py:33-34)
    o[1] = (float)(Q[1][best] / wv);
    o[2] = (float)(Q[2][best] / wv);
}

}  // namespace skimi

using namespace skimi;

extern "C" {

int skimi_pose_to_cameras(const float* pose_enc, int64_t rows, int32_t H, int32_t W, float* extrinsic, float* intrinsic,
                          void* stream) {
    SKIMI_CHECK_ARG(pose_enc && extrinsic && rows > 0 && H > 0 && W > 0, "skimi_pose_to_cameras: bad arguments");
    hipLaunchKernelGGL(pose_to_cameras_kernel, dim3((unsigned)cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream, pose_enc,
                       extrinsic, intrinsic, (long)rows, (float)H, (float)W);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

int skimi_unproject_depth(const float* depth, const float* extrinsic, const float* intrinsic, float* world_points,
                          int32_t frames, int32_t H, int32_t W, void* stream) {
    SKIMI_CHECK_ARG(depth && extrinsic && intrinsic && world_points && frames > 0 && H > 0 && W > 0,
                    "skimi_unproject_depth: bad arguments");
    const long total = (long)frames * H * W;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 16384);
    hipLaunchKernelGGL(unproject_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, extrinsic, intrinsic,
                       world_points, frames, H, W);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

int skimi_triangulate_dlt(const float* K, const float* R, const float* t, const float* keypoints, float* joints3d,
                          int64_t steps, int32_t views, int32_t joints, void* stream) {
    SKIMI_CHECK_ARG(K && R && t && keypoints && joints3d && steps > 0 && views >= 2 && joints > 0,
                    "skimi_triangulate_dlt: bad arguments (need >= 2 views)");
    const long n = steps * joints;
    hipLaunchKernelGGL(triangulate_dlt_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, K, R, t,
                       keypoints, joints3d, (long)steps, views, joints);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // extern "C"
