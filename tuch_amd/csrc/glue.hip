// Caller-side glue of the training step that the reference runs on the host:
//   * estimate_translation (tuch/utils/geometry.py:114-205): per sample, a weighted 3x3 least-squares
//     problem built from 24/25 joints, solved with numpy on the CPU inside a Python loop with two
//     .cpu() round trips per sample, twice per training step (train_module.py:171-180);
//   * rotation_matrix_to_angle_axis (torchgeometry 0.1.2, called at train_module.py:208-212 and
//     demo_smplify_dc.py:128-132): rotation matrix -> quaternion (four-branch form) -> angle-axis.
// Both are tiny: one thread per sample / matrix, no host round trip.
#include "common.h"

namespace {

constexpr int kBlock = 128;

// numpy promotes the reference's arithmetic to float64 (focal and centre are float64 arrays) and
// solves with LAPACK gesv; the same here: float64 normal equations, LU with partial pivoting.
__global__ __launch_bounds__(kBlock) void estimate_translation_kernel(
    const float* __restrict__ S,            // [B,J,3] model joints
    const float* __restrict__ kp,           // [B,J,3] 2D keypoints + confidence
    const uint8_t* __restrict__ has_anno,   // [B]: 1 -> joints [25,49), 0 -> joints [0,25)
    int B, int J, double focal, double img_size, float* __restrict__ trans)   // [B,3]
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    const int j0 = has_anno[i] ? 25 : 0, j1 = has_anno[i] ? J : 25;
    const double c = 0.5 * img_size;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
    double conf_sum = 0.0;
    for (int j = j0; j < j1; ++j) {
        const float* s = S + ((size_t)i * J + j) * 3;
        const float* k = kp + ((size_t)i * J + j) * 3;
        const double conf = (double)k[2];
        conf_sum += conf;
        const double w = (double)__builtin_sqrtf(k[2]);   // the reference takes the root in float32 (np.sqrt of a float32 array)
        // rows (x, y) of Q = W [F 0 (O - u); 0 F (O - v)], c = W [(u - O) Z - F X; (v - O) Z - F Y]
        for (int a = 0; a < 2; ++a) {
            const double q[3] = {a == 0 ? w * focal : 0.0, a == 1 ? w * focal : 0.0, w * (c - (double)k[a])};
            const double r = w * (((double)k[a] - c) * (double)s[2] - focal * (double)s[a]);
            for (int m = 0; m < 3; ++m) {
                for (int n = 0; n < 3; ++n) A[m][n] += q[m] * q[n];
                b[m] += q[m] * r;
            }
        }
    }
    float* out = trans + (size_t)i * 3;
    if (!(conf_sum > 0.0)) {                 // geometry.py:201: samples without confident joints keep zeros
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    // LU with partial pivoting on the augmented matrix
    double M[3][4] = {{A[0][0], A[0][1], A[0][2], b[0]}, {A[1][0], A[1][1], A[1][2], b[1]}, {A[2][0], A[2][1], A[2][2], b[2]}};
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        for (int r = col + 1; r < 3; ++r)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (piv != col)
            for (int n = 0; n < 4; ++n) { const double t = M[col][n]; M[col][n] = M[piv][n]; M[piv][n] = t; }
        for (int r = col + 1; r < 3; ++r) {
            const double f = M[r][col] / M[col][col];
            for (int n = col; n < 4; ++n) M[r][n] -= f * M[col][n];
        }
    }
    double x[3];
    for (int r = 2; r >= 0; --r) {
        double v = M[r][3];
        for (int n = r + 1; n < 3; ++n) v -= M[r][n] * x[n];
        x[r] = v / M[r][r];
    }
    out[0] = (float)x[0]; out[1] = (float)x[1]; out[2] = (float)x[2];
}

// torchgeometry 0.1.2 conversions.rotation_matrix_to_quaternion (eps = 1e-6, works on the TRANSPOSED
// matrix) followed by quaternion_to_angle_axis, in float32 like the original.
__global__ __launch_bounds__(kBlock) void rotmat_to_angle_axis_kernel(
    const float* __restrict__ R, int N, int row_stride,   // [N,3,row_stride]: 3 (3x3) or 4 (3x4 homogeneous)
    float* __restrict__ aa)                                // [N,3]
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float* m = R + (size_t)i * 3 * row_stride;
    // t[a][b] = element (a, b) of the transposed matrix = m[b][a]
    auto t = [&](int a, int b) { return m[b * row_stride + a]; };
    const bool d2 = t(2, 2) < 1e-6f;
    const bool d0_d1 = t(0, 0) > t(1, 1);
    const bool d0_nd1 = t(0, 0) < -t(1, 1);
    float q[4], tr;
    if (d2 && d0_d1) {
        tr = 1 + t(0, 0) - t(1, 1) - t(2, 2);
        q[0] = t(1, 2) - t(2, 1); q[1] = tr; q[2] = t(0, 1) + t(1, 0); q[3] = t(2, 0) + t(0, 2);
    } else if (d2) {
        tr = 1 - t(0, 0) + t(1, 1) - t(2, 2);
        q[0] = t(2, 0) - t(0, 2); q[1] = t(0, 1) + t(1, 0); q[2] = tr; q[3] = t(1, 2) + t(2, 1);
    } else if (d0_nd1) {
        tr = 1 - t(0, 0) - t(1, 1) + t(2, 2);
        q[0] = t(0, 1) - t(1, 0); q[1] = t(2, 0) + t(0, 2); q[2] = t(1, 2) + t(2, 1); q[3] = tr;
    } else {
        tr = 1 + t(0, 0) + t(1, 1) + t(2, 2);
        q[0] = tr; q[1] = t(1, 2) - t(2, 1); q[2] = t(2, 0) - t(0, 2); q[3] = t(0, 1) - t(1, 0);
    }
    const float s = 0.5f / __builtin_sqrtf(tr);          // the selected tr is >= 1 for any finite input
    for (int k = 0; k < 4; ++k) q[k] *= s;
    const float sin2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const float sn = __builtin_sqrtf(sin2);
    const float two_theta = 2.0f * (q[0] < 0.0f ? atan2f(-sn, -q[0]) : atan2f(sn, q[0]));
    const float k = sin2 > 0.0f ? two_theta / sn : 2.0f;
    float* o = aa + (size_t)i * 3;
    o[0] = q[1] * k; o[1] = q[2] * k; o[2] = q[3] * k;
}

}  // namespace

extern "C" int tuch_estimate_translation(const float* joints3d, const float* keypoints2d, const uint8_t* has_anno,
                                         int B, int J, float focal_length, float img_size, float* trans, void* stream)
{
    TUCH_REQUIRE(joints3d && keypoints2d && has_anno && trans, "tuch_estimate_translation: null pointer");
    TUCH_REQUIRE(B > 0 && J >= 25, "tuch_estimate_translation: bad sizes B=%d J=%d", B, J);
    hipLaunchKernelGGL(estimate_translation_kernel, dim3(ceil_div(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       joints3d, keypoints2d, has_anno, B, J, (double)focal_length, (double)img_size, trans);
    return tuch_check_launch("tuch_estimate_translation");
}

extern "C" int tuch_rotmat_to_angle_axis(const float* rotmat, int N, int row_stride, float* angle_axis, void* stream)
{
    TUCH_REQUIRE(rotmat && angle_axis, "tuch_rotmat_to_angle_axis: null pointer");
    TUCH_REQUIRE(N > 0 && (row_stride == 3 || row_stride == 4), "tuch_rotmat_to_angle_axis: bad sizes");
    hipLaunchKernelGGL(rotmat_to_angle_axis_kernel, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       rotmat, N, row_stride, angle_axis);
    return tuch_check_launch("tuch_rotmat_to_angle_axis");
}

// ---- HD points (tuch/train/loss.py:285: hd = Vert_Regressor[selected] @ verts) -----------------------
// The reference multiplies a dense [N_hd, 6890] regressor with the vertices; every row has three non-zeros
// (barycentric weights of the face the point was sampled from).  Here: point n of body body[n] is HD point
// hd[n]: out[n] = sum_k w[hd[n]][k] * verts[body[n]][idx[hd[n]][k]]; the adjoint scatters with atomics.
namespace {

__global__ __launch_bounds__(256) void hd_points_fwd_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ body, const int32_t* __restrict__ hd,
    const int32_t* __restrict__ idx, const float* __restrict__ w, int V, int N, float* __restrict__ out)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* vb = verts + (size_t)body[n] * V * 3;
    const int h = hd[n];
    float x = 0.f, y = 0.f, z = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wk = w[3 * (size_t)h + k];
        const float* p = vb + 3 * (size_t)idx[3 * (size_t)h + k];
        x = __builtin_fmaf(wk, p[0], x); y = __builtin_fmaf(wk, p[1], y); z = __builtin_fmaf(wk, p[2], z);
    }
    out[3 * (size_t)n] = x; out[3 * (size_t)n + 1] = y; out[3 * (size_t)n + 2] = z;
}

__global__ __launch_bounds__(256) void hd_points_bwd_kernel(
    const float* __restrict__ g, const int32_t* __restrict__ body, const int32_t* __restrict__ hd,
    const int32_t* __restrict__ idx, const float* __restrict__ w, int V, int N, float* __restrict__ g_verts)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float* gb = g_verts + (size_t)body[n] * V * 3;
    const int h = hd[n];
    const float gx = g[3 * (size_t)n], gy = g[3 * (size_t)n + 1], gz = g[3 * (size_t)n + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wk = w[3 * (size_t)h + k];
        float* p = gb + 3 * (size_t)idx[3 * (size_t)h + k];
        atomicAdd(p, wk * gx); atomicAdd(p + 1, wk * gy); atomicAdd(p + 2, wk * gz);
    }
}

}  // namespace

extern "C" int tuch_hd_points_fwd(const float* verts, const int32_t* body_of_point, const int32_t* hd_of_point,
                                  const int32_t* hd_idx, const float* hd_w, int V, int N, float* points, void* stream)
{
    TUCH_REQUIRE(verts && body_of_point && hd_of_point && hd_idx && hd_w && points, "tuch_hd_points_fwd: null pointer");
    TUCH_REQUIRE(V > 0 && N >= 0, "tuch_hd_points_fwd: bad sizes");
    if (N == 0) return TUCH_OK;
    hipLaunchKernelGGL(hd_points_fwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, verts,
                       body_of_point, hd_of_point, hd_idx, hd_w, V, N, points);
    return tuch_check_launch("tuch_hd_points_fwd");
}

extern "C" int tuch_hd_points_bwd(const float* grad_points, const int32_t* body_of_point, const int32_t* hd_of_point,
                                  const int32_t* hd_idx, const float* hd_w, int V, int N, float* grad_verts, void* stream)
{
    TUCH_REQUIRE(grad_points && body_of_point && hd_of_point && hd_idx && hd_w && grad_verts, "tuch_hd_points_bwd: null pointer");
    TUCH_REQUIRE(V > 0 && N >= 0, "tuch_hd_points_bwd: bad sizes");
    if (N == 0) return TUCH_OK;
    hipLaunchKernelGGL(hd_points_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, grad_points,
                       body_of_point, hd_of_point, hd_idx, hd_w, V, N, grad_verts);
    return tuch_check_launch("tuch_hd_points_bwd");
}
