// Adjoint of solid_angles / winding_numbers (tuch/utils/contact.py:49-147) for callers that differentiate through them.
// The reference itself only calls the two under torch.no_grad() (smplify/losses.py:79-82, train/loss.py:251-297,
// eft/loss.py:145-148, utils/segmentation.py:97), but they are plain differentiable torch ops there, and the Python mirror
// tuch.utils.contact keeps them so.  Not on any hot path: two straightforward passes over the (query, triangle) pairs.
//
// With a, b, c the triangle's corners relative to the query, la = |a| ...:
//     num = a . (b x c),   den = la lb lc + (a.b) lc + (a.c) lb + (b.c) la,   Omega = 2 atan2(num, den)
//     dOmega = k (den dnum - num dden),   k = 2 / (num^2 + den^2)
//     dnum/da = b x c (cyclic),   dden/da = (a / la)(lb lc + b.c) + b lc + c lb (cyclic)
// and the query receives minus the sum of the three corner gradients.  torch's conventions at the singular points are kept:
// d|a|/da = 0 at a = 0 (norm backward), and a pair with num = den = 0 (the query ON a corner) gives NaN (atan2 backward:
// 0 / 0), as in the reference.
// Both reductions run in a fixed order (no atomics): bit-reproducible.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr float kInvFourPi = 0.07957747154594767f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(const V3& p, const V3& q) { return V3{p.x - q.x, p.y - q.y, p.z - q.z}; }
__device__ __forceinline__ V3 cross(const V3& p, const V3& q) { return V3{p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x}; }
__device__ __forceinline__ float dot(const V3& p, const V3& q) { return p.x * q.x + p.y * q.y + p.z * q.z; }
__device__ __forceinline__ V3 axpy(float s, const V3& p, const V3& q) { return V3{s * p.x + q.x, s * p.y + q.y, s * p.z + q.z}; }
__device__ __forceinline__ V3 scale(float s, const V3& p) { return V3{s * p.x, s * p.y, s * p.z}; }

// gradients of g * Omega(query, triangle) with respect to the three corners (ga, gb, gc)
__device__ __forceinline__ void pair_adjoint(const V3& q, const V3& A, const V3& B, const V3& C, float g, V3& ga, V3& gb, V3& gc)
{
    const V3 a = sub(A, q), b = sub(B, q), c = sub(C, q);
    const float la = __builtin_sqrtf(dot(a, a)), lb = __builtin_sqrtf(dot(b, b)), lc = __builtin_sqrtf(dot(c, c));
    const V3 bc = cross(b, c), ca = cross(c, a), ab = cross(a, b);
    const float num = dot(a, bc);
    const float dab = dot(a, b), dac = dot(a, c), dbc = dot(b, c);
    const float den = la * lb * lc + dab * lc + dac * lb + dbc * la;
    const float k = 2.0f * g / (num * num + den * den);
    // unit vectors, zero at the origin (torch.norm's backward)
    const V3 ua = la > 0.0f ? scale(1.0f / la, a) : V3{0.f, 0.f, 0.f};
    const V3 ub = lb > 0.0f ? scale(1.0f / lb, b) : V3{0.f, 0.f, 0.f};
    const V3 uc = lc > 0.0f ? scale(1.0f / lc, c) : V3{0.f, 0.f, 0.f};
    const V3 da = axpy(lb * lc + dbc, ua, axpy(lc, b, scale(lb, c)));
    const V3 db = axpy(la * lc + dac, ub, axpy(lc, a, scale(la, c)));
    const V3 dc = axpy(la * lb + dab, uc, axpy(lb, a, scale(la, b)));
    ga = scale(k, axpy(-num, da, scale(den, bc)));
    gb = scale(k, axpy(-num, db, scale(den, ca)));
    gc = scale(k, axpy(-num, dc, scale(den, ab)));
}

// one thread per query, the triangles in order (wave-uniform: scalar loads)
__global__ __launch_bounds__(kBlock) void solid_angle_bwd_points_kernel(
    const float* __restrict__ points, const float* __restrict__ tris, const float* __restrict__ grad_out,
    const float* __restrict__ grad_w, int Q, int F, float* __restrict__ grad_points)
{
    const int b = blockIdx.y, i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= Q) return;
    const float* p = points + ((size_t)b * Q + i) * 3;
    const V3 q = {p[0], p[1], p[2]};
    const float gw = grad_w ? grad_w[(size_t)b * Q + i] * kInvFourPi : 0.0f;
    const float* go = grad_out ? grad_out + ((size_t)b * Q + i) * F : nullptr;
    V3 acc = {0.f, 0.f, 0.f};
    for (int f = 0; f < F; ++f) {
        const float* t = tris + ((size_t)b * F + f) * 9;
        V3 ga, gb, gc;
        pair_adjoint(q, V3{t[0], t[1], t[2]}, V3{t[3], t[4], t[5]}, V3{t[6], t[7], t[8]}, go ? go[f] : gw, ga, gb, gc);
        acc.x -= ga.x + gb.x + gc.x; acc.y -= ga.y + gb.y + gc.y; acc.z -= ga.z + gb.z + gc.z;
    }
    float* o = grad_points + ((size_t)b * Q + i) * 3;
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z;
}

// one thread per triangle, the queries in order
__global__ __launch_bounds__(kBlock) void solid_angle_bwd_triangles_kernel(
    const float* __restrict__ points, const float* __restrict__ tris, const float* __restrict__ grad_out,
    const float* __restrict__ grad_w, int Q, int F, float* __restrict__ grad_tris)
{
    const int b = blockIdx.y, f = blockIdx.x * kBlock + threadIdx.x;
    if (f >= F) return;
    const float* t = tris + ((size_t)b * F + f) * 9;
    const V3 A = {t[0], t[1], t[2]}, B = {t[3], t[4], t[5]}, C = {t[6], t[7], t[8]};
    V3 sa = {0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f};
    for (int i = 0; i < Q; ++i) {
        const float* p = points + ((size_t)b * Q + i) * 3;
        const float g = grad_out ? grad_out[((size_t)b * Q + i) * F + f] : grad_w[(size_t)b * Q + i] * kInvFourPi;
        V3 ga, gb, gc;
        pair_adjoint(V3{p[0], p[1], p[2]}, A, B, C, g, ga, gb, gc);
        sa.x += ga.x; sa.y += ga.y; sa.z += ga.z;
        sb.x += gb.x; sb.y += gb.y; sb.z += gb.z;
        sc.x += gc.x; sc.y += gc.y; sc.z += gc.z;
    }
    float* o = grad_tris + ((size_t)b * F + f) * 9;
    o[0] = sa.x; o[1] = sa.y; o[2] = sa.z; o[3] = sb.x; o[4] = sb.y; o[5] = sb.z; o[6] = sc.x; o[7] = sc.y; o[8] = sc.z;
}

}  // namespace

// grad_out [B,Q,F] (adjoint of tuch_solid_angles) or grad_w [B,Q] (adjoint of tuch_winding_numbers: w = sum_f Omega / 4 pi),
// exactly one of them; grad_points [B,Q,3] and / or grad_triangles [B,F,3,3] (either may be NULL).
extern "C" int tuch_solid_angles_bwd(const float* points, const float* triangles, const float* grad_out, const float* grad_w,
                                     int B, int Q, int F, float* grad_points, float* grad_triangles, void* stream)
{
    TUCH_REQUIRE(points && triangles && (grad_points || grad_triangles), "tuch_solid_angles_bwd: null pointer");
    TUCH_REQUIRE((grad_out != nullptr) != (grad_w != nullptr), "tuch_solid_angles_bwd: exactly one of grad_out / grad_w");
    TUCH_REQUIRE(B > 0 && B <= 65535 && Q > 0 && F > 0, "tuch_solid_angles_bwd: bad sizes");
    if (grad_points)
        hipLaunchKernelGGL(solid_angle_bwd_points_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, (hipStream_t)stream,
                           points, triangles, grad_out, grad_w, Q, F, grad_points);
    if (grad_triangles)
        hipLaunchKernelGGL(solid_angle_bwd_triangles_kernel, dim3(ceil_div(F, kBlock), B), dim3(kBlock), 0, (hipStream_t)stream,
                           points, triangles, grad_out, grad_w, Q, F, grad_triangles);
    return tuch_check_launch("tuch_solid_angles_bwd");
}
