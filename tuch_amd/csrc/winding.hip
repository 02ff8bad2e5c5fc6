// Generalized winding numbers on gfx950 (K2/K3 of SURVEY.md §2.2).
//
// Replaces tuch/utils/contact.py:49-147 (solid_angles + winding_numbers): for every
// query point q and triangle (a,b,c), with A=a-q, B=b-q, C=c-q,
//     omega = 2*atan2(A.(BxC), |A||B||C| + (A.B)|C| + (A.C)|B| + (B.C)|A|)
//     w(q)  = 1/(4 pi) * sum_f omega
// The reference materialises a [1,Q,F,3,3] tensor (3.4 GB at SMPL size); here
// nothing of size QxF exists.  Design (VALU-bound):
//   * one lane owns TWO queries held as float2 so that the ~50 add/mul/fma per
//     (query, triangle) issue as v_pk_*_f32 (the packed-FP32 rate is what the
//     157 TFLOP/s vector peak is quoted on);
//   * the triangle is wave-uniform: its nine floats are read with scalar loads
//     (SGPR broadcast), so the inner loop touches neither LDS nor the vector
//     memory path;
//   * F is split across blocks (grid.y) so that small batches still fill 256 CUs;
//     the per-split partial sums are reduced in fixed order by a second tiny
//     kernel => bit-reproducible results, no float atomics.
#include "common.h"
#include "model.h"
#include "workspace.h"
#include "tree_device.h"
#include <stdlib.h>

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr int kBlock = 256;
constexpr int kQueriesPerLane = 2;
constexpr int kQueriesPerBlock = kBlock * kQueriesPerLane;
constexpr int kStripBlock = 128;                              // strip / segment kernels
constexpr int kStripQueries = kStripBlock * kQueriesPerLane;

// atan on [0,1]: t * P(t^2), minimax, max abs error 9.6e-8 in float32.
__device__ __forceinline__ v2f atan_poly(v2f t)
{
    const v2f s = t * t;
    v2f p = splat2(0.0024567253421992064f);
    p = fma2(p, s, splat2(-0.014401361346244812f));
    p = fma2(p, s, splat2(0.03978123143315315f));
    p = fma2(p, s, splat2(-0.07234857976436615f));
    p = fma2(p, s, splat2(0.10498946160078049f));
    p = fma2(p, s, splat2(-0.14161229133605957f));
    p = fma2(p, s, splat2(0.19985906779766083f));
    p = fma2(p, s, splat2(-0.33332598209381104f));
    p = fma2(p, s, splat2(0.9999998807907104f));
    return p * t;
}

// atan2(y, x) for a pair.  atan2(0,0) = 0 as torch.atan2 returns (contact.py:105:
// a query that is a corner of the triangle has a zero vector => num = den = 0).
__device__ __forceinline__ v2f atan2_pair(v2f y, v2f x)
{
    v2f r;
    float t[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float ax = __builtin_fabsf(x[k]), ay = __builtin_fabsf(y[k]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-37f);
        const float mn = __builtin_fminf(ax, ay);
        t[k] = mn * __builtin_amdgcn_rcpf(mx);
    }
    const v2f p = atan_poly((v2f){t[0], t[1]});
    const v2f swapped = splat2(kHalfPi) - p;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float v = (__builtin_fabsf(y[k]) > __builtin_fabsf(x[k])) ? swapped[k] : p[k];
        v = (x[k] < 0.0f) ? (kPi - v) : v;
        r[k] = __builtin_copysignf(v, y[k]);
    }
    return r;
}

// atan2(num, den) when every lane of the wave has |num| < den/8 for both of its queries
// (far triangles, the overwhelming majority): atan(t) = t - t^3/3 + t^5/5, truncation error
// < t^7/7 <= 7e-8 at the gate and ~1e-20 for a typical far triangle (t ~ 1e-3).  Otherwise the whole wave takes the general path; the choice is wave-uniform.
__device__ __forceinline__ v2f half_angle(v2f num, v2f den)
{
    const bool big = !(__builtin_fabsf(num[0]) < 0.125f * den[0]) || !(__builtin_fabsf(num[1]) < 0.125f * den[1]);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
        const v2f t = num * (v2f){__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
        const v2f s = t * t;
        v2f p = fma2(s, splat2(0.2f), splat2(-1.0f / 3.0f));
        p = fma2(p, s, splat2(1.0f));
        return p * t;
    }
    return atan2_pair(num, den);
}

__device__ __forceinline__ v2f sqrt2(v2f a)
{
    return (v2f){__builtin_amdgcn_sqrtf(a[0]), __builtin_amdgcn_sqrtf(a[1])};
}

// Half solid angle atan2(num, den) of one (wave-uniform) triangle for a pair of queries.
__device__ __forceinline__ v2f half_solid_angle(const float* __restrict__ t, v2f qx, v2f qy, v2f qz)
{
    const v2f Ax = splat2(t[0]) - qx, Ay = splat2(t[1]) - qy, Az = splat2(t[2]) - qz;
    const v2f Bx = splat2(t[3]) - qx, By = splat2(t[4]) - qy, Bz = splat2(t[5]) - qz;
    const v2f Cx = splat2(t[6]) - qx, Cy = splat2(t[7]) - qy, Cz = splat2(t[8]) - qz;
    const v2f nA = sqrt2(fma2(Az, Az, fma2(Ay, Ay, Ax * Ax)));
    const v2f nB = sqrt2(fma2(Bz, Bz, fma2(By, By, Bx * Bx)));
    const v2f nC = sqrt2(fma2(Cz, Cz, fma2(Cy, Cy, Cx * Cx)));
    const v2f cx = fma2(By, Cz, -(Bz * Cy));
    const v2f cy = fma2(Bz, Cx, -(Bx * Cz));
    const v2f cz = fma2(Bx, Cy, -(By * Cx));
    const v2f num = fma2(Az, cz, fma2(Ay, cy, Ax * cx));
    const v2f dAB = fma2(Az, Bz, fma2(Ay, By, Ax * Bx));
    const v2f dBC = fma2(Bz, Cz, fma2(By, Cy, Bx * Cx));
    const v2f dAC = fma2(Az, Cz, fma2(Ay, Cy, Ax * Cx));
    v2f den = nA * nB * nC;
    den = fma2(dAB, nC, den);
    den = fma2(dAC, nB, den);
    den = fma2(dBC, nA, den);
    return half_angle(num, den);
}

// partial[b][split][q] = sum over the split's triangles of atan2(num, den)
__global__ __launch_bounds__(kBlock) void winding_partial_kernel(
    const float* __restrict__ points,   // [B,Q,3]
    const float* __restrict__ tris,     // [B,F,9]
    int Q, int F, int tris_per_split,
    float* __restrict__ partial)        // [B,S,Q]
{
    const int b = blockIdx.z, split = blockIdx.y, nsplit = gridDim.y;
    const int q0 = blockIdx.x * kQueriesPerBlock + threadIdx.x;
    const int q1 = q0 + kBlock;
    const float* pts = points + (size_t)b * Q * 3;
    // out-of-range lanes recompute query Q-1; their result is not stored
    const int c0 = q0 < Q ? q0 : Q - 1, c1 = q1 < Q ? q1 : Q - 1;
    const v2f qx = {pts[3 * c0 + 0], pts[3 * c1 + 0]};
    const v2f qy = {pts[3 * c0 + 1], pts[3 * c1 + 1]};
    const v2f qz = {pts[3 * c0 + 2], pts[3 * c1 + 2]};

    const int f_begin = split * tris_per_split;
    const int f_end = min(F, f_begin + tris_per_split);
    const float* t = tris + ((size_t)b * F + f_begin) * 9;
    v2f acc = splat2(0.0f);
    for (int f = f_begin; f < f_end; ++f, t += 9)
        acc += half_solid_angle(t, qx, qy, qz);

    float* out = partial + ((size_t)b * nsplit + split) * Q;
    if (q0 < Q) out[q0] = acc[0];
    if (q1 < Q) out[q1] = acc[1];
}

// w = (2 / 4pi) * sum_s partial, exterior = w <= thresh (losses.py:82, loss.py:262)
__global__ __launch_bounds__(kBlock) void winding_finalize_kernel(
    const float* __restrict__ partial, int Q, int stride, int nsplit, float thresh,   // partial [B,S,stride]
    float* __restrict__ w, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const float* p = partial + (size_t)b * nsplit * stride + q;
    float acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) acc += p[(size_t)s * stride];
    const float val = acc * (0.5f / kPi);
    if (w) w[(size_t)b * Q + q] = val;
    if (exterior) exterior[(size_t)b * Q + q] = val <= thresh ? 1 : 0;
}

// the same for partial sums stored in the cluster tree's query order: position i holds vertex qperm[i]
__global__ __launch_bounds__(kBlock) void winding_finalize_tree_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ qperm, int V, int Vp, int nsplit, float thresh,
    float* __restrict__ w, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= V) return;                              // positions >= V repeat the last vertex
    const float* p = partial + (size_t)b * nsplit * Vp + i;
    float acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) acc += p[(size_t)s * Vp];
    const float val = acc * (0.5f / kPi);
    const int q = qperm[i];
    if (w) w[(size_t)b * V + q] = val;
    if (exterior) exterior[(size_t)b * V + q] = val <= thresh ? 1 : 0;
}

// contact.py:49-109 materialised (API parity; small inputs): out[b][q][f] = 2*atan2(...)
__global__ __launch_bounds__(kBlock) void solid_angles_kernel(
    const float* __restrict__ points, const float* __restrict__ tris, int Q, int F,
    float* __restrict__ out)
{
    const int b = blockIdx.z;
    const int f = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const float* pt = points + ((size_t)b * Q + q) * 3;
    const float* t = tris + ((size_t)b * F + f) * 9;
    const v2f h = half_solid_angle(t, splat2(pt[0]), splat2(pt[1]), splat2(pt[2]));
    out[((size_t)b * Q + q) * F + f] = 2.0f * h[0];
}

// triangles[b][f] = verts[b][faces[f]]  (losses.py:81, loss.py:260)
__global__ __launch_bounds__(kBlock) void gather_triangles_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ faces, int V, int F,
    float* __restrict__ tris)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;   // one (face, corner) per thread
    if (i >= F * 3) return;
    const int v = faces[i];
    const float* src = verts + ((size_t)b * V + v) * 3;
    float* dst = tris + ((size_t)b * F * 3 + i) * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}


// ---- triangle-strip form ------------------------------------------------------------------
// The model's faces are walked as triangle strips (model.hip build_strips): consecutive
// triangles share two vertices, so per emitted triangle only ONE new vertex needs its
// difference vector and norm (1 sqrt instead of 3) and only two of the three dot products
// are new.  The triple product A.(BxC) equals X.n for ANY corner X of the triangle, with
// n = (b-a)x(c-a) independent of the query: n is computed once per (body, triangle) by
// gather_stream_kernel and arrives as wave-uniform scalars, so the numerator costs one dot
// product.  (n is built from edge vectors, which avoids the cancellation inside BxC for far
// triangles.)  Stream element = (x, y, z, sign, nx, ny, nz, -): sign 0 primes a strip, +-1
// emits the triangle of the last three elements with that orientation; n is already
// multiplied by the sign.  Three register slots are rotated by position modulo 3 (loop unrolled
// by 3, no register moves).
struct StreamElem { float x, y, z, sign, nx, ny, nz, pad; };

__global__ __launch_bounds__(kBlock) void gather_stream_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ vidx, const float* __restrict__ sign,
    int V, int L, int Lpad, StreamElem* __restrict__ out)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= Lpad) return;
    StreamElem e = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < L) {
        const float* vb = verts + (size_t)b * V * 3;
        const float* c = vb + 3 * vidx[p];
        e.x = c[0]; e.y = c[1]; e.z = c[2]; e.sign = sign[p];
        if (e.sign != 0.0f) {
            const float* a = vb + 3 * vidx[p - 2];
            const float* bb = vb + 3 * vidx[p - 1];
            const float ux = bb[0] - a[0], uy = bb[1] - a[1], uz = bb[2] - a[2];
            const float wx = c[0] - a[0], wy = c[1] - a[1], wz = c[2] - a[2];
            e.nx = e.sign * (uy * wz - uz * wy);
            e.ny = e.sign * (uz * wx - ux * wz);
            e.nz = e.sign * (ux * wy - uy * wx);
        }
    }
    out[(size_t)b * Lpad + p] = e;
}

struct Slot { v2f x, y, z, n; };

// atan2(num, den) as half_angle(), plus: den == 0 exactly happens when the query IS a corner of
// the triangle (its difference vector, hence one norm and two dot products, are exactly zero);
// the reference then evaluates atan2(0, 0) = 0 (contact.py:105), while X.n carries rounding noise.
__device__ __forceinline__ v2f half_angle_strip(v2f num, v2f den)
{
    const bool big = !(__builtin_fabsf(num[0]) < 0.125f * den[0]) || !(__builtin_fabsf(num[1]) < 0.125f * den[1]);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
        const v2f t = num * (v2f){__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
        const v2f s = t * t;
        v2f p = fma2(s, splat2(0.2f), splat2(-1.0f / 3.0f));
        p = fma2(p, s, splat2(1.0f));
        return p * t;
    }
    v2f r = atan2_pair(num, den);
    if (den[0] == 0.0f) r[0] = 0.0f;
    if (den[1] == 0.0f) r[1] = 0.0f;
    return r;
}

// One stream element; A = register slot of the new vertex (position mod 3).
template <int A>
__device__ __forceinline__ void strip_step(const StreamElem e, bool emit, Slot (&s)[3], v2f (&d)[3],
                                           v2f qx, v2f qy, v2f qz, v2f& acc)
{
    constexpr int Bq = (A + 1) % 3, Cq = (A + 2) % 3;      // slots of stream positions p-2 and p-1
    s[A].x = splat2(e.x) - qx;
    s[A].y = splat2(e.y) - qy;
    s[A].z = splat2(e.z) - qz;
    s[A].n = sqrt2(fma2(s[A].z, s[A].z, fma2(s[A].y, s[A].y, s[A].x * s[A].x)));
    // d[k] = dot of the two slots other than k
    d[Cq] = fma2(s[A].z, s[Bq].z, fma2(s[A].y, s[Bq].y, s[A].x * s[Bq].x));
    d[Bq] = fma2(s[A].z, s[Cq].z, fma2(s[A].y, s[Cq].y, s[A].x * s[Cq].x));
    if (emit && e.sign != 0.0f) {                            // wave-uniform
        const v2f num = fma2(s[A].z, splat2(e.nz), fma2(s[A].y, splat2(e.ny), s[A].x * splat2(e.nx)));
        v2f den = s[0].n * s[1].n * s[2].n;
        den = fma2(d[0], s[0].n, den);
        den = fma2(d[1], s[1].n, den);
        den = fma2(d[2], s[2].n, den);
        acc += half_angle_strip(num, den);
    }
}

// ---- the same strip walk with ONE query per lane (64 queries per wavefront) ------------------------
// Plain and packed FP32 instructions have the same per-float throughput on gfx950, so halving the
// block costs nothing per query; the smaller block tests "near" for fewer nodes of the tree.
struct Slot1 { float x, y, z, n; };

__device__ __forceinline__ float atan2_single(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-37f);
    const float mn = __builtin_fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const v2f p2 = atan_poly((v2f){t, t});
    float v = ay > ax ? kHalfPi - p2[0] : p2[0];
    v = x < 0.0f ? kPi - v : v;
    return __builtin_copysignf(v, y);
}

__device__ __forceinline__ float half_angle_strip1(float num, float den)
{
    const bool big = !(__builtin_fabsf(num) < 0.125f * den);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
        const float t = num * __builtin_amdgcn_rcpf(den);
        const float s = t * t;
        float p = __builtin_fmaf(s, 0.2f, -1.0f / 3.0f);
        p = __builtin_fmaf(p, s, 1.0f);
        return p * t;
    }
    const float r = atan2_single(num, den);
    return den == 0.0f ? 0.0f : r;
}

// Half solid angle atan2(num, den) of one (wave-uniform) triangle for one query per lane (per-triangle form
// of contact.py:79-105, used for the small closed segment meshes).
__device__ __forceinline__ float half_solid_angle1(const float* __restrict__ t, float qx, float qy, float qz)
{
    const float Ax = t[0] - qx, Ay = t[1] - qy, Az = t[2] - qz;
    const float Bx = t[3] - qx, By = t[4] - qy, Bz = t[5] - qz;
    const float Cx = t[6] - qx, Cy = t[7] - qy, Cz = t[8] - qz;
    const float nA = __builtin_amdgcn_sqrtf(__builtin_fmaf(Az, Az, __builtin_fmaf(Ay, Ay, Ax * Ax)));
    const float nB = __builtin_amdgcn_sqrtf(__builtin_fmaf(Bz, Bz, __builtin_fmaf(By, By, Bx * Bx)));
    const float nC = __builtin_amdgcn_sqrtf(__builtin_fmaf(Cz, Cz, __builtin_fmaf(Cy, Cy, Cx * Cx)));
    const float cx = __builtin_fmaf(By, Cz, -(Bz * Cy));
    const float cy = __builtin_fmaf(Bz, Cx, -(Bx * Cz));
    const float cz = __builtin_fmaf(Bx, Cy, -(By * Cx));
    const float num = __builtin_fmaf(Az, cz, __builtin_fmaf(Ay, cy, Ax * cx));
    const float dAB = __builtin_fmaf(Az, Bz, __builtin_fmaf(Ay, By, Ax * Bx));
    const float dBC = __builtin_fmaf(Bz, Cz, __builtin_fmaf(By, Cy, Bx * Cx));
    const float dAC = __builtin_fmaf(Az, Cz, __builtin_fmaf(Ay, Cy, Ax * Cx));
    float den = nA * nB * nC;
    den = __builtin_fmaf(dAB, nC, den);
    den = __builtin_fmaf(dAC, nB, den);
    den = __builtin_fmaf(dBC, nA, den);
    const bool big = !(__builtin_fabsf(num) < 0.125f * den);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
        const float tt = num * __builtin_amdgcn_rcpf(den);
        const float s2 = tt * tt;
        float p = __builtin_fmaf(s2, 0.2f, -1.0f / 3.0f);
        p = __builtin_fmaf(p, s2, 1.0f);
        return p * tt;
    }
    return atan2_single(num, den);
}

template <int A>
__device__ __forceinline__ void strip_step1(const StreamElem e, Slot1 (&s)[3], float (&d)[3],
                                            float qx, float qy, float qz, float& acc)
{
    constexpr int Bq = (A + 1) % 3, Cq = (A + 2) % 3;
    s[A].x = e.x - qx;
    s[A].y = e.y - qy;
    s[A].z = e.z - qz;
    s[A].n = __builtin_amdgcn_sqrtf(__builtin_fmaf(s[A].z, s[A].z, __builtin_fmaf(s[A].y, s[A].y, s[A].x * s[A].x)));
    d[Cq] = __builtin_fmaf(s[A].z, s[Bq].z, __builtin_fmaf(s[A].y, s[Bq].y, s[A].x * s[Bq].x));
    d[Bq] = __builtin_fmaf(s[A].z, s[Cq].z, __builtin_fmaf(s[A].y, s[Cq].y, s[A].x * s[Cq].x));
    if (e.sign != 0.0f) {                                    // wave-uniform
        const float num = __builtin_fmaf(s[A].z, e.nz, __builtin_fmaf(s[A].y, e.ny, s[A].x * e.nx));
        float den = s[0].n * s[1].n * s[2].n;
        den = __builtin_fmaf(d[0], s[0].n, den);
        den = __builtin_fmaf(d[1], s[1].n, den);
        den = __builtin_fmaf(d[2], s[2].n, den);
        acc += half_angle_strip1(num, den);
    }
}

// one run of the strip loop over stream elements [off, off+len), len % 3 == 0; the stream has
// three readable elements past its end for the prefetch
__device__ __forceinline__ void run_stream1(const StreamElem* __restrict__ st, int off, int len, Slot1 (&s)[3],
                                            float (&d)[3], float qx, float qy, float qz, float& acc)
{
    const StreamElem* p = st + off;
    const StreamElem* end = p + len;
    StreamElem n0 = p[0], n1 = p[1], n2 = p[2];
    for (; p < end; p += 3) {
        const StreamElem e0 = n0, e1 = n1, e2 = n2;
        n0 = p[3]; n1 = p[4]; n2 = p[5];
        strip_step1<0>(e0, s, d, qx, qy, qz, acc);
        strip_step1<1>(e1, s, d, qx, qy, qz, acc);
        strip_step1<2>(e2, s, d, qx, qy, qz, acc);
    }
}

__global__ __launch_bounds__(kStripBlock) void winding_strip_kernel(
    const float* __restrict__ points,            // [B,Q,3]
    const StreamElem* __restrict__ stream,       // [B,Lpad]
    int Q, int Lpad, int elems_per_split,        // Lpad % 3 == 0, elems_per_split % 3 == 0
    const int32_t* __restrict__ counts,          // [B] valid queries per body, or nullptr (= Q)
    float* __restrict__ partial)                 // [B,S,Q]
{
    // XCD-aware launch order: the body index varies fastest, so with the observed round-robin of
    // workgroups over the 8 XCDs all workgroups of body b run on XCD b % 8 and its stream is
    // fetched into ONE L2 instead of all eight (FETCH_SIZE 133 MB -> 19 MB, profiles/)
    const int b = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int q0 = blockIdx.z * kStripQueries + threadIdx.x, q1 = q0 + kStripBlock;
    if (counts && (int)(blockIdx.z * kStripQueries) >= counts[b]) return;   // padding of a ragged point set
    const float* pts = points + (size_t)b * Q * 3;
    const int c0 = q0 < Q ? q0 : Q - 1, c1 = q1 < Q ? q1 : Q - 1;
    const v2f qx = {pts[3 * c0 + 0], pts[3 * c1 + 0]};
    const v2f qy = {pts[3 * c0 + 1], pts[3 * c1 + 1]};
    const v2f qz = {pts[3 * c0 + 2], pts[3 * c1 + 2]};
    const int p_first = split * elems_per_split;
    const int p_end = min(Lpad - 3, p_first + elems_per_split);   // the prefetch reads one triple ahead
    const StreamElem* st = stream + (size_t)b * Lpad;
    Slot s[3];
    v2f d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k].x = s[k].y = s[k].z = s[k].n = splat2(0.0f);
        d[k] = splat2(0.0f);
    }
    v2f acc = splat2(0.0f);
    // the triple before the chunk only primes the slots (a strip may straddle the boundary)
    int p = max(p_first - 3, 0);
    StreamElem n0 = st[p], n1 = st[p + 1], n2 = st[p + 2];
    for (; p < p_end; p += 3) {
        const bool emit = p >= p_first;
        const StreamElem e0 = n0, e1 = n1, e2 = n2;
        n0 = st[p + 3]; n1 = st[p + 4]; n2 = st[p + 5];        // next triple in flight during the math
        strip_step<0>(e0, emit, s, d, qx, qy, qz, acc);
        strip_step<1>(e1, emit, s, d, qx, qy, qz, acc);
        strip_step<2>(e2, emit, s, d, qx, qy, qz, acc);
    }
    float* out = partial + ((size_t)b * nsplit + split) * Q;
    if (q0 < Q) out[q0] = acc[0];
    if (q1 < Q) out[q1] = acc[1];
}

// ---- hierarchical form (cluster_tree.hip) ------------------------------------------------------
// Posed bounding boxes of all tree nodes of one body: leaves from their strip elements, inner
// nodes bottom-up from their children.  One workgroup per body, boxes kept in LDS.

// Bounding volume of a node = 9 slabs (a 18-DOP): the coordinate axes and the six face diagonals
// x+-y, x+-z, y+-z (unnormalised sums: rounding is monotone, so a point of the convex hull can
// never test as outside).  Much tighter than a box around a limb that runs diagonally.
constexpr int kSlabs = 9;
constexpr int kSlabStride = 10;            // floats per half (9 + pad): node = [lo[10], hi[10]]

__device__ __forceinline__ void slab_project(float x, float y, float z, float (&p)[kSlabs])
{
    p[0] = x; p[1] = y; p[2] = z;
    p[3] = x + y; p[4] = x - y; p[5] = x + z; p[6] = x - z; p[7] = y + z; p[8] = y - z;
}

__global__ __launch_bounds__(kBoundsBlock) void tree_leaf_bounds_kernel(
    const StreamElem* __restrict__ stream, int T, const TreeNode* __restrict__ nodes, int N,
    const int32_t* __restrict__ height_off, const int32_t* __restrict__ height_nodes,
    float* __restrict__ bounds)                  // [B,N,2*kSlabStride]
{
    const int b = blockIdx.y;
    const StreamElem* st = stream + (size_t)b * T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = height_off[0] + blockIdx.x * (kBoundsBlock / 64) + wave;       // one leaf per wave
    if (i >= height_off[1]) return;
    const int node = height_nodes[i];
    const int off = nodes[node].ex_off, len = nodes[node].ex_len;
    float lo[kSlabs], hi[kSlabs];
#pragma unroll
    for (int k = 0; k < kSlabs; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; }
    for (int p = lane; p < len; p += 64) {
        const StreamElem e = st[off + p];
        float pr[kSlabs];
        slab_project(e.x, e.y, e.z, pr);
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) { lo[k] = fminf(lo[k], pr[k]); hi[k] = fmaxf(hi[k], pr[k]); }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], m));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m));
        }
    if (lane == 0) {
        float* o = bounds + ((size_t)b * N + node) * (2 * kSlabStride);
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) { o[k] = lo[k]; o[kSlabStride + k] = hi[k]; }
        o[kSlabs] = 0.0f;
        o[kSlabStride + kSlabs] = 0.0f;
    }
}

// Winding numbers by walking the cluster tree: a node whose posed bounding volume contains none of
// the wavefront's 64 queries contributes through its boundary cap (exactly the same solid angle), a
// leaf that does is summed face by face, an inner node that does is descended.  All decisions are
// wave-uniform.  One query per lane: plain and packed FP32 have the same per-float throughput on
// gfx950, and a 64-query block is "near" fewer nodes than a 128-query block (-12 % run time).
// The nodes above the frontier are handled without a pass of their own: every wavefront first tests
// the ancestors of its subtree top-down (a static list).  At the first ancestor that is far, the
// wavefront of the ancestor's FIRST frontier subtree adds the ancestor's cap and all others leave,
// so splitting the tree into many subtrees (load balance) does not cost one cap per far subtree.
// Queries: the model's own vertices in the tree's order (qperm, Q = V), or arbitrary points [B,Q,3]
// in the caller's order (qperm == nullptr; counts[b] of them are real): any order is exact, blocks of
// 64 consecutive points that are close in space are fast.
// kCount: also add the number of stream elements walked (leaf strips, caps) to stats[0], stats[1]
// (measurement only: tuch_winding_tree_work).
constexpr int kMaxAncestors = 8;           // deeper ancestors are simply never replaced by their caps
constexpr int kTreeQueries = 64;           // queries per wavefront

template <bool kCount>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void winding_tree_kernel(
    const float* __restrict__ verts,             // query points [B,Q,3]
    const StreamElem* __restrict__ stream,       // [B,T]
    const TreeNode* __restrict__ nodes, const float* __restrict__ bounds, int N,
    const int32_t* __restrict__ frontier, const int32_t* __restrict__ ancestors,   // [S], [S][kMaxAncestors]
    const int32_t* __restrict__ order,           // launch order over (subtree, 128-query block) pairs, or nullptr
    const int32_t* __restrict__ qperm, const int32_t* __restrict__ counts,
    int V, int T, int nsub, int num_bodies, float* __restrict__ partial,   // [B,S,qblocks*64]
    unsigned long long* __restrict__ stats)
{
    int walked_exact = 0, walked_cap = 0;
    // grid (8, pairs, B/8), x fastest: workgroups go round-robin to the 8 XCDs, so XCD x works on body
    // 8 z + x -- ONE body per XCD at a time, whose 1.1 MB posed stream stays in that XCD's 4 MB L2 (with all
    // bodies in flight at once, 8 streams competed for each L2 and were fetched twice).  Within a body the
    // (subtree, query block) pairs come in the model's launch order, long-running first.
    const int b = blockIdx.z * gridDim.x + blockIdx.x;
    if (b >= num_bodies) return;
    int sub, qb, i0;
    bool real = true;                                // padding entries of a ragged point set report w = 0
    if (qperm) {
        const int pair = __builtin_amdgcn_readfirstlane(order[blockIdx.y >> 1]);
        sub = pair >> 16;
        qb = (pair & 0xffff) * 2 + (blockIdx.y & 1);            // the two halves of the model's 128-blocks
        i0 = qperm[qb * kTreeQueries + threadIdx.x];
    } else {
        sub = blockIdx.y % nsub;
        qb = blockIdx.y / nsub;
        const int n = counts ? counts[b] : V;
        if (qb * kTreeQueries >= n) return;          // padding of a ragged point set (partial sums preset to 0)
        real = qb * kTreeQueries + (int)threadIdx.x < n;
        i0 = min(qb * kTreeQueries + (int)threadIdx.x, n - 1);
    }
    const float* pts = verts + (size_t)b * V * 3;
    const float qx = pts[3 * i0 + 0], qy = pts[3 * i0 + 1], qz = pts[3 * i0 + 2];
    const StreamElem* st = stream + (size_t)b * T;
    const float* bb = bounds + (size_t)b * N * (2 * kSlabStride);
    Slot1 s[3];
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k].x = s[k].y = s[k].z = s[k].n = 0.0f;
        d[k] = 0.0f;
    }
    float acc = 0.0f;
    // near = some query of the wavefront is inside all slabs of the node (wave-uniform):
    // outside = some slab separates the query from the node, max_k max(lo_k - p_k, p_k - hi_k) > 0
    auto is_near = [&](int node) {
        const float* box = bb + (size_t)node * (2 * kSlabStride);
        const float qp[kSlabs] = {qx, qy, qz, qx + qy, qx - qy, qx + qz, qx - qz, qy + qz, qy - qz};
        float out = -1.0f;
#pragma unroll
        for (int k = 0; k < kSlabs; ++k)
            out = __builtin_fmaxf(out, __builtin_fmaxf(box[k] - qp[k], qp[k] - box[kSlabStride + k]));
        return __builtin_amdgcn_ballot_w64(!(out > 0.0f)) != 0;
    };
    // partial sums are stored in the tree's query order (coalesced); the finalize kernel un-permutes
    const int qblocks = gridDim.y / nsub;
    float* out = partial + ((size_t)b * nsub + sub) * ((size_t)qblocks * kTreeQueries) + qb * kTreeQueries;
    int node = __builtin_amdgcn_readfirstlane(frontier[sub]);
    int end = __builtin_amdgcn_readfirstlane(nodes[node].skip);
    const int32_t* anc = ancestors + (size_t)sub * kMaxAncestors;
    for (int k = 0; k < kMaxAncestors; ++k) {
        const int x = __builtin_amdgcn_readfirstlane(anc[k]);
        if (x < 0) break;
        if (is_near(x)) continue;
        // far ancestor: its cap stands for all frontier subtrees below it
        const bool first = sub == 0 || __builtin_amdgcn_readfirstlane(frontier[sub - 1]) < x;
        if (first) {
            const TreeNode nd = nodes[x];
            run_stream1(st, nd.cap_off, nd.cap_len, s, d, qx, qy, qz, acc);
            if (kCount) walked_cap += nd.cap_len;
        }
        end = node;                                // nothing left to walk
        break;
    }
    while (node < end) {
        const TreeNode nd = nodes[node];
        const bool near = is_near(node);
        if (near && nd.ex_len == 0) {
            node = node + 1;
        } else {
            run_stream1(st, near ? nd.ex_off : nd.cap_off, near ? nd.ex_len : nd.cap_len, s, d, qx, qy, qz, acc);
            if (kCount) {
                walked_exact += near ? nd.ex_len : 0;
                walked_cap += near ? 0 : nd.cap_len;
            }
            node = nd.skip;
        }
        node = __builtin_amdgcn_readfirstlane(node);
    }
    out[threadIdx.x] = real ? acc : 0.0f;
    if (kCount && threadIdx.x == 0) {
        atomicAdd(stats, (unsigned long long)walked_exact);
        atomicAdd(stats + 1, (unsigned long long)walked_cap);
    }
}

// ---- body segments (tuch/utils/segmentation.py) ---------------------------------
// cap vertex of band c = mean of the band's boundary-loop vertices (segmentation.py:74-76)
__global__ __launch_bounds__(64) void cap_centroid_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ cap_off,
    const int32_t* __restrict__ cap_vidx, int V, int K, float* __restrict__ caps,   // [B,K,3]
    int32_t* __restrict__ seg_count_clear, int S)         // [B,S] or nullptr: cleared here (no memset node of its own)
{
    // one wave per (cap, body): lanes stride over the loop, shuffle-reduce
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (seg_count_clear && c == 0)
        for (int i = lane; i < S; i += 64) seg_count_clear[(size_t)b * S + i] = 0;
    const float* vb = verts + (size_t)b * V * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const int beg = cap_off[c], end = cap_off[c + 1];
    for (int k = beg + lane; k < end; k += 64) {
        const float* p = vb + 3 * cap_vidx[k];
        sx += p[0]; sy += p[1]; sz += p[2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_down(sx, o, 64); sy += __shfl_down(sy, o, 64); sz += __shfl_down(sz, o, 64);
    }
    if (lane == 0) {
        const float inv = 1.0f / (float)(end - beg);
        float* o = caps + ((size_t)b * K + c) * 3;
        o[0] = sx * inv; o[1] = sy * inv; o[2] = sz * inv;
    }
}

// closed-segment triangles: index < V -> body vertex, else cap vertex (segmentation.py:77)
__global__ __launch_bounds__(kBlock) void gather_segment_triangles_kernel(
    const float* __restrict__ verts, const float* __restrict__ caps,
    const int32_t* __restrict__ seg_faces, int V, int K, int Fs, float* __restrict__ tris)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= Fs * 3) return;
    const int v = seg_faces[i];
    const float* src = v < V ? verts + ((size_t)b * V + v) * 3 : caps + ((size_t)b * K + (v - V)) * 3;
    float* dst = tris + ((size_t)b * Fs * 3 + i) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// winding number of segment vertices w.r.t. their own closed segment (segmentation.py:81-99).
// The filter can only turn INTERIOR vertices exterior (losses.py:85-89, loss.py:264-266), so
// only those are tested: segment_compact_kernel builds, per (body, segment), the list of the
// segment's interior vertices; the winding kernel walks dense lists.  Blocks come from a
// host-built table (segment, first list position); the segment's faces are split over grid.y
// and reduced in fixed order by segment_finalize_kernel.  (List order comes from an atomic
// counter and may vary between runs; every entry's result does not.)
constexpr int kSegBlock = 64;     // one wave = 64 compacted queries per workgroup
constexpr int kSegChunk = 128;    // triangles staged in LDS per pass
constexpr int kSegSplits = 16;   // maximum; few (compacted) queries per segment: parallelism comes from the faces
int seg_splits(const tuch_contact_model* m)
{
    const int x = m->opt.seg_splits;
    return x < 1 ? 1 : (x > kSegSplits ? kSegSplits : x);
}

// list[b][seg_q_off[s] + k] = position (within the segment) of its k-th vertex that needs the test
__global__ __launch_bounds__(kBlock) void segment_compact_kernel(
    const uint8_t* __restrict__ exterior,      // [B,V] or nullptr (= test every segment vertex)
    const int32_t* __restrict__ seg_of_q, const int32_t* __restrict__ seg_q_off,
    const int32_t* __restrict__ seg_q_vidx, int V, int Qs_total, int S,
    int32_t* __restrict__ count,               // [B,S], zeroed by the caller
    int32_t* __restrict__ list)                // [B,Qs_total]
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= Qs_total) return;
    const int s = seg_of_q[q];
    const int local = q - seg_q_off[s];
    if (!exterior) {
        list[(size_t)b * Qs_total + q] = local;
        if (local == 0) count[b * S + s] = seg_q_off[s + 1] - seg_q_off[s];
        return;
    }
    // one atomic per (wavefront, segment) instead of one per vertex: the counters are few (B x S) and atomics on one
    // address serialise (~100 ns each); the lanes of a wavefront mostly belong to one segment
    bool mine = !exterior[(size_t)b * V + seg_q_vidx[q]];
    const int lane = threadIdx.x & 63;
    while (unsigned long long todo = __builtin_amdgcn_ballot_w64(mine)) {
        const int s0 = __builtin_amdgcn_readlane(s, __builtin_ctzll(todo));
        const unsigned long long group = __builtin_amdgcn_ballot_w64(mine && s == s0);
        int base = 0;
        if (lane == __builtin_ctzll(group)) base = atomicAdd(&count[b * S + s0], __builtin_popcountll(group));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(group));
        if (mine && s == s0) {
            list[(size_t)b * Qs_total + seg_q_off[s] + base + __builtin_popcountll(group & ((1ull << lane) - 1ull))] = local;
            mine = false;
        }
    }
}

__global__ __launch_bounds__(kSegBlock) void segment_winding_kernel(
    const float* __restrict__ verts, const float* __restrict__ seg_tris,
    const int32_t* __restrict__ seg_blocks, const int32_t* __restrict__ seg_q_off,
    const int32_t* __restrict__ seg_q_vidx, const int32_t* __restrict__ seg_f_off,
    const int32_t* __restrict__ count, const int32_t* __restrict__ list, int V, int Fs_total,
    int Qs_total, int S, float* __restrict__ partial)     // [B,kSegSplits,Qs_total] by list position
{
    // bodies vary fastest in the launch order: consecutive workgroups go to consecutive XCDs, and
    // the heavy table entries (big segments) would otherwise all land on the same one or two XCDs
    const int b = blockIdx.x, split = blockIdx.y;
    const int s = seg_blocks[2 * blockIdx.z], k_start = seg_blocks[2 * blockIdx.z + 1];
    const int n = count[b * S + s];
    if (k_start >= n) return;
    const int q_beg = seg_q_off[s];
    const int32_t* mine = list + (size_t)b * Qs_total + q_beg;
    // one query per lane: the compacted lists are short (a few dozen interior vertices per body and
    // segment), 64-query blocks waste far fewer lanes than 128-query blocks
    const int k0 = k_start + threadIdx.x;
    const int v0 = seg_q_vidx[q_beg + mine[min(k0, n - 1)]];
    const float* vb = verts + (size_t)b * V * 3;
    const float qx = vb[3 * v0 + 0], qy = vb[3 * v0 + 1], qz = vb[3 * v0 + 2];
    const int f_seg = seg_f_off[s], f_cnt = seg_f_off[s + 1] - f_seg;
    const int nsplit = gridDim.y;
    const int per = (f_cnt + nsplit - 1) / nsplit;
    const int f_beg = f_seg + split * per, f_end = min(f_seg + f_cnt, f_beg + per);
    // Few waves are alive here (compacted queries), so a per-triangle scalar load would expose its
    // full memory latency every iteration.  Stage the chunk's triangles in LDS with coalesced
    // vector loads (many in flight), then read them back as broadcasts.
    __shared__ float sT[kSegChunk * 9];
    float acc = 0.0f;
    for (int chunk = f_beg; chunk < f_end; chunk += kSegChunk) {
        const int cn = min(kSegChunk, f_end - chunk);
        const float* src = seg_tris + ((size_t)b * Fs_total + chunk) * 9;
        __syncthreads();
        for (int i = threadIdx.x; i < cn * 9; i += kSegBlock) sT[i] = src[i];
        __syncthreads();
        for (int f = 0; f < cn; ++f) {
            float tri[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) tri[e] = sT[f * 9 + e];
            acc += half_solid_angle1(tri, qx, qy, qz);
        }
    }
    float* out = partial + ((size_t)b * nsplit + split) * Qs_total + q_beg;
    if (k0 < n) out[k0] = acc;
}

// vertices that are NOT exterior to their own segment are re-marked exterior in the body
// flags (losses.py:87-89, loss.py:265-266)
__global__ __launch_bounds__(kBlock) void segment_finalize_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ seg_of_q,
    const int32_t* __restrict__ seg_q_off, const int32_t* __restrict__ seg_q_vidx,
    const int32_t* __restrict__ count, const int32_t* __restrict__ list, int V, int Qs_total, int S,
    int nsplit, float thresh, float* __restrict__ seg_w, uint8_t* __restrict__ seg_ext, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;       // a list position
    if (q >= Qs_total) return;
    const int s = seg_of_q[q];
    const int k = q - seg_q_off[s];
    if (k >= count[b * S + s]) return;
    float acc = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) acc += partial[((size_t)b * nsplit + sp) * Qs_total + q];
    const float w = acc * (0.5f / kPi);
    const int qq = seg_q_off[s] + list[(size_t)b * Qs_total + q];   // the vertex's slot in the segment tables
    const size_t o = (size_t)b * Qs_total + qq;
    if (seg_w) seg_w[o] = w;
    if (seg_ext) seg_ext[o] = w <= thresh;
    if (exterior && !(w <= thresh)) exterior[(size_t)b * V + seg_q_vidx[qq]] = 1;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// option winding_strips = 0 selects the plain per-triangle kernel (A/B measurements)
bool use_strips(const tuch_contact_model* m) { return m->opt.winding_strips != 0; }

// Number of stream chunks for the strip kernel: fill the 256 CUs x 4 SIMDs x 8 resident waves an
// integral number of times (tail effect), keep chunks long enough to amortise the priming triple.
int choose_strip_splits(int B, int Q, int L)
{
    const long slots = 256L * 4 * 8;
    const long base = (long)B * ceil_div(Q, kStripQueries) * (kStripBlock / 64);
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 32; ++s) {
        if (s > 1 && L / s < 384) break;
        const double rounds = (double)(base * s) / slots;
        const double eff = rounds / (double)((long)((base * s + slots - 1) / slots));
        if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    return best;
}

struct ExteriorLayout {
    size_t tris, partial, bounds, stats, caps, seg_tris, seg_partial, seg_count, seg_list, body_flags, ray, ray_bytes, total;
    int lpad;
    int tree_frontier;         // frontier used by the hierarchical path (-1: flat path)
    int tree_subs;
};

// Inside test by ray crossings (ray_winding.hip): option winding_ray = 0 never, 1 (default) when only the flags are
// wanted, 2 also when the caller asks for w (reported as crossings - fan angles)
int ray_mode(const tuch_contact_model* m)
{
    if (!tuch_ray_available(m)) return 0;
    return m->opt.winding_ray;
}

// option winding_tree = 0 keeps the flat strip walk (A/B measurements)
bool use_tree(const tuch_contact_model* m)
{
    if (m->tree_nodes <= 0) return false;
    return m->opt.winding_tree != 0;
}

// smallest frontier (set of subtrees, one workgroup column each) that yields enough wavefronts to
// balance the uneven subtree costs over 256 CUs
int choose_frontier(const tuch_contact_model* m, int B)
{
    const long target = m->opt.tree_waves;   // counted in 128-query blocks: two wavefronts each
    int f = 0;
    while (f + 1 < m->tree_num_frontiers &&
           (long)B * m->tree_qblocks * (m->tree_frontier_off_host[f + 1] - m->tree_frontier_off_host[f]) < target) ++f;
    return f;
}

inline int strip_lpad(int L) { return ceil_div(L, 3) * 3 + 6; }

int choose_splits(int B, int Q, int F);

ExteriorLayout exterior_layout(const tuch_contact_model* m, int B)
{
    ExteriorLayout l;
    size_t o = 0;
    l.lpad = strip_lpad(m->strip_len);
    // triangle buffer or strip stream, whichever is larger (both forms are supported)
    const size_t tri_bytes = (size_t)B * m->F * 9 * sizeof(float);
    size_t strip_bytes = (size_t)B * l.lpad * sizeof(StreamElem);
    int max_splits = choose_splits(B, m->V, m->F) > choose_strip_splits(B, m->V, strip_lpad(m->strip_len))
                         ? choose_splits(B, m->V, m->F) : choose_strip_splits(B, m->V, strip_lpad(m->strip_len));
    l.tree_frontier = -1;
    l.tree_subs = 0;
    if (m->tree_nodes > 0) {       // sized for both paths: the switch is read per call
        l.tree_frontier = choose_frontier(m, B);
        l.tree_subs = m->tree_frontier_off_host[l.tree_frontier + 1] - m->tree_frontier_off_host[l.tree_frontier];
        const size_t tree_bytes = (size_t)B * (m->tree_stream_len + 3) * sizeof(StreamElem);
        if (tree_bytes > strip_bytes) strip_bytes = tree_bytes;
        if (l.tree_subs > max_splits) max_splits = l.tree_subs;
    }
    l.tris = tuch_ws_take(o, tri_bytes > strip_bytes ? tri_bytes : strip_bytes);
    l.partial = tuch_ws_take(o, (size_t)B * max_splits * (m->V + 128) * sizeof(float));
    l.bounds = tuch_ws_take(o, (size_t)B * (m->tree_nodes > 0 ? m->tree_nodes : 1) * 2 * kSlabStride * sizeof(float));
    l.stats = tuch_ws_take(o, 256);
    l.caps = tuch_ws_take(o, (size_t)B * (m->num_caps > 0 ? m->num_caps : 1) * 3 * sizeof(float));
    // triangles of the closed segments, or (ray form) faces + boundary entries; partial sums, two arrays in the ray form
    const int seg_entries = m->seg_ray_total > m->seg_f_total ? m->seg_ray_total : m->seg_f_total;
    l.seg_tris = tuch_ws_take(o, ((size_t)B * (seg_entries > 0 ? seg_entries : 1) + 1) * 9 * sizeof(float));
    l.seg_partial = o; o += align256(2 * (size_t)B * kSegSplits * (m->seg_q_total > 0 ? m->seg_q_total : 1) * sizeof(float) +
                                      ((size_t)B * (m->num_seg_blocks > 0 ? m->num_seg_blocks : 1) * 2 + 4) * sizeof(int32_t));
    l.seg_count = tuch_ws_take(o, (size_t)B * (m->num_segments > 0 ? m->num_segments : 1) * sizeof(int32_t));
    l.seg_list = tuch_ws_take(o, (size_t)B * (m->seg_q_total > 0 ? m->seg_q_total : 1) * sizeof(int32_t));
    l.body_flags = tuch_ws_take(o, (size_t)B * m->V);        // the body test's flags, kept for the fused segment filter
    { tuch_ws_pause nested; l.ray_bytes = tuch_ray_workspace_bytes(m, B, 0); }
    l.ray = tuch_ws_take(o, l.ray_bytes);
    l.total = o;
    return l;
}

int choose_splits(int B, int Q, int F)
{
    // enough blocks for >= ~8 waves per SIMD over 256 CUs, at least ~512 triangles per split
    const int qblocks = ceil_div(Q, kQueriesPerBlock);
    int s = 1;
    while (s < 16 && (long)B * qblocks * s < 2048 && F / (s * 2) >= 512) s *= 2;
    return s;
}

// the posed strip stream of the tree (leaf strips + caps) and the slabs of every node
void launch_tree_boxes(const tuch_contact_model* m, const float* verts, int B, StreamElem* st, float* bounds, hipStream_t s)
{
    const int T = m->tree_stream_len + 3;
    hipLaunchKernelGGL(gather_stream_kernel, dim3(ceil_div(T, kBlock), B), dim3(kBlock), 0, s, verts,
                       (const int32_t*)m->tree_vidx, (const float*)m->tree_sign, m->V, m->tree_stream_len, T, st);
    hipLaunchKernelGGL(tree_leaf_bounds_kernel, dim3(ceil_div(m->tree_leaves, kBoundsBlock / 64), B),
                       dim3(kBoundsBlock), 0, s, (const StreamElem*)st, T, (const TreeNode*)m->tree_node,
                       m->tree_nodes, (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, bounds);
    hipLaunchKernelGGL(tree_inner_bounds_kernel<kSlabStride>, dim3(B), dim3(kBoundsBlock),
                       tree_inner_bounds_lds<kSlabStride>(m->tree_nodes), s, (const TreeNode*)m->tree_node, m->tree_nodes,
                       (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, m->tree_heights,
                       bounds);
}

// gather the posed stream, box every node, walk the tree: partial sums into ws + l.partial
void launch_tree_walk(const tuch_contact_model* m, const ExteriorLayout& l, const float* verts, int B, char* ws,
                      unsigned long long* stats, hipStream_t s)
{
    StreamElem* st = (StreamElem*)(ws + l.tris);
    const int T = m->tree_stream_len + 3;
    float* bounds = (float*)(ws + l.bounds);
    launch_tree_boxes(m, verts, B, st, bounds, s);
    const int f0 = m->tree_frontier_off_host[l.tree_frontier];
    // the model's query blocks hold 128 vertices (m->tree_qblocks of them): two wavefronts each
    const dim3 grid(B < 8 ? B : 8, 2 * l.tree_subs * m->tree_qblocks, ceil_div(B, 8));
    const int32_t* frontier = (const int32_t*)m->tree_frontier_nodes + f0;
    const int32_t* ancestors = (const int32_t*)m->tree_ancestors + (size_t)f0 * kMaxAncestors;
    const int32_t* order = (const int32_t*)m->tree_launch_order + (size_t)f0 * m->tree_qblocks;
    if (stats)
        hipLaunchKernelGGL(winding_tree_kernel<true>, grid, dim3(64), 0, s, verts, (const StreamElem*)st,
                           (const TreeNode*)m->tree_node, (const float*)bounds, m->tree_nodes, frontier, ancestors, order,
                           (const int32_t*)m->tree_qperm, (const int32_t*)nullptr, m->V, T, l.tree_subs, B, (float*)(ws + l.partial), stats);
    else
        hipLaunchKernelGGL(winding_tree_kernel<false>, grid, dim3(64), 0, s, verts, (const StreamElem*)st,
                           (const TreeNode*)m->tree_node, (const float*)bounds, m->tree_nodes, frontier, ancestors, order,
                           (const int32_t*)m->tree_qperm, (const int32_t*)nullptr, m->V, T, l.tree_subs, B, (float*)(ws + l.partial), stats);
}

}  // namespace

extern "C" size_t tuch_winding_workspace_bytes(int B, int Q, int F)
{
    if (B <= 0 || Q <= 0 || F <= 0) return 0;
    return (size_t)B * choose_splits(B, Q, F) * Q * sizeof(float);
}

extern "C" int tuch_winding_numbers(const float* points, const float* triangles, int B, int Q, int F,
                                    float* w, uint8_t* exterior, float exterior_thresh,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(points && triangles && (w || exterior), "tuch_winding_numbers: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && Q > 0 && F > 0, "tuch_winding_numbers: bad sizes B=%d Q=%d F=%d", B, Q, F);
    const int nsplit = choose_splits(B, Q, F);
    const size_t need = (size_t)B * nsplit * Q * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        tuch_set_error("tuch_winding_numbers: workspace %zu < %zu bytes", workspace_bytes, need);
        return TUCH_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int per_split = ceil_div(F, nsplit);
    dim3 grid(ceil_div(Q, kQueriesPerBlock), nsplit, B);
    hipLaunchKernelGGL(winding_partial_kernel, grid, dim3(kBlock), 0, s, points, triangles, Q, F,
                       per_split, (float*)workspace);
    hipLaunchKernelGGL(winding_finalize_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, s,
                       (const float*)workspace, Q, Q, nsplit, exterior_thresh, w, exterior);
    return tuch_check_launch("tuch_winding_numbers");
}

extern "C" int tuch_solid_angles(const float* points, const float* triangles, int B, int Q, int F,
                                 float* out, void* stream)
{
    TUCH_REQUIRE(points && triangles && out, "tuch_solid_angles: null pointer");
    TUCH_REQUIRE(B > 0 && Q > 0 && F > 0 && F <= 65535 && B <= 65535, "tuch_solid_angles: bad sizes");
    hipLaunchKernelGGL(solid_angles_kernel, dim3(ceil_div(Q, kBlock), F, B), dim3(kBlock), 0,
                       (hipStream_t)stream, points, triangles, Q, F, out);
    return tuch_check_launch("tuch_solid_angles");
}

extern "C" int tuch_gather_triangles(const float* verts, const int32_t* faces, int B, int V, int F,
                                     float* triangles, void* stream)
{
    TUCH_REQUIRE(verts && faces && triangles, "tuch_gather_triangles: null pointer");
    TUCH_REQUIRE(B > 0 && V > 0 && F > 0, "tuch_gather_triangles: bad sizes");
    hipLaunchKernelGGL(gather_triangles_kernel, dim3(ceil_div(F * 3, kBlock), B), dim3(kBlock), 0,
                       (hipStream_t)stream, verts, faces, V, F, triangles);
    return tuch_check_launch("tuch_gather_triangles");
}

// ---- model-level entry points ------------------------------------------------------
extern "C" size_t tuch_exterior_workspace_bytes(const tuch_contact_model* m, int B)
{
    if (!m || B <= 0) return 0;
    tuch_ws_scope scope(m->opt.canary != 0);
    return exterior_layout(m, B).total;
}

// exterior flags of tuch/smplify/losses.py:81-89 and tuch/train/loss.py:260-266:
//   exterior = winding_numbers(verts, verts[faces]).le(thresh), then every vertex that is
//   interior to its own closed body segment is re-marked exterior.
extern "C" int tuch_exterior_flags(const tuch_contact_model* m, const float* verts, int B,
                                   int apply_segments, float thresh, float* w, uint8_t* exterior,
                                   float* seg_w, uint8_t* seg_exterior,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(m && verts && exterior, "tuch_exterior_flags: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_exterior_flags: bad batch %d", B);
    tuch_ws_scope scope(m->opt.canary != 0);
    const ExteriorLayout l = scope.record(0, [&] { return exterior_layout(m, B); });
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_exterior_flags: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    if (tuch_ray_available(m)) scope.record(l.ray, [&] { tuch_ray_layout_touch(m, B, 0); return 0; });
    scope.arm(workspace, m->canary_hits, (hipStream_t)stream);
    char* ws = (char*)workspace;
    float* tris = (float*)(ws + l.tris);
    hipStream_t s = (hipStream_t)stream;
    int rc = TUCH_OK;
    const int ray = ray_mode(m);
    const bool segments = apply_segments && m->num_segments > 0;
    const bool body_by_rays = ray == 2 || (ray == 1 && !w);
    const bool segments_by_rays = segments && m->seg_link_off && (ray == 2 || (ray == 1 && !seg_w));
    // flags only, by rays, leaf-assisted: the whole segment filter is one launch behind the body test (ray_winding.hip)
    const bool segments_fused = segments && segments_by_rays && body_by_rays && !seg_w && !seg_exterior &&
                                tuch_ray_segment_fused_available(m);
    if (segments && !segments_fused) {
        // what the segment pass needs of the vertices alone goes first, off the critical chain behind the body test
        if (m->num_caps > 0) {
            hipLaunchKernelGGL(cap_centroid_kernel, dim3(m->num_caps, B), dim3(64), 0, s,
                               verts, (const int32_t*)m->cap_off, (const int32_t*)m->cap_vidx, m->V,
                               m->num_caps, (float*)(ws + l.caps), (int32_t*)(ws + l.seg_count), m->num_segments);
        } else if (hipMemsetAsync(ws + l.seg_count, 0, (size_t)B * m->num_segments * sizeof(int32_t), s) != hipSuccess) {
            tuch_set_error("tuch_exterior_flags: hipMemsetAsync failed");
            return TUCH_ERR_HIP;
        }
        if (segments_by_rays)
            tuch_ray_segment_prepare(m, verts, (const float*)(ws + l.caps), body_by_rays && m->seg_elem_mask, B,
                                     (float*)(ws + l.seg_tris), s);
    }
    if (body_by_rays) {
        rc = tuch_ray_exterior_verts(m, verts, B, thresh, exterior, w, ws + l.ray, s, nullptr,
                                     segments_fused ? (uint8_t*)(ws + l.body_flags) : (uint8_t*)nullptr);
        if (rc != TUCH_OK) return rc;
    } else if (use_strips(m) && use_tree(m)) {
        launch_tree_walk(m, l, verts, B, ws, nullptr, s);
        hipLaunchKernelGGL(winding_finalize_tree_kernel, dim3(ceil_div(m->V, kBlock), B), dim3(kBlock), 0, s,
                           (const float*)(ws + l.partial), (const int32_t*)m->tree_qperm, m->V,
                           2 * m->tree_qblocks * kTreeQueries, l.tree_subs, thresh, w, exterior);
    } else if (use_strips(m) && m->strip_len > 0) {
        StreamElem* st = (StreamElem*)tris;
        hipLaunchKernelGGL(gather_stream_kernel, dim3(ceil_div(l.lpad, kBlock), B), dim3(kBlock), 0, s, verts,
                           (const int32_t*)m->strip_vidx, (const float*)m->strip_sign, m->V, m->strip_len,
                           l.lpad, st);
        const int nsplit = choose_strip_splits(B, m->V, l.lpad);
        const int per_split = ceil_div(ceil_div(l.lpad - 6, nsplit), 3) * 3;
        hipLaunchKernelGGL(winding_strip_kernel, dim3(B, nsplit, ceil_div(m->V, kStripQueries)), dim3(kStripBlock),
                           0, s, verts, (const StreamElem*)st, m->V, l.lpad, per_split, (const int32_t*)nullptr,
                           (float*)(ws + l.partial));
        hipLaunchKernelGGL(winding_finalize_kernel, dim3(ceil_div(m->V, kBlock), B), dim3(kBlock), 0, s,
                           (const float*)(ws + l.partial), m->V, m->V, nsplit, thresh, w, exterior);
    } else {
        hipLaunchKernelGGL(gather_triangles_kernel, dim3(ceil_div(m->F * 3, kBlock), B), dim3(kBlock), 0, s,
                           verts, (const int32_t*)m->faces, m->V, m->F, tris);
        rc = tuch_winding_numbers(verts, tris, B, m->V, m->F, w, exterior, thresh, ws + l.partial,
                                  l.bounds - l.partial, stream);
        if (rc != TUCH_OK) return rc;
    }
    if (segments_fused) {
        rc = tuch_ray_segment_flags_one(m, verts, (const uint8_t*)(ws + l.body_flags), tuch_ray_segment_counts(m, B, ws + l.ray), B, thresh,
                                        exterior, s);
        if (rc != TUCH_OK) return rc;
    } else if (segments) {
        float* caps = (float*)(ws + l.caps);
        float* seg_tris = (float*)(ws + l.seg_tris);
        float* seg_partial = (float*)(ws + l.seg_partial);
        int32_t* seg_count = (int32_t*)(ws + l.seg_count);
        int32_t* seg_list = (int32_t*)(ws + l.seg_list);
        const bool all = seg_w || seg_exterior;     // the detailed outputs want every segment vertex
        hipLaunchKernelGGL(segment_compact_kernel, dim3(ceil_div(m->seg_q_total, kBlock), B), dim3(kBlock), 0, s,
                           all ? (const uint8_t*)nullptr : (const uint8_t*)exterior, (const int32_t*)m->seg_of_q,
                           (const int32_t*)m->seg_q_off, (const int32_t*)m->seg_q_vidx, m->V, m->seg_q_total,
                           m->num_segments, seg_count, seg_list);
        // by ray crossings under the same rule as the body test (ray_mode: when only flags are wanted, or always with
        // TUCH_WINDING_RAY=2)
        if (segments_by_rays) {
            // the body's inside test, when it ran by ray crossings just above, has left the crossings of every vertex
            // with the body faces of its segments
            rc = tuch_ray_segment_flags(m, verts, caps, seg_count, seg_list,
                                        body_by_rays ? tuch_ray_segment_counts(m, B, ws + l.ray) : nullptr, B, seg_splits(m),
                                        thresh, seg_tris, (int32_t*)seg_partial, seg_w, seg_exterior, exterior, s);
            if (rc != TUCH_OK) return rc;
            return tuch_check_launch("tuch_exterior_flags");
        }
        hipLaunchKernelGGL(gather_segment_triangles_kernel, dim3(ceil_div(m->seg_f_total * 3, kBlock), B),
                           dim3(kBlock), 0, s, verts, (const float*)caps, (const int32_t*)m->seg_faces,
                           m->V, m->num_caps, m->seg_f_total, seg_tris);
        hipLaunchKernelGGL(segment_winding_kernel, dim3(B, seg_splits(m), m->num_seg_blocks), dim3(kSegBlock),
                           0, s, verts, (const float*)seg_tris, (const int32_t*)m->seg_blocks,
                           (const int32_t*)m->seg_q_off, (const int32_t*)m->seg_q_vidx,
                           (const int32_t*)m->seg_f_off, (const int32_t*)seg_count, (const int32_t*)seg_list,
                           m->V, m->seg_f_total, m->seg_q_total, m->num_segments, seg_partial);
        hipLaunchKernelGGL(segment_finalize_kernel, dim3(ceil_div(m->seg_q_total, kBlock), B), dim3(kBlock), 0, s,
                           (const float*)seg_partial, (const int32_t*)m->seg_of_q, (const int32_t*)m->seg_q_off,
                           (const int32_t*)m->seg_q_vidx, (const int32_t*)seg_count, (const int32_t*)seg_list,
                           m->V, m->seg_q_total, m->num_segments, seg_splits(m), thresh, seg_w, seg_exterior, exterior);
    }
    return tuch_check_launch("tuch_exterior_flags");
}

// Measurement aid: walk the tree for `verts` and report how many stream elements the wavefronts
// actually stepped through.  out_host = {leaf-strip elements, cap elements, wavefronts, elements of
// the flat strip stream (what every 128-query block would step through without the tree)}.
// One element step serves 64 queries (one wavefront).  Synchronises the stream.
extern "C" int tuch_winding_tree_work(const tuch_contact_model* m, const float* verts, int B,
                                      void* workspace, size_t workspace_bytes, unsigned long long* out_host, void* stream)
{
    TUCH_REQUIRE(m && verts && out_host, "tuch_winding_tree_work: null pointer");
    TUCH_REQUIRE(m->tree_nodes > 0, "tuch_winding_tree_work: the model has no cluster tree");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_winding_tree_work: bad batch %d", B);
    const ExteriorLayout l = exterior_layout(m, B);
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_winding_tree_work: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* stats = (unsigned long long*)(ws + l.stats);
    if (hipMemsetAsync(stats, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return TUCH_ERR_HIP;
    launch_tree_walk(m, l, verts, B, ws, stats, s);
    if (hipMemcpyAsync(out_host, stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        tuch_set_error("tuch_winding_tree_work: copy back failed");
        return TUCH_ERR_HIP;
    }
    out_host[2] = (unsigned long long)B * l.tree_subs * m->tree_qblocks * 2;
    out_host[3] = (unsigned long long)m->strip_len;
    return tuch_check_launch("tuch_winding_tree_work");
}

// Measurement aid for the ray-crossing inside test (ray_winding.hip): out_host = {strip elements walked by all
// wavefronts (64 queries each), wavefronts launched}.  Synchronises the stream.
extern "C" int tuch_ray_work(const tuch_contact_model* m, const float* verts, int B, void* workspace,
                             size_t workspace_bytes, unsigned long long* out_host, void* stream)
{
    TUCH_REQUIRE(m && verts && out_host, "tuch_ray_work: null pointer");
    TUCH_REQUIRE(tuch_ray_available(m), "tuch_ray_work: the model has no cluster tree / vertex rings");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_ray_work: bad batch %d", B);
    const ExteriorLayout l = exterior_layout(m, B);
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_ray_work: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    return tuch_ray_exterior_verts(m, verts, B, 0.99f, nullptr, nullptr, (char*)workspace + l.ray, (hipStream_t)stream, out_host);
}

// winding numbers of ARBITRARY query points against the model's mesh posed by `verts`
// (tuch/train/loss.py:297: HD points offset along the face normals).  points [B,Q,3]; counts [B]
// (device, optional) = number of meaningful points per body when the set is ragged and padded
// to Q; w / exterior of padded entries are 0 / 1.
struct PointsLayout { size_t stream, bounds, partial, total; int frontier, nsub, qblocks; };

static PointsLayout points_layout(const tuch_contact_model* m, int B, int Q)
{
    PointsLayout l;
    l.qblocks = ceil_div(Q, kTreeQueries);
    // enough wavefronts to balance the uneven subtree walks (as choose_frontier, for Q points per body)
    int f = 0;
    while (f + 1 < m->tree_num_frontiers &&
           (long)B * l.qblocks * (m->tree_frontier_off_host[f + 1] - m->tree_frontier_off_host[f]) < 65536L) ++f;   // wavefronts
    l.frontier = f;
    l.nsub = m->tree_frontier_off_host[f + 1] - m->tree_frontier_off_host[f];
    size_t o = 0;
    l.stream = tuch_ws_take(o, (size_t)B * (m->tree_stream_len + 3) * sizeof(StreamElem));
    l.bounds = tuch_ws_take(o, (size_t)B * m->tree_nodes * 2 * kSlabStride * sizeof(float));
    l.partial = tuch_ws_take(o, (size_t)B * l.nsub * l.qblocks * kTreeQueries * sizeof(float));
    l.total = o;
    return l;
}

extern "C" size_t tuch_winding_points_workspace_bytes(const tuch_contact_model* m, int B, int Q)
{
    if (!m || B <= 0 || Q <= 0) return 0;
    tuch_ws_scope scope(m->opt.canary != 0);
    const int lpad = strip_lpad(m->strip_len);
    const size_t flat = align256((size_t)B * lpad * sizeof(StreamElem)) +
                        align256((size_t)B * choose_strip_splits(B, Q, lpad) * Q * sizeof(float));
    if (m->tree_nodes <= 0) return flat;
    size_t tree = points_layout(m, B, Q).total;
    const size_t ray = tuch_ray_workspace_bytes(m, B, Q);
    if (ray > tree) tree = ray;
    return tree > flat ? tree : flat;
}

extern "C" int tuch_winding_points(const tuch_contact_model* m, const float* verts, const float* points,
                                   const int32_t* counts, int B, int Q, float thresh, float* w,
                                   uint8_t* exterior, void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(m && verts && points && (w || exterior), "tuch_winding_points: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && Q > 0, "tuch_winding_points: bad sizes B=%d Q=%d", B, Q);
    const size_t need = tuch_winding_points_workspace_bytes(m, B, Q);
    if (!workspace || workspace_bytes < need) {
        tuch_set_error("tuch_winding_points: workspace %zu < %zu bytes", workspace_bytes, need);
        return TUCH_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int ray = ray_mode(m);
    tuch_ws_scope scope(m->opt.canary != 0);
    if (ray == 2 || (ray == 1 && !w)) {   // off-surface points: the winding number is the integer crossing count
        scope.record(0, [&] { tuch_ray_layout_touch(m, B, Q); return 0; });
        scope.arm(workspace, m->canary_hits, s);
        return tuch_ray_exterior_points(m, verts, points, counts, B, Q, thresh, exterior, w, workspace, s);
    }
    if (use_strips(m) && use_tree(m)) {
        // hierarchical walk (cluster tree + boundary caps) with the caller's points as queries
        const PointsLayout l = scope.record(0, [&] { return points_layout(m, B, Q); });
        scope.arm(workspace, m->canary_hits, s);
        char* ws = (char*)workspace;
        StreamElem* st = (StreamElem*)(ws + l.stream);
        float* bounds = (float*)(ws + l.bounds);
        float* partial = (float*)(ws + l.partial);
        const int T = m->tree_stream_len + 3, stride = l.qblocks * kTreeQueries;
        if (counts && hipMemsetAsync(partial, 0, (size_t)B * l.nsub * stride * sizeof(float), s) != hipSuccess) {
            tuch_set_error("tuch_winding_points: hipMemsetAsync failed");
            return TUCH_ERR_HIP;
        }
        launch_tree_boxes(m, verts, B, st, bounds, s);
        const int f0 = m->tree_frontier_off_host[l.frontier];
        hipLaunchKernelGGL(winding_tree_kernel<false>, dim3(B < 8 ? B : 8, l.nsub * l.qblocks, ceil_div(B, 8)), dim3(64), 0, s, points,
                           (const StreamElem*)st, (const TreeNode*)m->tree_node, (const float*)bounds, m->tree_nodes,
                           (const int32_t*)m->tree_frontier_nodes + f0,
                           (const int32_t*)m->tree_ancestors + (size_t)f0 * kMaxAncestors, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, counts, Q, T, l.nsub, B, partial, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(winding_finalize_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, s,
                           (const float*)partial, Q, stride, l.nsub, thresh, w, exterior);
        return tuch_check_launch("tuch_winding_points");
    }
    const int lpad = strip_lpad(m->strip_len);
    StreamElem* st = (StreamElem*)workspace;
    float* partial = (float*)((char*)workspace + align256((size_t)B * lpad * sizeof(StreamElem)));
    const int nsplit = choose_strip_splits(B, Q, lpad);
    if (counts && hipMemsetAsync(partial, 0, (size_t)B * nsplit * Q * sizeof(float), s) != hipSuccess) {
        tuch_set_error("tuch_winding_points: hipMemsetAsync failed");
        return TUCH_ERR_HIP;
    }
    hipLaunchKernelGGL(gather_stream_kernel, dim3(ceil_div(lpad, kBlock), B), dim3(kBlock), 0, s, verts,
                       (const int32_t*)m->strip_vidx, (const float*)m->strip_sign, m->V, m->strip_len, lpad, st);
    const int per_split = ceil_div(ceil_div(lpad - 6, nsplit), 3) * 3;
    hipLaunchKernelGGL(winding_strip_kernel, dim3(B, nsplit, ceil_div(Q, kStripQueries)), dim3(kStripBlock), 0, s,
                       points, (const StreamElem*)st, Q, lpad, per_split, counts, partial);
    hipLaunchKernelGGL(winding_finalize_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, s,
                       (const float*)partial, Q, Q, nsplit, thresh, w, exterior);
    return tuch_check_launch("tuch_winding_points");
}
