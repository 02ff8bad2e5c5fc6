// Inside test by signed ray crossings on gfx950 -- the exterior flags of tuch/smplify/losses.py:81-82 and
// tuch/train/loss.py:260-262,297 (`winding_numbers(...).le(0.99)`) without summing solid angles.
//
// For a point p off a closed oriented mesh M the winding number w(p) = 1/(4 pi) sum_f Omega_f(p) is an INTEGER:
// the signed number of crossings of any ray from p with M (+1 where the ray leaves through a face along its
// outward normal, -1 where it enters).  That covers the HD points of loss.py:297 (1 mm off the surface).
// For a VERTEX v of M the reference's sum skips the faces around v (their terms are atan2(0,0) = 0,
// contact.py:105), i.e. it is the solid angle of M' = M minus star(v), a surface with the one-ring of v as its
// boundary.  Close M' with a fan S = {(a, r_j, r_j+1)} over the ring with an apex a = v + delta u, a != v: then
//     sum_{f in M'} Omega_f(v) + sum_{g in S} Omega_g(v) = 4 pi N,    N = signed ray crossings of M' u S from v,
// so   w_ref(v) = N - 1/(4 pi) sum_{g in S} Omega_g(v):  an integer from ~1.5 k cheap crossing tests (no sqrt, no
// atan) plus valence(v) ~ 6 solid angles -- instead of ~2.5 k solid-angle steps of the cluster-tree walk
// (winding.hip) and ~14 k of the flat walk.  Both fan terms depend on the DIRECTION u only (the solid-angle
// formula is homogeneous of degree 0 in every corner vector; a ray from the origin hits the triangle (delta u, B, C)
// iff its direction lies in the cone spanned by u, B, C), so delta never appears.
//
// Crossings are counted along ONE fixed direction for all queries, in sheared coordinates
//     x' = x - kx z,  y' = y - ky z,  z' = z        (ray = +z', i.e. direction (kx, ky, 1) in space)
// where a triangle is hit iff the origin of the query-relative (x', y') plane lies inside its projection and the
// hit is in front.  Edge functions e(P,Q) = fl(Px Qy) - fl(Py Qx) are evaluated WITHOUT fma: they are exactly
// antisymmetric, so two triangles sharing an edge always agree on which side of it the ray passes, and the
// ties e == 0 are broken by the lexicographic order of the endpoints (the rasteriser's top-left rule): every ray
// is assigned to exactly one of the triangles around an edge or a vertex -- the count is the exact integer
// winding number of the float-coordinate mesh.  The walk uses the model's cluster tree (cluster_tree.hip): a
// node is entered only if the ray of some query of the wavefront can meet its 9-slab volume; far nodes cost
// nothing (no boundary caps are needed: a ray that misses a node's volume crosses none of its faces).
//
// The result equals thresholding the reference's float sum wherever that sum is not within its own rounding
// noise (~1e-5) of the threshold; tests/test_gpu_contact.py compares the two on every fixture.
#include "common.h"
#include "model.h"
#include "tree_device.h"
#include <stdlib.h>

#pragma clang fp contract(off)      // edge functions must stay mul, mul, sub (exact antisymmetry)

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr int kBlock = 256;
constexpr int kRayQueries = 64;               // one query per lane
constexpr float kShearX = 0.3217f, kShearY = 0.4331f;
// apex direction of the closing fan (any direction that is not +-ray and not in a ring face's plane)
constexpr float kFanX = 0.8191f, kFanY = 0.3467f, kFanZ = 0.4571f;
constexpr int kSlabs = 9;
constexpr int kSlabStride = 10;               // as winding.hip: node = [lo[10], hi[10]]

struct RayElem { float x, y, z, sign; };      // sheared position of a strip vertex, orientation of the triangle it closes

__device__ __forceinline__ float shear_x(float x, float z) { return __builtin_fmaf(-kShearX, z, x); }
__device__ __forceinline__ float shear_y(float y, float z) { return __builtin_fmaf(-kShearY, z, y); }

// ---- per call: sheared leaf strips + node slabs ------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ray_stream_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ vidx, const float* __restrict__ sign,
    int V, int L, int Lpad, RayElem* __restrict__ out)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= Lpad) return;
    RayElem e = {0.f, 0.f, 0.f, 0.f};
    if (p < L) {
        const float* c = verts + ((size_t)b * V + vidx[p]) * 3;
        e.x = shear_x(c[0], c[2]); e.y = shear_y(c[1], c[2]); e.z = c[2]; e.sign = sign[p];
    }
    out[(size_t)b * Lpad + p] = e;
}

__device__ __forceinline__ void slab_project(float x, float y, float z, float (&p)[kSlabs])
{
    p[0] = x; p[1] = y; p[2] = z;
    p[3] = x + y; p[4] = x - y; p[5] = x + z; p[6] = x - z; p[7] = y + z; p[8] = y - z;
}

// slabs of every leaf in sheared coordinates, one wave per (leaf, body); widened by a few ulps so that a ray
// decided by the float edge functions to pass on the leaf's side of a shared edge can never test as missing it
__global__ __launch_bounds__(kBoundsBlock) void ray_leaf_bounds_kernel(
    const RayElem* __restrict__ stream, int T, const TreeNode* __restrict__ nodes, int N,
    const int32_t* __restrict__ height_off, const int32_t* __restrict__ height_nodes, float* __restrict__ bounds)
{
    const int b = blockIdx.y;
    const RayElem* st = stream + (size_t)b * T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = height_off[0] + blockIdx.x * (kBoundsBlock / 64) + wave;
    if (i >= height_off[1]) return;
    const int node = height_nodes[i];
    const int off = nodes[node].ex_off, len = nodes[node].ex_len;
    float lo[kSlabs], hi[kSlabs];
#pragma unroll
    for (int k = 0; k < kSlabs; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; }
    for (int p = lane; p < len; p += 64) {
        const RayElem e = st[off + p];
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
        for (int k = 0; k < kSlabs; ++k) {
            const float pad = 4e-7f * fmaxf(fabsf(lo[k]), fabsf(hi[k])) + 1e-9f;
            o[k] = lo[k] - pad;
            o[kSlabStride + k] = hi[k] + pad;
        }
        o[kSlabs] = 0.0f;
        o[kSlabStride + kSlabs] = 0.0f;
    }
}

// ---- crossing test --------------------------------------------------------------------------------------------
struct P3 { float x, y, z; };

__device__ __forceinline__ float edge_fn(const P3& p, const P3& q) { return p.x * q.y - p.y * q.x; }   // no fma (pragma)

// tie-break for e(P,Q) == 0: is the origin "left of or on" the directed edge P -> Q?  Lexicographic order of the
// endpoints; exactly one of (P,Q), (Q,P) answers true.
__device__ __forceinline__ bool left_of(float e, const P3& p, const P3& q)
{
    return e > 0.0f || (e == 0.0f && (p.x < q.x || (p.x == q.x && p.y < q.y)));
}

// signed crossing of the +z ray from the origin with the triangle (a, b, c) given relative to the query;
// ea = e(b,c), eb = e(c,a), ec = e(a,b) (edge opposite each corner).  +1: the ray leaves through the front of
// (a,b,c) (det(a,b,c) > 0), -1: enters, 0: no hit in front.
__device__ __forceinline__ int crossing(const P3& a, const P3& b, const P3& c, float ea, float eb, float ec)
{
    const bool la = left_of(ea, b, c), lb = left_of(eb, c, a), lc = left_of(ec, a, b);
    const bool pos = la && lb && lc, neg = !la && !lb && !lc;
    if (!(pos || neg)) return 0;
    const float numz = ea * a.z + eb * b.z + ec * c.z;          // = det(a,b,c): sign of the crossing, and of the depth
    if (pos) return numz > 0.0f ? 1 : 0;
    return numz < 0.0f ? -1 : 0;
}

// careful form of the crossing test for lanes with an edge function that is exactly zero (the ray passes through an
// edge or a corner: tie rules) -- which includes every triangle that has the query itself as a corner
template <bool kSkipIncident>
__device__ __forceinline__ int crossing_with_ties(const P3 a, const P3 b, const P3 c, float ea, float eb, float ec)
{
    if (kSkipIncident) {
        // faces around the query vertex: the reference's atan2(0,0) = 0 terms; the closing fan stands in for them
        const bool za = (a.x == 0.0f) & (a.y == 0.0f) & (a.z == 0.0f);
        const bool zb = (b.x == 0.0f) & (b.y == 0.0f) & (b.z == 0.0f);
        const bool zc = (c.x == 0.0f) & (c.y == 0.0f) & (c.z == 0.0f);
        if (za | zb | zc) return 0;
    }
    return crossing(a, b, c, ea, eb, ec);
}

// One leaf strip, elements [off, off+len), len % 3 == 0, three readable elements past the end.  Register slots are
// rotated by position modulo 3 like the solid-angle walk; e[k] = edge function of the edge opposite slot k in
// stream order (p-2 -> p-1 -> p).  Fast path (most elements): two new edge functions, min3 / max3, one ballot.
// Elements whose projection holds the ray of some lane: depth test, +-1; lanes with an exact tie (and, for vertex
// queries, the faces around the query itself) take the careful form.
template <int A, bool kSkipIncident>
__device__ __forceinline__ void ray_step(const RayElem el, P3 (&s)[3], float (&e)[3], float qx, float qy, float qz, int& count)
{
    constexpr int Bq = (A + 1) % 3, Cq = (A + 2) % 3;            // slots of stream positions p-2 and p-1
    s[A].x = el.x - qx;
    s[A].y = el.y - qy;
    s[A].z = el.z - qz;
    // e[A] = e(Bq -> Cq) is carried over from the previous triangle; the two edges at the new vertex:
    e[Bq] = edge_fn(s[Cq], s[A]);
    e[Cq] = edge_fn(s[A], s[Bq]);
    // The origin may be inside the projection (ties included) iff all edge functions are >= 0 or all <= 0, i.e. iff
    // mn * mx >= 0: ONE compare + ballot branch per element.  (Measured on gfx950, tools/ubench/valu_rate2.hip: the
    // two-compare form `mn >= 0 | mx <= 0` costs ~22 cycles per element in compare -> mask -> scalar OR -> branch, the
    // product form ~11; a plain FP32 op 2.3.)  The element is wave-uniform (scalar loads), so skipping the two priming
    // vertices of a strip is a scalar branch.
    if (el.sign != 0.0f) {
        const float mn = __builtin_fminf(__builtin_fminf(e[0], e[1]), e[2]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(e[0], e[1]), e[2]);
        const bool cand = mn * mx >= 0.0f;
        if (__builtin_amdgcn_ballot_w64(cand)) {                  // a few hits per ray
            // triangle (Bq, Cq, A): det = sum of (edge function opposite a corner) x (that corner's depth)
            const float numz = e[Bq] * s[Bq].z + e[Cq] * s[Cq].z + e[A] * s[A].z;
            // generic position: all edge functions of one sign; the hit is in front iff det has that sign, and the
            // crossing is +1 (leaving through the front) for the positive orientation, -1 for the negative one
            int c = ((mn > 0.0f) & (numz > 0.0f)) - ((mx < 0.0f) & (numz < 0.0f));
            // exact ties.  The faces around a query vertex (two edge functions through the origin) always tie, but
            // their det is exactly 0 (the zero corner), so the generic form already counts them as 0: the careful
            // form is only needed where det != 0, i.e. for rays through an edge or a corner of some OTHER triangle
            const bool tie = cand & ((mn == 0.0f) | (mx == 0.0f)) & (numz != 0.0f);
            if (__builtin_amdgcn_ballot_w64(tie)) {
                if (tie) c = crossing_with_ties<kSkipIncident>(s[Bq], s[Cq], s[A], e[Bq], e[Cq], e[A]);
            }
            count += el.sign > 0.0f ? c : -c;
        }
    }
}

template <bool kSkipIncident>
__device__ __forceinline__ void ray_run(const RayElem* __restrict__ st, int off, int len, P3 (&s)[3], float (&e)[3],
                                        float qx, float qy, float qz, int& count)
{
    const RayElem* p = st + off;
    const RayElem* end = p + len;
    RayElem n0 = p[0], n1 = p[1], n2 = p[2];
    for (; p < end; p += 3) {
        const RayElem e0 = n0, e1 = n1, e2 = n2;
        n0 = p[3]; n1 = p[4]; n2 = p[5];
        ray_step<0, kSkipIncident>(e0, s, e, qx, qy, qz, count);
        ray_step<1, kSkipIncident>(e1, s, e, qx, qy, qz, count);
        ray_step<2, kSkipIncident>(e2, s, e, qx, qy, qz, count);
    }
}

// ---- which leaves can the rays of a 64-query block meet? -------------------------------------------------------------
// A ray can only cross faces of a leaf whose 9-slab volume it meets.  Slabs whose functional does not change along
// the ray (x, y, x+y, x-y in the sheared frame) bound the query from both sides, those that grow along it (z, x+z,
// y+z) only from above, those that shrink (x-z, y-z) only from below.  One wavefront per (block, body):
//   stage 1, lanes over LEAVES: the ranges of the block's 64 queries against every leaf's slabs (a superset);
//   stage 2, lanes over QUERIES: the surviving leaves tested query by query (wave-uniform leaf, scalar loads).
// Output: the strip ranges (ex_off, ex_len) of the leaves some ray of the block can meet, and how many.  No tree
// descent: with ~215 leaves the flat test is four rounds of 64 lanes, and nothing in it waits on a parent's verdict.
constexpr int kMaxChunks = 16;                // most wavefronts per query block in ray_strips_kernel (TUCH_RAY_CHUNKS, default 8)

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

template <bool kVerts>
__global__ __launch_bounds__(64) void ray_near_kernel(
    const float* __restrict__ pts, const TreeNode* __restrict__ nodes, const float* __restrict__ bounds, int N,
    const int32_t* __restrict__ leaf_nodes, int num_leaves, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ counts, int Q, int qblocks, int2* __restrict__ lists, int32_t* __restrict__ list_len)
{
    const int qb = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    int i0;
    if (kVerts) {
        i0 = qperm[qb * kRayQueries + lane];
    } else {
        const int n = counts ? counts[b] : Q;
        if (qb * kRayQueries >= n) {
            if (lane == 0) list_len[(size_t)b * qblocks + qb] = 0;
            return;
        }
        i0 = min(qb * kRayQueries + lane, n - 1);
    }
    const float* q3 = pts + ((size_t)b * Q + i0) * 3;
    const float qz = q3[2];
    const float qx = shear_x(q3[0], qz), qy = shear_y(q3[1], qz);
    const float q4 = qx + qy, q5 = qx - qy, q6 = qx + qz, q7 = qx - qz, q8 = qy + qz, q9 = qy - qz;
    // ranges of the block (wave-uniform after the reductions)
    const float bx0 = wave_min(qx), bx1 = wave_max(qx), by0 = wave_min(qy), by1 = wave_max(qy);
    const float b40 = wave_min(q4), b41 = wave_max(q4), b50 = wave_min(q5), b51 = wave_max(q5);
    const float bz0 = wave_min(qz), b60 = wave_min(q6), b71 = wave_max(q7), b80 = wave_min(q8), b91 = wave_max(q9);
    const float* bb = bounds + (size_t)b * N * (2 * kSlabStride);
    int2* list = lists + ((size_t)b * qblocks + qb) * num_leaves;
    int cnt = 0;
    for (int base = 0; base < num_leaves; base += 64) {
        const int leaf = base + lane;
        const int node = leaf_nodes[leaf < num_leaves ? leaf : num_leaves - 1];
        const float* lo = bb + (size_t)node * (2 * kSlabStride);
        const float* hi = lo + kSlabStride;
        bool pass = leaf < num_leaves;
        pass = pass && bx0 <= hi[0] && bx1 >= lo[0] && by0 <= hi[1] && by1 >= lo[1];
        pass = pass && b40 <= hi[3] && b41 >= lo[3] && b50 <= hi[4] && b51 >= lo[4];
        pass = pass && bz0 <= hi[2] && b60 <= hi[5] && b71 >= lo[6] && b80 <= hi[7] && b91 >= lo[8];
        unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int nd = __builtin_amdgcn_readlane(node, j);
            const float* l2 = bb + (size_t)nd * (2 * kSlabStride);
            const float* h2 = l2 + kSlabStride;
            float out = __builtin_fmaxf(l2[0] - qx, qx - h2[0]);
            out = __builtin_fmaxf(out, __builtin_fmaxf(l2[1] - qy, qy - h2[1]));
            out = __builtin_fmaxf(out, __builtin_fmaxf(l2[3] - q4, q4 - h2[3]));
            out = __builtin_fmaxf(out, __builtin_fmaxf(l2[4] - q5, q5 - h2[4]));
            out = __builtin_fmaxf(out, qz - h2[2]);
            out = __builtin_fmaxf(out, q6 - h2[5]);
            out = __builtin_fmaxf(out, l2[6] - q7);
            out = __builtin_fmaxf(out, q8 - h2[7]);
            out = __builtin_fmaxf(out, l2[8] - q9);
            if (__builtin_amdgcn_ballot_w64(!(out > 0.0f))) {
                const TreeNode t = nodes[nd];
                if (lane == 0) list[cnt] = make_int2(t.ex_off, t.ex_len);
                ++cnt;
            }
        }
    }
    if (lane == 0) list_len[(size_t)b * qblocks + qb] = cnt;
}

// Crossing counts: wavefront c of a query block walks the strips c, c + kChunks, ... of the block's list.  Queries:
// the model's vertices in tree order (qperm != nullptr; faces around the query are skipped) or arbitrary points
// [B,Q,3] in the caller's order (counts[b] of them real).  Grid (8, blocks x kChunks, B/8): workgroups go round-robin
// to the 8 XCDs, so XCD x works on body 8 z + x (its 0.2 MB sheared stream stays in that L2).
// kCount: elements walked are added to stats[0] (measurement).
template <bool kVerts, bool kCount>
__global__ __launch_bounds__(64) void ray_strips_kernel(
    const float* __restrict__ pts, const RayElem* __restrict__ stream, const int2* __restrict__ lists,
    const int32_t* __restrict__ list_len, int num_leaves, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ counts, int Q, int T, int qblocks, int num_bodies, int kChunks,
    int32_t* __restrict__ partial, unsigned long long* __restrict__ stats)
{
    const int b = blockIdx.z * gridDim.x + blockIdx.x;
    if (b >= num_bodies) return;
    const int qb = blockIdx.y / kChunks, c = blockIdx.y % kChunks;
    const int cnt = __builtin_amdgcn_readfirstlane(list_len[(size_t)b * qblocks + qb]);
    if (c >= cnt) return;
    int i0;
    if (kVerts) {
        i0 = qperm[qb * kRayQueries + threadIdx.x];
    } else {
        const int n = counts ? counts[b] : Q;
        i0 = min(qb * kRayQueries + (int)threadIdx.x, n - 1);
    }
    const float* q3 = pts + ((size_t)b * Q + i0) * 3;
    const float qz = q3[2];
    const float qx = shear_x(q3[0], qz), qy = shear_y(q3[1], qz);
    const RayElem* st = stream + (size_t)b * T;
    const int2* list = lists + ((size_t)b * qblocks + qb) * num_leaves;
    P3 s[3];
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k].x = s[k].y = s[k].z = 0.0f; e[k] = 0.0f; }
    int count = 0, walked = 0;
    for (int j = c; j < cnt; j += kChunks) {
        const int off = __builtin_amdgcn_readfirstlane(list[j].x), len = __builtin_amdgcn_readfirstlane(list[j].y);
        ray_run<kVerts>(st, off, len, s, e, qx, qy, qz, count);
        if (kCount) walked += len;
    }
    partial[((size_t)b * kChunks + c) * ((size_t)qblocks * kRayQueries) + qb * kRayQueries + threadIdx.x] = count;
    if (kCount && threadIdx.x == 0) atomicAdd(stats, (unsigned long long)walked);
}

// atan2 with a degree-8 minimax atan on [0,1] (max abs error 1e-7), octant fix-up; atan2(0,0) = 0
__device__ __forceinline__ float fast_atan2(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-37f);
    const float t = __builtin_fminf(ax, ay) / mx;
    const float s = t * t;
    float p = 0.0024567253421992064f;
    p = p * s + -0.014401361346244812f;
    p = p * s + 0.03978123143315315f;
    p = p * s + -0.07234857976436615f;
    p = p * s + 0.10498946160078049f;
    p = p * s + -0.14161229133605957f;
    p = p * s + 0.19985906779766083f;
    p = p * s + -0.33332598209381104f;
    p = p * s + 0.9999998807907104f;
    float v = p * t;
    v = ay > ax ? 1.57079632679489661923f - v : v;
    v = x < 0.0f ? kPi - v : v;
    return __builtin_copysignf(v, y);
}

// half solid angle atan2(num, den) of the triangle with corner vectors a, b, c (contact.py:79-105)
__device__ __forceinline__ float half_solid_angle(const P3& a, const P3& b, const P3& c)
{
    const float na = __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    const float nb = __builtin_sqrtf(b.x * b.x + b.y * b.y + b.z * b.z);
    const float nc = __builtin_sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
    const float cx = b.y * c.z - b.z * c.y, cy = b.z * c.x - b.x * c.z, cz = b.x * c.y - b.y * c.x;
    const float num = a.x * cx + a.y * cy + a.z * cz;
    const float dab = a.x * b.x + a.y * b.y + a.z * b.z;
    const float dbc = b.x * c.x + b.y * c.y + b.z * c.z;
    const float dac = a.x * c.x + a.y * c.y + a.z * c.z;
    const float den = na * nb * nc + dab * nc + dac * nb + dbc * na;
    return fast_atan2(num, den);
}

// vertices: N = sum of the subtree counts + crossings of the closing fan; w = N - (sum of the fan's half angles) / (2 pi)
__global__ __launch_bounds__(kBlock) void ray_finalize_verts_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ partial, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ ring_off, const int32_t* __restrict__ ring_vidx, const int32_t* __restrict__ list_len,
    int V, int stride, int kChunks, float thresh, float* __restrict__ w_out, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;     // position in tree order
    if (i >= V) return;
    const int v = qperm[i];
    int n = 0;
    const int chunks = min(kChunks, list_len[(size_t)b * (stride / kRayQueries) + i / kRayQueries]);
    for (int sp = 0; sp < chunks; ++sp) n += partial[((size_t)b * kChunks + sp) * stride + i];
    const float* vb = verts + (size_t)b * V * 3;
    const float vx = vb[3 * v], vy = vb[3 * v + 1], vz = vb[3 * v + 2];
    const float qx = shear_x(vx, vz), qy = shear_y(vy, vz);
    // the apex direction of the fan, in space and sheared (ray frame)
    const P3 u = {kFanX, kFanY, kFanZ};
    const P3 us = {shear_x(kFanX, kFanZ), shear_y(kFanY, kFanZ), kFanZ};
    const int lo = ring_off[v], cnt = ring_off[v + 1] - lo;
    float half_sum = 0.0f;
    // previous ring vertex (j = cnt-1) to start the cycle
    int r = ring_vidx[lo + cnt - 1];
    P3 pb = {vb[3 * r] - vx, vb[3 * r + 1] - vy, vb[3 * r + 2] - vz};                                 // space, for the angle
    P3 sb = {shear_x(vb[3 * r], vb[3 * r + 2]) - qx, shear_y(vb[3 * r + 1], vb[3 * r + 2]) - qy, vb[3 * r + 2] - vz};
    for (int j = 0; j < cnt; ++j) {
        r = ring_vidx[lo + j];
        const P3 pc = {vb[3 * r] - vx, vb[3 * r + 1] - vy, vb[3 * r + 2] - vz};
        const P3 sc = {shear_x(vb[3 * r], vb[3 * r + 2]) - qx, shear_y(vb[3 * r + 1], vb[3 * r + 2]) - qy, vb[3 * r + 2] - vz};
        // fan triangle (u, previous, current) replaces the face (v, previous, current)
        half_sum += half_solid_angle(u, pb, pc);
        n += crossing(us, sb, sc, edge_fn(sb, sc), edge_fn(sc, us), edge_fn(us, sb));
        pb = pc;
        sb = sc;
    }
    const float w = (float)n - half_sum * (0.5f / kPi);
    const size_t o = (size_t)b * V + v;
    if (w_out) w_out[o] = w;
    if (exterior) exterior[o] = w <= thresh;
}

__global__ __launch_bounds__(kBlock) void ray_finalize_points_kernel(
    const int32_t* __restrict__ partial, const int32_t* __restrict__ counts, const int32_t* __restrict__ list_len,
    int Q, int stride, int kChunks, float thresh, float* __restrict__ w_out, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= Q) return;
    int n = 0;
    if (!counts || i < counts[b]) {
        const int chunks = min(kChunks, list_len[(size_t)b * (stride / kRayQueries) + i / kRayQueries]);
        for (int sp = 0; sp < chunks; ++sp) n += partial[((size_t)b * kChunks + sp) * stride + i];
    }
    const float w = (float)n;
    const size_t o = (size_t)b * Q + i;
    if (w_out) w_out[o] = w;
    if (exterior) exterior[o] = w <= thresh;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RayLayout { size_t stream, bounds, partial, lists, list_len, stats, total; int T, qblocks; };

}  // namespace

bool tuch_ray_available(const tuch_contact_model* m)
{
    if (!m || m->tree_nodes <= 0 || !m->ring_off || m->tree_exact_len <= 0) return false;
    const char* e = getenv("TUCH_WINDING_RAY");
    return !e || atoi(e) != 0;
}

static int ray_chunks()
{
    const char* e = getenv("TUCH_RAY_CHUNKS");
    const int c = e ? atoi(e) : 8;
    return c < 1 ? 1 : (c > kMaxChunks ? kMaxChunks : c);
}

static RayLayout full_layout(const tuch_contact_model* m, int B, int Q, bool verts)
{
    RayLayout l;
    l.qblocks = verts ? 2 * m->tree_qblocks : ceil_div(Q, kRayQueries);
    // leaf strips are the first part of the tree stream; the caps behind them are never read
    l.T = ceil_div(m->tree_exact_len, 3) * 3 + 6;
    size_t o = 0;
    l.stream = o;   o += align256((size_t)B * l.T * sizeof(RayElem));
    l.bounds = o;   o += align256((size_t)B * m->tree_nodes * 2 * kSlabStride * sizeof(float));
    l.partial = o;  o += align256((size_t)B * kMaxChunks * l.qblocks * kRayQueries * sizeof(int32_t));
    l.lists = o;    o += align256((size_t)B * l.qblocks * m->tree_leaves * sizeof(int2));
    l.list_len = o; o += align256((size_t)B * l.qblocks * sizeof(int32_t));
    l.stats = o;    o += 256;
    l.total = o;
    return l;
}

size_t tuch_ray_workspace_bytes(const tuch_contact_model* m, int B, int Q)
{
    if (!m || m->tree_nodes <= 0 || B <= 0) return 0;
    const size_t a = full_layout(m, B, m->V, true).total;
    const size_t b = Q > 0 ? full_layout(m, B, Q, false).total : 0;
    return a > b ? a : b;
}

// sheared leaf strips and the slabs of every LEAF (inner nodes are not used by the flat near test)
static void launch_ray_boxes(const tuch_contact_model* m, const RayLayout& l, const float* verts, int B, char* ws, hipStream_t s)
{
    RayElem* st = (RayElem*)(ws + l.stream);
    float* bounds = (float*)(ws + l.bounds);
    hipLaunchKernelGGL(ray_stream_kernel, dim3(ceil_div(l.T, kBlock), B), dim3(kBlock), 0, s, verts,
                       (const int32_t*)m->tree_vidx, (const float*)m->tree_sign, m->V, m->tree_exact_len, l.T, st);
    hipLaunchKernelGGL(ray_leaf_bounds_kernel, dim3(ceil_div(m->tree_leaves, kBoundsBlock / 64), B), dim3(kBoundsBlock), 0, s,
                       (const RayElem*)st, l.T, (const TreeNode*)m->tree_node, m->tree_nodes,
                       (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, bounds);
}

int tuch_ray_exterior_verts(const tuch_contact_model* m, const float* verts, int B, float thresh, uint8_t* exterior,
                            float* w, void* workspace, hipStream_t s, unsigned long long* stats_host)
{
    const RayLayout l = full_layout(m, B, m->V, true);
    char* ws = (char*)workspace;
    launch_ray_boxes(m, l, verts, B, ws, s);
    int32_t* partial = (int32_t*)(ws + l.partial);
    int2* lists = (int2*)(ws + l.lists);
    int32_t* list_len = (int32_t*)(ws + l.list_len);
    unsigned long long* stats = (unsigned long long*)(ws + l.stats);
    // the leaves are the height-0 entries of the tree's height table (tree_height_off_host[0] == 0)
    const int32_t* leaf_nodes = (const int32_t*)m->tree_height_nodes;
    hipLaunchKernelGGL(ray_near_kernel<true>, dim3(l.qblocks, B), dim3(64), 0, s, verts, (const TreeNode*)m->tree_node,
                       (const float*)(ws + l.bounds), m->tree_nodes, leaf_nodes, m->tree_leaves, (const int32_t*)m->tree_qperm,
                       (const int32_t*)nullptr, m->V, l.qblocks, lists, list_len);
    const int kChunks = ray_chunks();
    const dim3 grid(B < 8 ? B : 8, l.qblocks * kChunks, ceil_div(B, 8));
    if (stats_host) {
        if (hipMemsetAsync(stats, 0, sizeof(unsigned long long), s) != hipSuccess) return TUCH_ERR_HIP;
        hipLaunchKernelGGL((ray_strips_kernel<true, true>), grid, dim3(64), 0, s, verts, (const RayElem*)(ws + l.stream),
                           (const int2*)lists, (const int32_t*)list_len, m->tree_leaves, (const int32_t*)m->tree_qperm,
                           (const int32_t*)nullptr, m->V, l.T, l.qblocks, B, kChunks, partial, stats);
    } else {
        hipLaunchKernelGGL((ray_strips_kernel<true, false>), grid, dim3(64), 0, s, verts, (const RayElem*)(ws + l.stream),
                           (const int2*)lists, (const int32_t*)list_len, m->tree_leaves, (const int32_t*)m->tree_qperm,
                           (const int32_t*)nullptr, m->V, l.T, l.qblocks, B, kChunks, partial, stats);
    }
    if (w || exterior)
        hipLaunchKernelGGL(ray_finalize_verts_kernel, dim3(ceil_div(m->V, kBlock), B), dim3(kBlock), 0, s, verts,
                           (const int32_t*)partial, (const int32_t*)m->tree_qperm, (const int32_t*)m->ring_off,
                           (const int32_t*)m->ring_vidx, (const int32_t*)list_len, m->V, l.qblocks * kRayQueries, kChunks, thresh, w,
                           exterior);
    if (stats_host) {
        if (hipMemcpyAsync(stats_host, stats, sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return TUCH_ERR_HIP;
        stats_host[1] = (unsigned long long)B * l.qblocks * kChunks;
    }
    return tuch_check_launch("tuch_ray_exterior_verts");
}

int tuch_ray_exterior_points(const tuch_contact_model* m, const float* verts, const float* points, const int32_t* counts,
                             int B, int Q, float thresh, uint8_t* exterior, float* w, void* workspace, hipStream_t s)
{
    const RayLayout l = full_layout(m, B, Q, false);
    char* ws = (char*)workspace;
    int32_t* partial = (int32_t*)(ws + l.partial);
    int2* lists = (int2*)(ws + l.lists);
    int32_t* list_len = (int32_t*)(ws + l.list_len);
    launch_ray_boxes(m, l, verts, B, ws, s);
    hipLaunchKernelGGL(ray_near_kernel<false>, dim3(l.qblocks, B), dim3(64), 0, s, points, (const TreeNode*)m->tree_node,
                       (const float*)(ws + l.bounds), m->tree_nodes, (const int32_t*)m->tree_height_nodes, m->tree_leaves,
                       (const int32_t*)nullptr, counts, Q, l.qblocks, lists, list_len);
    const int kChunks = ray_chunks();
    hipLaunchKernelGGL((ray_strips_kernel<false, false>), dim3(B < 8 ? B : 8, l.qblocks * kChunks, ceil_div(B, 8)), dim3(64), 0, s,
                       points, (const RayElem*)(ws + l.stream), (const int2*)lists, (const int32_t*)list_len, m->tree_leaves,
                       (const int32_t*)nullptr, counts, Q, l.T, l.qblocks, B, kChunks, partial, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(ray_finalize_points_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, s,
                       (const int32_t*)partial, counts, (const int32_t*)list_len, Q, l.qblocks * kRayQueries, kChunks, thresh, w, exterior);
    return tuch_check_launch("tuch_ray_exterior_points");
}
