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
#include "workspace.h"
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
constexpr int kNearSlabs = 13;                // of the 18 slab values of a leaf, those the near test uses

struct RayElem { float x, y, z, sign; };      // sheared position of a strip vertex; `sign` holds the model's word for the
                                              // element (tree_sign_word): bits 0-23 the orientation of the triangle it
                                              // closes as the INTEGER +1 / -1 / 0, bits 24-31 the segments that list it

__device__ __forceinline__ float shear_x(float x, float z) { return __builtin_fmaf(-kShearX, z, x); }
__device__ __forceinline__ float shear_y(float y, float z) { return __builtin_fmaf(-kShearY, z, y); }

// ---- per call: sheared leaf strips + node slabs ------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ray_stream_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ vidx, const float* __restrict__ sign,
    int V, int L, int Lpad, RayElem* __restrict__ out, uint4* __restrict__ zeroed, size_t zeroed_n)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    // also clears the counters of the kernels that follow (a separate 6 MB fill in front of the chain was 12 us)
    for (size_t g = ((size_t)b * gridDim.x + blockIdx.x) * kBlock + threadIdx.x; g < zeroed_n;
         g += (size_t)gridDim.x * gridDim.y * kBlock)
        zeroed[g] = make_uint4(0u, 0u, 0u, 0u);
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

// minimum over the 16 lanes of a DPP row, left in every lane: quad swaps, then the two mirror controls -- single VALU
// instructions, no LDS crossbar (a 64-lane shuffle reduction of the 18 slab values was ~200 instructions per leaf)
template <int kCtrl>
__device__ __forceinline__ float dpp_move(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), kCtrl, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_min(float v)
{
    v = fminf(v, dpp_move<0xB1>(v));        // quad_perm [1,0,3,2]
    v = fminf(v, dpp_move<0x4E>(v));        // quad_perm [2,3,0,1]
    v = fminf(v, dpp_move<0x141>(v));       // row_half_mirror
    v = fminf(v, dpp_move<0x140>(v));       // row_mirror
    return v;
}

// slabs of every leaf in sheared coordinates, 16 lanes per (leaf, body): a leaf strip is a few dozen elements, so a
// whole wavefront per leaf was mostly idle lanes behind four dependent rounds of loads; widened by a few ulps so that
// a ray decided by the float edge functions to pass on the leaf's side of a shared edge can never test as missing it
// kPose: the kernel also WRITES the sheared strip (each 16-lane group poses the run of its own leaf; the leaves' runs
// tile the stream, checked at model creation) and clears the counters of the kernels that follow -- ray_stream_kernel's
// work, without a launch of its own ahead of this one on the step's serial chain.
// bits of the largest half-precision number <= x / the smallest >= x (|x| beyond the half range saturates outwards)
__device__ __forceinline__ uint32_t half_below(float x)
{
    const _Float16 h = (_Float16)x;
    uint32_t u = __builtin_bit_cast(uint16_t, h);
    if ((float)h > x) u = (u & 0x8000u) ? u + 1 : (u == 0 ? 0x8001u : u - 1);
    return u;
}
__device__ __forceinline__ uint32_t half_above(float x)
{
    const _Float16 h = (_Float16)x;
    uint32_t u = __builtin_bit_cast(uint16_t, h);
    if ((float)h < x) u = (u & 0x8000u) ? (u == 0x8000u ? 0x0001u : u - 1) : u + 1;
    return u;
}
// where the tables of one call's `bounds` region start (floats): [B][13][L] | records [B][L][8 dwords] | centres [B][4]
__device__ __host__ __forceinline__ size_t near_records_at(int B, int L) { return (size_t)B * kNearSlabs * L; }
__device__ __host__ __forceinline__ size_t near_centres_at(int B, int L) { return near_records_at(B, L) + (size_t)B * L * 8; }

// ---- the query side of the one-launch crossing kernel (round 6, ray_cross_kernel below) -------------------------------
// QRec: a vertex as a ray origin, in tree order (slot = position): its sheared coordinates -- the SAME bits as the strip
// element of that vertex, shear_x / shear_y of the same three floats: the faces around a query are recognised by exact
// zeros -- and tag = slot | (segments of the vertex) << 24; slot field 0xffffff: a padding lane of the last block.
// ranges: per block of 64 slots the 13 extreme values the leaves' records are tested against (ray_near_kernel's block
// ranges), relative to the body's reference point like the records.
struct QRec { float x, y, z; int32_t tag; };
constexpr int kNoSlot = 0xffffff;
struct RayPre { int blocks, qblocks; QRec* qrec; float4* ranges; const int32_t* vmask; };

__device__ __forceinline__ void ray_block_prepass(const float* __restrict__ vb, int first_vertex,
                                                  const int32_t* __restrict__ qperm, const int32_t* __restrict__ vmask,
                                                  int Q, int qb, int lane, QRec* __restrict__ qrec, float4* __restrict__ ranges)
{
    const int slot = qb * kRayQueries + lane;
    const float* q3 = vb + 3 * (size_t)qperm[slot];          // (the table is padded: lanes behind Q repeat a vertex)
    const float* c0 = vb + 3 * (size_t)first_vertex;         // the body's reference point: the first element of its strip
    const float qz = q3[2], qx = shear_x(q3[0], qz), qy = shear_y(q3[1], qz);
    const float cz = c0[2], cx = shear_x(c0[0], cz), cy = shear_y(c0[1], cz);
    QRec r;
    r.x = qx; r.y = qy; r.z = qz;
    r.tag = slot < Q ? (slot | (vmask ? vmask[slot] << 24 : 0)) : kNoSlot;
    qrec[slot] = r;
    const float rx = qx - cx, ry = qy - cy, rz = qz - cz;
    const float r4 = rx + ry, r5 = rx - ry, r6 = rx + rz, r7 = rx - rz, r8 = ry + rz, r9 = ry - rz;
    const float bx0 = wave_min_uniform(rx), bx1 = wave_max_uniform(rx), by0 = wave_min_uniform(ry), by1 = wave_max_uniform(ry);
    const float b40 = wave_min_uniform(r4), b41 = wave_max_uniform(r4), b50 = wave_min_uniform(r5), b51 = wave_max_uniform(r5);
    const float bz0 = wave_min_uniform(rz), b60 = wave_min_uniform(r6), b71 = wave_max_uniform(r7);
    const float b80 = wave_min_uniform(r8), b91 = wave_max_uniform(r9);
    if (lane == 0) {
        ranges[0] = make_float4(bx0, bx1, by0, by1);
        ranges[1] = make_float4(b40, b41, b50, b51);
        ranges[2] = make_float4(bz0, b60, b71, b80);
        ranges[3] = make_float4(b91, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ float fan_terms(const float* __restrict__ vb, int v, bool real, int& n,
                                           const int32_t* __restrict__ ring_off, const int32_t* __restrict__ ring_vidx);
// kPose, leaf_blocks > 0: the workgroups behind the first leaf_blocks compute the closing fans of 256 vertices each
// (fan_terms: they need the vertices only).  This launch is the first of the inside test's chain, with a handful of
// workgroups per body on an otherwise idle chip; inside ray_finalize_verts_kernel, behind the crossing kernel, the fans
// were 15 us of the step's critical chain, as workgroups of ray_near_kernel's launch they stretched that launch from 53
// to 80 us (its own workgroups run beside the search, whose wavefronts hold the vector units).
template <bool kPose>
__global__ __launch_bounds__(kBoundsBlock) void ray_leaf_bounds_kernel(
    const RayElem* __restrict__ stream, int T, const TreeNode* __restrict__ nodes, int N,
    const int32_t* __restrict__ height_off, const int32_t* __restrict__ height_nodes, float* __restrict__ bounds,
    const float* __restrict__ verts, const int32_t* __restrict__ vidx, const float* __restrict__ sign, int V, int Lexact,
    RayElem* __restrict__ stream_out, uint4* __restrict__ zeroed, size_t zeroed_n,
    int leaf_blocks = 0, const int32_t* __restrict__ qperm = nullptr, const int32_t* __restrict__ ring_off = nullptr,
    const int32_t* __restrict__ ring_vidx = nullptr, float2* __restrict__ fans = nullptr, int fan_stride = 0,
    RayPre pre = RayPre{0, 0, nullptr, nullptr, nullptr})
{
    if (kPose && pre.blocks > 0 && (int)blockIdx.x >= (int)gridDim.x - pre.blocks) {
        // the query side of the one-launch crossing kernel (ray_cross_kernel): four query blocks per workgroup
        for (size_t g = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kBoundsBlock + threadIdx.x; g < zeroed_n;
             g += (size_t)gridDim.x * gridDim.y * kBoundsBlock)
            zeroed[g] = make_uint4(0u, 0u, 0u, 0u);
        const int qb = ((int)blockIdx.x - ((int)gridDim.x - pre.blocks)) * (kBoundsBlock / 64) + (int)(threadIdx.x >> 6);
        if (qb < pre.qblocks)
            ray_block_prepass(verts + (size_t)blockIdx.y * V * 3, vidx[0], qperm, pre.vmask, V, qb, threadIdx.x & 63,
                              pre.qrec + (size_t)blockIdx.y * pre.qblocks * kRayQueries,
                              pre.ranges + ((size_t)blockIdx.y * pre.qblocks + qb) * 4);
        return;
    }
    if (kPose && leaf_blocks > 0 && (int)blockIdx.x >= leaf_blocks) {
        for (size_t g = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kBoundsBlock + threadIdx.x; g < zeroed_n;
             g += (size_t)gridDim.x * gridDim.y * kBoundsBlock)
            zeroed[g] = make_uint4(0u, 0u, 0u, 0u);
        const int i = ((int)blockIdx.x - leaf_blocks) * kBoundsBlock + (int)threadIdx.x;        // position in tree order
        const bool real = i < V;
        const int v = qperm[real ? i : V - 1];
        int cr = 0;
        const float half = fan_terms(verts + (size_t)blockIdx.y * V * 3, v, real, cr, ring_off, ring_vidx);
        if (i < fan_stride) fans[(size_t)blockIdx.y * fan_stride + i] = make_float2(__int_as_float(cr), half);
        return;
    }
    const int b = blockIdx.y;
    const RayElem* st = stream + (size_t)b * T;
    const int group = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int i = height_off[0] + blockIdx.x * (kBoundsBlock / 16) + group;
    const bool real = i < height_off[1];                // all lanes stay: the DPP rows need them
    const int node = height_nodes[real ? i : height_off[0]];
    const int off = nodes[node].ex_off, len = real ? nodes[node].ex_len : 0;
    if (kPose) {
        for (size_t g = ((size_t)b * gridDim.x + blockIdx.x) * kBoundsBlock + threadIdx.x; g < zeroed_n;
             g += (size_t)gridDim.x * gridDim.y * kBoundsBlock)
            zeroed[g] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0)                            // the padding behind the last run (three readable elements past the end)
            for (int p = Lexact + (int)threadIdx.x; p < T; p += kBoundsBlock) stream_out[(size_t)b * T + p] = RayElem{0.f, 0.f, 0.f, 0.f};
    }
    float lo[kSlabs], nhi[kSlabs];                      // minima of the projections and of their negatives
#pragma unroll
    for (int k = 0; k < kSlabs; ++k) { lo[k] = 3.0e38f; nhi[k] = 3.0e38f; }
    for (int p = sub; p < len; p += 16) {
        RayElem e;
        if (kPose) {
            const float* c = verts + ((size_t)b * V + vidx[off + p]) * 3;
            e.x = shear_x(c[0], c[2]); e.y = shear_y(c[1], c[2]); e.z = c[2]; e.sign = sign[off + p];
            stream_out[(size_t)b * T + off + p] = e;
        } else {
            e = st[off + p];
        }
        float pr[kSlabs];
        slab_project(e.x, e.y, e.z, pr);
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) { lo[k] = fminf(lo[k], pr[k]); nhi[k] = fminf(nhi[k], -pr[k]); }
    }
#pragma unroll
    for (int k = 0; k < kSlabs; ++k) { lo[k] = row_min(lo[k]); nhi[k] = row_min(nhi[k]); }
    if (real && sub == 0) {
        // the 13 slab values the near test uses, padded for the roundings of the projections (round 2-3: also stored in
        // single precision, one array per value over the leaves; since round 4 ray_near_kernel reads the records alone)
        const int L = height_off[1] - height_off[0], leaf = i - height_off[0];
        float lo_p[kSlabs], hi_p[kSlabs];
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) {
            const float hi = -nhi[k];
            const float pad = 4e-7f * fmaxf(fabsf(lo[k]), fabsf(hi)) + 1e-9f;
            lo_p[k] = lo[k] - pad;
            hi_p[k] = hi + pad;
        }
        // ... as the 32-byte RECORD both stages of the near test read: half precision, relative to a
        // reference point of the body (the first element of its strip, so that the numbers stay small wherever the body
        // stands), lower bounds rounded down and upper bounds up after a pad for the roundings of the subtraction.
        float c[3];
        if (kPose) {
            const float* c0 = verts + ((size_t)b * V + vidx[0]) * 3;
            c[0] = shear_x(c0[0], c0[2]); c[1] = shear_y(c0[1], c0[2]); c[2] = c0[2];
        } else {
            const RayElem e0 = st[0];
            c[0] = e0.x; c[1] = e0.y; c[2] = e0.z;
        }
        float pc[kSlabs];
        slab_project(c[0], c[1], c[2], pc);
        uint32_t lo_h[kSlabs], hi_h[kSlabs];
#pragma unroll
        for (int k = 0; k < kSlabs; ++k) {
            const float pad = 4e-6f * (fabsf(lo_p[k]) + fabsf(hi_p[k]) + fabsf(pc[k])) + 1e-7f;
            lo_h[k] = half_below(lo_p[k] - pc[k] - pad);
            hi_h[k] = half_above(hi_p[k] - pc[k] + pad);
        }
        uint4* r = reinterpret_cast<uint4*>(bounds + near_records_at(gridDim.y, L)) + ((size_t)b * L + leaf) * 2;
        r[0] = make_uint4(lo_h[0] | lo_h[1] << 16, lo_h[3] | lo_h[4] << 16, lo_h[6] | lo_h[8] << 16, hi_h[0] | hi_h[1] << 16);
        r[1] = make_uint4(hi_h[2] | hi_h[3] << 16, hi_h[4] | hi_h[5] << 16, hi_h[7] | (uint32_t)leaf << 16, (uint32_t)node);  // + who it is
        if (leaf == 0) *reinterpret_cast<float4*>(bounds + near_centres_at(gridDim.y, L) + (size_t)b * 4) = make_float4(c[0], c[1], c[2], 0.f);
    }
}

// the query side alone (meshes whose leaf runs do not tile the strip: the boxes are then two launches of their own)
__global__ __launch_bounds__(kBoundsBlock) void ray_prepass_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ vidx, const int32_t* __restrict__ qperm, int V, RayPre pre)
{
    const int qb = (int)blockIdx.x * (kBoundsBlock / 64) + (int)(threadIdx.x >> 6);
    if (qb < pre.qblocks)
        ray_block_prepass(verts + (size_t)blockIdx.y * V * 3, vidx[0], qperm, pre.vmask, V, qb, threadIdx.x & 63,
                          pre.qrec + (size_t)blockIdx.y * pre.qblocks * kRayQueries,
                          pre.ranges + ((size_t)blockIdx.y * pre.qblocks + qb) * 4);
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

// the same through the branch-free generic form, the careful one only for lanes with an exact tie (wave-uniform test)
__device__ __forceinline__ int crossing_mostly_generic(const P3& a, const P3& b, const P3& c, float ea, float eb, float ec)
{
    const float numz = ea * a.z + eb * b.z + ec * c.z;
    const float mn = __builtin_fminf(__builtin_fminf(ea, eb), ec), mx = __builtin_fmaxf(__builtin_fmaxf(ea, eb), ec);
    int n = (int)(__builtin_fminf(mn, numz) > 0.0f) - (int)(__builtin_fmaxf(mx, numz) < 0.0f);
    const bool tie = (mn * mx == 0.0f) & (numz != 0.0f);
    if (__builtin_amdgcn_ballot_w64(tie)) {
        if (tie) n = crossing(a, b, c, ea, eb, ec);
    }
    return n;
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
// stream order (p-2 -> p-1 -> p).  Per element: two new edge functions, the depth determinant, and a branch-free
// +-1 (the rays of a wavefront all pass the leaf's slabs, so most elements are hit by SOME lane: a "does any lane
// hit" branch would almost always be taken).  Lanes with an exact tie take the careful form.
// kSeg: `em` = the segments that list the element's triangle as a face (bit s, s < 8); qmask = the segments the lane's
// query vertex belongs to: the crossing also goes into per-segment counts (the segment test of winding.hip needs, for
// every segment vertex, its crossings with the faces of its own segments only).  Eight counters as biased bytes of two
// words: bits 0-3 of (em & qmask) are spread to the byte positions by one multiplication.
constexpr uint32_t kSegBias = 0x80808080u;
// a strip vertex relative to the query, (x, y) as a register pair: the two products of an edge function are ONE packed
// multiplication with crossed halves (v_pk_mul_f32 rounds each product like v_mul_f32: the same bits, still exactly
// antisymmetric), and the subtraction of the query is one packed add
struct S3 { v2f xy; float z; };
__device__ __forceinline__ float edge_fn(const S3& p, const S3& q)
{
    v2f r;                                                        // (p.x q.y, p.y q.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(p.xy), "v"(q.xy));
    float d;                                                      // (in assembly: left to the vectoriser, the difference is
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(r.x), "v"(r.y));    // paired with the z subtraction behind three v_mov)
    return d;
}
__device__ __forceinline__ P3 p3(const S3& a) { return P3{a.xy.x, a.xy.y, a.z}; }

template <int A, bool kSkipIncident, bool kSeg>
__device__ __forceinline__ void ray_step(const RayElem el, S3 (&s)[3], float (&e)[3], v2f qxy, float qz, int& count,
                                         int qmask, uint32_t& pa, uint32_t& pb)
{
    constexpr int Bq = (A + 1) % 3, Cq = (A + 2) % 3;            // slots of stream positions p-2 and p-1
    s[A].xy = (v2f){el.x, el.y} - qxy;
    s[A].z = el.z - qz;
    // e[A] = e(Bq -> Cq) is carried over from the previous triangle; the two edges at the new vertex:
    e[Bq] = edge_fn(s[Cq], s[A]);
    e[Cq] = edge_fn(s[A], s[Bq]);
    // the element's fourth word: orientation (integer, bits 0-23) and segments (bits 24-31), see RayElem.  gfx950 has no
    // scalar float compare -- `sign > 0` and `sign != 0` on the wave-uniform element cost a vector compare each
    const int word = __float_as_int(el.sign);
    if ((word & 0xffffff) != 0) {                                 // wave-uniform: the two priming vertices of a strip
        // triangle (Bq, Cq, A): det = sum of (edge function opposite a corner) x (that corner's depth)
        const float numz = __builtin_fmaf(e[A], s[A].z, __builtin_fmaf(e[Cq], s[Cq].z, e[Bq] * s[Bq].z));
        const float mn = __builtin_fminf(__builtin_fminf(e[0], e[1]), e[2]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(e[0], e[1]), e[2]);
        // generic position: the origin is inside the projection iff the edge functions have one sign, the hit is in
        // front iff det has that sign too; +1 (leaving through the front) for the positive orientation, -1 for the
        // negative one.  The faces around a query vertex (two edge functions through the origin, det exactly 0)
        // count 0 here, as they must.
        int c = (int)(__builtin_fminf(mn, numz) > 0.0f) - (int)(__builtin_fmaxf(mx, numz) < 0.0f);
        // exact ties: some edge function is zero (mn * mx == 0: all are >= 0 or all <= 0 then) and det is not, i.e. the
        // ray passes through an edge or a corner of a triangle that does not contain the query
        const bool edge_zero = mn * mx == 0.0f;
        if (__builtin_amdgcn_ballot_w64(edge_zero)) {             // the wavefronts of a query's own leaf; else rare
            const bool tie = edge_zero & (numz != 0.0f);
            if (__builtin_amdgcn_ballot_w64(tie)) {
                if (tie) c = crossing_with_ties<kSkipIncident>(p3(s[Bq]), p3(s[Cq]), p3(s[A]), e[Bq], e[Cq], e[A]);
            }
        }
        // one full-rate instruction (left to the compiler, c * sign + count becomes a 64-bit multiply-add: quarter rate);
        // the 24-bit multiply reads bits 0-23 of the word: the orientation
        asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(count) : "v"(c), "s"(word));
        if (kSeg && ((uint32_t)word >> 24) != 0) {                // wave-uniform
            // few lanes belong to one of the element's segments AND cross it: the counters' arithmetic (32-bit integer
            // multiplications: quarter rate) only when some lane has something to add
            const uint32_t t = ((uint32_t)word >> 24) & (uint32_t)qmask;
            if (__builtin_amdgcn_ballot_w64((t != 0) & (c != 0))) {
                uint32_t cs;
                asm("v_mul_i32_i24 %0, %1, %2" : "=v"(cs) : "v"(c), "s"(word));
                pa += cs * (((t & 15u) * 0x00204081u) & 0x01010101u);
                pb += cs * ((((t >> 4) & 15u) * 0x00204081u) & 0x01010101u);
            }
        }
    }
}

template <bool kSkipIncident, bool kSeg>
__device__ __forceinline__ void ray_run(const RayElem* __restrict__ st, int off, int len,
                                        S3 (&s)[3], float (&e)[3], v2f qxy, float qz, int& count,
                                        int qmask, uint32_t& pa, uint32_t& pb)
{
    const RayElem* p = st + off;
    RayElem n0 = p[0], n1 = p[1], n2 = p[2];
    // (a 32-bit count, opaque to the loop optimiser: compared on the scalar unit; the pointer form `p < end` is a 64-bit
    // VECTOR compare on gfx950)
    for (int left = len; left > 0; left -= 3, p += 3) {
        asm("" : "+s"(left));
        const RayElem e0 = n0, e1 = n1, e2 = n2;
        n0 = p[3]; n1 = p[4]; n2 = p[5];
        ray_step<0, kSkipIncident, kSeg>(e0, s, e, qxy, qz, count, qmask, pa, pb);
        ray_step<1, kSkipIncident, kSeg>(e1, s, e, qxy, qz, count, qmask, pa, pb);
        ray_step<2, kSkipIncident, kSeg>(e2, s, e, qxy, qz, count, qmask, pa, pb);
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
constexpr int kFallbackChunks = 8;            // wavefronts per query block when a body falls back to block-major order

__device__ __forceinline__ float wave_min(float v) { return wave_min_uniform(v); }      // common.h: DPP, no LDS trips
__device__ __forceinline__ float wave_max(float v) { return wave_max_uniform(v); }

// One (query block, leaf) the block has to visit: which of its 64 rays pass the leaf's slabs.
// (RayEntry, RayTile, RayBody: model.h)

// (half in the low / high 16 bits of s) - q and q - (half of s) in single precision, one instruction each; the largest of three
__device__ __forceinline__ float lo_minus(uint32_t s, float q)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(q));
    return d;
}
__device__ __forceinline__ float hi_minus(uint32_t s, float q)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(q));
    return d;
}
__device__ __forceinline__ float minus_lo(float q, uint32_t s)
{
    float d;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(q));
    return d;
}
__device__ __forceinline__ float minus_hi(float q, uint32_t s)
{
    float d;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(q));
    return d;
}
__device__ __forceinline__ float max3(float a, float b, float c)
{
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// One unit of work of ray_leaf_kernel: up to 64 rays (pairs[first .. first + n)) against one leaf's strip run.
// per body: number of tiles, whether the pair list overflowed (the body is then walked block-major)

// One workgroup of kWaves wavefronts per (block of 64 queries, body); every wavefront holds the block's 64 queries.
//   ranges : the 13 block ranges (kWaves = 4: each reduced by one of the wavefronts, shared through LDS);
//   rounds : the wavefronts take one chunk of 64 leaves each, lanes over LEAVES: block ranges against the slabs
//            (stage 1); the records of the passing leaves go to a queue in LDS; the queue is then dealt out evenly,
//            two entries per wavefront and trip: every ray against the record (stage 2).
// A trip of stage 2 is a dependent chain (LDS read -> 19 VALU -> ballot -> branch, ~450 clocks measured with the cycle
// counter) and the blocks through the trunk pass five times the average number of leaves: the time of the kernel is
// that of its slowest workgroup plus the rounds of workgroups the chip needs.  With one wavefront testing a block's
// leaves one at a time it was 59 us at batch 64 and 41 us at batch 8.  Four wavefronts and two chains per trip: 16 us
// at batch 8.  At batch 64 the search runs beside this kernel on another stream and leaves 7 wave slots and < 4 KB of
// LDS per CU (v2v.hip, leave_room): there the one-wavefront form (2 KB) is used -- workgroups of four with an 8 KB
// queue waited for LDS until the search had drained (202 us, tools/graph_timeline.py).
template <bool kVerts, int kWaves>
__global__ __launch_bounds__(64 * kWaves) void ray_near_kernel(
    const float* __restrict__ pts, const TreeNode* __restrict__ nodes, const float* __restrict__ bounds, int N,
    int num_leaves, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ counts, int Q, int qblocks, RayEntry* __restrict__ lists, int32_t* __restrict__ list_len,
    int32_t* __restrict__ leaf_cnt,             // [B][num_leaves], zeroed: rays per leaf
    unsigned long long* __restrict__ stats,     // measurement (or nullptr): [1] += (ray, element) pairs inside a listed
                                                // leaf's slabs, [2] += 64 x elements listed
    const int32_t* __restrict__ ring_off = nullptr, const int32_t* __restrict__ ring_vidx = nullptr,
    float2* __restrict__ fans = nullptr)        // kVerts: workgroups behind the query blocks compute the vertices' closing fans
{
    if (kVerts && (int)blockIdx.x >= qblocks) {
        // the closing fans of 64 kWaves vertices (fan_terms): vector work that needs the vertices only, beside this launch's
        // own workgroups, which mostly wait (as a kernel of its own, or inside ray_finalize_verts_kernel behind the
        // crossing kernel, it was 18 - 28 us on the step's critical chain)
        const int i = ((int)blockIdx.x - qblocks) * (64 * kWaves) + (int)threadIdx.x;        // position in tree order
        const bool real = i < Q;
        const int v = qperm[real ? i : Q - 1];
        int cr = 0;
        const float half = fan_terms(pts + (size_t)blockIdx.y * Q * 3, v, real, cr, ring_off, ring_vidx);
        if (i < qblocks * kRayQueries) fans[(size_t)blockIdx.y * qblocks * kRayQueries + i] = make_float2(__int_as_float(cr), half);
        return;
    }
    __builtin_amdgcn_s_setprio(3);               // (see ray_tiles_fill_kernel)
    const int qb = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = kWaves > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    __shared__ uint4 queue[64 * kWaves][2];     // records of the round's leaves that passed stage 1
    __shared__ float range[16];
    __shared__ int queued[2], slots;            // queue entries (rounds in turn); list entries written (slots are taken from it)
    int i0;
    bool real;                                  // padding lanes repeat a query; they are left out of the masks
    if (kVerts) {
        i0 = qperm[qb * kRayQueries + lane];
        real = qb * kRayQueries + lane < Q;
    } else {
        const int n = counts ? counts[b] : Q;
        if (qb * kRayQueries >= n) {
            if (threadIdx.x == 0) list_len[(size_t)b * qblocks + qb] = 0;
            return;
        }
        i0 = min(qb * kRayQueries + lane, n - 1);
        real = qb * kRayQueries + lane < n;
    }
    if (threadIdx.x == 0) { queued[0] = 0; queued[1] = 0; slots = 0; }
    // every load that does not wait for the block's ranges is issued here: the body's reference point and the slab
    // values / records of the wavefront's first chunk of leaves
    const float4 ctr = *reinterpret_cast<const float4*>(bounds + near_centres_at(gridDim.y, num_leaves) + (size_t)b * 4);
    const uint4* recs = reinterpret_cast<const uint4*>(bounds + near_records_at(gridDim.y, num_leaves)) + (size_t)b * num_leaves * 2;
    // stage 1 reads the leaves' 32-byte RECORDS only (half precision, relative to the body's reference point, bounds rounded
    // outwards: what stage 2 tests every ray against) -- with the 13 single-precision slab values beside them every
    // block fetched 84 bytes per leaf, 257 MB per launch at batch 64 out of the L2s: the kernel's time.  A leaf that some
    // ray passes in stage 2 passes stage 1 (the block's range holds the ray's value, the record is the same).
    struct Chunk { uint4 r0, r1; };
    auto load_chunk = [&](int base) {
        Chunk c;
        const int lc = min(base + lane, num_leaves - 1);
        c.r0 = recs[lc * 2];
        c.r1 = recs[lc * 2 + 1];
        return c;
    };
    Chunk nxt = load_chunk(wave * 64);
    const float* q3 = pts + ((size_t)b * Q + i0) * 3;
    const float qz = q3[2];
    const float qx = shear_x(q3[0], qz), qy = shear_y(q3[1], qz);
    // the queries relative to the body's reference point, as the records are
    const float rx = qx - ctr.x, ry = qy - ctr.y, rz = qz - ctr.z;
    const float r4 = rx + ry, r5 = rx - ry, r6 = rx + rz, r7 = rx - rz, r8 = ry + rz, r9 = ry - rz;
    // the block's ranges of these nine values, the side each record bound is compared with
    float bx0, bx1, by0, by1, b40, b41, b50, b51, bz0, b60, b71, b80, b91;
    if (kWaves == 4) {
        if (wave == 0) {
            const float a = wave_min(rx), c = wave_max(rx), d = wave_min(ry), e = wave_max(ry);
            if (lane == 0) { range[0] = a; range[1] = c; range[2] = d; range[3] = e; }
        } else if (wave == 1) {
            const float a = wave_min(r4), c = wave_max(r4), d = wave_min(r5), e = wave_max(r5);
            if (lane == 0) { range[4] = a; range[5] = c; range[6] = d; range[7] = e; }
        } else if (wave == 2) {
            const float a = wave_min(rz), c = wave_min(r6), d = wave_max(r7);
            if (lane == 0) { range[8] = a; range[9] = c; range[10] = d; }
        } else {
            const float a = wave_min(r8), c = wave_max(r9);
            if (lane == 0) { range[11] = a; range[12] = c; }
        }
        __syncthreads();
        bx0 = range[0]; bx1 = range[1]; by0 = range[2]; by1 = range[3]; b40 = range[4]; b41 = range[5]; b50 = range[6];
        b51 = range[7]; bz0 = range[8]; b60 = range[9]; b71 = range[10]; b80 = range[11]; b91 = range[12];
    } else {
        bx0 = wave_min(rx); bx1 = wave_max(rx); by0 = wave_min(ry); by1 = wave_max(ry);
        b40 = wave_min(r4); b41 = wave_max(r4); b50 = wave_min(r5); b51 = wave_max(r5);
        bz0 = wave_min(rz); b60 = wave_min(r6); b71 = wave_max(r7); b80 = wave_min(r8); b91 = wave_max(r9);
    }
    RayEntry* list = lists + ((size_t)b * qblocks + qb) * num_leaves;
    unsigned long long useful = 0, listed = 0;
    // record: s0 = (lo0,lo1) (lo3,lo4) (lo6,lo8) (hi0,hi1)   s1 = (hi2,hi3) (hi4,hi5) (hi7,leaf) node
    // > 0: outside some slab.  x1 .. : the value set against the slab's LOWER bound, x0 .. : against its UPPER bound (a
    // ray's value twice; the block's largest and smallest)
    auto outside13 = [&](const uint4& s0, const uint4& s1, float x1, float x0, float y1, float y0, float p1, float p0,
                         float m1, float m0, float z0, float xz0, float xz1, float yz0, float yz1) {
        float out = max3(lo_minus(s0.x, x1), minus_lo(x0, s0.w), hi_minus(s0.x, y1));
        out = max3(out, minus_hi(y0, s0.w), lo_minus(s0.y, p1));
        out = max3(out, minus_hi(p0, s1.x), hi_minus(s0.y, m1));
        out = max3(out, minus_lo(m0, s1.y), minus_lo(z0, s1.x));
        out = max3(out, minus_hi(xz0, s1.y), lo_minus(s0.z, xz1));
        return max3(out, minus_lo(yz0, s1.z), hi_minus(s0.z, yz1));
    };
    auto outside = [&](const uint4& s0, const uint4& s1) {
        return outside13(s0, s1, rx, rx, ry, ry, r4, r4, r5, r5, rz, r6, r7, r8, r9);
    };
    int nslots = 0;                             // one wavefront: the list's length in a (scalar) register, no LDS atomic
    auto emit = [&](const uint4& s1, unsigned long long hit) {
        if (!hit) return;
        const int leaf = (int)(s1.z >> 16), nd = (int)s1.w;
        if (kWaves == 1) ++nslots;
        if (lane == 0) {
            const int slot = kWaves == 1 ? nslots - 1 : atomicAdd(&slots, 1);
            list[slot] = RayEntry{leaf, nd, (uint32_t)hit, (uint32_t)(hit >> 32)};
            atomicAdd(&leaf_cnt[(size_t)b * num_leaves + leaf], __builtin_popcountll(hit));
        }
        if (stats) {
            const int len = nodes[nd].ex_len;
            useful += (unsigned long long)__builtin_popcountll(hit) * len;
            listed += 64ull * len;
        }
    };
    // (one wavefront: its LDS accesses are served in order, nothing to wait for between the stages)
    auto sync = [&] { if (kWaves > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    for (int round = 0; round * 64 * kWaves < num_leaves; ++round) {
        const int base = (round * kWaves + wave) * 64;
        const Chunk c = nxt;
        if (base + 64 * kWaves < num_leaves) nxt = load_chunk(base + 64 * kWaves);
        const bool pass = (base + lane < num_leaves) &
                          !(outside13(c.r0, c.r1, bx1, bx0, by1, by0, b41, b40, b51, b50, bz0, b60, b71, b80, b91) > 0.0f);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask) {
            int at = 0;
            if (kWaves > 1) {
                if (lane == 0) at = atomicAdd(&queued[round & 1], __builtin_popcountll(mask));
                at = __builtin_amdgcn_readfirstlane(at);
            }
            at += __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (pass) { queue[at][0] = c.r0; queue[at][1] = c.r1; }
        }
        sync();
        const int n = kWaves > 1 ? queued[round & 1] : __builtin_popcountll(mask);
        if (kWaves > 1 && threadIdx.x == 0) queued[(round + 1) & 1] = 0;    // (last read before this round's first barrier)
        if (kWaves == 1) {
            // one wavefront per block (large batches): FOUR records per trip -- the trip is a dependent chain (LDS read ->
            // 19 vector instructions -> ballot -> branch) and the blocks through the trunk have ~200 of them
            for (int i = 0; i < n; i += 4) {
                const int k1 = min(i + 1, n - 1), k2 = min(i + 2, n - 1), k3 = min(i + 3, n - 1);
                const uint4 a0 = queue[i][0], a1 = queue[i][1], c0 = queue[k1][0], c1 = queue[k1][1];
                const uint4 d0 = queue[k2][0], d1 = queue[k2][1], f0 = queue[k3][0], f1 = queue[k3][1];
                const float out_a = outside(a0, a1), out_c = outside(c0, c1), out_d = outside(d0, d1), out_f = outside(f0, f1);
                const unsigned long long hit_a = __builtin_amdgcn_ballot_w64(real && !(out_a > 0.0f));
                const unsigned long long hit_c = __builtin_amdgcn_ballot_w64(real && !(out_c > 0.0f) && i + 1 < n);
                const unsigned long long hit_d = __builtin_amdgcn_ballot_w64(real && !(out_d > 0.0f) && i + 2 < n);
                const unsigned long long hit_f = __builtin_amdgcn_ballot_w64(real && !(out_f > 0.0f) && i + 3 < n);
                emit(a1, hit_a);
                emit(c1, hit_c);
                emit(d1, hit_d);
                emit(f1, hit_f);
            }
        } else
        for (int i = wave * 2; i < n; i += kWaves * 2) {
            const int k = i + 1 < n ? i + 1 : i;
            const uint4 a0 = queue[i][0], a1 = queue[i][1], c0 = queue[k][0], c1 = queue[k][1];
            const float out_a = outside(a0, a1), out_c = outside(c0, c1);
            const unsigned long long hit_a = __builtin_amdgcn_ballot_w64(real && !(out_a > 0.0f));
            const unsigned long long hit_c = __builtin_amdgcn_ballot_w64(real && !(out_c > 0.0f) && k != i);
            emit(a1, hit_a);
            emit(c1, hit_c);
        }
        sync();
    }
    if (stats && lane == 0) { atomicAdd(stats + 1, useful); atomicAdd(stats + 2, listed); }
    if (threadIdx.x == 0) list_len[(size_t)b * qblocks + qb] = kWaves == 1 ? nslots : slots;
}

// Per body: where every leaf's rays start in the pair list (exclusive scan of the counts) and the table of tiles
// (leaf, 64 of its rays).  A body whose pairs do not fit `cap` is marked: it is walked block-major instead.
constexpr int kTilesBlock = 256;
__global__ __launch_bounds__(kTilesBlock) void ray_tiles_kernel(
    const int32_t* __restrict__ leaf_cnt, const TreeNode* __restrict__ nodes, const int32_t* __restrict__ leaf_nodes,
    int num_leaves, int cap, int max_tiles, int fallback_tiles, int32_t* __restrict__ leaf_off,
    RayTile* __restrict__ tiles, RayBody* __restrict__ body)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (num_leaves + kTilesBlock - 1) / kTilesBlock;
    const int l0 = min(t * per, num_leaves), l1 = min(l0 + per, num_leaves);
    const int32_t* cnt = leaf_cnt + (size_t)b * num_leaves;
    int pairs = 0, ntile = 0;
    for (int l = l0; l < l1; ++l) { pairs += cnt[l]; ntile += (cnt[l] + 63) >> 6; }
    __shared__ int sp[kTilesBlock], st[kTilesBlock];
    sp[t] = pairs;
    st[t] = ntile;
    __syncthreads();
    for (int d = 1; d < kTilesBlock; d <<= 1) {          // inclusive scans
        const int ap = t >= d ? sp[t - d] : 0, at = t >= d ? st[t - d] : 0;
        __syncthreads();
        sp[t] += ap;
        st[t] += at;
        __syncthreads();
    }
    const int total_pairs = sp[kTilesBlock - 1], total_tiles = st[kTilesBlock - 1];
    const bool overflow = total_pairs > cap || total_tiles > max_tiles;
    if (t == 0) body[b] = RayBody{overflow ? fallback_tiles : total_tiles, overflow ? 1 : 0};
    if (overflow) return;
    int off = sp[t] - pairs, tile = st[t] - ntile;
    RayTile* out = tiles + (size_t)b * max_tiles;
    for (int l = l0; l < l1; ++l) {
        leaf_off[(size_t)b * num_leaves + l] = off;
        const TreeNode nd = nodes[leaf_nodes[l]];
        for (int k = 0; k < cnt[l]; k += 64) out[tile++] = RayTile{nd.ex_off, nd.ex_len, off + k, min(64, cnt[l] - k)};
        off += cnt[l];
    }
}

// The rays of every leaf: slot (= position of the query in its body's order) of each set bit of every entry.
// The order within a leaf's range comes from an atomic counter and may vary between runs; the crossing counts
// are integer sums and do not depend on it.
__global__ __launch_bounds__(64) void ray_fill_kernel(
    const RayEntry* __restrict__ lists, const int32_t* __restrict__ list_len, const RayBody* __restrict__ body,
    const int32_t* __restrict__ leaf_off, int num_leaves, int qblocks, int cap, int32_t* __restrict__ leaf_fill,
    int32_t* __restrict__ pairs)
{
    const int qb = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (body[b].overflow) return;
    const int cnt = list_len[(size_t)b * qblocks + qb];
    const RayEntry* list = lists + ((size_t)b * qblocks + qb) * num_leaves;
    int32_t* out = pairs + (size_t)b * cap;
    for (int base = 0; base < cnt; base += 64) {
        // lanes over entries: reserve the entry's range in its leaf (64 atomics in flight)
        const int j = base + lane;
        RayEntry e = RayEntry{0, 0, 0u, 0u};
        int dst = 0;
        if (j < cnt) {
            e = list[j];
            const int n = __builtin_popcount(e.mask_lo) + __builtin_popcount(e.mask_hi);
            dst = leaf_off[(size_t)b * num_leaves + e.leaf] + atomicAdd(&leaf_fill[(size_t)b * num_leaves + e.leaf], n);
        }
        // lanes over rays: entry by entry, the set lanes write their slot
        const int m = min(64, cnt - base);
        for (int k = 0; k < m; ++k) {
            const uint32_t lo = __builtin_amdgcn_readlane(e.mask_lo, k), hi = __builtin_amdgcn_readlane(e.mask_hi, k);
            const int d = __builtin_amdgcn_readlane(dst, k);
            const unsigned long long mask = ((unsigned long long)hi << 32) | lo;
            if ((mask >> lane) & 1ull)
                out[d + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = qb * kRayQueries + lane;
        }
    }
}

// ray_tiles_kernel + ray_fill_kernel in one launch: kFillSplit one-wave workgroups per body.  Each scans the body's
// counts for itself (430 numbers, seven per lane and a wavefront scan); the first one writes the tile table; the body's
// entries -- all its query blocks' lists laid end to end through a prefix of their lengths -- are dealt out 64 at a time.
// (As two launches with one wavefront per query block: 10 + 31 us at batch 64 and 7 + 14 us at batch 8, most of it the
// launch rate of 6912 one-wavefront workgroups and three dependent global round trips in each; as ONE workgroup per body,
// fill counters in LDS: 61 us -- the write loop below is ~25 instructions per entry and a body's 4300 entries kept one CU
// busy that long.)
// ONE-wavefront workgroups: beside the nearest-vertex search -- 55 k one-wave workgroups that take every wave slot as it
// falls free -- a workgroup of four wavefronts waits until a CU has four free slots AT ONCE: the 256-thread form of
// this kernel took 24 us alone and 77-95 us in the replayed step, on the chain in front of the crossing kernel.
constexpr int kTilesFillBlock = 64, kFillSplit = 64, kFillMaxBlocks = 2048;
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}
__global__ __launch_bounds__(kTilesFillBlock) void ray_tiles_fill_kernel(
    const int32_t* __restrict__ leaf_cnt, const TreeNode* __restrict__ nodes, const int32_t* __restrict__ leaf_nodes,
    int num_leaves, int cap, int max_tiles, int fallback_tiles, RayTile* __restrict__ tiles, RayBody* __restrict__ body,
    const RayEntry* __restrict__ lists, const int32_t* __restrict__ list_len, int qblocks, int32_t* __restrict__ leaf_fill,
    int32_t* __restrict__ pairs, int list_stride)
{
    __builtin_amdgcn_s_setprio(3);                  // a chain of short dependent steps: ask for the issue slots first
    extern __shared__ int32_t dyn[];                // off[num_leaves] | tile[num_leaves] | blk[qblocks + 1]
    int32_t* off_s = dyn;
    int32_t* tile_s = dyn + num_leaves;
    int32_t* blk = dyn + 2 * num_leaves;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int per = (num_leaves + 63) / 64;
    const int l0 = min(lane * per, num_leaves), l1 = min(l0 + per, num_leaves);
    const int perq = (qblocks + 63) / 64;
    const int q0 = min(lane * perq, qblocks), q1 = min(q0 + perq, qblocks);
    const int32_t* cnt = leaf_cnt + (size_t)b * num_leaves;
    const int32_t* len = list_len + (size_t)b * qblocks;
    int npairs = 0, ntile = 0, nent = 0;
    for (int l = l0; l < l1; ++l) { const int c = cnt[l]; npairs += c; ntile += (c + 63) >> 6; }
    for (int q = q0; q < q1; ++q) nent += len[q];
    const int sp = wave_inclusive_scan(npairs, lane), st = wave_inclusive_scan(ntile, lane), sq = wave_inclusive_scan(nent, lane);
    const int total_pairs = __builtin_amdgcn_readlane(sp, 63), total_tiles = __builtin_amdgcn_readlane(st, 63);
    const int total = __builtin_amdgcn_readlane(sq, 63);
    const bool overflow = total_pairs > cap || total_tiles > max_tiles;
    if (lane == 0 && blockIdx.y == 0) body[b] = RayBody{overflow ? fallback_tiles : total_tiles, overflow ? 1 : 0};
    if (overflow) return;
    {
        int off = sp - npairs, tile = st - ntile;
        for (int l = l0; l < l1; ++l) {
            const int c = cnt[l];
            off_s[l] = off;
            tile_s[l] = tile;
            off += c;
            tile += (c + 63) >> 6;
        }
        int ent = sq - nent;
        for (int q = q0; q < q1; ++q) { blk[q] = ent; ent += len[q]; }
        if (lane == 0) blk[qblocks] = total;
    }
    __syncthreads();
    // the tile table: every workgroup of the body writes the tiles of a few leaves (one leaf per lane: two dependent loads
    // and a short loop) -- written by the body's first workgroup alone, seven leaves per lane one after the other, it was
    // the longest chain in the launch
    for (int l = (int)blockIdx.y + (int)gridDim.y * lane; l < num_leaves; l += (int)gridDim.y * 64) {
        const TreeNode nd = nodes[leaf_nodes[l]];
        const int ex_off = nd.ex_off, ex_len = nd.ex_len;
        const int c = cnt[l], off = off_s[l];
        RayTile* out = tiles + (size_t)b * max_tiles + tile_s[l];
        for (int k = 0; k < c; k += 64) *out++ = RayTile{ex_off, ex_len, off + k, min(64, c - k)};
    }
    int32_t* fill = leaf_fill + (size_t)b * num_leaves;
    int32_t* out = pairs + (size_t)b * cap;
    for (int base = (int)blockIdx.y * 64; base < total; base += (int)gridDim.y * 64) {
        // lanes over entries: reserve the entry's range in its leaf (64 atomics in flight)
        const int g = base + lane;
        RayEntry e = RayEntry{0, 0, 0u, 0u};
        int dst = 0, qb = 0;
        if (g < total) {
            int lo = 0, hi = qblocks;                       // the block with blk[qb] <= g < blk[qb + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (blk[mid] <= g) lo = mid; else hi = mid;
            }
            qb = lo;
            e = lists[((size_t)b * qblocks + qb) * list_stride + (g - blk[qb])];
            dst = off_s[e.leaf] + atomicAdd(&fill[e.leaf], __builtin_popcount(e.mask_lo) + __builtin_popcount(e.mask_hi));
        }
        // lanes over rays: entry by entry, the set lanes write their slot
        const int m = min(64, total - base);
        for (int k = 0; k < m; ++k) {
            const uint32_t lo = __builtin_amdgcn_readlane(e.mask_lo, k), hi = __builtin_amdgcn_readlane(e.mask_hi, k);
            const int d = __builtin_amdgcn_readlane(dst, k), q = __builtin_amdgcn_readlane(qb, k);
            const unsigned long long mask = ((unsigned long long)hi << 32) | lo;
            if ((mask >> lane) & 1ull)
                out[d + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = q * kRayQueries + lane;
        }
    }
}

// Crossing counts, leaf-major: a wavefront takes a tile -- one leaf and up to 64 of the rays that pass its slabs,
// whichever query blocks they come from -- walks the leaf's strip run and adds each ray's crossings to its query's
// count.  (Block-major, 64 neighbouring queries against every leaf any of them meets, only ~22 % of the lanes
// can have a crossing at all: measured, tuch_ray_work.)  Queries: the model's vertices in tree order (qperm != nullptr;
// faces around the query are skipped) or arbitrary points [B,Q,3] in the caller's order.
// Grid (G = min(B, 8), workers): workgroups go round-robin to the 8 XCDs, so the wavefronts of column x all sit on one
// XCD; together they work through bodies x, x + G, x + 2 G, ... one after the other (a body's 0.2 MB sheared stream
// stays in that XCD's L2, and a body with many tiles is shared by all the column's wavefronts).
// The next tile (record, ray slot, query coordinates: a chain of dependent loads) is fetched while the current one is
// walked.  A body marked overflow is walked block-major: tile t = (query block t / 8, every 8th leaf of the block's
// list from t % 8).
// kCount: elements walked are added to stats[0] (measurement).
// kSeg (vertex queries of a model with segments): the crossings with the faces of the lane's own segments are counted
// as well (seg_count[b][slot][2], four signed byte counters per word; the elements' segments ride in the stream's
// fourth word -- tree_sign_word --, the vertices' in seg_vmask).
template <bool kVerts, bool kCount, bool kSeg>
__global__ __launch_bounds__(64) void ray_leaf_kernel(
    const float* __restrict__ pts, const RayElem* __restrict__ stream, const RayTile* __restrict__ tiles,
    const RayBody* __restrict__ body, const int32_t* __restrict__ pairs, const RayEntry* __restrict__ lists,
    const int32_t* __restrict__ list_len, const TreeNode* __restrict__ nodes, int num_leaves,
    const int32_t* __restrict__ qperm, const int32_t* __restrict__ counts, int Q, int T, int qblocks, int num_bodies,
    int cap, int max_tiles, int32_t* __restrict__ count, unsigned long long* __restrict__ stats,
    const int32_t* __restrict__ vmask, int32_t* __restrict__ seg_count)
{
    const int lane = threadIdx.x;
    struct Work { int b, off, len, slot, qmask; bool active; float qx, qy, qz; };     // b < 0: nothing left; len < 0: block-major tile `off`
    // The tiles of the column's bodies x, x + G, ... form one sequence; wavefront y takes every gridDim.y-th of it (a
    // work counter instead serialises: ~100 ns per atomic on one address, measured).
    int b_cur = blockIdx.x, base = 0, g = blockIdx.y;       // current body, tiles before it, next position in the sequence
    auto fetch = [&]() {
        Work w;
        w.b = -1; w.off = w.len = w.slot = 0; w.qmask = 0; w.active = false; w.qx = w.qy = w.qz = 0.0f;
        while (b_cur < num_bodies) {
            const int b = b_cur;
            const int nt = __builtin_amdgcn_readfirstlane(body[b].tiles);
            if (g >= base + nt) { base += nt; b_cur += gridDim.x; continue; }
            const int t = g - base;
            g += gridDim.y;
            const float* pb = pts + (size_t)b * Q * 3;
            w.b = b;
            int i0;
            if (!__builtin_amdgcn_readfirstlane(body[b].overflow)) {
                const RayTile tile = tiles[(size_t)b * max_tiles + t];
                const int n = __builtin_amdgcn_readfirstlane(tile.n), first = __builtin_amdgcn_readfirstlane(tile.first);
                w.active = lane < n;
                w.slot = pairs[(size_t)b * cap + first + (w.active ? lane : 0)];
                i0 = kVerts ? qperm[w.slot] : w.slot;
                w.off = __builtin_amdgcn_readfirstlane(tile.ex_off);
                w.len = __builtin_amdgcn_readfirstlane(tile.ex_len);
            } else {
                const int nq = kVerts ? Q : (counts ? counts[b] : Q);
                w.slot = (t / kFallbackChunks) * kRayQueries + lane;
                w.active = w.slot < nq;
                i0 = kVerts ? qperm[w.slot] : min(w.slot, nq - 1);
                w.off = t;                     // block-major tile: (query block, chunk) packed, marked by len < 0
                w.len = -1;
            }
            w.qz = pb[3 * i0 + 2];
            w.qx = shear_x(pb[3 * i0], w.qz);
            w.qy = shear_y(pb[3 * i0 + 1], w.qz);
            if (kSeg) w.qmask = vmask[w.slot];
            break;
        }
        return w;
    };
    int walked = 0;
    Work next = fetch();
    while (next.b >= 0) {
        const Work w = next;
        next = fetch();
        S3 s[3];
        float e[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { s[k].xy = (v2f){0.0f, 0.0f}; s[k].z = 0.0f; e[k] = 0.0f; }
        const v2f qxy = {w.qx, w.qy};
        int crossings = 0;
        uint32_t pa = kSegBias, pb = kSegBias;
        const RayElem* st = stream + (size_t)w.b * T;
        if (w.len >= 0) {
            ray_run<kVerts, kSeg>(st, w.off, w.len, s, e, qxy, w.qz, crossings, w.qmask, pa, pb);
            if (kCount) walked += w.len;
        } else {
            const int qb = w.off / kFallbackChunks, c = w.off % kFallbackChunks;
            const int cnt = __builtin_amdgcn_readfirstlane(list_len[(size_t)w.b * qblocks + qb]);
            const RayEntry* list = lists + ((size_t)w.b * qblocks + qb) * num_leaves;
            for (int j = c; j < cnt; j += kFallbackChunks) {
                const TreeNode nd = nodes[__builtin_amdgcn_readfirstlane(list[j].node)];
                ray_run<kVerts, kSeg>(st, nd.ex_off, nd.ex_len, s, e, qxy, w.qz, crossings, w.qmask, pa, pb);
                if (kCount) walked += nd.ex_len;
            }
        }
        if (w.active && crossings != 0) atomicAdd(&count[((size_t)w.b * qblocks) * kRayQueries + w.slot], crossings);
        if (kSeg && w.active && (pa != kSegBias || pb != kSegBias)) {             // few lanes
            // two words of four byte-wide counters per ray; a tile's (biased) byte sums are added as ONE integer: the
            // running word may borrow across its bytes, the final word is the exact sum all the same (seg_leaf_count)
            int32_t* sc = seg_count + 2 * (((size_t)w.b * qblocks) * kRayQueries + w.slot);
            if (pa != kSegBias) atomicAdd(sc, (int32_t)(pa - kSegBias));
            if (pb != kSegBias) atomicAdd(sc + 1, (int32_t)(pb - kSegBias));
        }
    }
    if (kCount && lane == 0) atomicAdd(stats, (unsigned long long)walked);
}

// ---- the middle of the inside test as ONE launch (round 6) ---------------------------------------------------------------
// ray_near_kernel -> ray_tiles_fill_kernel -> ray_leaf_kernel are a chain of three launches on the step's critical path
// (57 + 41 + 121 us in the replayed step at batch 64), the first two of them bookkeeping: which leaves the rays of a query
// block can meet (lists in global memory), the lists regrouped by leaf into tiles of 64 rays (counters, an exclusive scan,
// a fill pass).  Turned round, nothing has to leave the wavefront: ONE wavefront per (leaf, body)
//   1. lanes over QUERY BLOCKS: the block's ranges (ray_block_prepass, beside the leaf bounds in the chain's first launch)
//      against the leaf's record -- the same 13 comparisons as ray_near_kernel's stage 1, the same superset;
//   2. lanes over QUERIES, the passing blocks three at a time (independent loads and tests in flight): every ray against
//      the record -- ray_near_kernel's stage 2 --, the rays that pass are appended to a ring in LDS (coordinates + tag);
//   3. whenever the ring holds 64 rays, and at the end for the rest: the leaf's strip run is walked for them (ray_run) and
//      the crossings are added to the queries' counts.
// The (ray, leaf) pairs walked are exactly those of the three-launch form, so are the counts (integer sums).  No lists, no
// pair table, no overflow fallback; the leaf's record, strip range and the strip itself stay in scalar registers / the
// scalar cache for all of the leaf's tiles.
// Grid (G = min(B, 8), rows): workgroup (x, y) = body x + G (y / L), leaf y % L -- column x sits on one XCD (workgroups go
// round-robin to the XCDs), which works through bodies x, x + G, ... in turn: a body's stream and query records stay in
// that XCD's L2.
constexpr int kCrossRing = 256, kCrossTrip = 3;
template <bool kSeg>
__global__ __launch_bounds__(64) void ray_cross_kernel(
    const RayElem* __restrict__ stream, const TreeNode* __restrict__ nodes, const float* __restrict__ bounds, int num_leaves,
    int nsplit,                                   // wavefronts that share a leaf: wavefront z takes the query blocks z, z + nsplit, ...
    const QRec* __restrict__ qrec, const float4* __restrict__ ranges, int T, int qblocks, int num_bodies,
    int32_t* __restrict__ count, int32_t* __restrict__ seg_count)
{
    __shared__ QRec ring[kCrossRing];
    const int lane = threadIdx.x;
    const int unit = blockIdx.y / nsplit, z = blockIdx.y % nsplit;
    const int b = blockIdx.x + gridDim.x * (unit / num_leaves);
    if (b >= num_bodies) return;
    const int leaf = unit % num_leaves;
    const uint4* recs = reinterpret_cast<const uint4*>(bounds + near_records_at(num_bodies, num_leaves)) + ((size_t)b * num_leaves + leaf) * 2;
    const uint4 s0 = recs[0], s1 = recs[1];
    const float4 ctr = *reinterpret_cast<const float4*>(bounds + near_centres_at(num_bodies, num_leaves) + (size_t)b * 4);
    const TreeNode nd = nodes[(int)s1.w];
    const int ex_off = __builtin_amdgcn_readfirstlane(nd.ex_off), ex_len = __builtin_amdgcn_readfirstlane(nd.ex_len);
    const RayElem* st = stream + (size_t)b * T;
    const QRec* qb_base = qrec + (size_t)b * qblocks * kRayQueries;
    const float4* rg = ranges + (size_t)b * qblocks * 4;
    const int mine = qblocks > z ? (qblocks - z + nsplit - 1) / nsplit : 0;      // own query blocks: z + nsplit j, j < mine
    // > 0: outside some slab (ray_near_kernel's test, word for word)
    auto outside13 = [&](float x1, float x0, float y1, float y0, float p1, float p0, float m1, float m0, float z0, float xz0,
                         float xz1, float yz0, float yz1) {
        float out = max3(lo_minus(s0.x, x1), minus_lo(x0, s0.w), hi_minus(s0.x, y1));
        out = max3(out, minus_hi(y0, s0.w), lo_minus(s0.y, p1));
        out = max3(out, minus_hi(p0, s1.x), hi_minus(s0.y, m1));
        out = max3(out, minus_lo(m0, s1.y), minus_lo(z0, s1.x));
        out = max3(out, minus_hi(xz0, s1.y), lo_minus(s0.z, xz1));
        return max3(out, minus_lo(yz0, s1.z), hi_minus(s0.z, yz1));
    };
    auto ray_outside = [&](const QRec& q) {
        const float rx = q.x - ctr.x, ry = q.y - ctr.y, rz = q.z - ctr.z;
        return outside13(rx, rx, ry, ry, rx + ry, rx + ry, rx - ry, rx - ry, rz, rx + rz, rx - rz, ry + rz, ry - rz);
    };
    int head = 0, tail = 0;                       // ring positions (wave-uniform)
    int base = -64;                               // first of the 64 own blocks `todo` is about
    unsigned long long todo = 0;                  // own blocks that passed stage 1 and are not yet fetched
    bool more = true, loaded = false;
    QRec q[kCrossTrip];                           // the trip in flight: its loads were issued one trip ahead
    bool have[kCrossTrip];
#pragma unroll
    for (int j = 0; j < kCrossTrip; ++j) { q[j] = QRec{0.f, 0.f, 0.f, kNoSlot}; have[j] = false; }
    auto fetch = [&]() {                          // the next (up to) three passing blocks: one 16-byte load per lane each
        int k[kCrossTrip];
#pragma unroll
        for (int j = 0; j < kCrossTrip; ++j) {
            have[j] = todo != 0;
            k[j] = have[j] ? z + nsplit * (base + (int)__builtin_ctzll(todo)) : (j ? k[j - 1] : z);
            if (have[j]) todo &= todo - 1;
        }
#pragma unroll
        for (int j = 0; j < kCrossTrip; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(qb_base + (size_t)k[j] * kRayQueries + lane);
            q[j].x = v.x; q[j].y = v.y; q[j].z = v.z; q[j].tag = __float_as_int(v.w);
        }
        loaded = true;
    };
    for (;;) {
        if (tail - head < 64 && (more || loaded)) {
            if (loaded) {
                // this trip's records are here (or on their way); the NEXT trip's loads go out before they are tested
                QRec qc[kCrossTrip];
                bool hc[kCrossTrip];
#pragma unroll
                for (int j = 0; j < kCrossTrip; ++j) { qc[j] = q[j]; hc[j] = have[j]; }
                loaded = false;
                if (todo) fetch();
                unsigned long long hit[kCrossTrip];
#pragma unroll
                for (int j = 0; j < kCrossTrip; ++j)
                    hit[j] = __builtin_amdgcn_ballot_w64(hc[j] && (qc[j].tag & kNoSlot) != kNoSlot && !(ray_outside(qc[j]) > 0.0f));
#pragma unroll
                for (int j = 0; j < kCrossTrip; ++j) {
                    if (hit[j]) {
                        const int pos = tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit[j], 0));
                        if ((hit[j] >> lane) & 1ull) ring[pos & (kCrossRing - 1)] = qc[j];
                        tail += __builtin_popcountll(hit[j]);
                    }
                }
                continue;
            }
            if (!todo) {
                base += 64;
                if (base >= mine) { more = false; continue; }
                const int j = base + lane;
                const float4* g = rg + (size_t)(z + nsplit * min(j, mine - 1)) * 4;
                const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
                // (bx0, bx1, by0, by1) (b40, b41, b50, b51) (bz0, b60, b71, b80) (b91): the lower bounds of the record
                // against the block's largest value, the upper ones against its smallest
                const bool pass = (j < mine) &
                                  !(outside13(g0.y, g0.x, g0.w, g0.z, g1.y, g1.x, g1.w, g1.z, g2.x, g2.y, g2.z, g2.w, g3.x) > 0.0f);
                todo = __builtin_amdgcn_ballot_w64(pass);
                continue;
            }
            fetch();
            continue;
        }
        const int n = min(64, tail - head);
        if (n <= 0) break;
        __builtin_amdgcn_wave_barrier();
        const bool active = lane < n;
        const QRec w = ring[(head + (active ? lane : 0)) & (kCrossRing - 1)];
        __builtin_amdgcn_wave_barrier();
        head += n;
        const int slot = w.tag & kNoSlot;
        S3 s[3];
        float e[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { s[u].xy = (v2f){0.0f, 0.0f}; s[u].z = 0.0f; e[u] = 0.0f; }
        int crossings = 0;
        uint32_t pa = kSegBias, pb = kSegBias;
        ray_run<true, kSeg>(st, ex_off, ex_len, s, e, (v2f){w.x, w.y}, w.z, crossings, (int)((uint32_t)w.tag >> 24), pa, pb);
        if (active && crossings != 0) atomicAdd(&count[((size_t)b * qblocks) * kRayQueries + slot], crossings);
        if (kSeg && active && (pa != kSegBias || pb != kSegBias)) {             // few lanes (see ray_leaf_kernel)
            int32_t* sc = seg_count + 2 * (((size_t)b * qblocks) * kRayQueries + slot);
            if (pa != kSegBias) atomicAdd(sc, (int32_t)(pa - kSegBias));
            if (pb != kSegBias) atomicAdd(sc + 1, (int32_t)(pb - kSegBias));
        }
    }
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

// One cone triangle (apex direction u, then x, y) of a closing fan / closing chain: its signed ray crossing and its half
// solid angle, BOTH signed by ONE determinant.  The two only mean something together: where u comes to lie in the plane
// of (query, x, y) -- a fan direction tangent to a face of the star -- the crossing flips by one and the half angle
// jumps by 2 pi / 2, and crossing - half / (2 pi) is continuous across the flip.  Computing det(u, x, y) twice (sheared
// for the crossing, unsheared inside the angle; equal in exact arithmetic, the shear has determinant 1) let the two
// signs disagree when the determinant is within rounding of zero: w off by exactly one, about once in 3 million cone
// triangles (found on the irregular-topology body, tests/test_gpu_contact.py::test_big_batches_...).  Exactly zero counts
// as positive for both.  s*: corners relative to the query in the sheared ray frame; u, pb, pc: the same in space.
__device__ __forceinline__ void cone_term(const P3& us, const P3& sb, const P3& sc, const P3& u, const P3& pb, const P3& pc,
                                          int& crossings, float& half)
{
    const float ea = edge_fn(sb, sc), eb = edge_fn(sc, us), ec = edge_fn(us, sb);
    const float d = ea * us.z + eb * sb.z + ec * sc.z;
    const bool la = left_of(ea, sb, sc), lb = left_of(eb, sc, us), lc = left_of(ec, us, sb);
    const bool front = d >= 0.0f;
    crossings = (la & lb & lc) ? (int)front : ((!la & !lb & !lc) ? -(int)!front : 0);
    const float nu = __builtin_sqrtf(u.x * u.x + u.y * u.y + u.z * u.z);
    const float nb = __builtin_sqrtf(pb.x * pb.x + pb.y * pb.y + pb.z * pb.z);
    const float nc = __builtin_sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
    const float cx = pb.y * pc.z - pb.z * pc.y, cy = pb.z * pc.x - pb.x * pc.z, cz = pb.x * pc.y - pb.y * pc.x;
    const float num = __builtin_fabsf(u.x * cx + u.y * cy + u.z * cz);
    const float dub = u.x * pb.x + u.y * pb.y + u.z * pb.z;
    const float dbc = pb.x * pc.x + pb.y * pc.y + pb.z * pc.z;
    const float duc = u.x * pc.x + u.y * pc.y + u.z * pc.z;
    const float den = nu * nb * nc + dub * nc + duc * nb + dbc * nu;
    half = fast_atan2(front ? num : -num, den);
}

// The closing fan of a VERTEX: n += its signed crossings, returns the sum of its half angles.  w = N - half_sum / (2 pi),
// N = count + crossings of the fan (fan_winding below).  The fan needs the vertices only -- not the crossing counts --: in
// the loss path it is computed by extra workgroups of ray_near_kernel's launch (whose own workgroups are chains of dependent
// steps that leave the vector units idle) and ray_finalize_verts_kernel only reads two words per vertex.
// One vertex per lane (`real` = false: a padding lane; all lanes of the wavefront must call -- long rings are shared out).
__device__ __forceinline__ float fan_terms(const float* __restrict__ vb, int v, bool real, int& n,
                                           const int32_t* __restrict__ ring_off, const int32_t* __restrict__ ring_vidx)
{
    const float vx = vb[3 * v], vy = vb[3 * v + 1], vz = vb[3 * v + 2];
    const float qx = shear_x(vx, vz), qy = shear_y(vy, vz);
    // the apex direction of the fan, in space and sheared (ray frame)
    const P3 u_dir = {kFanX, kFanY, kFanZ};
    const P3 us = {shear_x(kFanX, kFanZ), shear_y(kFanY, kFanZ), kFanZ};
    const int lo = ring_off[v], cnt = real ? ring_off[v + 1] - lo : 0;
    float half_sum = 0.0f;
    constexpr int kLongRing = 16;
    constexpr int kShortRing = 8;
    if (cnt > 0 && cnt <= kShortRing) {
        // the usual case (SMPL: valence 6, a few 7 - 9): ALL ring vertices' ids in one round of loads, all their coordinates
        // in the next -- with four at a time and the cycle's last vertex first this was six dependent rounds
        int rr[kShortRing];
        float cc[kShortRing][3];
#pragma unroll
        for (int u = 0; u < kShortRing; ++u) rr[u] = ring_vidx[lo + min(u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < kShortRing; ++u) { cc[u][0] = vb[3 * rr[u]]; cc[u][1] = vb[3 * rr[u] + 1]; cc[u][2] = vb[3 * rr[u] + 2]; }
        // slots behind the ring repeat its last vertex: slot kShortRing - 1 always holds the cycle's predecessor of vertex 0
        P3 pb = {cc[kShortRing - 1][0] - vx, cc[kShortRing - 1][1] - vy, cc[kShortRing - 1][2] - vz};
        P3 sb = {shear_x(cc[kShortRing - 1][0], cc[kShortRing - 1][2]) - qx, shear_y(cc[kShortRing - 1][1], cc[kShortRing - 1][2]) - qy,
                 cc[kShortRing - 1][2] - vz};
#pragma unroll
        for (int u = 0; u < kShortRing; ++u) {
            if (u < cnt) {
                const P3 pc = {cc[u][0] - vx, cc[u][1] - vy, cc[u][2] - vz};
                const P3 sc = {shear_x(cc[u][0], cc[u][2]) - qx, shear_y(cc[u][1], cc[u][2]) - qy, cc[u][2] - vz};
                int cr;
                float hf;
                cone_term(us, sb, sc, u_dir, pb, pc, cr, hf);
                half_sum += hf;
                n += cr;
                pb = pc;
                sb = sc;
            }
        }
    } else if (cnt > 0 && cnt <= kLongRing) {
        // previous ring vertex (j = cnt-1) to start the cycle
        int r = ring_vidx[lo + cnt - 1];
        P3 pb = {vb[3 * r] - vx, vb[3 * r + 1] - vy, vb[3 * r + 2] - vz};
        P3 sb = {shear_x(vb[3 * r], vb[3 * r + 2]) - qx, shear_y(vb[3 * r + 1], vb[3 * r + 2]) - qy, vb[3 * r + 2] - vz};
        // four ring vertices at a time: their ids, then their coordinates, are fetched together (the loop is
        // otherwise a chain of dependent gathers: id -> coordinates -> next id ...)
        for (int j0 = 0; j0 < cnt; j0 += 4) {
            int rr[4];
            float cc[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) rr[u] = ring_vidx[lo + min(j0 + u, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) { cc[u][0] = vb[3 * rr[u]]; cc[u][1] = vb[3 * rr[u] + 1]; cc[u][2] = vb[3 * rr[u] + 2]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < cnt) {
                    const P3 pc = {cc[u][0] - vx, cc[u][1] - vy, cc[u][2] - vz};
                    const P3 sc = {shear_x(cc[u][0], cc[u][2]) - qx, shear_y(cc[u][1], cc[u][2]) - qy, cc[u][2] - vz};
                    int cr;
                    float hf;
                    cone_term(us, sb, sc, u_dir, pb, pc, cr, hf);
                    half_sum += hf;
                    n += cr;
                    pb = pc;
                    sb = sc;
                }
            }
        }
    }
    // long rings (the poles of a lat-long sphere; SMPL has none above 16) are shared out over the wavefront: one fan
    // triangle per lane, sums by butterfly -- one lane walking 80 triangles would hold up the whole launch
    unsigned long long todo = __builtin_amdgcn_ballot_w64(cnt > kLongRing);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int llo = __builtin_amdgcn_readlane(lo, src), lcnt = __builtin_amdgcn_readlane(cnt, src);
        const float lvx = __shfl(vx, src), lvy = __shfl(vy, src), lvz = __shfl(vz, src);
        float h = 0.0f;
        int cr = 0;
        for (int j = lane; j < lcnt; j += 64) {
            const int r_prev = ring_vidx[llo + (j == 0 ? lcnt - 1 : j - 1)], r_cur = ring_vidx[llo + j];
            const float bx = vb[3 * r_prev], by = vb[3 * r_prev + 1], bz = vb[3 * r_prev + 2];
            const float cx = vb[3 * r_cur], cy = vb[3 * r_cur + 1], cz = vb[3 * r_cur + 2];
            const float lqx = shear_x(lvx, lvz), lqy = shear_y(lvy, lvz);
            const P3 pb = {bx - lvx, by - lvy, bz - lvz}, pc = {cx - lvx, cy - lvy, cz - lvz};
            const P3 sb = {shear_x(bx, bz) - lqx, shear_y(by, bz) - lqy, bz - lvz}, sc = {shear_x(cx, cz) - lqx, shear_y(cy, cz) - lqy, cz - lvz};
            int c1;
            float h1;
            cone_term(us, sb, sc, u_dir, pb, pc, c1, h1);
            h += h1;
            cr += c1;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { h += __shfl_xor(h, m); cr += __shfl_xor(cr, m); }
        if (lane == src) { half_sum = h; n += cr; }
    }
    return half_sum;
}
__device__ __forceinline__ float fan_winding(const float* __restrict__ vb, int v, bool real, int n,
                                             const int32_t* __restrict__ ring_off, const int32_t* __restrict__ ring_vidx)
{
    const float half_sum = fan_terms(vb, v, real, n, ring_off, ring_vidx);
    return (float)n - half_sum * (0.5f / kPi);
}

// vertices: the body test's flags from the crossing counts
__global__ __launch_bounds__(kBlock) void ray_finalize_verts_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ count, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ ring_off, const int32_t* __restrict__ ring_vidx,
    int V, int stride, float thresh, float* __restrict__ w_out, uint8_t* __restrict__ exterior,
    uint8_t* __restrict__ exterior_copy,                 // or nullptr: the same flags once more (the two-launch segment
                                                         // filter reads them while it re-marks `exterior`)
    const float2* __restrict__ fans)                     // or nullptr: [B][stride] (fan crossings as int bits, half-angle sum)
{                                                        // left by ray_near_kernel's fan workgroups
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;     // position in tree order
    const bool real = i < V;                             // all lanes stay: long rings need the whole wavefront
    const int v = qperm[real ? i : V - 1];
    const int n = count[(size_t)b * stride + (real ? i : V - 1)];
    float w;
    if (fans) {
        const float2 f = fans[(size_t)b * stride + (real ? i : V - 1)];
        w = (float)(n + __float_as_int(f.x)) - f.y * (0.5f / kPi);           // fan_winding's arithmetic
    } else
        w = fan_winding(verts + (size_t)b * V * 3, v, real, n, ring_off, ring_vidx);
    const size_t o = (size_t)b * V + v;
    if (real) {
        if (w_out) w_out[o] = w;
        if (exterior) exterior[o] = w <= thresh;
        if (exterior_copy) exterior_copy[o] = w <= thresh;
    }
}

__global__ __launch_bounds__(kBlock) void ray_finalize_points_kernel(
    const int32_t* __restrict__ count, const int32_t* __restrict__ counts,
    int Q, int stride, float thresh, float* __restrict__ w_out, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= Q) return;
    const int n = (!counts || i < counts[b]) ? count[(size_t)b * stride + i] : 0;
    const float w = (float)n;
    const size_t o = (size_t)b * Q + i;
    if (w_out) w_out[o] = w;
    if (exterior) exterior[o] = w <= thresh;
}

// ---- the segment filter by ray crossings ------------------------------------------------------------------------
// has_self_isect (segmentation.py:81-99): winding number of a segment's vertices w.r.t. its own "closed" mesh T
// (segment faces + cap fans).  The reference sums the solid angles of the faces that do not contain the query v
// (the others are atan2(0,0) = 0): of the chain T' = T minus star(v).  Close T' with cones from its boundary to an
// apex v + delta u: the result is a 2-cycle, its winding number around v is the signed number of ray crossings, so
//     w_ref(v) = crossings(T') + sum over the closing chain C of [crossing - half angle / 2 pi] of the cone triangle,
//     C = -boundary(T') = links of star(v) - boundary(T),
// exactly the fan of the body test when T is a closed manifold (boundary(T) = 0, the links form the ring) -- but with no
// such assumption: the reference's caps need not close the segment consistently (segmentation.py:56-66 takes the loops
// as the asset lists them), and the synthetic segments do leave a few dozen boundary edges.  As for the fan, the cone
// terms depend on the direction u only.  A segment vertex on no face of its segment has no links: w = crossings - the
// boundary terms.
// The segment meshes are small (hundreds of faces) and the queries few (the segment's interior vertices, compacted
// by winding.hip), so the layout is that of the solid-angle kernel it replaces: one wavefront per 64 queries and
// face split, triangles staged through LDS; ~30 plain operations per (query, triangle) instead of ~90.
constexpr int kSegRayChunk = 128;     // triangles staged in LDS per pass

// The entries a segment's crossing kernel walks, posed and sheared, 9 floats each: a face (three corners; index < V ->
// body vertex, else cap vertex, segmentation.py:77) or a boundary edge x -> y of the segment mesh with its (negative)
// multiplicity: [x', y', multiplicity, marker, 0].
constexpr float kConeMarker = 3.0e38f;
__global__ __launch_bounds__(kBlock) void segment_shear_entries_kernel(
    const float* __restrict__ verts, const float* __restrict__ caps,
    const int32_t* __restrict__ ent, int V, int K, int E, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= E) return;
    const int id[3] = {ent[3 * i], ent[3 * i + 1], ent[3 * i + 2]};
    float* dst = out + ((size_t)b * E + i) * 9;
    const int corners = id[2] < 0 ? 2 : 3;
    for (int k = 0; k < corners; ++k) {
        const float* src = id[k] < V ? verts + ((size_t)b * V + id[k]) * 3 : caps + ((size_t)b * K + (id[k] - V)) * 3;
        dst[3 * k] = shear_x(src[0], src[2]); dst[3 * k + 1] = shear_y(src[1], src[2]); dst[3 * k + 2] = src[2];
    }
    if (corners == 2) { dst[6] = (float)id[2]; dst[7] = kConeMarker; dst[8] = 0.0f; }
}

// Work items of the crossing kernel: (body, segment, block of 64 compacted queries), only those that exist -- a grid
// over the worst case (every segment vertex interior) is ten times larger and costs more in dispatch than in work.
// One workgroup; items[i] = (body * S + segment, first query), items_total[0] = how many, items_total[1] = face splits.
constexpr int kItemsBlock = 256;
__global__ __launch_bounds__(kItemsBlock) void segment_items_kernel(
    const int32_t* __restrict__ count, int BS, int2* __restrict__ items, int32_t* __restrict__ items_total,
    int max_split, int grid)
{
    __shared__ int sc[kItemsBlock];
    __shared__ int base;
    const int t = threadIdx.x;
    if (t == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < BS; c0 += kItemsBlock) {
        const int e = c0 + t;
        const int nb = e < BS ? (count[e] + 63) >> 6 : 0;
        sc[t] = nb;
        __syncthreads();
        for (int d = 1; d < kItemsBlock; d <<= 1) {          // inclusive scan
            const int add = t >= d ? sc[t - d] : 0;
            __syncthreads();
            sc[t] += add;
            __syncthreads();
        }
        const int off = base + sc[t] - nb;
        for (int k = 0; k < nb; ++k) items[off + k] = make_int2(e, 64 * k);
        __syncthreads();
        if (t == kItemsBlock - 1) base += sc[t];
        __syncthreads();
    }
    if (t == 0) {
        items_total[0] = base;
        // as many face splits as keep (items x splits) within 7/8 of one round of the crossing kernel's grid: a second,
        // mostly empty round cost more than the longer loops of fewer splits (16 splits: 110.3, 12: 112.1 k it/s at batch 64)
        const int fit = base > 0 ? (grid - grid / 8) / base : max_split;
        items_total[1] = fit < 1 ? 1 : (fit > max_split ? max_split : fit);
    }
}

__global__ __launch_bounds__(64) void segment_ray_kernel(
    const float* __restrict__ verts, const float* __restrict__ entries,
    const int2* __restrict__ items, const int32_t* __restrict__ items_total, const int32_t* __restrict__ seg_q_off,
    const int32_t* __restrict__ seg_q_vidx, const int32_t* __restrict__ ent_off,
    const int32_t* __restrict__ count, const int32_t* __restrict__ list, int V, int E,
    int Qs_total, int S, int nsplit, int32_t* __restrict__ partial,     // [B,nsplit,Qs_total] by list position: crossings
    float* __restrict__ partial_half)                                   // same shape: half angles of the boundary cones
{
    const P3 u_dir = {kFanX, kFanY, kFanZ};
    const P3 us = {shear_x(kFanX, kFanZ), shear_y(kFanY, kFanZ), kFanZ};
    __shared__ float sT[kSegRayChunk * 9];
    const int ns = items_total[1];                                      // face splits in use (<= nsplit, the array stride)
    const int units = items_total[0] * ns;                              // (item, face split)
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int2 item = items[unit / ns];
        const int split = unit % ns;
        const int b = item.x / S, s = item.x % S, k_start = item.y;
        const int n = count[item.x];
        const int q_beg = seg_q_off[s];
        const int32_t* mine = list + (size_t)b * Qs_total + q_beg;
        const int k0 = k_start + threadIdx.x;
        const int v0 = seg_q_vidx[q_beg + mine[min(k0, n - 1)]];
        const float* vb = verts + (size_t)b * V * 3;
        const float qz = vb[3 * v0 + 2];
        const float qx = shear_x(vb[3 * v0], qz), qy = shear_y(vb[3 * v0 + 1], qz);
        const int f_seg = ent_off[s], f_cnt = ent_off[s + 1] - f_seg;
        const int per = (f_cnt + ns - 1) / ns;
        const int f_beg = f_seg + split * per, f_end = min(f_seg + f_cnt, f_beg + per);
        int crossings = 0;
        float half_sum = 0.0f;
        for (int chunk = f_beg; chunk < f_end; chunk += kSegRayChunk) {
            const int cn = min(kSegRayChunk, f_end - chunk);
            const float* src = entries + ((size_t)b * E + chunk) * 9;
            __syncthreads();
            for (int i = threadIdx.x; i < cn * 9; i += 64) sT[i] = src[i];
            __syncthreads();
            for (int f = 0; f < cn; ++f) {
                const float* t = sT + f * 9;
                if (t[7] != kConeMarker) {                              // a face (wave-uniform: LDS broadcast)
                    const P3 a = {t[0] - qx, t[1] - qy, t[2] - qz}, bb = {t[3] - qx, t[4] - qy, t[5] - qz}, c = {t[6] - qx, t[7] - qy, t[8] - qz};
                    const float ea = edge_fn(bb, c), eb = edge_fn(c, a), ec = edge_fn(a, bb);
                    const float numz = ea * a.z + eb * bb.z + ec * c.z;
                    const float mn = __builtin_fminf(__builtin_fminf(ea, eb), ec), mx = __builtin_fmaxf(__builtin_fmaxf(ea, eb), ec);
                    int n1 = (int)(__builtin_fminf(mn, numz) > 0.0f) - (int)(__builtin_fmaxf(mx, numz) < 0.0f);
                    // exact ties with a definite depth; the faces around the query (det exactly 0) count 0 above
                    const bool edge_zero = mn * mx == 0.0f;
                    if (__builtin_amdgcn_ballot_w64(edge_zero)) {
                        const bool tie = edge_zero & (numz != 0.0f);
                        if (__builtin_amdgcn_ballot_w64(tie)) {
                            if (tie) n1 = crossing_with_ties<true>(a, bb, c, ea, eb, ec);
                        }
                    }
                    crossings += n1;
                } else {
                    // a boundary edge x -> y: the cone triangle (u, x, y), its apex a direction; edges at the query
                    // itself cancel against the spokes of its star and are left out
                    const P3 bb = {t[0] - qx, t[1] - qy, t[2] - qz}, c = {t[3] - qx, t[4] - qy, t[5] - qz};
                    const bool at_query = ((bb.x == 0.0f) & (bb.y == 0.0f) & (bb.z == 0.0f)) | ((c.x == 0.0f) & (c.y == 0.0f) & (c.z == 0.0f));
                    if (!at_query) {
                        const int mult = (int)t[6];
                        // back from the sheared frame for the angle
                        const P3 pb = {__builtin_fmaf(kShearX, bb.z, bb.x), __builtin_fmaf(kShearY, bb.z, bb.y), bb.z};
                        const P3 pc = {__builtin_fmaf(kShearX, c.z, c.x), __builtin_fmaf(kShearY, c.z, c.y), c.z};
                        int c1;
                        float h1;
                        cone_term(us, bb, c, u_dir, pb, pc, c1, h1);
                        crossings += mult * c1;
                        half_sum += (float)mult * h1;
                    }
                }
            }
        }
        const size_t o = ((size_t)b * nsplit + split) * Qs_total + q_beg;
        if (k0 < n) { partial[o + k0] = crossings; partial_half[o + k0] = half_sum; }
    }
}

// counter s (0..7) of a ray's two packed words: the words are sums of signed per-byte contributions (|total| < 128 per
// byte), so the bytes are peeled off from the low end, each one's sign carried into the rest
__device__ __forceinline__ int seg_leaf_count(const int32_t* words, int s)
{
    int32_t x = words[s >> 2];
    int c = 0;
    for (int k = 0; k <= (s & 3); ++k) {
        c = (int)(int8_t)(x & 255);
        x = (x - c) >> 8;
    }
    return c;
}

// vertices that are NOT exterior to their own segment are re-marked exterior in the body flags
// (losses.py:87-89, loss.py:265-266)
__global__ __launch_bounds__(kBlock) void segment_ray_finalize_kernel(
    const float* __restrict__ verts, const float* __restrict__ caps, const int32_t* __restrict__ partial,
    const float* __restrict__ partial_half,
    const int32_t* __restrict__ seg_of_q, const int32_t* __restrict__ seg_q_off, const int32_t* __restrict__ seg_q_vidx,
    const int32_t* __restrict__ link_off, const int32_t* __restrict__ link,
    const int32_t* __restrict__ count, const int32_t* __restrict__ list,
    const int32_t* __restrict__ leaf_counts,      // [B][slots][2] crossings with the segments' body faces, or nullptr
    const int32_t* __restrict__ vpos, int slots, int V, int K,
    int Qs_total, int S, int nsplit, const int32_t* __restrict__ items_total, float thresh, float* __restrict__ seg_w,
    uint8_t* __restrict__ seg_ext, uint8_t* __restrict__ exterior)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;       // a list position
    const int ns = items_total[1];                         // face splits the crossing kernel used (<= nsplit, the stride)
    // all lanes stay to the end: long link lists below are shared out over the wavefront
    bool active = q < Qs_total;
    const int s = active ? seg_of_q[q] : 0;
    const int k = q - seg_q_off[s];
    active = active && k < count[b * S + s];
    int n = 0;
    float half_sum = 0.0f;
    int qq = 0, v = 0;
    if (active) {
        for (int sp = 0; sp < ns; ++sp) {
            n += partial[((size_t)b * nsplit + sp) * Qs_total + q];
            half_sum += partial_half[((size_t)b * nsplit + sp) * Qs_total + q];
        }
        qq = seg_q_off[s] + list[(size_t)b * Qs_total + q];   // the vertex's slot in the segment tables
        v = seg_q_vidx[qq];
        if (leaf_counts) n += seg_leaf_count(leaf_counts + 2 * ((size_t)b * slots + vpos[v]), s);
    }
    const float* vb = verts + (size_t)b * V * 3;
    const float* cb = caps + (size_t)b * K * 3;
    const float vx = vb[3 * v], vy = vb[3 * v + 1], vz = vb[3 * v + 2];
    const float qx = shear_x(vx, vz), qy = shear_y(vy, vz);
    const P3 u_dir = {kFanX, kFanY, kFanZ};
    const P3 us = {shear_x(kFanX, kFanZ), shear_y(kFanY, kFanZ), kFanZ};
    // cone triangle (apex direction u, x, y) of one link x -> y of a vertex's star
    auto cone = [&](const float* p0, const float* p1, float ax, float ay, float az, float sx, float sy, float& h, int& c) {
        const P3 pb = {p0[0] - ax, p0[1] - ay, p0[2] - az}, pc = {p1[0] - ax, p1[1] - ay, p1[2] - az};
        const P3 sb = {shear_x(p0[0], p0[2]) - sx, shear_y(p0[1], p0[2]) - sy, p0[2] - az};
        const P3 sc = {shear_x(p1[0], p1[2]) - sx, shear_y(p1[1], p1[2]) - sy, p1[2] - az};
        int c1;
        float h1;
        cone_term(us, sb, sc, u_dir, pb, pc, c1, h1);
        h += h1;
        c += c1;
    };
    const int e0 = active ? link_off[qq] : 0, e1 = active ? link_off[qq + 1] : 0;
    constexpr int kLongLinks = 16;
    if (e1 - e0 <= kLongLinks) {
        // two links at a time (their four corner positions are fetched together)
        for (int e = e0; e < e1; e += 2) {
            int id[4];
            float pp[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) id[u] = link[2 * min(e + (u >> 1), e1 - 1) + (u & 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* p = id[u] < V ? vb + 3 * (size_t)id[u] : cb + 3 * (size_t)(id[u] - V);
                pp[u][0] = p[0]; pp[u][1] = p[1]; pp[u][2] = p[2];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (e + h < e1) cone(pp[2 * h], pp[2 * h + 1], vx, vy, vz, qx, qy, half_sum, n);
        }
    }
    // long stars (the poles of a lat-long sphere have 80 links; SMPL has none above 16): one link per lane, sums by
    // butterfly -- one lane walking 80 cones alone held the whole launch up (14 us or 50, depending on whether a pole
    // was interior in that iteration)
    unsigned long long todo = __builtin_amdgcn_ballot_w64(e1 - e0 > kLongLinks);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int le0 = __builtin_amdgcn_readlane(e0, src), le1 = __builtin_amdgcn_readlane(e1, src);
        const float lvx = __shfl(vx, src), lvy = __shfl(vy, src), lvz = __shfl(vz, src);
        const float lqx = __shfl(qx, src), lqy = __shfl(qy, src);
        float h = 0.0f;
        int cr = 0;
        for (int e = le0 + lane; e < le1; e += 64) {
            const int i0 = link[2 * e], i1 = link[2 * e + 1];
            const float* p0 = i0 < V ? vb + 3 * (size_t)i0 : cb + 3 * (size_t)(i0 - V);
            const float* p1 = i1 < V ? vb + 3 * (size_t)i1 : cb + 3 * (size_t)(i1 - V);
            const float a0[3] = {p0[0], p0[1], p0[2]}, a1[3] = {p1[0], p1[1], p1[2]};
            cone(a0, a1, lvx, lvy, lvz, lqx, lqy, h, cr);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { h += __shfl_xor(h, m); cr += __shfl_xor(cr, m); }
        if (lane == src) { half_sum += h; n += cr; }
    }
    if (!active) return;
    const float w = (float)n - half_sum * (0.5f / kPi);
    const size_t o = (size_t)b * Qs_total + qq;
    if (seg_w) seg_w[o] = w;
    if (seg_ext) seg_ext[o] = w <= thresh;
    if (exterior && !(w <= thresh)) exterior[(size_t)b * V + v] = 1;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RayLayout {
    size_t stream, bounds, lists, list_len, zeroed, zeroed_bytes, leaf_cnt, leaf_fill, count, leaf_off, tiles, body, pairs,
        stats, fans, qrec, ranges, total;
    int T, qblocks, cap, max_tiles, workers, columns;
    size_t seg_count;        // per-segment crossing counts of the vertices (models with seg_elem_mask), zeroed with `count`
};

// ---- the segment filter behind the body test (flags only, leaf-assisted models) ------------------------------------------
// cap centroids -> sheared entries -> compaction of the interior vertices -> work items -> crossings -> cones + flags were
// six launches in round 2 (two ahead of the body test, four behind it: ~60 us of the step's serial chain at batch 64 for a
// few hundred interior vertices per body), two in round 3 (segment_cross_kernel: one workgroup per (segment, body, 1/16 of
// the entries), partial sums to global memory; segment_flags_kernel: sums + cones + flags: 48 + 18 us), one since round 4.
constexpr int kSegFusedMaxQ = 4096;      // vertices of one segment (the synthetic head: ~1150)
constexpr int kSegFusedMaxCaps = 8;

// cap centroids of the caps c_lo .. c_hi -> s_caps (segmentation.py:74-76), a wavefront per cap
template <int kWaves>
__device__ __forceinline__ void seg_fused_caps(const float* __restrict__ vb, const int32_t* __restrict__ cap_off,
                                               const int32_t* __restrict__ cap_vidx, int c_lo, int c_hi, float* s_caps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = c_lo + wave; c < c_hi; c += kWaves) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        const int beg = cap_off[c], end = cap_off[c + 1];
        for (int k = beg + lane; k < end; k += 64) {
            const float* p = vb + 3 * cap_vidx[k];
            sx += p[0]; sy += p[1]; sz += p[2];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sx += __shfl_down(sx, o, 64); sy += __shfl_down(sy, o, 64); sz += __shfl_down(sz, o, 64); }
        if (lane == 0) {
            const float inv = 1.0f / (float)(end - beg);
            s_caps[3 * (c - c_lo)] = sx * inv; s_caps[3 * (c - c_lo) + 1] = sy * inv; s_caps[3 * (c - c_lo) + 2] = sz * inv;
        }
    }
}

// ---- ONE launch (round 4) ---------------------------------------------------------------------------------------------
// In the two launches of round 3 every one of the 16 slice workgroups of a (segment, body) compacted the segment's interior
// vertices again, and the slices' sums travelled through global memory to the second launch.  Here kSegOneZ workgroups of eight wavefronts share a (segment, body) by VERTICES, not by entries:
//   A. the segment's interior vertices by the body test's flags (the copy nobody writes), compacted in list order -- the
//      same list in every workgroup; workgroup z owns the groups of 64 of them with index = z (mod kSegOneZ); none -> done;
//   B. the cap centroids;
//   C. ALL entries of the segment (cap faces, boundary edges), posed and sheared 256 at a time in LDS; the wavefronts share
//      out (own group) x (sub-slice of the entries): one group -> eight sub-slices, eight groups -> one each; lane = vertex,
//      entries by LDS broadcast; the sub-slices' sums meet in LDS;
//   D. per own interior vertex: those sums in sub-slice order + the crossings with the segment's body faces the body's own
//      inside test has counted (seg_leaf_count) + the cones of its links -> w; not exterior to its own segment ->
//      exterior[v] = 1 (losses.py:87-89, loss.py:265-266).
// Nothing crosses workgroups: no partial sums in global memory, no arrival counter, no second launch.
// (Measured on the way, solo at batch 64, against 63 us for the two launches: one workgroup per (segment, body) with four
// wavefronts, round 3: 81 us; 128 vertices per workgroup with the body winding numbers recomputed from the crossing counts
// so that nothing waits for the flags: 89 us -- interior vertices are scattered over a segment's list, every chunk that
// holds one walks all entries with a handful of live lanes; one workgroup of SIXTEEN wavefronts: 44 us at every batch size --
// a body whose arm lies in its trunk keeps one compute unit busy with ~80 k instructions; four workgroups sharing the
// ENTRIES with the last arrival finishing (partial sums + fence + ticket per workgroup): 74 us.)
constexpr int kSegOneBlock = 512, kSegOneWaves = kSegOneBlock / 64, kSegOneChunk = 256, kSegOneZ = 8;
constexpr int kSegOneOwn = kSegFusedMaxQ / kSegOneZ;      // interior vertices a workgroup can own (groups of 64 dealt round-robin)
#ifdef TUCH_SEG_CLOCKS
// diagnostic build only (tools/diag/seg_clocks.py): s_memrealtime (100 MHz) of every block at the phase boundaries of
// segment_one_kernel: [block][0] start, [1] interior vertices compacted, [2] caps, [3] entries walked, [4] end
__device__ unsigned long long g_seg_clocks[8192][8];
#define SEG_CLOCK(i) do { if (threadIdx.x == 0) { const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (blk < 8192) g_seg_clocks[blk][i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int tuch_debug_seg_clocks(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seg_clocks), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? TUCH_OK : TUCH_ERR_HIP;
}
#else
#define SEG_CLOCK(i) do { } while (0)
#endif

__global__ __launch_bounds__(kSegOneBlock) void segment_one_kernel(
    const float* __restrict__ verts, const uint8_t* __restrict__ body_flags, uint8_t* __restrict__ exterior,
    const int32_t* __restrict__ seg_q_off, const int32_t* __restrict__ seg_q_vidx,
    const int32_t* __restrict__ cap_range, const int32_t* __restrict__ cap_off, const int32_t* __restrict__ cap_vidx,
    const int32_t* __restrict__ ent_off, const int32_t* __restrict__ ent,
    const int32_t* __restrict__ link_off, const int32_t* __restrict__ link, const int32_t* __restrict__ leaf_counts,
    const int32_t* __restrict__ vpos, int slots, int V, float thresh)
{
    __shared__ int32_t s_list[kSegFusedMaxQ];
    __shared__ int32_t s_cnt[kSegOneOwn];
    __shared__ float s_half[kSegOneOwn];
    __shared__ float sT[kSegOneChunk * 9];
    __shared__ float s_caps[kSegFusedMaxCaps * 3];
    __shared__ int32_t s_wave[kSegOneWaves];
    __shared__ int32_t s_n;
    const int sg = blockIdx.x, b = blockIdx.y, z = blockIdx.z, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* vb = verts + (size_t)b * V * 3;
    uint8_t* eb = exterior + (size_t)b * V;
    const int q_beg = seg_q_off[sg], nq = seg_q_off[sg + 1] - q_beg;
    SEG_CLOCK(0);
    // A.  All of a thread's flags are requested before the first is looked at (the rounds of the scan then only touch LDS).
    {
        const uint8_t* fb = body_flags + (size_t)b * V;
        constexpr int kRounds = kSegFusedMaxQ / kSegOneBlock;
        uint32_t interior = 0;                              // bit r: vertex r * block + t is interior
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int q = r * kSegOneBlock + t;
            if (q < nq && fb[seg_q_vidx[q_beg + q]] == 0) interior |= 1u << r;
        }
        if (t == 0) s_n = 0;
        __syncthreads();
        for (int r = 0; r * kSegOneBlock < nq; ++r) {
            const bool in = (interior >> r) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) s_wave[wave] = __builtin_popcountll(bal);
            __syncthreads();
            int base = s_n;
            for (int w2 = 0; w2 < wave; ++w2) base += s_wave[w2];
            if (in) s_list[base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = r * kSegOneBlock + t;
            __syncthreads();
            if (t == 0) { int add = 0; for (int w2 = 0; w2 < kSegOneWaves; ++w2) add += s_wave[w2]; s_n += add; }
            __syncthreads();
        }
    }
    const int n = s_n;
    const int groups = (n + 63) >> 6;
    const int mine = groups > z ? (groups - z + kSegOneZ - 1) / kSegOneZ : 0;      // own groups: z, z + Z, ...
    SEG_CLOCK(1);
    if (mine == 0) return;
    // B.
    const int c_lo = cap_range[sg];
    seg_fused_caps<kSegOneWaves>(vb, cap_off, cap_vidx, c_lo, cap_range[sg + 1], s_caps);
    SEG_CLOCK(2);
    // C.
    const P3 u_dir = {kFanX, kFanY, kFanZ};
    const P3 us = {shear_x(kFanX, kFanZ), shear_y(kFanY, kFanZ), kFanZ};
    const int e_beg = ent_off[sg], e_end = ent_off[sg + 1];
    const int slices = mine >= kSegOneWaves ? 1 : kSegOneWaves / mine;          // wavefronts per own group
    for (int j0 = 0; j0 < mine; j0 += kSegOneWaves) {                           // (one round unless n > 1024)
        const int j = slices > 1 ? wave / slices : j0 + wave;                    // own group, sub-slice of the entries
        const int sl = slices > 1 ? wave % slices : 0;
        const bool have = j < mine && (slices == 1 || wave < mine * slices);
        const int k = (z + (have ? j : 0) * kSegOneZ) * 64 + lane;               // position in the compacted list
        const int v0 = seg_q_vidx[q_beg + s_list[min(k, n - 1)]];
        const float qz = vb[3 * (size_t)v0 + 2];
        const float qx = shear_x(vb[3 * (size_t)v0], qz), qy = shear_y(vb[3 * (size_t)v0 + 1], qz);
        int crossings = 0;
        float half_sum = 0.0f;
        for (int chunk = e_beg; chunk < e_end; chunk += kSegOneChunk) {
            const int cn = min(kSegOneChunk, e_end - chunk);
            __syncthreads();
            for (int i = t; i < cn; i += kSegOneBlock) {
                const int id[3] = {ent[3 * (size_t)(chunk + i)], ent[3 * (size_t)(chunk + i) + 1], ent[3 * (size_t)(chunk + i) + 2]};
                float* dst = sT + i * 9;
                const int corners = id[2] < 0 ? 2 : 3;
                for (int c = 0; c < corners; ++c) {
                    const float* p = id[c] < V ? vb + 3 * (size_t)id[c] : s_caps + 3 * (id[c] - V - c_lo);
                    dst[3 * c] = shear_x(p[0], p[2]); dst[3 * c + 1] = shear_y(p[1], p[2]); dst[3 * c + 2] = p[2];
                }
                if (corners == 2) { dst[6] = (float)id[2]; dst[7] = kConeMarker; dst[8] = 0.0f; }
            }
            __syncthreads();
            if (!have) continue;
            for (int f = sl; f < cn; f += slices) {
                const float* e = sT + f * 9;
                if (e[7] != kConeMarker) {                              // a cap face (wave-uniform: LDS broadcast)
                    const P3 a = {e[0] - qx, e[1] - qy, e[2] - qz}, bb = {e[3] - qx, e[4] - qy, e[5] - qz}, c = {e[6] - qx, e[7] - qy, e[8] - qz};
                    const float ea = edge_fn(bb, c), eb2 = edge_fn(c, a), ec = edge_fn(a, bb);
                    const float numz = ea * a.z + eb2 * bb.z + ec * c.z;
                    const float mn = __builtin_fminf(__builtin_fminf(ea, eb2), ec), mx = __builtin_fmaxf(__builtin_fmaxf(ea, eb2), ec);
                    int n1 = (int)(__builtin_fminf(mn, numz) > 0.0f) - (int)(__builtin_fmaxf(mx, numz) < 0.0f);
                    const bool edge_zero = mn * mx == 0.0f;
                    if (__builtin_amdgcn_ballot_w64(edge_zero)) {
                        const bool tie = edge_zero & (numz != 0.0f);
                        if (__builtin_amdgcn_ballot_w64(tie)) {
                            if (tie) n1 = crossing_with_ties<true>(a, bb, c, ea, eb2, ec);
                        }
                    }
                    crossings += n1;
                } else {                                                // a boundary edge x -> y: the cone (u, x, y)
                    const P3 bb = {e[0] - qx, e[1] - qy, e[2] - qz}, c = {e[3] - qx, e[4] - qy, e[5] - qz};
                    const bool at_query = ((bb.x == 0.0f) & (bb.y == 0.0f) & (bb.z == 0.0f)) | ((c.x == 0.0f) & (c.y == 0.0f) & (c.z == 0.0f));
                    if (!at_query) {
                        const int mult = (int)e[6];
                        const P3 pb = {__builtin_fmaf(kShearX, bb.z, bb.x), __builtin_fmaf(kShearY, bb.z, bb.y), bb.z};
                        const P3 pc = {__builtin_fmaf(kShearX, c.z, c.x), __builtin_fmaf(kShearY, c.z, c.y), c.z};
                        int c1;
                        float h1;
                        cone_term(us, bb, c, u_dir, pb, pc, c1, h1);
                        crossings += mult * c1;
                        half_sum += (float)mult * h1;
                    }
                }
            }
        }
        // sums: [sub-slice][own group][lane] with several sub-slices (mine x slices <= 8 wavefronts), else [own group][lane]
        if (have) {
            const int at = slices > 1 ? (sl * mine + j) * 64 + lane : j * 64 + lane;
            s_cnt[at] = crossings;
            s_half[at] = half_sum;
        }
        if (slices > 1) break;
    }
    __syncthreads();
    SEG_CLOCK(3);
    // D. one thread per own interior vertex (all lanes of a wavefront stay: long link lists are shared out over it)
    auto pos = [&](int id, float (&o)[3]) {                // a vertex id of the segment tables (>= V: cap vertex)
        const float* p = id < V ? vb + 3 * (size_t)id : s_caps + 3 * (id - V - c_lo);
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    };
    for (int j = wave; j < mine; j += kSegOneWaves) {       // (wave-uniform)
        const int k = (z + j * kSegOneZ) * 64 + lane;
        const bool active = k < n;
        const int qq = q_beg + s_list[active ? k : 0];
        const int v = seg_q_vidx[qq];
        int cnt = 0;
        float half_sum = 0.0f;
        if (active) {
            if (slices > 1) {
                for (int s2 = 0; s2 < slices; ++s2) { cnt += s_cnt[(s2 * mine + j) * 64 + lane]; half_sum += s_half[(s2 * mine + j) * 64 + lane]; }
            } else {
                cnt = s_cnt[j * 64 + lane];
                half_sum = s_half[j * 64 + lane];
            }
            cnt += seg_leaf_count(leaf_counts + 2 * ((size_t)b * slots + vpos[v]), sg);
        }
        const float vx = vb[3 * (size_t)v], vy = vb[3 * (size_t)v + 1], vz = vb[3 * (size_t)v + 2];
        const float qx = shear_x(vx, vz), qy = shear_y(vy, vz);
        auto cone = [&](const float (&p0)[3], const float (&p1)[3], float ax, float ay, float az, float sx, float sy, float& h, int& c) {
            const P3 pb = {p0[0] - ax, p0[1] - ay, p0[2] - az}, pc = {p1[0] - ax, p1[1] - ay, p1[2] - az};
            const P3 sb = {shear_x(p0[0], p0[2]) - sx, shear_y(p0[1], p0[2]) - sy, p0[2] - az};
            const P3 sc = {shear_x(p1[0], p1[2]) - sx, shear_y(p1[1], p1[2]) - sy, p1[2] - az};
            int c1;
            float h1;
            cone_term(us, sb, sc, u_dir, pb, pc, c1, h1);
            h += h1;
            c += c1;
        };
        const int e0 = active ? link_off[qq] : 0, e1 = active ? link_off[qq + 1] : 0;
        constexpr int kLongLinks = 16;
        if (e1 - e0 <= kLongLinks) {
            for (int e = e0; e < e1; e += 2) {             // two links at a time: their four corners are fetched together
                int id[4];
                float pp[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) id[u] = link[2 * min(e + (u >> 1), e1 - 1) + (u & 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) pos(id[u], pp[u]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (e + h < e1) cone(pp[2 * h], pp[2 * h + 1], vx, vy, vz, qx, qy, half_sum, cnt);
            }
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(e1 - e0 > kLongLinks);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int le0 = __builtin_amdgcn_readlane(e0, src), le1 = __builtin_amdgcn_readlane(e1, src);
            const float lvx = __shfl(vx, src), lvy = __shfl(vy, src), lvz = __shfl(vz, src);
            const float lqx = __shfl(qx, src), lqy = __shfl(qy, src);
            float h = 0.0f;
            int cr = 0;
            for (int e = le0 + lane; e < le1; e += 64) {
                float a0[3], a1[3];
                pos(link[2 * e], a0);
                pos(link[2 * e + 1], a1);
                cone(a0, a1, lvx, lvy, lvz, lqx, lqy, h, cr);
            }
#pragma unroll
            for (int m2 = 32; m2 >= 1; m2 >>= 1) { h += __shfl_xor(h, m2); cr += __shfl_xor(cr, m2); }
            if (lane == src) { half_sum += h; cnt += cr; }
        }
        if (active) {
            const float w = (float)cnt - half_sum * (0.5f / kPi);
            if (!(w <= thresh)) eb[v] = 1;
        }
    }
    SEG_CLOCK(4);
}

}  // namespace

// can the segment filter run as the one launch above?  (flags only; the caller checks that)
bool tuch_ray_segment_fused_available(const tuch_contact_model* m)
{
    return m && m->seg_cap_off && m->seg_cap_range && m->seg_elem_mask && m->seg_q_max <= kSegFusedMaxQ && m->opt.seg_fused != 0;
}

// The segment filter as ONE launch (segment_one_kernel): kSegOneZ workgroups per (segment, body) read body_flags (the body
// test's own flags: a copy nobody writes during the call) and write ones into `exterior` [B,V]; leaf_counts =
// tuch_ray_segment_counts of the SAME vertices.
int tuch_ray_segment_flags_one(const tuch_contact_model* m, const float* verts, const uint8_t* body_flags,
                               const int32_t* leaf_counts, int B, float thresh, uint8_t* exterior, hipStream_t s)
{
    hipLaunchKernelGGL(segment_one_kernel, dim3(m->num_segments, B, kSegOneZ), dim3(kSegOneBlock), 0, s, verts, body_flags, exterior,
                       (const int32_t*)m->seg_q_off, (const int32_t*)m->seg_q_vidx, (const int32_t*)m->seg_cap_range,
                       (const int32_t*)m->cap_off, (const int32_t*)m->cap_vidx, (const int32_t*)m->seg_cap_off,
                       (const int32_t*)m->seg_cap_ent, (const int32_t*)m->seg_link_off, (const int32_t*)m->seg_link, leaf_counts,
                       (const int32_t*)m->seg_vpos, 2 * m->tree_qblocks * kRayQueries, m->V, thresh);
    return tuch_check_launch("tuch_ray_segment_flags_one");
}

bool tuch_ray_available(const tuch_contact_model* m)
{
    if (!m || m->tree_nodes <= 0 || !m->ring_off || m->tree_exact_len <= 0) return false;
    // ray_near_kernel's records hold the leaf in 16 bits (2 M faces; a mesh beyond that takes the solid-angle form)
    if (m->tree_leaves > 65535) return false;
    return m->opt.winding_ray != 0;
}

// room in the pair list, in (ray, leaf) pairs per query (a ray passes the slabs of ~4-5 leaves; option ray_pair_cap)
static int ray_pair_cap(const tuch_contact_model* m) { return m->opt.ray_pair_cap < 1 ? 1 : m->opt.ray_pair_cap; }

static RayLayout full_layout(const tuch_contact_model* m, int B, int Q, bool verts)
{
    RayLayout l;
    l.qblocks = verts ? 2 * m->tree_qblocks : ceil_div(Q, kRayQueries);
    // leaf strips are the first part of the tree stream; the caps behind them are never read
    l.T = ceil_div(m->tree_exact_len, 3) * 3 + 6;
    const int L = m->tree_leaves;
    const long cap = (long)ray_pair_cap(m) * Q;
    l.cap = (int)(cap < 0x3fffffffL ? cap : 0x3fffffffL);
    // tiles of a body: its pairs in 64s + one ragged tile per leaf; never fewer than the block-major fallback needs
    l.max_tiles = l.cap / 64 + L;
    if (l.max_tiles < l.qblocks * kFallbackChunks) l.max_tiles = l.qblocks * kFallbackChunks;
    // one column of wavefronts per XCD; four times what the chip holds at once (256 CUs x 4 SIMDs x 8), so that a
    // wavefront's share is about one tile and the hardware balances the rest (TUCH_RAY_WAVES)
    l.columns = B < 8 ? B : 8;
    const int waves = m->opt.ray_waves > 0 ? m->opt.ray_waves : 32768;
    l.workers = waves / l.columns > 0 ? waves / l.columns : 1;
    size_t o = 0;
    l.stream = tuch_ws_take(o, (size_t)B * l.T * sizeof(RayElem));
    l.bounds = tuch_ws_take(o, (size_t)B * m->tree_nodes * 2 * kSlabStride * sizeof(float));
    l.lists = tuch_ws_take(o, (size_t)B * l.qblocks * L * sizeof(RayEntry));
    l.list_len = tuch_ws_take(o, (size_t)B * l.qblocks * sizeof(int32_t));
    l.zeroed = o;                                                  // one memset: rays per leaf, fill cursors, crossing counts
    // (the span ray_stream_kernel clears in one go stays contiguous: no guards inside it, one behind it)
    l.leaf_cnt = tuch_ws_take(o, (size_t)B * L * sizeof(int32_t), false);
    l.leaf_fill = tuch_ws_take(o, (size_t)B * L * sizeof(int32_t), false);
    l.count = tuch_ws_take(o, (size_t)B * l.qblocks * kRayQueries * sizeof(int32_t), false);
    l.seg_count = o;
    if (verts && m->seg_elem_mask) o += align256(2 * (size_t)B * l.qblocks * kRayQueries * sizeof(int32_t));
    l.zeroed_bytes = o - l.zeroed;
    (void)tuch_ws_take(o, 0);
    l.leaf_off = tuch_ws_take(o, (size_t)B * L * sizeof(int32_t));
    l.tiles = tuch_ws_take(o, (size_t)B * l.max_tiles * sizeof(RayTile));
    l.body = tuch_ws_take(o, (size_t)B * sizeof(RayBody));
    l.pairs = tuch_ws_take(o, (size_t)B * l.cap * sizeof(int32_t));
    l.stats = tuch_ws_take(o, 256);
    l.fans = verts ? tuch_ws_take(o, (size_t)B * l.qblocks * kRayQueries * sizeof(float2)) : 0;
    l.qrec = verts ? tuch_ws_take(o, (size_t)B * l.qblocks * kRayQueries * sizeof(QRec)) : 0;
    l.ranges = verts ? tuch_ws_take(o, (size_t)B * l.qblocks * 4 * sizeof(float4)) : 0;
    l.total = o;
    return l;
}

void tuch_ray_layout_touch(const tuch_contact_model* m, int B, int Q)
{
    (void)full_layout(m, B, Q > 0 ? Q : m->V, Q <= 0);
}

size_t tuch_ray_workspace_bytes(const tuch_contact_model* m, int B, int Q)
{
    if (!m || m->tree_nodes <= 0 || B <= 0) return 0;
    const size_t a = full_layout(m, B, m->V, true).total;
    const size_t b = Q > 0 ? full_layout(m, B, Q, false).total : 0;
    return a > b ? a : b;
}

// where the vertices' closing fans are computed (option ray_fans): 1 by workgroups of the chain's first launch, 2 of
// ray_near_kernel's, 0 (and meshes without rings, or whose leaf runs do not tile the stream) inside the finalize kernel
static bool fans_in_bounds_launch(const tuch_contact_model* m) { return m->ring_off && m->opt.ray_fans == 1 && m->tree_leaf_runs_tile; }
static bool cross_in_one_launch(const tuch_contact_model* m);
static bool fans_in_near_launch(const tuch_contact_model* m) { return m->ring_off && m->opt.ray_fans == 2 && !cross_in_one_launch(m); }

// the vertices' inside test with its middle as one launch (ray_cross_kernel; option ray_cross, default on): slots must fit
// the 24-bit field of QRec::tag
static bool cross_in_one_launch(const tuch_contact_model* m)
{
    return m->opt.ray_cross != 0 && 2L * m->tree_qblocks * kRayQueries < kNoSlot;
}

// sheared leaf strips and the slabs of every LEAF (inner nodes are not used by the flat near test)
static void launch_ray_boxes(const tuch_contact_model* m, const RayLayout& l, const float* verts, int B, char* ws, hipStream_t s,
                             bool one_launch, bool query_side = false)
{
    RayElem* st = (RayElem*)(ws + l.stream);
    float* bounds = (float*)(ws + l.bounds);
    const RayPre pre{query_side ? ceil_div(l.qblocks, kBoundsBlock / 64) : 0, l.qblocks, (QRec*)(ws + l.qrec), (float4*)(ws + l.ranges),
                     (const int32_t*)m->seg_vmask};
    // one launch: every leaf poses its own run of the strip.  (Not for the points form with its far larger span of
    // counters to clear -- [B][Q] with Q = all HD points: the leaf grid has too few workgroups for that, 100 us)
    if (one_launch && m->tree_leaf_runs_tile) {
        const int leaf_blocks = ceil_div(m->tree_leaves, kBoundsBlock / 16);
        const bool fans = fans_in_bounds_launch(m);
        hipLaunchKernelGGL(ray_leaf_bounds_kernel<true>, dim3(leaf_blocks + (fans ? ceil_div(m->V, kBoundsBlock) : 0) + pre.blocks, B),
                           dim3(kBoundsBlock), 0,
                           s, (const RayElem*)st, l.T, (const TreeNode*)m->tree_node, m->tree_nodes,
                           (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, bounds, verts,
                           (const int32_t*)m->tree_vidx, (const float*)m->tree_sign_word, m->V, m->tree_exact_len, st,
                           (uint4*)(ws + l.zeroed), l.zeroed_bytes / sizeof(uint4), fans ? leaf_blocks : 0,
                           (const int32_t*)m->tree_qperm, (const int32_t*)m->ring_off, (const int32_t*)m->ring_vidx,
                           (float2*)(ws + l.fans), l.qblocks * kRayQueries, pre);
        return;
    }
    if (query_side)
        hipLaunchKernelGGL(ray_prepass_kernel, dim3(pre.blocks, B), dim3(kBoundsBlock), 0, s, verts, (const int32_t*)m->tree_vidx,
                           (const int32_t*)m->tree_qperm, m->V, pre);
    hipLaunchKernelGGL(ray_stream_kernel, dim3(ceil_div(l.T, kBlock), B), dim3(kBlock), 0, s, verts,
                       (const int32_t*)m->tree_vidx, (const float*)m->tree_sign_word, m->V, m->tree_exact_len, l.T, st,
                       (uint4*)(ws + l.zeroed), l.zeroed_bytes / sizeof(uint4));
    hipLaunchKernelGGL(ray_leaf_bounds_kernel<false>, dim3(ceil_div(m->tree_leaves, kBoundsBlock / 16), B), dim3(kBoundsBlock), 0, s,
                       (const RayElem*)st, l.T, (const TreeNode*)m->tree_node, m->tree_nodes,
                       (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, bounds,
                       (const float*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, 0, 0, (RayElem*)nullptr,
                       (uint4*)nullptr, (size_t)0);
}

// near leaves -> rays per leaf -> tiles -> crossing counts (count[b][slot])
template <bool kVerts>
static int launch_ray_counts(const tuch_contact_model* m, const RayLayout& l, const float* verts, const float* queries,
                             const int32_t* counts, int B, int Q, char* ws, hipStream_t s, unsigned long long* stats)
{
    const bool one = kVerts && !stats && cross_in_one_launch(m);
    launch_ray_boxes(m, l, verts, B, ws, s, kVerts, one);  // clears the counters (l.zeroed) as well
    const int L = m->tree_leaves;
    const TreeNode* nodes = (const TreeNode*)m->tree_node;
    if (one) {
        const int columns = B < 8 ? B : 8;
        // wavefronts per leaf: a leaf through the trunk meets five times the average number of rays -- one wavefront's chain
        // of ~20 trips and 6 walks is what a small batch waits for (option ray_cross_split, 0: by batch size)
        int nsplit = m->opt.ray_cross_split > 0 ? m->opt.ray_cross_split : (B <= 8 ? 4 : B <= 32 ? 2 : 1);
        if (nsplit > l.qblocks) nsplit = l.qblocks;
        const dim3 grid(columns, ceil_div(B, columns) * L * nsplit);
        int32_t* seg_cnt = (int32_t*)(ws + l.seg_count);
        if (m->seg_elem_mask)
            hipLaunchKernelGGL(ray_cross_kernel<true>, grid, dim3(64), 0, s, (const RayElem*)(ws + l.stream), nodes, (const float*)(ws + l.bounds), L,
                               nsplit, (const QRec*)(ws + l.qrec), (const float4*)(ws + l.ranges), l.T, l.qblocks, B,
                               (int32_t*)(ws + l.count), seg_cnt);
        else
            hipLaunchKernelGGL(ray_cross_kernel<false>, grid, dim3(64), 0, s, (const RayElem*)(ws + l.stream), nodes, (const float*)(ws + l.bounds), L,
                               nsplit, (const QRec*)(ws + l.qrec), (const float4*)(ws + l.ranges), l.T, l.qblocks, B,
                               (int32_t*)(ws + l.count), seg_cnt);
        return TUCH_OK;
    }
    // the leaves are the height-0 entries of the tree's height table (tree_height_off_host[0] == 0)
    const int32_t* leaf_nodes = (const int32_t*)m->tree_height_nodes;
    const int32_t* qperm = kVerts ? (const int32_t*)m->tree_qperm : nullptr;
    RayEntry* lists = (RayEntry*)(ws + l.lists);
    int32_t* list_len = (int32_t*)(ws + l.list_len);
    int32_t* leaf_cnt = (int32_t*)(ws + l.leaf_cnt);
    int32_t* leaf_off = (int32_t*)(ws + l.leaf_off);
    RayTile* tiles = (RayTile*)(ws + l.tiles);
    RayBody* body = (RayBody*)(ws + l.body);
    int32_t* pairs = (int32_t*)(ws + l.pairs);
    // four wavefronts per query block while the blocks alone do not fill the chip (see the kernel)
    // vertices of a closed manifold (rings): their closing fans ride in this launch, as workgroups behind the query blocks
    const bool fans = kVerts && fans_in_near_launch(m);
    float2* fan_out = fans ? (float2*)(ws + l.fans) : nullptr;
    if ((long)l.qblocks * B <= 2048)
        hipLaunchKernelGGL((ray_near_kernel<kVerts, 4>), dim3(l.qblocks + (fans ? ceil_div(Q, 256) : 0), B), dim3(256), 0, s, queries, nodes,
                           (const float*)(ws + l.bounds), m->tree_nodes, L, qperm, counts, Q, l.qblocks, lists, list_len,
                           leaf_cnt, stats, (const int32_t*)m->ring_off, (const int32_t*)m->ring_vidx, fan_out);
    else
        hipLaunchKernelGGL((ray_near_kernel<kVerts, 1>), dim3(l.qblocks + (fans ? ceil_div(Q, 64) : 0), B), dim3(64), 0, s, queries, nodes,
                           (const float*)(ws + l.bounds), m->tree_nodes, L, qperm, counts, Q, l.qblocks, lists, list_len,
                           leaf_cnt, stats, (const int32_t*)m->ring_off, (const int32_t*)m->ring_vidx, fan_out);
    const size_t tf_lds = (2 * (size_t)L + l.qblocks + 1) * sizeof(int32_t);
    if (l.qblocks <= kFillMaxBlocks && tf_lds <= 48u * 1024)
        // (one-wave workgroups per body: 64, more for small batches -- the kernel is a chain of dependent steps per workgroup, and a
        // body's ~5000 list entries over 256 workgroups are one pass each: batch 8 0.180 -> 0.173 ms per step, batch 16 0.203 -> 0.199,
        // batch 32 / 64 unchanged / slower with more)
        hipLaunchKernelGGL(ray_tiles_fill_kernel, dim3(B, std::min(256, std::max(kFillSplit, 2048 / B))), dim3(kTilesFillBlock), tf_lds, s, (const int32_t*)leaf_cnt,
                           nodes, leaf_nodes, L, l.cap, l.max_tiles, l.qblocks * kFallbackChunks, tiles, body,
                           (const RayEntry*)lists, (const int32_t*)list_len, l.qblocks, (int32_t*)(ws + l.leaf_fill), pairs, L);
    else {
    hipLaunchKernelGGL(ray_tiles_kernel, dim3(B), dim3(kTilesBlock), 0, s, (const int32_t*)leaf_cnt, nodes, leaf_nodes, L, l.cap,
                       l.max_tiles, l.qblocks * kFallbackChunks, leaf_off, tiles, body);
    hipLaunchKernelGGL(ray_fill_kernel, dim3(l.qblocks, B), dim3(64), 0, s, (const RayEntry*)lists, (const int32_t*)list_len,
                       (const RayBody*)body, (const int32_t*)leaf_off, L, l.qblocks, l.cap, (int32_t*)(ws + l.leaf_fill), pairs);
    }
    const dim3 grid(l.columns, l.workers);
    const bool seg = kVerts && m->seg_elem_mask;
    const int32_t* vmask = (const int32_t*)m->seg_vmask;
    int32_t* seg_count = (int32_t*)(ws + l.seg_count);
#define TUCH_LAUNCH_RAY_LEAF(COUNT, SEG)                                                                                      \
    hipLaunchKernelGGL((ray_leaf_kernel<kVerts, COUNT, SEG>), grid, dim3(64), 0, s, queries, (const RayElem*)(ws + l.stream),  \
                       (const RayTile*)tiles, (const RayBody*)body, (const int32_t*)pairs, (const RayEntry*)lists,            \
                       (const int32_t*)list_len, nodes, L, qperm, counts, Q, l.T, l.qblocks, B, l.cap, l.max_tiles,           \
                       (int32_t*)(ws + l.count), stats, vmask, seg_count)
    if (stats) {
        if (seg) TUCH_LAUNCH_RAY_LEAF(true, kVerts); else TUCH_LAUNCH_RAY_LEAF(true, false);
    } else {
        if (seg) TUCH_LAUNCH_RAY_LEAF(false, kVerts); else TUCH_LAUNCH_RAY_LEAF(false, false);
    }
#undef TUCH_LAUNCH_RAY_LEAF
    return TUCH_OK;
}

int tuch_ray_exterior_verts(const tuch_contact_model* m, const float* verts, int B, float thresh, uint8_t* exterior,
                            float* w, void* workspace, hipStream_t s, unsigned long long* stats_host, uint8_t* exterior_copy)
{
    const RayLayout l = full_layout(m, B, m->V, true);
    char* ws = (char*)workspace;
    unsigned long long* stats = stats_host ? (unsigned long long*)(ws + l.stats) : nullptr;
    if (stats && hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), s) != hipSuccess) return TUCH_ERR_HIP;
    const int rc = launch_ray_counts<true>(m, l, verts, verts, nullptr, B, m->V, ws, s, stats);
    if (rc != TUCH_OK) return rc;
    if (w || exterior)
        hipLaunchKernelGGL(ray_finalize_verts_kernel, dim3(ceil_div(m->V, kBlock), B), dim3(kBlock), 0, s, verts,
                           (const int32_t*)(ws + l.count), (const int32_t*)m->tree_qperm, (const int32_t*)m->ring_off,
                           (const int32_t*)m->ring_vidx, m->V, l.qblocks * kRayQueries, thresh, w, exterior, exterior_copy,
                           (const float2*)(fans_in_bounds_launch(m) || fans_in_near_launch(m) ? ws + l.fans : nullptr));
    if (stats_host) {
        RayBody bodies[8];
        const int nb = B < 8 ? B : 8;
        if (hipMemcpyAsync(stats_host, stats, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(bodies, ws + l.body, nb * sizeof(RayBody), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return TUCH_ERR_HIP;
        unsigned long long tiles = 0;
        for (int b = 0; b < nb; ++b) tiles += bodies[b].tiles;
        stats_host[3] = tiles * B / nb;            // wavefront tiles (extrapolated from the first bodies)
    }
    return tuch_check_launch("tuch_ray_exterior_verts");
}

const int32_t* tuch_ray_segment_counts(const tuch_contact_model* m, int B, const void* workspace)
{
    if (!m->seg_elem_mask) return nullptr;
    return (const int32_t*)((const char*)workspace + full_layout(m, B, m->V, true).seg_count);
}

int tuch_ray_exterior_points(const tuch_contact_model* m, const float* verts, const float* points, const int32_t* counts,
                             int B, int Q, float thresh, uint8_t* exterior, float* w, void* workspace, hipStream_t s)
{
    const RayLayout l = full_layout(m, B, Q, false);
    char* ws = (char*)workspace;
    const int rc = launch_ray_counts<false>(m, l, verts, points, counts, B, Q, ws, s, nullptr);
    if (rc != TUCH_OK) return rc;
    hipLaunchKernelGGL(ray_finalize_points_kernel, dim3(ceil_div(Q, kBlock), B), dim3(kBlock), 0, s,
                       (const int32_t*)(ws + l.count), counts, Q, l.qblocks * kRayQueries, thresh, w, exterior);
    return tuch_check_launch("tuch_ray_exterior_points");
}

// The segment filter (winding.hip: caps, compacted interior vertices per (body, segment) in seg_count / seg_list) by
// ray crossings.  leaf_counts (tuch_ray_segment_counts of the SAME verts, or nullptr): the crossings with the body faces
// of the segments are taken from there and only cap faces and boundary edges are walked here.
// Scratch of the caller: seg_entries [B,seg_ray_total,9] floats filled by tuch_ray_segment_prepare (same `assisted`), seg_partial
// (two partial arrays + the work items).
// the part that needs only the vertices and the cap centroids (the caller runs it ahead of the body's inside test, off the
// critical chain): the posed, sheared entries.  assisted as in tuch_ray_segment_flags (leaf counts will be available).
void tuch_ray_segment_prepare(const tuch_contact_model* m, const float* verts, const float* caps, int assisted, int B,
                              float* seg_entries, hipStream_t s)
{
    const bool as = assisted && m->seg_cap_off;
    const int E = as ? m->seg_cap_total : m->seg_ray_total;
    hipLaunchKernelGGL(segment_shear_entries_kernel, dim3(ceil_div(E > 0 ? E : 1, kBlock), B), dim3(kBlock), 0, s,
                       verts, caps, (const int32_t*)(as ? m->seg_cap_ent : m->seg_ray_ent), m->V, m->num_caps, E, seg_entries);
}

int tuch_ray_segment_flags(const tuch_contact_model* m, const float* verts, const float* caps, const int32_t* seg_count,
                           const int32_t* seg_list, const int32_t* leaf_counts, int B, int nsplit, float thresh,
                           float* seg_entries, int32_t* seg_partial, float* seg_w, uint8_t* seg_ext, uint8_t* exterior,
                           hipStream_t s)
{
    const bool assisted = leaf_counts && m->seg_cap_off;
    const int E = assisted ? m->seg_cap_total : m->seg_ray_total;
    const int32_t* ent_off = (const int32_t*)(assisted ? m->seg_cap_off : m->seg_ray_off);
    float* partial_half = (float*)(seg_partial + (size_t)B * nsplit * m->seg_q_total);
    // work items behind the two partial arrays (B * num_seg_blocks int2 + 1 counter; the caller sizes seg_partial for it)
    int2* items = (int2*)(partial_half + (size_t)B * nsplit * m->seg_q_total);
    int32_t* items_total = (int32_t*)(items + (size_t)B * m->num_seg_blocks);
    const long worst = (long)B * m->num_seg_blocks * nsplit;
    const int grid = (int)(worst < 8192 ? worst : 8192);
    hipLaunchKernelGGL(segment_items_kernel, dim3(1), dim3(kItemsBlock), 0, s, seg_count, B * m->num_segments, items, items_total,
                       nsplit, grid);
    hipLaunchKernelGGL(segment_ray_kernel, dim3((unsigned)grid), dim3(64), 0, s, verts,
                       (const float*)seg_entries, (const int2*)items, (const int32_t*)items_total, (const int32_t*)m->seg_q_off,
                       (const int32_t*)m->seg_q_vidx, ent_off, seg_count, seg_list, m->V, E,
                       m->seg_q_total, m->num_segments, nsplit, seg_partial, partial_half);
    hipLaunchKernelGGL(segment_ray_finalize_kernel, dim3(ceil_div(m->seg_q_total, kBlock), B), dim3(kBlock), 0, s, verts, caps,
                       (const int32_t*)seg_partial, (const float*)partial_half, (const int32_t*)m->seg_of_q,
                       (const int32_t*)m->seg_q_off, (const int32_t*)m->seg_q_vidx, (const int32_t*)m->seg_link_off,
                       (const int32_t*)m->seg_link, seg_count, seg_list, assisted ? leaf_counts : nullptr,
                       (const int32_t*)m->seg_vpos, 2 * m->tree_qblocks * kRayQueries, m->V,
                       m->num_caps, m->seg_q_total, m->num_segments, nsplit, (const int32_t*)items_total, thresh, seg_w, seg_ext,
                       exterior);
    return tuch_check_launch("tuch_ray_segment_flags");
}
