// The HD-mesh branch of RegressorLoss.contact_loss (tuch/train/loss.py:274-301) as one device pipeline.
//
// Reference, per valid body: vertices in contact or inside (:278) -> faces touching them (:279-280, a
// [3F x n_c] boolean broadcast) -> HD points sampled from those faces (:281, an [N_hd x n_f] broadcast) ->
// positions = dense regressor rows x vertices (:285) -> masked nearest HD point (:288-291, n x n) -> inside test
// of the points moved 1 mm along their face normals (:295-297, n x F x 3 x 3) -> pull / push terms (:299-315).
//
// Here: a persistent tuch_hd_model holds the regressor as its three non-zeros per row, the points sorted by the
// surface patch (cluster-tree leaf) of their face, and a CSR table vertex -> the HD points it supports (for the adjoint).  Every call is a fixed sequence of kernels over
// buffers of the worst-case size (every HD point selected), the actual counts stay on the device: no host
// synchronisation, no allocation, capturable in a hipGraph.
//   hd_select   flags per face -> selected points of each body compacted in order (chunk counts + scan) + slot table
//   hd_points   positions, positions + 1 mm normal, mask ids of the selected points
//   search      v2v_indexed_kernel (v2v.hip) with per-body counts
//   inside      tuch_winding_points: integer ray-crossing counts (ray_winding.hip), or the solid-angle walk
//   hd_terms    pull / push sums per body, fixed-order reduction
// Adjoint: point gradients (the partner side through float atomics on the points, low contention) and then a
// GATHER per vertex over the CSR table -- no atomics on the vertices, where ~18 point corners meet.
#include "common.h"
#include "model.h"
#include "workspace.h"
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <numeric>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct tuch_hd_model {
    const tuch_contact_model* cm;
    int N, V, F;
    int K;               // non-zeros per regressor row (3: barycentric samples; up to 8 for a general sparse regressor)
    int32_t* idx;        // [N,K] supporting vertices (sorted order; rows with fewer non-zeros padded with weight 0)
    float* w;            // [N,K] weights
    int32_t* face;       // [N] face the point was sampled from (loss.py:87 faces_vert_is_sampled_from)
    int32_t* tv;         // [N] template vertex = first vertex of that face (loss.py:88 geovec_verts)
    int32_t* mask_id;    // [N] row / column of the geodesic mask the point inherits: tv, or its position in tree order
    int32_t* orig;       // [N] index of the point in the caller's order
    int32_t* by_orig;    // [N] inverse: where the caller's point r sits in the sorted order
    int32_t* v_off;      // [V+1] CSR: vertex -> entries (point * 8 + corner), non-zero weights only
    int32_t* v_ent;
    int32_t* offsets;    // [kMaxBatch+1] = b * N (device): where body b's slots start
    int tree_order;      // mask ids are tree positions (the model's mask in tree order is used)
    std::vector<int32_t>* order_host;   // sorted -> original index
    // the search of the selected points and their inside test only share their input: the inside test runs on a stream of
    // the model's own beside the search (option hd_overlap; fork / join through events, capturable)
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    std::mutex* enqueue;                // one forward call at a time enqueues on `side`
};

namespace {

constexpr int kMaxBatch = 4096;
constexpr int kSel = 1024;

template <typename T>
int upload(T** dst, const T* src, size_t count)
{
    return tuch_table_upload((void**)dst, src, count * sizeof(T));
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- selection (loss.py:278-281) -------------------------------------------------------------------------------
// candidate vertices -> face flags; then the selected HD points of every body compacted in their order: chunks of
// kSel points are counted, the chunk counts prefix-summed by every chunk for itself (one wavefront, a count per lane), and
// the points scattered to their slots
__global__ __launch_bounds__(256) void hd_face_flags_kernel(
    const uint8_t* __restrict__ exterior, const float* __restrict__ min_d2, const uint8_t* __restrict__ valid,
    const int32_t* __restrict__ faces, int V, int F, float eucl2, uint8_t* __restrict__ face_flag,
    uint8_t* __restrict__ vert_flag)
{
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    // (cleared here, set by hd_points_kernel for the vertices that support a selected point: the adjoint's gather per
    // vertex skips the other ~85 %)
    for (int v = f; v < V; v += gridDim.x * 256) vert_flag[(size_t)b * V + v] = 0;
    if (f >= F) return;
    bool any = false;
    if (!valid || valid[b]) {
        const uint8_t* ext = exterior + (size_t)b * V;
        const float* md = min_d2 + (size_t)b * V;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = faces[3 * f + k];
            any = any || (md[v] < eucl2) || (ext[v] == 0);           // :278
        }
    }
    face_flag[(size_t)b * F + f] = any;
}

// per chunk of kSel points: how many are selected (points in the sorted order), and the first selected point of the
// chunk in the CALLER's order (the same index range read as caller indices): what torch.min reports for an all-inf column
// is the first selected point in the caller's order = the first hit of the first chunk that has one
__global__ __launch_bounds__(kSel) void hd_count_kernel(
    const uint8_t* __restrict__ face_flag, const int32_t* __restrict__ hd_face, const int32_t* __restrict__ by_orig, int F, int N,
    int chunks, int32_t* __restrict__ chunk_cnt, int32_t* __restrict__ chunk_first)
{
    __shared__ int wave_sum[kSel / 64], wave_first[kSel / 64];
    const int b = blockIdx.y, c = blockIdx.x, n = c * kSel + threadIdx.x;
    const uint8_t* fb = face_flag + (size_t)b * F;
    const bool take = n < N && fb[hd_face[n]];                                       // :281
    const bool take_r = n < N && fb[hd_face[by_orig[n]]];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(take), mr = __builtin_amdgcn_ballot_w64(take_r);
    if ((threadIdx.x & 63) == 0) {
        wave_sum[threadIdx.x >> 6] = __builtin_popcountll(m);
        wave_first[threadIdx.x >> 6] = mr ? n + (int)__builtin_ctzll(mr) : 0x7fffffff;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0, first = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < kSel / 64; ++k) { t += wave_sum[k]; first = min(first, wave_first[k]); }
        chunk_cnt[(size_t)b * chunks + c] = t;
        chunk_first[(size_t)b * chunks + c] = first;
    }
}

__global__ __launch_bounds__(kSel) void hd_scatter_kernel(
    const uint8_t* __restrict__ face_flag, const int32_t* __restrict__ hd_face, const int32_t* __restrict__ chunk_cnt, int F, int N,
    int chunks, int32_t* __restrict__ sel, int32_t* __restrict__ slot, int32_t* __restrict__ counts)
{
    __shared__ int wave_sum[kSel / 64];
    __shared__ int s_base, s_total;
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = c * kSel + tid;
    // the chunk counts before this chunk and in all: one wavefront, a count per lane (a loop of dependent scalar loads
    // over the ~40 chunks, and one 64-bit atomic per chunk on ONE address per body for the first selected point, were
    // what this kernel spent its 35 us on)
    if (wave == 0) {
        int before = 0, all = 0;
        for (int k0 = 0; k0 < chunks; k0 += 64) {
            const int k = k0 + lane;
            const int t = k < chunks ? chunk_cnt[(size_t)b * chunks + k] : 0;
            before += k < c ? t : 0;
            all += t;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
        if (lane == 0) { s_base = before; s_total = all; }
    }
    const bool take = n < N && face_flag[(size_t)b * F + hd_face[n]];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
    const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sum[wave] = __builtin_popcountll(m);
    __syncthreads();
    int wbase = s_base;
#pragma unroll
    for (int k = 0; k < kSel / 64; ++k) wbase += k < wave ? wave_sum[k] : 0;
    if (n < N) {
        const int s = take ? wbase + before : -1;
        slot[(size_t)b * N + n] = s;
        if (take) sel[(size_t)b * N + s] = n;
    }
    if (tid == 0 && c == 0) counts[b] = s_total;
}

// ---- positions (loss.py:285, :295-296) ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hd_points_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ sel, const int32_t* __restrict__ counts,
    const int32_t* __restrict__ idx, const float* __restrict__ w, int K, const int32_t* __restrict__ hd_face,
    const int32_t* __restrict__ faces, const int32_t* __restrict__ mask_id, int V, int N,
    const int32_t* __restrict__ chunk_first, const int32_t* __restrict__ by_orig, const int32_t* __restrict__ slot, int chunks,
    int32_t* __restrict__ first_slot, float* __restrict__ pts, float* __restrict__ offs, int32_t* __restrict__ vid,
    uint8_t* __restrict__ vert_flag)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // the selected point that comes first in the caller's order: smallest first hit over the chunks
        int first = 0x7fffffff;
        for (int c = threadIdx.x; c < chunks; c += 64) first = min(first, chunk_first[(size_t)b * chunks + c]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) first = min(first, __shfl_xor(first, o));
        if (threadIdx.x == 0) first_slot[b] = first < N ? slot[(size_t)b * N + by_orig[first]] : 0;
    }
    if (k >= counts[b]) return;
    const size_t o = (size_t)b * N + k;
    const int n = sel[o];
    const float* vb = verts + (size_t)b * V * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    // the K non-zeros of the regressor row (loss.py:285 multiplies the dense matrix; K = 3 for barycentric samples), all
    // requested before the first is used
    float wk[8];
    int sk[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cc = c < K ? c : K - 1;
        wk[c] = c < K ? w[(size_t)K * n + cc] : 0.0f;
        sk[c] = idx[(size_t)K * n + cc];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c >= K) break;
        const float wc = wk[c];
        const int sv = sk[c];
        vert_flag[(size_t)b * V + sv] = 1;
        const float* p = vb + 3 * (size_t)sv;
        x = __builtin_fmaf(wc, p[0], x); y = __builtin_fmaf(wc, p[1], y); z = __builtin_fmaf(wc, p[2], z);
    }
    pts[3 * o] = x; pts[3 * o + 1] = y; pts[3 * o + 2] = z;
    // unit face normal of the posed face the point was sampled from (loss.py:30-41)
    const int f = hd_face[n];
    const float* a = vb + 3 * (size_t)faces[3 * f];
    const float* c1 = vb + 3 * (size_t)faces[3 * f + 1];
    const float* c2 = vb + 3 * (size_t)faces[3 * f + 2];
    const float ux = c1[0] - a[0], uy = c1[1] - a[1], uz = c1[2] - a[2];
    const float vx = c2[0] - a[0], vy = c2[1] - a[1], vz = c2[2] - a[2];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float inv = 0.001f / __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
    offs[3 * o] = x + inv * nx; offs[3 * o + 1] = y + inv * ny; offs[3 * o + 2] = z + inv * nz;
    vid[o] = mask_id[n];
}

// ---- terms (loss.py:299-315) and their adjoint -------------------------------------------------------------------
struct Term { float value, dd; };

__device__ __forceinline__ Term contact_term(float d, bool exterior)
{
    const float weight = exterior ? 0.005f : 1.0f, scale = exterior ? 0.005f : 0.04f;
    const float th = tanhf(d / scale);
    Term t = {weight * th * th, 2.0f * weight * th * (1.0f - th * th) / scale};
    return t;
}

__global__ __launch_bounds__(1024) void hd_terms_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner, const uint8_t* __restrict__ ext,
    const int32_t* __restrict__ counts, int N, float* __restrict__ terms)
{
    __shared__ float smem[2][16];
    const int b = blockIdx.x;
    const float* pb = pts + (size_t)b * N * 3;
    float in_sum = 0.0f, ex_sum = 0.0f;
    const int n = counts[b];
    for (int k = threadIdx.x; k < n; k += 1024) {
        const int p = partner[(size_t)b * N + k];
        const float dx = pb[3 * k] - pb[3 * p], dy = pb[3 * k + 1] - pb[3 * p + 1], dz = pb[3 * k + 2] - pb[3 * p + 2];
        const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
        const bool e = ext[(size_t)b * N + k] != 0;
        const Term t = contact_term(d, e);
        if (e) ex_sum += t.value; else in_sum += t.value;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { in_sum += __shfl_down(in_sum, o, 64); ex_sum += __shfl_down(ex_sum, o, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { smem[0][wave] = in_sum; smem[1][wave] = ex_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, c = 0.0f;
        for (int k = 0; k < 16; ++k) { a += smem[0][k]; c += smem[1][k]; }
        terms[2 * b] = a; terms[2 * b + 1] = c;
    }
}

// gradient on the points, G[b][k] = own side + what the points that picked k as their partner send it.  A workgroup owns
// kGradSpan consecutive points of one body and keeps their sums in LDS: it walks ALL the body's points (four per thread and
// trip, their loads requested together), takes those whose partner lies in its span (the partner's side of the term: LDS
// atomics) and those in the span themselves (own side).  Nothing is added up in global memory.  (Before: float atomics on
// a zeroed [B,N,3] array -- device-scope atomics on this part are resolved behind the per-XCD L2s, 1 M of them took 82 us.)
constexpr int kGradSpan = 1024;
constexpr int kGradBlock = 1024;
// kFixed (deterministic mode, common.h): the sums are 64-bit fixed-point integers (LDS integer atomics: associative, so the
// order of arrival does not matter) converted once at the end; the vertex gather behind this kernel has a fixed order
// anyway, so the whole HD adjoint then reproduces bit for bit
template <bool kFixed>
__global__ __launch_bounds__(kGradBlock) void hd_grad_points_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner, const uint8_t* __restrict__ ext,
    const int32_t* __restrict__ counts, const float* __restrict__ gscale, int N, float* __restrict__ G)
{
    using Acc = typename std::conditional<kFixed, long long, float>::type;
    __shared__ Acc acc[3 * kGradSpan];
    const int b = blockIdx.y, n = counts[b];
    const int lo = blockIdx.x * kGradSpan;
    if (lo >= n) return;
    const int hi = min(n, lo + kGradSpan);
    for (int i = threadIdx.x; i < 3 * kGradSpan; i += kGradBlock) acc[i] = 0;
    __syncthreads();
    auto add = [&](int i, float x) {
        if constexpr (kFixed) atomicAdd((unsigned long long*)&acc[i], (unsigned long long)__double2ll_rn((double)x * kFixedScale));
        else atomicAdd(&acc[i], x);
    };
    const float* pb = pts + (size_t)b * N * 3;
    const int32_t* qb = partner + (size_t)b * N;
    const uint8_t* eb = ext + (size_t)b * N;
    const float g_in = gscale[2 * b], g_ex = gscale[2 * b + 1];
    constexpr int kU = 2;
    for (int k0 = threadIdx.x; k0 < n; k0 += kU * kGradBlock) {
        int k[kU], p[kU];
        bool e[kU], use[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            k[u] = k0 + u * kGradBlock;
            const bool in = k[u] < n;
            p[u] = in ? qb[k[u]] : 0;
            e[u] = in ? eb[k[u]] != 0 : false;
            use[u] = in;
        }
        float a[kU][3], c[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            use[u] = use[u] && ((k[u] >= lo && k[u] < hi) || (p[u] >= lo && p[u] < hi));
            const int ku = use[u] ? k[u] : 0, pu = use[u] ? p[u] : 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) { a[u][d] = pb[3 * ku + d]; c[u][d] = pb[3 * pu + d]; }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const float g = e[u] ? g_ex : g_in;
            const float dx = a[u][0] - c[u][0], dy = a[u][1] - c[u][1], dz = a[u][2] - c[u][2];
            const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
            if (!use[u] || g == 0.0f || !(d > 0.0f)) continue;   // torch.norm's backward at 0 is 0
            const Term t = contact_term(d, e[u]);
            const float s = g * t.dd / d;
            const float ox = s * dx, oy = s * dy, oz = s * dz;
            if (k[u] >= lo && k[u] < hi) { add(3 * (k[u] - lo), ox); add(3 * (k[u] - lo) + 1, oy); add(3 * (k[u] - lo) + 2, oz); }
            if (p[u] >= lo && p[u] < hi) { add(3 * (p[u] - lo), -ox); add(3 * (p[u] - lo) + 1, -oy); add(3 * (p[u] - lo) + 2, -oz); }
        }
    }
    __syncthreads();
    float* gb = G + 3 * ((size_t)b * N + lo);
    for (int i = threadIdx.x; i < 3 * (hi - lo); i += kGradBlock) {
        if constexpr (kFixed) gb[i] = fixed_value(acc[i]);
        else gb[i] = acc[i];
    }
}

// adjoint of the regressor rows: a gather per vertex over the points it supports
__global__ __launch_bounds__(256) void hd_grad_verts_kernel(
    const float* __restrict__ G, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ v_off, const int32_t* __restrict__ v_ent, const float* __restrict__ w, int K, int V, int N,
    const uint8_t* __restrict__ vert_flag, float* __restrict__ grad_verts)
{
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const bool real = v < V;                          // all lanes stay: the long lists below need the whole wavefront
    const bool touched = real && vert_flag[(size_t)b * V + v] != 0;      // supports a selected point at all?
    const int e0 = touched ? v_off[v] : 0, e1 = touched ? v_off[v + 1] : 0;
    const int32_t* sb = slot + (size_t)b * N;
    const float* gb = G + 3 * (size_t)b * N;
    float x = 0.f, y = 0.f, z = 0.f;
    auto add = [&](int ent, int s) {                 // entry = point * 8 + corner; s = the point's slot in this body
        if (s < 0) return;
        const float wc = w[(size_t)K * (ent >> 3) + (ent & 7)];
        const float* g = gb + 3 * (size_t)s;
        x = __builtin_fmaf(wc, g[0], x); y = __builtin_fmaf(wc, g[1], y); z = __builtin_fmaf(wc, g[2], z);
    };
    constexpr int kLong = 64;
    if (e1 - e0 <= kLong) {
        // four entries at a time: entries, then slots, are fetched together (otherwise a chain of dependent gathers)
        for (int e = e0; e < e1; e += 4) {
            int ent[4], s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ent[u] = v_ent[min(e + u, e1 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = e + u < e1 ? sb[ent[u] >> 3] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u) add(ent[u], s[u]);
        }
    }
    // vertices with very many HD points (the poles of a lat-long sphere) are shared out over the wavefront: one lane
    // walking hundreds of entries would hold up the whole launch.  Lane-strided partial sums, then a butterfly:
    // a fixed order.
    unsigned long long todo = __builtin_amdgcn_ballot_w64(e1 - e0 > kLong);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int le0 = __builtin_amdgcn_readlane(e0, src), le1 = __builtin_amdgcn_readlane(e1, src);
        float px = 0.f, py = 0.f, pz = 0.f;
        for (int e = le0 + lane; e < le1; e += 64) {
            const int ent = v_ent[e], s = sb[ent >> 3];
            if (s < 0) continue;
            const float wc = w[(size_t)K * (ent >> 3) + (ent & 7)];
            const float* g = gb + 3 * (size_t)s;
            px = __builtin_fmaf(wc, g[0], px); py = __builtin_fmaf(wc, g[1], py); pz = __builtin_fmaf(wc, g[2], pz);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { px += __shfl_xor(px, m); py += __shfl_xor(py, m); pz += __shfl_xor(pz, m); }
        if (lane == src) { x = px; y = py; z = pz; }
    }
    if (!real) return;
    float* o = grad_verts + 3 * ((size_t)b * V + v);
    o[0] = x; o[1] = y; o[2] = z;
}

struct Saved { size_t counts, first, sel, slot, pts, partner, ext, vert_flag, total; };
Saved saved_layout(int B, int N, int V)
{
    Saved l;
    size_t o = 0;
    l.counts = tuch_ws_take(o, (size_t)B * sizeof(int32_t));
    l.first = tuch_ws_take(o, (size_t)B * sizeof(int32_t));
    l.sel = tuch_ws_take(o, (size_t)B * N * sizeof(int32_t));
    l.slot = tuch_ws_take(o, (size_t)B * N * sizeof(int32_t));
    l.pts = tuch_ws_take(o, (size_t)B * N * 3 * sizeof(float));
    l.partner = tuch_ws_take(o, (size_t)B * N * sizeof(int32_t));
    l.ext = tuch_ws_take(o, (size_t)B * N);
    l.vert_flag = tuch_ws_take(o, (size_t)B * V);
    l.total = o;
    return l;
}

struct Grad { size_t points, total; };
Grad grad_layout(int B, int N)
{
    Grad l;
    size_t o = 0;
    l.points = tuch_ws_take(o, (size_t)B * N * 3 * sizeof(float));
    l.total = o;
    return l;
}

struct Work { size_t offs, vid, min_d2, flags, chunk_cnt, chunk_first, search, winding, winding_bytes, total; };
Work work_layout(const tuch_hd_model* hm, int B)
{
    Work l;
    const int N = hm->N;
    size_t o = 0;
    l.offs = tuch_ws_take(o, (size_t)B * N * 3 * sizeof(float));
    l.vid = tuch_ws_take(o, (size_t)B * N * sizeof(int32_t));
    l.min_d2 = tuch_ws_take(o, (size_t)B * N * sizeof(float));
    l.flags = tuch_ws_take(o, (size_t)B * hm->F);
    l.chunk_cnt = tuch_ws_take(o, (size_t)B * ceil_div(N, kSel) * sizeof(int32_t));
    l.chunk_first = tuch_ws_take(o, (size_t)B * ceil_div(N, kSel) * sizeof(int32_t));
    size_t search_bytes, winding_bytes;
    { tuch_ws_pause nested; search_bytes = std::max(tuch_v2v_min_indexed_workspace_bytes(B, N), tuch_hd_search_workspace_bytes(B, N)); winding_bytes = tuch_winding_points_workspace_bytes(hm->cm, B, N); }
    l.search = tuch_ws_take(o, search_bytes);
    l.winding = tuch_ws_take(o, winding_bytes);
    l.winding_bytes = winding_bytes;
    l.total = o;
    return l;
}

}  // namespace

extern "C" void tuch_hd_model_destroy(tuch_hd_model* hm)
{
    if (!hm) return;
    void* dev[] = {hm->idx, hm->w, hm->face, hm->tv, hm->mask_id, hm->orig, hm->by_orig, hm->v_off, hm->v_ent, hm->offsets};
    for (void* p : dev) tuch_table_free(p);
    if (hm->side) (void)hipStreamDestroy(hm->side);
    if (hm->ev_fork) (void)hipEventDestroy(hm->ev_fork);
    if (hm->ev_join) (void)hipEventDestroy(hm->ev_join);
    delete hm->enqueue;
    delete hm->order_host;
    free(hm);
}

// hd_idx / hd_w [N,3]: the three non-zeros of every row of the HD vertex regressor; hd_face [N]: the face each point
// was sampled from.  The contact model must outlive the HD model.
extern "C" int tuch_hd_model_create_k(tuch_hd_model** out, const tuch_contact_model* cm, int N, int K, const int32_t* hd_idx,
                                      const float* hd_w, const int32_t* hd_face);
extern "C" int tuch_hd_model_create(tuch_hd_model** out, const tuch_contact_model* cm, int N, const int32_t* hd_idx,
                                    const float* hd_w, const int32_t* hd_face)
{
    return tuch_hd_model_create_k(out, cm, N, 3, hd_idx, hd_w, hd_face);
}

// The same for a regressor with up to K <= 8 non-zeros per row (hd_idx / hd_w [N,K]; rows with fewer: weight 0, any valid id).
extern "C" int tuch_hd_model_create_k(tuch_hd_model** out, const tuch_contact_model* cm, int N, int K, const int32_t* hd_idx,
                                      const float* hd_w, const int32_t* hd_face)
{
    TUCH_REQUIRE(out && cm && hd_idx && hd_w && hd_face && N > 0, "tuch_hd_model_create: bad arguments");
    TUCH_REQUIRE(K >= 1 && K <= 8, "tuch_hd_model_create: %d non-zeros per row (1..8)", K);
    TUCH_REQUIRE(cm->mask_bits, "tuch_hd_model_create: the contact model has no geodesic mask");
    const int V = cm->V, F = cm->F;
    std::vector<int32_t> faces((size_t)F * 3);
    if (tuch_table_download(faces.data(), cm->faces, faces.size() * sizeof(int32_t)) != TUCH_OK) {
        tuch_set_error("tuch_hd_model_create: cannot read the model's faces");
        return TUCH_ERR_HIP;
    }
    for (int n = 0; n < N; ++n) {
        TUCH_REQUIRE(hd_face[n] >= 0 && hd_face[n] < F, "tuch_hd_model_create: face id %d out of range", hd_face[n]);
        for (int c = 0; c < K; ++c)
            TUCH_REQUIRE(hd_idx[(size_t)K * n + c] >= 0 && hd_idx[(size_t)K * n + c] < V, "tuch_hd_model_create: vertex id out of range");
    }
    // The HD points are a set (the loss sums over them): keep them sorted by the surface patch of their face, so that
    // consecutive selected points are neighbours in space (coherent query blocks for the inside test, tight row boxes
    // for the search).  The model's own tree decides (not a rebuilt one).
    std::vector<int32_t> order(N);
    std::iota(order.begin(), order.end(), 0);
    // Within a patch: by the mask vertex (first vertex of the face, loss.py:88), then by face -- consecutive points that
    // inherit the same mask row form the runs hd_search.hip fetches one mask word for.
    const bool tree = cm->tree_nodes > 0 && cm->tree_face_leaf_host;
    if (tree)
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const int fa = hd_face[a], fb = hd_face[b];
            const int la = cm->tree_face_leaf_host[fa], lb = cm->tree_face_leaf_host[fb];
            if (la != lb) return la < lb;
            const int va = faces[3 * (size_t)fa], vb = faces[3 * (size_t)fb];
            if (va != vb) return va < vb;
            return fa < fb;
        });
    std::vector<int32_t> pos(V);
    const bool tree_mask = tree && cm->tree_mask_bits && cm->tree_qperm_host;
    if (tree_mask)
        for (int i = 0; i < V; ++i) pos[cm->tree_qperm_host[i]] = i;
    std::vector<int32_t> idx((size_t)N * K), face(N), tv(N), mask_id(N);
    std::vector<float> w((size_t)N * K);
    for (int k = 0; k < N; ++k) {
        const int n = order[k];
        for (int c = 0; c < K; ++c) { idx[(size_t)K * k + c] = hd_idx[(size_t)K * n + c]; w[(size_t)K * k + c] = hd_w[(size_t)K * n + c]; }
        face[k] = hd_face[n];
        tv[k] = faces[3 * (size_t)hd_face[n]];                       // loss.py:88: first vertex of the face
        mask_id[k] = tree_mask ? pos[tv[k]] : tv[k];
    }
    // CSR tables
    std::vector<int32_t> v_off(V + 1, 0);
    for (int k = 0; k < N; ++k) {
        for (int c = 0; c < K; ++c)
            if (w[(size_t)K * k + c] != 0.0f) ++v_off[idx[(size_t)K * k + c] + 1];
    }
    for (int v = 0; v < V; ++v) v_off[v + 1] += v_off[v];
    TUCH_REQUIRE((long)N * 8 < 0x7fffffffL, "tuch_hd_model_create: too many HD points");
    std::vector<int32_t> v_ent((size_t)v_off[V] + 1), vf(v_off.begin(), v_off.end() - 1);
    for (int k = 0; k < N; ++k) {
        for (int c = 0; c < K; ++c)
            if (w[(size_t)K * k + c] != 0.0f) v_ent[vf[idx[(size_t)K * k + c]]++] = k * 8 + c;
    }
    std::vector<int32_t> offsets(kMaxBatch + 1);
    for (int b = 0; b <= kMaxBatch; ++b) offsets[b] = (int32_t)((long)b * N < 0x7fffffffL ? (long)b * N : 0x7fffffffL);
    tuch_hd_model* hm = (tuch_hd_model*)calloc(1, sizeof(tuch_hd_model));
    hm->cm = cm; hm->N = N; hm->V = V; hm->F = F; hm->K = K; hm->tree_order = tree_mask ? 1 : 0;
    hm->order_host = new std::vector<int32_t>(order);
    hm->enqueue = new std::mutex();
    if (!tuch_host_tables() &&
        (hipStreamCreateWithFlags(&hm->side, hipStreamNonBlocking) != hipSuccess ||
         hipEventCreateWithFlags(&hm->ev_fork, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&hm->ev_join, hipEventDisableTiming) != hipSuccess)) {
        tuch_set_error("tuch_hd_model_create: cannot create the side stream / events");
        tuch_hd_model_destroy(hm);
        *out = nullptr;
        return TUCH_ERR_HIP;
    }
    int rc = upload(&hm->idx, idx.data(), idx.size());
    if (rc == TUCH_OK) rc = upload(&hm->w, w.data(), w.size());
    if (rc == TUCH_OK) rc = upload(&hm->face, face.data(), face.size());
    if (rc == TUCH_OK) rc = upload(&hm->tv, tv.data(), tv.size());
    if (rc == TUCH_OK) rc = upload(&hm->mask_id, mask_id.data(), mask_id.size());
    if (rc == TUCH_OK) rc = upload(&hm->orig, order.data(), order.size());
    if (rc == TUCH_OK) {
        std::vector<int32_t> inverse(N);
        for (int k = 0; k < N; ++k) inverse[order[k]] = k;
        rc = upload(&hm->by_orig, inverse.data(), inverse.size());
    }
    if (rc == TUCH_OK) rc = upload(&hm->v_off, v_off.data(), v_off.size());
    if (rc == TUCH_OK) rc = upload(&hm->v_ent, v_ent.data(), v_ent.size());
    if (rc == TUCH_OK) rc = upload(&hm->offsets, offsets.data(), offsets.size());
    if (rc != TUCH_OK) {
        tuch_hd_model_destroy(hm);
        *out = nullptr;
        return rc;
    }
    *out = hm;
    return TUCH_OK;
}

extern "C" int tuch_hd_model_info(const tuch_hd_model* hm, int* N, int32_t* order_host)
{
    TUCH_REQUIRE(hm, "tuch_hd_model_info: null model");
    if (N) *N = hm->N;
    if (order_host) memcpy(order_host, hm->order_host->data(), sizeof(int32_t) * hm->N);
    return TUCH_OK;
}

extern "C" size_t tuch_hd_contact_saved_bytes(const tuch_hd_model* hm, int B)
{
    if (!hm || B <= 0) return 0;
    tuch_ws_scope scope(hm->cm->opt.canary != 0);
    return saved_layout(B, hm->N, hm->V).total;
}

extern "C" size_t tuch_hd_contact_workspace_bytes(const tuch_hd_model* hm, int B)
{
    if (!hm || B <= 0) return 0;
    tuch_ws_scope scope(hm->cm->opt.canary != 0);
    const size_t f = work_layout(hm, B).total;
    const size_t g = grad_layout(B, hm->N).total;       // adjoint: point gradients
    return f > g ? f : g;
}

// terms[b] = {sum over interior HD points of tanh^2(d/0.04), sum over exterior ones of 0.005 tanh^2(d/0.005)} for
// every body with valid[b] != 0 (others 0, 0); inputs are the vertex-level results of the same vertices
// (tuch_exterior_flags with the segment filter, tuch_v2v_min_model).  `saved` is kept by the caller for the adjoint.
extern "C" int tuch_hd_contact_fwd(const tuch_hd_model* hm, const float* verts, const uint8_t* exterior, const float* min_d2,
                                   const int32_t* partner, const uint8_t* valid, int B, float euclthres, float thresh,
                                   float* terms, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                   void* stream)
{
    TUCH_REQUIRE(hm && verts && exterior && min_d2 && partner && terms && saved && workspace, "tuch_hd_contact_fwd: null pointer");
    TUCH_REQUIRE(B > 0 && B <= kMaxBatch && (long)B * hm->N < 0x7fffffffL, "tuch_hd_contact_fwd: bad batch %d", B);
    const int N = hm->N, V = hm->V;
    tuch_ws_scope saved_scope(hm->cm->opt.canary != 0), scope(hm->cm->opt.canary != 0);
    const Saved sl = saved_scope.record(0, [&] { return saved_layout(B, N, V); });
    const Work wl = scope.record(0, [&] { return work_layout(hm, B); });
    if (saved_bytes < sl.total || workspace_bytes < wl.total) {
        tuch_set_error("tuch_hd_contact_fwd: saved %zu < %zu or workspace %zu < %zu bytes", saved_bytes, sl.total,
                       workspace_bytes, wl.total);
        return TUCH_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    saved_scope.arm(saved, hm->cm->canary_hits, s);
    scope.arm(workspace, hm->cm->canary_hits, s);
    char* sv = (char*)saved;
    char* ws = (char*)workspace;
    int32_t* counts = (int32_t*)(sv + sl.counts);
    int32_t* first = (int32_t*)(sv + sl.first);
    int32_t* sel = (int32_t*)(sv + sl.sel);
    int32_t* slot = (int32_t*)(sv + sl.slot);
    float* pts = (float*)(sv + sl.pts);
    int32_t* part = (int32_t*)(sv + sl.partner);
    uint8_t* ext = (uint8_t*)(sv + sl.ext);
    float* offs = (float*)(ws + wl.offs);
    int32_t* vid = (int32_t*)(ws + wl.vid);
    uint8_t* flags = (uint8_t*)(ws + wl.flags);
    int32_t* chunk_cnt = (int32_t*)(ws + wl.chunk_cnt);
    int32_t* chunk_first = (int32_t*)(ws + wl.chunk_first);
    const int chunks = ceil_div(N, kSel);
    const uint64_t* bits = hm->tree_order ? hm->cm->tree_mask_bits : hm->cm->mask_bits;
    hipLaunchKernelGGL(hd_face_flags_kernel, dim3(ceil_div(hm->F, 256), B), dim3(256), 0, s, exterior, min_d2, valid,
                       (const int32_t*)hm->cm->faces, V, hm->F, euclthres * euclthres, flags, (uint8_t*)(sv + sl.vert_flag));
    hipLaunchKernelGGL(hd_count_kernel, dim3(chunks, B), dim3(kSel), 0, s, (const uint8_t*)flags, (const int32_t*)hm->face,
                       (const int32_t*)hm->by_orig, hm->F, N, chunks, chunk_cnt, chunk_first);
    hipLaunchKernelGGL(hd_scatter_kernel, dim3(chunks, B), dim3(kSel), 0, s, (const uint8_t*)flags, (const int32_t*)hm->face,
                       (const int32_t*)chunk_cnt, hm->F, N, chunks, sel, slot, counts);
    const dim3 pgrid(ceil_div(N, 256), B);
    hipLaunchKernelGGL(hd_points_kernel, pgrid, dim3(256), 0, s, verts, (const int32_t*)sel, (const int32_t*)counts,
                       (const int32_t*)hm->idx, (const float*)hm->w, hm->K, (const int32_t*)hm->face,
                       (const int32_t*)hm->cm->faces, (const int32_t*)hm->mask_id, V, N, (const int32_t*)chunk_first,
                       (const int32_t*)hm->by_orig, (const int32_t*)slot, chunks, first, pts, offs, vid, (uint8_t*)(sv + sl.vert_flag));
    // (seeding the search from the vertex-level partners was tried: the seeds are excellent where they exist -- median
    // ratio to the final distance 1.00 -- but the nearest admissible HD point is ~10 cm away, so a column block still has
    // to visit ~40 % of the rows, and building the seeds cost more than the sampling pass they replace)
    // the inside test of the points beside their search: fork here, join in front of the terms
    const bool overlap = hm->cm->opt.hd_overlap != 0;
    std::unique_lock<std::mutex> lock(*hm->enqueue, std::defer_lock);
    hipStream_t ws_stream = s;
    if (overlap) {
        lock.lock();
        if (hipEventRecord(hm->ev_fork, s) != hipSuccess || hipStreamWaitEvent(hm->side, hm->ev_fork, 0) != hipSuccess) {
            tuch_set_error("tuch_hd_contact_fwd: fork onto the side stream failed");
            return TUCH_ERR_HIP;
        }
        ws_stream = hm->side;
    }
    int rc = tuch_winding_points(hm->cm, verts, offs, counts, B, N, thresh, nullptr, ext, ws + wl.winding,
                                 wl.winding_bytes, (void*)ws_stream);
    if (rc == TUCH_OK)
        rc = hm->cm->opt.hd_search
                 ? tuch_hd_search(pts, vid, hm->offsets, counts, first, bits, B, V, N, (float*)(ws + wl.min_d2), part,
                                  ws + wl.search, s, hm->cm->opt.hd_search_waves)
                 : tuch_v2v_min_indexed_seeded(pts, vid, hm->offsets, counts, nullptr, nullptr, first, bits, B, V, N,
                                               (float*)(ws + wl.min_d2), part, ws + wl.search, s);
    if (overlap) {
        // (joined whatever happened above: a captured fork must not be left open)
        if (hipEventRecord(hm->ev_join, hm->side) != hipSuccess || hipStreamWaitEvent(s, hm->ev_join, 0) != hipSuccess) {
            tuch_set_error("tuch_hd_contact_fwd: join of the side stream failed");
            return TUCH_ERR_HIP;
        }
        lock.unlock();
    }
    if (rc != TUCH_OK) return rc;
    hipLaunchKernelGGL(hd_terms_kernel, dim3(B), dim3(1024), 0, s, (const float*)pts, (const int32_t*)part,
                       (const uint8_t*)ext, (const int32_t*)counts, N, terms);
    return tuch_check_launch("tuch_hd_contact_fwd");
}

// grad_verts [B,V,3] (overwritten) = d (sum_b grad_terms[b,0] * terms[b,0] + grad_terms[b,1] * terms[b,1]) / d verts
extern "C" int tuch_hd_contact_bwd(const tuch_hd_model* hm, const void* saved, const float* grad_terms, int B,
                                   float* grad_verts, void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(hm && saved && grad_terms && grad_verts && workspace, "tuch_hd_contact_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && B <= kMaxBatch, "tuch_hd_contact_bwd: bad batch %d", B);
    const int N = hm->N, V = hm->V;
    tuch_ws_scope scope(hm->cm->opt.canary != 0);
    const Saved sl = saved_layout(B, N, hm->V);
    const Grad gl = scope.record(0, [&] { return grad_layout(B, N); });
    if (workspace_bytes < gl.total) {
        tuch_set_error("tuch_hd_contact_bwd: workspace %zu < %zu bytes", workspace_bytes, gl.total);
        return TUCH_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    scope.arm(workspace, hm->cm->canary_hits, s);
    const char* sv = (const char*)saved;
    float* G = (float*)((char*)workspace + gl.points);
    if (tuch_deterministic())
        hipLaunchKernelGGL(hd_grad_points_kernel<true>, dim3(ceil_div(N, kGradSpan), B), dim3(kGradBlock), 0, s,
                           (const float*)(sv + sl.pts), (const int32_t*)(sv + sl.partner), (const uint8_t*)(sv + sl.ext),
                           (const int32_t*)(sv + sl.counts), grad_terms, N, G);
    else
        hipLaunchKernelGGL(hd_grad_points_kernel<false>, dim3(ceil_div(N, kGradSpan), B), dim3(kGradBlock), 0, s,
                           (const float*)(sv + sl.pts), (const int32_t*)(sv + sl.partner), (const uint8_t*)(sv + sl.ext),
                           (const int32_t*)(sv + sl.counts), grad_terms, N, G);
    hipLaunchKernelGGL(hd_grad_verts_kernel, dim3(ceil_div(V, 256), B), dim3(256), 0, s, (const float*)G,
                       (const int32_t*)(sv + sl.slot), (const int32_t*)hm->v_off, (const int32_t*)hm->v_ent,
                       (const float*)hm->w, hm->K, V, N, (const uint8_t*)(sv + sl.vert_flag), grad_verts);
    return tuch_check_launch("tuch_hd_contact_bwd");
}

// inspection (tests): counts [B] and, per body, the caller-order indices of the selected points in slot order
extern "C" int tuch_hd_contact_selection(const tuch_hd_model* hm, const void* saved, int B, int32_t* counts_host,
                                         int32_t* selected_host /* [B,N] caller-order index per slot, -1 padded */)
{
    TUCH_REQUIRE(hm && saved && counts_host, "tuch_hd_contact_selection: null pointer");
    const int N = hm->N;
    tuch_ws_scope scope(hm->cm->opt.canary != 0);        // the layout the forward call used
    const Saved sl = saved_layout(B, N, hm->V);
    const char* sv = (const char*)saved;
    if (hipMemcpy(counts_host, sv + sl.counts, sizeof(int32_t) * B, hipMemcpyDeviceToHost) != hipSuccess) return TUCH_ERR_HIP;
    if (selected_host) {
        std::vector<int32_t> sel((size_t)B * N);
        if (hipMemcpy(sel.data(), sv + sl.sel, sel.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return TUCH_ERR_HIP;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < N; ++k)
                selected_host[(size_t)b * N + k] = k < counts_host[b] ? (*hm->order_host)[sel[(size_t)b * N + k]] : -1;
    }
    return TUCH_OK;
}

// inspection (tests): per slot of every body (slot order of tuch_hd_contact_selection) the caller-order index of the
// partner found by the search (-1 behind the body's count) and the exterior flag of the point
extern "C" int tuch_hd_contact_details(const tuch_hd_model* hm, const void* saved, int B, int32_t* partner_host, uint8_t* ext_host)
{
    TUCH_REQUIRE(hm && saved && partner_host && ext_host, "tuch_hd_contact_details: null pointer");
    const int N = hm->N;
    tuch_ws_scope scope(hm->cm->opt.canary != 0);        // the layout the forward call used
    const Saved sl = saved_layout(B, N, hm->V);
    const char* sv = (const char*)saved;
    std::vector<int32_t> counts(B), sel((size_t)B * N), part((size_t)B * N);
    if (hipMemcpy(counts.data(), sv + sl.counts, sizeof(int32_t) * B, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(sel.data(), sv + sl.sel, sel.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(part.data(), sv + sl.partner, part.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(ext_host, sv + sl.ext, (size_t)B * N, hipMemcpyDeviceToHost) != hipSuccess)
        return TUCH_ERR_HIP;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < N; ++k) {
            const int32_t p = part[(size_t)b * N + k];
            partner_host[(size_t)b * N + k] =
                k < counts[b] && p >= 0 && p < counts[b] ? (*hm->order_host)[sel[(size_t)b * N + p]] : -1;
        }
    return TUCH_OK;
}
