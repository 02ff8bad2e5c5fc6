// Nearest geodesically-far vertex per vertex (K1 of SURVEY.md §2.2).
//
// Replaces the chain  batch_pairwise_dist (tuch/utils/contact.py:23-47)
//   -> P[:, ~geomask] = inf -> argmin/min over dim 1
// of tuch/smplify/losses.py:76-78,92-93 and tuch/train/loss.py:255-257,269-270,
// which materialises three [1,V,V] float matrices (570 MB) and a 47.5 MB bool
// mask per body.  Here:
//   * the mask is bit-packed once per model, transposed so that the 64 mask
//     bits of (row j, columns 64*w..64*w+63) are ONE 64-bit word at
//     bits[w][j]: a wave whose lanes are 64 consecutive columns fetches it with
//     a scalar load and applies it with a single v_cndmask (the word IS the
//     lane mask) -- no per-lane bit tests;
//   * vertex j is wave-uniform (scalar loads), every lane owns two columns held
//     as a float2 so the distance arithmetic issues as v_pk_*_f32;
//   * squared distances are direct differences (more accurate than the
//     reference's |x|^2+|y|^2-2x.y; see DESIGN.md "Parity");
//   * rows are split across blocks; partial (min, argmin) pairs are merged in
//     ascending row order with a strict '<' => first-index tie rule of
//     torch.argmin, all-masked column -> (inf, 0).
#include "common.h"
#include "model.h"
#include "workspace.h"
#include "tree_device.h"
#include <stdlib.h>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kColsPerWave = 128;                 // two 64-column blocks per wave
constexpr int kColsPerBlock = kColsPerWave * kWaves;

__device__ __forceinline__ float select_by_lane_mask(float if_clear, float if_set, uint64_t lane_mask)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(lane_mask));
    return r;
}

// minimum of four distances: the operands come out of select_by_lane_mask (opaque to the compiler, which would put a
// canonicalising v_max in front of every fminf of them); none is a NaN
__device__ __forceinline__ float min4_raw(float a, float b, float c, float d)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32_e32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}

// minimum of eight such distances: four instructions
__device__ __forceinline__ float min8_raw(const float* d)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3\n\tv_min3_f32 %0, %0, %4, %5\n\tv_min3_f32 %0, %0, %6, %7\n\tv_min_f32_e32 %0, %0, %8"
        : "=&v"(r) : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
    return r;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t x)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// geomask [V][V] bytes -> bits[w][j], bit k of the word = geomask[j][64*w + k]
__global__ __launch_bounds__(kBlock) void pack_mask_kernel(
    const uint8_t* __restrict__ mask, int V, int W, uint64_t* __restrict__ bits)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int w = blockIdx.y;
    if (j >= V) return;
    uint64_t word = 0;
    const int c0 = w * 64;
    if (c0 < V) {
        const uint8_t* row = mask + (size_t)j * V + c0;
        const int n = min(64, V - c0);
        for (int k = 0; k < n; ++k) word |= (uint64_t)(row[k] != 0) << k;
    }
    bits[(size_t)w * V + j] = word;
    (void)W;
}

__global__ __launch_bounds__(kBlock) void v2v_partial_kernel(
    const float* __restrict__ verts,      // [B,V,3]
    const uint64_t* __restrict__ bits,    // [W][V], W even
    int V, int rows_per_split,
    float* __restrict__ part_min,         // [B,S,V]
    int* __restrict__ part_arg)           // [B,S,V]
{
    // body index fastest in the launch order (XCD b % 8 keeps body b's vertices in one L2)
    const int b = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // first 64-column block of this wave; readfirstlane tells the compiler it is wave-uniform so
    // that the mask words come in through scalar loads (otherwise: one VMEM round trip per row)
    const int w0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.z * kWaves + wave) * 2);
    if (w0 * 64 >= V) return;                                  // wave-uniform: no columns left
    const int i0 = w0 * 64 + lane, i1 = i0 + 64;
    const float* vb = verts + (size_t)b * V * 3;
    const int c0 = i0 < V ? i0 : V - 1, c1 = i1 < V ? i1 : V - 1;
    const v2f px = {vb[3 * c0 + 0], vb[3 * c1 + 0]};
    const v2f py = {vb[3 * c0 + 1], vb[3 * c1 + 1]};
    const v2f pz = {vb[3 * c0 + 2], vb[3 * c1 + 2]};

    const int j_begin = split * rows_per_split;
    const int j_end = min(V, j_begin + rows_per_split);
    const uint64_t* m0 = bits + (size_t)w0 * V;
    const uint64_t* m1 = m0 + V;
    const float inf = __builtin_inff();
    float best0 = inf, best1 = inf;
    int arg0 = 0, arg1 = 0;
    auto row = [&](int j, uint64_t k0, uint64_t k1, float vx, float vy, float vz) {
        const v2f dx = px - splat2(vx), dy = py - splat2(vy), dz = pz - splat2(vz);
        const v2f d = fma2(dz, dz, fma2(dy, dy, dx * dx));
        const float d0 = select_by_lane_mask(inf, d[0], k0);
        const float d1 = select_by_lane_mask(inf, d[1], k1);
        // running minima improve O(log V) times per column: skip the four selects unless some lane
        // of the wave improves (wave-uniform branch; results unchanged)
        const bool up0 = d0 < best0, up1 = d1 < best1;
        if (__builtin_amdgcn_ballot_w64(up0 || up1)) {
            if (up0) { best0 = d0; arg0 = j; }
            if (up1) { best1 = d1; arg1 = j; }
        }
    };
    int j = j_begin;
    for (; j + 4 <= j_end; j += 4) {      // four rows per trip: the scalar loads are issued together
        uint64_t k0[4], k1[4];
        float c[12];
#pragma unroll
        for (int u = 0; u < 4; ++u) { k0[u] = m0[j + u]; k1[u] = m1[j + u]; }
#pragma unroll
        for (int u = 0; u < 12; ++u) c[u] = vb[3 * j + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) row(j + u, k0[u], k1[u], c[3 * u], c[3 * u + 1], c[3 * u + 2]);
    }
    for (; j < j_end; ++j) row(j, m0[j], m1[j], vb[3 * j], vb[3 * j + 1], vb[3 * j + 2]);
    const size_t o = ((size_t)b * nsplit + split) * V;
    if (i0 < V) { part_min[o + i0] = best0; part_arg[o + i0] = arg0; }
    if (i1 < V) { part_min[o + i1] = best1; part_arg[o + i1] = arg1; }
}

__global__ __launch_bounds__(kBlock) void v2v_merge_kernel(
    const float* __restrict__ part_min, const int* __restrict__ part_arg, int V, int nsplit,
    float* __restrict__ out_min, int* __restrict__ out_arg)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= V) return;
    float best = __builtin_inff();
    int arg = 0;
    for (int s = 0; s < nsplit; ++s) {
        const size_t o = ((size_t)b * nsplit + s) * V + i;
        const float m = part_min[o];
        if (m < best) { best = m; arg = part_arg[o]; }
    }
    if (out_min) out_min[(size_t)b * V + i] = best;
    if (out_arg) out_arg[(size_t)b * V + i] = arg;
}

// contact.py:23-47 materialised (API parity; small inputs only):
// P[b][i][j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j, same formula as the reference.
__global__ __launch_bounds__(kBlock) void pairwise_kernel(
    const float* __restrict__ x, const float* __restrict__ y, int nx, int ny, int squared,
    float* __restrict__ out)
{
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= ny) return;
    const float* xi = x + ((size_t)b * nx + i) * 3;
    const float* yj = y + ((size_t)b * ny + j) * 3;
    const float xx = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
    const float yy = yj[0] * yj[0] + yj[1] * yj[1] + yj[2] * yj[2];
    const float zz = xi[0] * yj[0] + xi[1] * yj[1] + xi[2] * yj[2];
    float p = xx + yy - 2.0f * zz;
    if (!squared) p = __builtin_sqrtf(p);
    out[((size_t)b * nx + i) * ny + j] = p;
}

// Adjoint of pairwise_kernel for callers that differentiate through the materialised matrix
// (tuch/smplify/losses.py:76-78 -> :115-116, tuch/eft/loss.py:142): with g = dL/dP,
//   dL/dx_i = 2 (x_i sum_j g_ij - sum_j g_ij y_j),   dL/dy_j = 2 (y_j sum_i g_ij - sum_i g_ij x_i);
// for !squared g is first divided by 2 sqrt(p), p recomputed by the forward's formula (torch's sqrt adjoint:
// the same inf/NaN where p <= 0).  Both reductions run in a fixed order: no atomics, bit-reproducible.
__device__ __forceinline__ float pairwise_adjoint_weight(const float* xi, const float* yj, float g, int squared)
{
    if (squared) return g;
    const float xx = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
    const float yy = yj[0] * yj[0] + yj[1] * yj[1] + yj[2] * yj[2];
    const float zz = xi[0] * yj[0] + xi[1] * yj[1] + xi[2] * yj[2];
    return g / (2.0f * __builtin_sqrtf(xx + yy - 2.0f * zz));
}

// one workgroup per row i: the row of g is read coalesced, (sum g, sum g y) reduced over the wavefronts
__global__ __launch_bounds__(kBlock) void pairwise_bwd_x_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g, int nx, int ny, int squared,
    float* __restrict__ gx)
{
    const int b = blockIdx.y, i = blockIdx.x;
    const float* xi = x + ((size_t)b * nx + i) * 3;
    const float xv[3] = {xi[0], xi[1], xi[2]};
    const float* grow = g + ((size_t)b * nx + i) * ny;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < ny; j += kBlock) {
        const float* yj = y + ((size_t)b * ny + j) * 3;
        const float yv[3] = {yj[0], yj[1], yj[2]};
        const float w = pairwise_adjoint_weight(xv, yv, grow[j], squared);
        acc[0] += w; acc[1] += w * yv[0]; acc[2] += w * yv[1]; acc[3] += w * yv[2];
    }
    __shared__ float part[kBlock / 64][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[k] += __shfl_xor(acc[k], m);
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 4; ++k) part[threadIdx.x >> 6][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        float s0 = 0.f, sk = 0.f;
        for (int w = 0; w < kBlock / 64; ++w) { s0 += part[w][0]; sk += part[w][1 + threadIdx.x]; }
        gx[((size_t)b * nx + i) * 3 + threadIdx.x] = 2.0f * (xv[threadIdx.x] * s0 - sk);
    }
}

// one workgroup per 64 columns: lane = column j, the wavefronts interleave the rows (x_i is wavefront-uniform),
// 256-byte row segments of g per wavefront load
__global__ __launch_bounds__(kBlock) void pairwise_bwd_y_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g, int nx, int ny, int squared,
    float* __restrict__ gy)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const bool live = j < ny;
    const float* yj = y + ((size_t)b * ny + (live ? j : ny - 1)) * 3;
    const float yv[3] = {yj[0], yj[1], yj[2]};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = wave; i < nx; i += kBlock / 64) {
        const float* xi = x + ((size_t)b * nx + i) * 3;
        const float xv[3] = {xi[0], xi[1], xi[2]};
        const float gij = live ? g[((size_t)b * nx + i) * ny + j] : 0.f;
        const float w = pairwise_adjoint_weight(xv, yv, gij, squared);
        acc[0] += w; acc[1] += w * xv[0]; acc[2] += w * xv[1]; acc[3] += w * xv[2];
    }
    __shared__ float part[kBlock / 64][4][64];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[wave][k][lane] = acc[k];
    __syncthreads();
    if (wave == 0 && live) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < kBlock / 64; ++w)
            for (int k = 0; k < 4; ++k) t[k] += part[w][k][lane];
        for (int k = 0; k < 3; ++k) gy[((size_t)b * ny + j) * 3 + k] = 2.0f * (yv[k] * t[0] - t[1 + k]);
    }
}

// Ragged variant for resampled (HD) point sets, tuch/train/loss.py:288-291: body b owns points
// off[b]..off[b+1]; point a carries the template vertex vid[a] whose geodesic-mask row/column
// it inherits (geovec_verts, loss.py:88).  Column a: min over the body's rows r with
// geomask[vid[r]][vid[a]].  argmin is the row index relative to the body's first point.
constexpr int kIndexedWaves = 8;             // wavefronts per workgroup: each takes an eighth of the rows
constexpr int kIndexedChunk = 32;            // rows per bounding box
constexpr float kIndexedSlack = 0.999999f;   // lower bounds are deflated by 1e-6 (rounding of the two sums)

// box of every chunk of kIndexedChunk consecutive points of a body: [B][max_chunks][8] = (lo xyz, -, hi xyz, -).
// The caller keeps the points of a body sorted by surface patch, so a chunk is a small patch.
__global__ __launch_bounds__(256) void v2v_indexed_boxes_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ counts, int max_chunks,
    float* __restrict__ boxes)
{
    const int b = blockIdx.y;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int beg = off[b], n = counts ? counts[b] : off[b + 1] - beg;
    if (chunk >= max_chunks || chunk * kIndexedChunk >= n) return;
    const int r = chunk * kIndexedChunk + (lane & (kIndexedChunk - 1));
    const float* p = pts + 3 * (size_t)(beg + min(r, n - 1));
    float lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[0], p[1], p[2]};
#pragma unroll
    for (int m = kIndexedChunk / 2; m >= 1; m >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], m));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m));
        }
    if (lane == 0) {
        float* o = boxes + ((size_t)b * max_chunks + chunk) * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f;
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
    }
}

// (Two rows per packed FP32 instruction -- rows rewritten in SoA groups of four by the boxes kernel, chunks are
// 32-aligned so there are no ragged heads -- was measured: 507 us instead of 490.  Like the tree kernel this one is
// held by its scalar side and the load -> gather -> compare chain of a trip, not by the vector arithmetic.)
// 64 columns per workgroup, the rows split over its 8 wavefronts.  Pass 1: every wavefront evaluates every
// eighth row of its share, the minima are merged in LDS: an upper bound for every column.  Pass 2: every
// wavefront walks the 32-row chunks of its share and skips those whose box is farther from all its columns
// than their current minima.  Candidates at the minimum are never skipped (strict test) and ties go to the
// smaller row, so the result is the first-index argmin of torch.min regardless of the order of evaluation.
__global__ __launch_bounds__(64 * kIndexedWaves) void v2v_indexed_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ vid, const int32_t* __restrict__ off,
    const int32_t* __restrict__ counts, const float* __restrict__ seed_best, const int32_t* __restrict__ seed_arg,
    const int32_t* __restrict__ all_masked_arg,
    const uint64_t* __restrict__ bits, int V, const float* __restrict__ boxes, int max_chunks,
    float* __restrict__ out_min, int32_t* __restrict__ out_arg)
{
    const int b = blockIdx.y;
    // wave-uniform bounds (readfirstlane lets the row data below come in through scalar loads)
    const int beg = __builtin_amdgcn_readfirstlane(off[b]);
    const int n = counts ? __builtin_amdgcn_readfirstlane(counts[b]) : __builtin_amdgcn_readfirstlane(off[b + 1]) - beg;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a = blockIdx.x * 64 + lane;
    if ((int)(blockIdx.x * 64) >= n) return;
    const int ac = min(a, n - 1);
    const float px = pts[3 * (size_t)(beg + ac)], py = pts[3 * (size_t)(beg + ac) + 1], pz = pts[3 * (size_t)(beg + ac) + 2];
    const int va = vid[beg + ac];
    // the column's bit of a row's mask word sits in one 32-bit half of it: only that half is gathered
    const uint32_t* col = reinterpret_cast<const uint32_t*>(bits + (size_t)(va >> 6) * V) + ((va >> 5) & 1);
    const uint32_t bit = 1u << (va & 31);
    const float inf = __builtin_inff();
    float best = inf;
    int arg = 0;
    // optional seeds: the distance to some admissible row (any real row works: the result is the lexicographic
    // (distance, row) minimum whatever the search starts from) -- with near-final bounds pass 1 is not needed
    bool sample = true;
    if (seed_best) {
        best = seed_best[beg + ac];
        arg = seed_arg[beg + ac];
        sample = __builtin_amdgcn_ballot_w64(!(best < inf)) != 0;     // some column without a seed: sample after all
    }
    const float* rp = pts + 3 * (size_t)beg;
    const int32_t* rv = vid + beg;
    auto dist = [&](uint32_t word, float qx, float qy, float qz) {
        const float dx = px - qx, dy = py - qy, dz = pz - qz;
        const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        return (word & bit) ? d : inf;
    };
    auto take = [&](int r, float d) {
        if (d < best || (d == best && d < inf && r < arg)) { best = d; arg = r; }
    };
    auto row = [&](int r, int v, float qx, float qy, float qz) {
        const float d = dist(col[2 * (size_t)v], qx, qy, qz);
        if (__builtin_amdgcn_ballot_w64(d <= best)) take(r, d);              // wave-uniform
    };
    // kTrip rows r0, r0+step, ...: coordinates and template vertices through scalar loads, the per-lane
    // gathers of the mask words (word = column's 32-vertex block x row's vertex) issued together so that
    // their latency overlaps.  ONE compare + ballot branch per trip: on gfx950 a compare -> mask -> branch costs as
    // much as four FP32 ops (tools/ubench/valu_rate2.hip), and improvements are rare once the bounds have settled.
    constexpr int kTrip = 8;
    auto rows_trip = [&](int r0, int step) {
        float q[3 * kTrip];
        int v[kTrip];
        uint32_t w[kTrip];
        // one base address per trip: the 8 ids and 24 coordinates then come in as a few wide scalar loads
        const int32_t* vp = rv + (size_t)r0;
        const float* qp = rp + 3 * (size_t)r0;
#pragma unroll
        for (int u = 0; u < kTrip; ++u) {
            v[u] = vp[u * step];
            q[3 * u] = qp[3 * u * step]; q[3 * u + 1] = qp[3 * u * step + 1]; q[3 * u + 2] = qp[3 * u * step + 2];
        }
        // consecutive points usually inherit the same template vertex (samples of one face): the gather is
        // repeated only when the row's vertex changes (wave-uniform test)
        w[0] = col[2 * (size_t)v[0]];
#pragma unroll
        for (int u = 1; u < kTrip; ++u) {
            if (v[u] != v[u - 1]) w[u] = col[2 * (size_t)v[u]];
            else w[u] = w[u - 1];
        }
        float d[kTrip];
#pragma unroll
        for (int u = 0; u < kTrip; ++u) d[u] = dist(w[u], q[3 * u], q[3 * u + 1], q[3 * u + 2]);
        float m = d[0];
#pragma unroll
        for (int u = 1; u < kTrip; ++u) m = __builtin_fminf(m, d[u]);
        if (__builtin_amdgcn_ballot_w64(m <= best)) {
#pragma unroll
            for (int u = 0; u < kTrip; ++u) take(r0 + u * step, d[u]);      // in row order: ties go to the smaller row
        }
    };
    // chunks are dealt to the wavefronts round-robin: the chunks that survive the pruning are neighbours
    // (the contact partner's patch), contiguous shares would leave them all to one wavefront
    const int chunks = (n + kIndexedChunk - 1) / kIndexedChunk;
    __shared__ float sbest[kIndexedWaves][64];
    __shared__ int sarg[kIndexedWaves][64];
    auto merge = [&]() {                                 // lexicographic (d, row) minimum over the wavefronts
        sbest[wave][lane] = best;
        sarg[wave][lane] = arg;
        __syncthreads();
        for (int w = 0; w < kIndexedWaves; ++w) {
            const float d = sbest[w][lane];
            const int r = sarg[w][lane];
            if (d < best || (d == best && d < inf && r < arg)) { best = d; arg = r; }
        }
        __syncthreads();
    };
    for (int c = wave; sample && c < chunks; c += kIndexedWaves) {                                       // pass 1
        const int r = c * kIndexedChunk;
        if (r + 24 < n) {
            float q[12];
            int v[4];
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = rv[r + 8 * u];
                q[3 * u] = rp[3 * (r + 8 * u)]; q[3 * u + 1] = rp[3 * (r + 8 * u) + 1]; q[3 * u + 2] = rp[3 * (r + 8 * u) + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = col[2 * (size_t)v[u]];
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = dist(w[u], q[3 * u], q[3 * u + 1], q[3 * u + 2]);
            if (__builtin_amdgcn_ballot_w64(__builtin_fminf(__builtin_fminf(d[0], d[1]), __builtin_fminf(d[2], d[3])) <= best)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) take(r + 8 * u, d[u]);
            }
        } else {
            for (int rr = r; rr < n; rr += 8) row(rr, rv[rr], rp[3 * rr], rp[3 * rr + 1], rp[3 * rr + 2]);
        }
    }
    merge();
    const float* bx = boxes + (size_t)b * max_chunks * 8;
    for (int c = wave; c < chunks; c += kIndexedWaves) {                                                 // pass 2
        const float* box = bx + (size_t)c * 8;
        const float ex = __builtin_fmaxf(__builtin_fmaxf(box[0] - px, px - box[4]), 0.0f);
        const float ey = __builtin_fmaxf(__builtin_fmaxf(box[1] - py, py - box[5]), 0.0f);
        const float ez = __builtin_fmaxf(__builtin_fmaxf(box[2] - pz, pz - box[6]), 0.0f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) * kIndexedSlack;
        if (__builtin_amdgcn_ballot_w64(lb <= best) == 0) continue;
        int r = c * kIndexedChunk;
        const int re = min(n, r + kIndexedChunk);
        for (; r + kTrip <= re; r += kTrip) rows_trip(r, 1);
        for (; r < re; ++r) row(r, rv[r], rp[3 * r], rp[3 * r + 1], rp[3 * r + 2]);
    }
    merge();
    if (wave == 0 && a < n) {
        out_min[beg + a] = best;
        // every row masked: torch.min of an all-inf column returns index 0 (of the caller's ORIGINAL order, which the
        // caller may pass as all_masked_arg[b] when it keeps the points in another order)
        out_arg[beg + a] = best < inf ? arg : (all_masked_arg ? all_masked_arg[b] : 0);
    }
}

// ---- tree-pruned form ----------------------------------------------------------------------
// Same result as v2v_partial/merge, but most rows are never touched.  Vertices are renumbered in
// the cluster tree's order (cluster_tree.hip: the vertices of a leaf are consecutive, a block of 128
// columns is a compact patch) and the mask is packed in that numbering.  A wavefront owns 64
// columns and walks the tree: a node is skipped when the mask rules out every (column, row) pair
// below it (static, per model) or when no column can improve, i.e. for every lane the squared
// distance from its column to the node's posed box exceeds that column's current minimum
// (exact: the box distance is a lower bound of every row distance below the node).  Minima start
// from a seed (the nearest admissible leaf by box distance) and are shared between the subtree
// walks of a body through 64-bit (distance bits, row) keys merged with atomicMin, so the final
// key is the lexicographic minimum over all rows attaining the minimum: deterministic, ties go
// to the smallest row in tree order.
constexpr int kTreeCols = 64;
constexpr float kPruneSlack = 0.999999f;      // lower bounds are deflated by 1e-6: rounding of the two sums

#ifdef TUCH_SCAN_COUNTS
// diagnostic build only (tools/diag/scan_counts.py builds a second library with -DTUCH_SCAN_COUNTS): where the scan's
// instructions go.  [0] wavefronts, [1] leaf-per-lane trips, [2] candidate leaves (box against box), [3] leaves that reach
// their rows (per-column test), [4] trips of eight rows, [5] of them skipped by the mask words, [6] of them taking the
// update branch, [7] trips of four, [8] skipped, [9] taking, [10] wavefronts that return at once (no admissible row)
__device__ unsigned long long g_scan_counts[32];
#define SCAN_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_scan_counts[i], 1ull); } while (0)
extern "C" int tuch_debug_scan_counts(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_counts), sizeof(unsigned long long) * 32) != hipSuccess) return TUCH_ERR_HIP;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_counts), z, sizeof(z)) != hipSuccess) return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}
#else
#define SCAN_COUNT(i) do { } while (0)
#endif

#ifdef TUCH_SCAN_CLOCKS
// diagnostic build only (tools/diag/scan_clocks.py): per wavefront of the scan its start and end (s_memrealtime, 100 MHz),
// its place on the chip (HW_ID) and the trips / candidates it went through
constexpr int kScanClockSlots = 1 << 17;
// [4..7]: s_memtime ticks of the whole wavefront, in flushes, in rows walked on the spot, in the leaf-per-lane tests (with their loads)
__device__ unsigned long long g_scan_clocks[kScanClockSlots][8];
#define SCAN_TICK() __builtin_amdgcn_s_memtime()
extern "C" int tuch_debug_scan_clocks(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_clocks), sizeof(unsigned long long) * 8 * kScanClockSlots) == hipSuccess ? TUCH_OK : TUCH_ERR_HIP;
}
#endif

__device__ __forceinline__ uint64_t v2v_key(float d, int j)
{
    return ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)j;
}

// minimum over the 16 lanes of a DPP row, left in every lane (quad swaps, then the two mirror controls: single VALU
// instructions, no LDS crossbar)
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

// rows in tree order + box of every leaf's rows; 16 lanes per (leaf, body): a leaf is a few dozen rows, a whole wavefront
// per leaf was mostly idle lanes behind its chain of dependent loads
__device__ __forceinline__ void v2v_rows_body(
    int bx, const float* __restrict__ verts, int V, int Vp, const int32_t* __restrict__ qperm,
    const int32_t* __restrict__ rows, const int32_t* __restrict__ height_off,
    const int32_t* __restrict__ height_nodes, int N,
    float* __restrict__ prow,                    // [B,Vp,3]
    float* __restrict__ bounds,                  // [B,N,8]
    float* __restrict__ leafbox,                 // [B,L,8] or nullptr: the leaf boxes once more, by leaf index, with the
                                                 // leaf's row range (first | count << 20) in the last padding word
    const int32_t* __restrict__ leaf_group,      // with prow_g: first group of four rows of every leaf ([L + 1])
    float* __restrict__ prow_g, int G)           // [B,G,12] or nullptr: the rows once more, every leaf's rows padded to
{                                                // groups of four, a group = x[4] y[4] z[4]; box word [3] = first group
    const int b = blockIdx.y;
    const int group = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int i = height_off[0] + bx * (kBoundsBlock / 16) + group;
    const float* vb = verts + (size_t)b * V * 3;
    float* pb = prow + (size_t)b * Vp * 3;
    if (bx == 0 && threadIdx.x < 64)             // padding columns repeat the last vertex
        for (int j = V + (int)threadIdx.x; j < Vp; j += 64) {
            const int v = qperm[j];
            pb[3 * j] = vb[3 * v]; pb[3 * j + 1] = vb[3 * v + 1]; pb[3 * j + 2] = vb[3 * v + 2];
        }
    const bool real = i < height_off[1];         // all lanes stay: the DPP rows need them
    const int node = height_nodes[real ? i : height_off[0]];
    const int off = rows[2 * node], len = real ? rows[2 * node + 1] : 0;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, nhi[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    const int g0 = prow_g && real ? leaf_group[i - height_off[0]] : 0;
    float* pg = prow_g ? prow_g + ((size_t)b * G + g0) * 12 : nullptr;
    if (prow_g)                                  // the padding rows of the last group (their mask words are 0)
        for (int k = len + sub; k < ((len + 3) & ~3); k += 16) {
            float* o = pg + (k >> 2) * 12 + (k & 3);
            o[0] = 0.0f; o[4] = 0.0f; o[8] = 0.0f;
        }
    for (int j = off + sub; j < off + len; j += 16) {
        const int v = qperm[j];
        const float x = vb[3 * v], y = vb[3 * v + 1], z = vb[3 * v + 2];
        pb[3 * j] = x; pb[3 * j + 1] = y; pb[3 * j + 2] = z;
        if (prow_g) {
            float* o = pg + ((j - off) >> 2) * 12 + ((j - off) & 3);
            o[0] = x; o[4] = y; o[8] = z;
        }
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        nhi[0] = fminf(nhi[0], -x); nhi[1] = fminf(nhi[1], -y); nhi[2] = fminf(nhi[2], -z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = row_min(lo[k]); nhi[k] = row_min(nhi[k]); }
    if (real && sub == 0) {
        float* o = bounds + ((size_t)b * N + node) * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f;
        o[4] = -nhi[0]; o[5] = -nhi[1]; o[6] = -nhi[2]; o[7] = 0.0f;
        if (leafbox) {
            const int L = height_off[1] - height_off[0];
            float* q = leafbox + ((size_t)b * L + (i - height_off[0])) * 8;
            q[0] = lo[0]; q[1] = lo[1]; q[2] = lo[2]; q[3] = __int_as_float(g0);
            q[4] = -nhi[0]; q[5] = -nhi[1]; q[6] = -nhi[2]; q[7] = __int_as_float(off | (len << 20));
        }
    }
}

__global__ __launch_bounds__(kBoundsBlock) void v2v_rows_kernel(
    const float* __restrict__ verts, int V, int Vp, const int32_t* __restrict__ qperm, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ height_off, const int32_t* __restrict__ height_nodes, int N, float* __restrict__ prow,
    float* __restrict__ bounds, float* __restrict__ leafbox, const int32_t* __restrict__ leaf_group, float* __restrict__ prow_g,
    int G)
{
    v2v_rows_body(blockIdx.x, verts, V, Vp, qperm, rows, height_off, height_nodes, N, prow, bounds, leafbox, leaf_group, prow_g, G);
}

// One column per lane (64 columns per wavefront): the union of the columns' search balls is smaller for 64
// neighbours than for 128.  (Packed FP32 over pairs of ROWS was tried -- rows in groups of four, SoA, so that
// (x_j, x_j+1) sit in even-aligned scalar pairs: 22 % fewer VALU instructions, but the handling of ranges that do
// not start on a group boundary costs scalar instructions, and the CU's one scalar unit is this kernel's second
// bottleneck: 1.35e8 SALU + 0.21e8 SMEM per launch = 0.61 M cycles per CU; 289-311 us instead of 257.)
struct Column {
    float px, py, pz, best;
    int arg;
};

// rows [j0, j0+n) against the wave's 64 columns; ties keep the smaller row
// reach: the lanes that can still improve in this range (within reach of its box); a group of four rows none of
// which is allowed for any of them is skipped on the scalar unit
__device__ __forceinline__ void v2v_rows(Column& c, const float* __restrict__ pb, const uint64_t* __restrict__ m0,
                                         int j0, int n, uint64_t reach = ~0ull)
{
    const float inf = __builtin_inff();
    auto dist = [&](uint64_t k0, float vx, float vy, float vz) {
        const float dx = c.px - vx, dy = c.py - vy, dz = c.pz - vz;
        return select_by_lane_mask(inf, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)), k0);
    };
    auto take = [&](int j, float d) {
        if (d < c.best || (d == c.best && d < inf && j < c.arg)) { c.best = d; c.arg = j; }
    };
    auto row = [&](int j, uint64_t k0, float vx, float vy, float vz) {
        const float d = dist(k0, vx, vy, vz);
        if (__builtin_amdgcn_ballot_w64(d <= c.best)) take(j, d);              // rare, wave-uniform
    };
    int j = j0;
    const int j_end = j0 + n;
    // running pointers: the scalar unit is this kernel's second bottleneck, and re-deriving two 64-bit addresses from
    // the row index costs it eight instructions per trip
    const uint64_t* mp = m0 + j0;
    const float* cp = pb + 3 * (size_t)j0;
    // eight rows per trip while they last: the CU's one scalar unit is this loop's tightest resource (the loads, the test
    // on the mask words, the pointer bumps and branches of a trip are scalar instructions: 0.8 busy against 0.7 for the
    // vector units), and a trip of eight needs 18 of them where two trips of four need 28
    for (; j + 8 <= j_end; j += 8, mp += 8, cp += 24) {
        uint64_t k0[8];
        float v[24];
#pragma unroll
        for (int u = 0; u < 8; ++u) k0[u] = mp[u];
#pragma unroll
        for (int u = 0; u < 24; ++u) v[u] = cp[u];
        asm volatile("" :: "s"(v[0]), "s"(v[8]), "s"(v[16]));
        if (((k0[0] | k0[1] | k0[2] | k0[3] | k0[4] | k0[5] | k0[6] | k0[7]) & reach) == 0) continue;
        float d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = dist(k0[u], v[3 * u], v[3 * u + 1], v[3 * u + 2]);
        const float m = __builtin_fminf(min4_raw(d[0], d[1], d[2], d[3]), min4_raw(d[4], d[5], d[6], d[7]));
        if (__builtin_amdgcn_ballot_w64(m <= c.best)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) take(j + u, d[u]);                     // in row order: ties go to the smaller row
        }
    }
    for (; j + 4 <= j_end; j += 4, mp += 4, cp += 12) {
        uint64_t k0[4];
        float v[12];
#pragma unroll
        for (int u = 0; u < 4; ++u) k0[u] = mp[u];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = cp[u];
        // both loads are in flight before the test waits: one scalar-memory latency per trip, not two (the empty asm
        // keeps the compiler from sinking the coordinate loads behind the branch).  Requesting the NEXT trip's rows
        // during this trip's arithmetic (two register sets, the loop unrolled by two) was slower, 0.677 against
        // 0.656 ms per step: more scalar instructions, and the other wavefronts of the SIMD already cover the latency.
        asm volatile("" :: "s"(v[0]), "s"(v[4]), "s"(v[8]));
        if (((k0[0] | k0[1] | k0[2] | k0[3]) & reach) == 0) continue;
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = dist(k0[u], v[3 * u], v[3 * u + 1], v[3 * u + 2]);
        // ONE compare + ballot branch per four rows (a compare -> mask -> branch costs as much as four FP32 ops on
        // gfx950, tools/ubench/valu_rate2.hip); improvements are rare once the bounds have settled
        const float m = min4_raw(d[0], d[1], d[2], d[3]);
        if (__builtin_amdgcn_ballot_w64(m <= c.best)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) take(j + u, d[u]);                     // in row order: ties go to the smaller row
        }
    }
    for (; j < j_end; ++j) row(j, m0[j], pb[3 * j], pb[3 * j + 1], pb[3 * j + 2]);
}

// The same over a leaf's rows stored in groups of four (x[4] y[4] z[4], padded; mg: the mask words by padded row): two
// rows per packed FP32 instruction -- (x_j, x_j+1) sit in even-aligned scalar register pairs, and a leaf always starts
// on a group, so there is no ragged head or tail to handle on the scalar unit (what sank the first attempt, see
// above).  Same arithmetic per row (v_pk_mul / v_pk_fma round like their scalar forms): the same keys.
__device__ __forceinline__ void v2v_rows_packed(Column& c, const float* __restrict__ pg, const uint64_t* __restrict__ mg,
                                                int j0, int ngroups, uint64_t reach)
{
    const float inf = __builtin_inff();
    const v2f px = splat2(c.px), py = splat2(c.py), pz = splat2(c.pz);
    auto dist2 = [&](float xa, float xb, float ya, float yb, float za, float zb) {
        const v2f dx = px - (v2f){xa, xb}, dy = py - (v2f){ya, yb}, dz = pz - (v2f){za, zb};
        return fma2(dz, dz, fma2(dy, dy, dx * dx));
    };
    auto take = [&](int j, float d) {
        if (d < c.best || (d == c.best && d < inf && j < c.arg)) { c.best = d; c.arg = j; }
    };
    // ONE induction variable (the byte offset into the mask words; the rows' is 3/2 of it): the scalar unit issues as many
    // instructions in this loop as the vector units, and four running 64-bit quantities cost it eleven per trip
    const char* mb = reinterpret_cast<const char*>(mg);
    const char* pb = reinterpret_cast<const char*>(pg);
    uint32_t om = 0;
    const uint32_t om_end = (uint32_t)(ngroups >> 1) * 64u;
    for (; om < om_end; om += 64u) {
        asm("" : "+s"(om));                      // opaque: keeps the loop optimiser from splitting it into three again
        const uint64_t* mgo = reinterpret_cast<const uint64_t*>(mb + om);
        const float* pgo = reinterpret_cast<const float*>(pb + (om + (om >> 1)));
        uint64_t k0[8];
        float v[24];
#pragma unroll
        for (int u = 0; u < 8; ++u) k0[u] = mgo[u];
#pragma unroll
        for (int u = 0; u < 24; ++u) v[u] = pgo[u];
        asm volatile("" :: "s"(v[0]), "s"(v[8]), "s"(v[16]));
        SCAN_COUNT(4);
#ifdef TUCH_SCAN_COUNTS
        if ((k0[0] & k0[1] & k0[2] & k0[3] & k0[4] & k0[5] & k0[6] & k0[7]) == ~0ull) SCAN_COUNT(13);       // no mask needed at all
        if (((k0[0] & k0[1] & k0[2] & k0[3] & k0[4] & k0[5] & k0[6] & k0[7]) & reach) == reach) SCAN_COUNT(14);  // ... for the lanes in reach
#endif
        // (without this test -- nine scalar instructions per trip against 37 vector instructions for one trip in eight --
        // the kernel takes the same time: 129.5 against 127.8 us)
        if (((k0[0] | k0[1] | k0[2] | k0[3] | k0[4] | k0[5] | k0[6] | k0[7]) & reach) == 0) { SCAN_COUNT(5); continue; }
        const v2f d01 = dist2(v[0], v[1], v[4], v[5], v[8], v[9]), d23 = dist2(v[2], v[3], v[6], v[7], v[10], v[11]);
        const v2f d45 = dist2(v[12], v[13], v[16], v[17], v[20], v[21]), d67 = dist2(v[14], v[15], v[18], v[19], v[22], v[23]);
        float d[8];
        d[0] = select_by_lane_mask(inf, d01.x, k0[0]); d[1] = select_by_lane_mask(inf, d01.y, k0[1]);
        d[2] = select_by_lane_mask(inf, d23.x, k0[2]); d[3] = select_by_lane_mask(inf, d23.y, k0[3]);
        d[4] = select_by_lane_mask(inf, d45.x, k0[4]); d[5] = select_by_lane_mask(inf, d45.y, k0[5]);
        d[6] = select_by_lane_mask(inf, d67.x, k0[6]); d[7] = select_by_lane_mask(inf, d67.y, k0[7]);
        const float m = min8_raw(d);
        if (__builtin_amdgcn_ballot_w64(m <= c.best)) {                       // rare once the bounds have settled
            SCAN_COUNT(6);
            // the eight updates in row order amount to ONE with (m, first row attaining m): ties go to the smaller row
            int first = 7;
#pragma unroll
            for (int u = 6; u >= 0; --u) first = d[u] == m ? u : first;
            take(j0 + (int)(om >> 3) + first, m);
        }
    }
    if (ngroups & 1) {
        SCAN_COUNT(7);
        const uint64_t* mgo = reinterpret_cast<const uint64_t*>(mb + om);
        const float* pgo = reinterpret_cast<const float*>(pb + (om + (om >> 1)));
        uint64_t k0[4];
        float v[12];
#pragma unroll
        for (int u = 0; u < 4; ++u) k0[u] = mgo[u];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = pgo[u];
        asm volatile("" :: "s"(v[0]), "s"(v[4]), "s"(v[8]));
        if (((k0[0] | k0[1] | k0[2] | k0[3]) & reach) == 0) { SCAN_COUNT(8); return; }
        const v2f d01 = dist2(v[0], v[1], v[4], v[5], v[8], v[9]), d23 = dist2(v[2], v[3], v[6], v[7], v[10], v[11]);
        float d[4];
        d[0] = select_by_lane_mask(inf, d01.x, k0[0]); d[1] = select_by_lane_mask(inf, d01.y, k0[1]);
        d[2] = select_by_lane_mask(inf, d23.x, k0[2]); d[3] = select_by_lane_mask(inf, d23.y, k0[3]);
        const float m = min4_raw(d[0], d[1], d[2], d[3]);
        if (__builtin_amdgcn_ballot_w64(m <= c.best)) {
            SCAN_COUNT(9);
            int first = 3;
#pragma unroll
            for (int u = 2; u >= 0; --u) first = d[u] == m ? u : first;
            take(j0 + (int)(om >> 3) + first, m);
        }
    }
}

// squared distance from the lane's column to a box
__device__ __forceinline__ float box_dist2(const Column& c, const float* __restrict__ box)
{
    const float ex = __builtin_fmaxf(__builtin_fmaxf(box[0] - c.px, c.px - box[4]), 0.0f);
    const float ey = __builtin_fmaxf(__builtin_fmaxf(box[1] - c.py, c.py - box[5]), 0.0f);
    const float ez = __builtin_fmaxf(__builtin_fmaxf(box[2] - c.pz, c.pz - box[6]), 0.0f);
    return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
}

// seed: descend to the admissible leaf nearest to the block's box, evaluate its rows.
// grid (B, 64-column blocks); the static mask table is kept per 64-column block.
__global__ __launch_bounds__(64) void v2v_seed_kernel(
    const float* __restrict__ prow, int V, int Vp, const uint64_t* __restrict__ bits,
    const TreeNode* __restrict__ nodes, const int32_t* __restrict__ rows, const float* __restrict__ bounds,
    const uint64_t* __restrict__ masked, int N,
    const int32_t* __restrict__ hint,            // [B,Vp] a row per column (tree order) from an earlier call, or nullptr
    uint64_t* __restrict__ keys,                 // [B,Vp]
    float* __restrict__ colbox,                  // [B][column blocks][8] or nullptr: the box of the block's 64 columns
    const float* __restrict__ leafbox,           // [B][L][8] + masked_leaf [column blocks][L]: the leaf to seed from is found
    const uint64_t* __restrict__ masked_leaf, int L)   // among the leaves themselves (no inner boxes needed), or nullptr
{
    const int b = blockIdx.x, qb = blockIdx.y, lane = threadIdx.x;
    const float* pb = prow + (size_t)b * Vp * 3;
    const int i0 = qb * kTreeCols + lane;
    Column c;
    c.px = pb[3 * i0]; c.py = pb[3 * i0 + 1]; c.pz = pb[3 * i0 + 2];
    c.best = __builtin_inff();
    c.arg = 0;
    if (colbox) {
        float lo[3] = {c.px, c.py, c.pz}, hi[3] = {c.px, c.py, c.pz};
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], m));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m));
            }
        if (lane == 0) {
            float* o = colbox + ((size_t)b * gridDim.y + qb) * 8;
            o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
        }
    }
    const uint64_t* mk = masked + (size_t)qb * N;     // per node: the lanes with an allowed row below it
    bool have = false;
    if (hint) {
        // the partner found for this column by an earlier call (e.g. the previous iteration of a fit): still an
        // admissible row, so its current distance is a valid -- and usually almost final -- upper bound
        const int j = hint[(size_t)b * Vp + i0];
        if (j >= 0 && j < V && ((bits[(size_t)qb * V + j] >> lane) & 1)) {
            const float dx = c.px - pb[3 * j], dy = c.py - pb[3 * j + 1], dz = c.pz - pb[3 * j + 2];
            c.best = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            c.arg = j;
            have = true;
        }
    }
    // Every column that has an allowed row at all already holds a bound: the descent to the nearest admissible leaf and
    // its rows would add nothing the walk does not find (22 -> 8 us at the head of the search's chain in an iterative fit).
    if ((mk[0] & ~__builtin_amdgcn_ballot_w64(have)) != 0) {
        float lo[3] = {c.px, c.py, c.pz}, hi[3] = {c.px, c.py, c.pz};
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], m));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m));
            }
        if (leafbox) {
            // flat form: one leaf per lane, the nearest of those that hold an allowed row for a column still without a
            // bound (any admissible rows make a valid seed; the leaf scans do not need the inner nodes' boxes at all, so
            // tree_inner_bounds_kernel is no longer in their chain)
            const uint64_t need = mk[0] & ~__builtin_amdgcn_ballot_w64(have);
            const float inf = __builtin_inff();
            float gbest = inf;
            int lbest = 0x7fffffff;
            for (int base = 0; base < L; base += 64) {
                const int li = base + lane;
                float g = inf;
                if (li < L && (masked_leaf[(size_t)qb * L + li] & need) != 0) {
                    const float* box = leafbox + ((size_t)b * L + li) * 8;
                    g = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float e = fmaxf(fmaxf(box[k] - hi[k], lo[k] - box[4 + k]), 0.0f);
                        g = fmaf(e, e, g);
                    }
                }
                if (g < gbest) { gbest = g; lbest = li; }
            }
            float gmin = gbest;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) gmin = fminf(gmin, __shfl_xor(gmin, m));
            if (gmin < inf) {
                int pick = gbest == gmin ? lbest : 0x7fffffff;          // the smallest leaf index among the nearest
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) pick = min(pick, __shfl_xor(pick, m));
                pick = __builtin_amdgcn_readfirstlane(pick);
                const int range = __builtin_amdgcn_readfirstlane(__float_as_int(leafbox[((size_t)b * L + pick) * 8 + 7]));
                v2v_rows(c, pb, bits + (size_t)qb * V, range & 0xfffff, range >> 20);
            }
            keys[(size_t)b * Vp + i0] = v2v_key(c.best, c.arg);
            return;
        }
        const float* bb = bounds + (size_t)b * N * 8;
        auto gap2 = [&](int node) {                 // squared distance between the block's box and the node's box
            const float* box = bb + (size_t)node * 8;
            float g = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = fmaxf(fmaxf(box[k] - hi[k], lo[k] - box[4 + k]), 0.0f);
                g = fmaf(e, e, g);
            }
            return g;
        };
        int node = 0;
        bool ok = mk[0] != 0;
        while (ok) {
            const TreeNode nd = nodes[node];
            if (nd.c0 < 0) break;
            const bool a0 = mk[nd.c0] != 0, a1 = mk[nd.c1] != 0;
            if (a0 && a1) {
                const float g0 = gap2(nd.c0), g1 = gap2(nd.c1);
                node = __builtin_amdgcn_readfirstlane(g1 < g0 ? nd.c1 : nd.c0);
            } else if (a0 || a1) {
                node = a0 ? nd.c0 : nd.c1;
            } else {
                ok = false;
            }
        }
        if (ok) v2v_rows(c, pb, bits + (size_t)qb * V, rows[2 * node], rows[2 * node + 1]);
    }
    keys[(size_t)b * Vp + i0] = v2v_key(c.best, c.arg);
}

// v2v_rows_kernel and the seed of an iterative fit in ONE launch.  With partner hints from the previous call the seed is
// one gather per column; it needs the posed vertices, not the rows kernel's output (a row in tree order is
// verts[qperm[row]]), so the two have nothing to wait for in each other: workgroups [0, row_blocks) pose the rows and box
// the leaves, the others seed four column blocks each (behind the rows kernel and beside ray_near the seed took 26 us of
// the search's head).  No fall-back descent here: a column whose hint is not an admissible row starts without a bound
// (slower, never wrong) -- calls without hints take the two launches.
__global__ __launch_bounds__(kBoundsBlock) void v2v_rows_seed_kernel(
    const float* __restrict__ verts, int V, int Vp, const int32_t* __restrict__ qperm, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ height_off, const int32_t* __restrict__ height_nodes, int N, float* __restrict__ prow,
    float* __restrict__ bounds, float* __restrict__ leafbox, const int32_t* __restrict__ leaf_group, float* __restrict__ prow_g,
    int G, int row_blocks, const uint64_t* __restrict__ bits, const int32_t* __restrict__ hint, uint64_t* __restrict__ keys,
    float* __restrict__ colbox, uint4* __restrict__ zero, size_t zero_n16,
    const uint8_t* __restrict__ prev_exterior, float cap2)      // capped search (tuch_v2v_min_model_capped), or nullptr
{
    // a buffer the CALLER wants cleared before the kernels it enqueues behind this call run (SMPLify-DC stage 2: the vertex
    // gradient the tail scatters into, its arrival counter, the region pairs' keys -- a fill launch of 5 us in front of them
    // otherwise): one 16-byte store per thread, every workgroup of this launch takes part
    if (zero) {
        const size_t stride = (size_t)gridDim.x * gridDim.y * kBoundsBlock;
        for (size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kBoundsBlock + threadIdx.x; i < zero_n16; i += stride)
            zero[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if ((int)blockIdx.x < row_blocks) {
        v2v_rows_body(blockIdx.x, verts, V, Vp, qperm, rows, height_off, height_nodes, N, prow, bounds, leafbox, leaf_group,
                      prow_g, G);
        return;
    }
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int blocks = Vp / kTreeCols;
    const int qb = ((int)blockIdx.x - row_blocks) * (kBoundsBlock / 64) + ((int)threadIdx.x >> 6);
    if (qb >= blocks) return;
    const float* vb = verts + (size_t)b * V * 3;
    const int i0 = qb * kTreeCols + lane;
    const int v0 = qperm[i0];
    const float px = vb[3 * v0], py = vb[3 * v0 + 1], pz = vb[3 * v0 + 2];
    float best = __builtin_inff();
    int arg = 0;
    const int j = hint[(size_t)b * Vp + i0];
    if (j >= 0 && j < V && ((bits[(size_t)qb * V + j] >> lane) & 1)) {
        const int vj = qperm[j];
        const float dx = px - vb[3 * vj], dy = py - vb[3 * vj + 1], dz = pz - vb[3 * vj + 2];
        best = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        arg = j;
        // a column the previous call found OUTSIDE the body only matters to the caller if its partner is within the cap:
        // its bound starts at the cap (key = (cap^2, the hint): "nothing closer found" is recognisable, and the hint stays a
        // real admissible row).  Columns without an admissible hint, and those predicted inside, search without a cap.
        if (prev_exterior && i0 < V && prev_exterior[(size_t)b * V + v0] != 0) best = fminf(best, cap2);
    }
    keys[(size_t)b * Vp + i0] = v2v_key(best, arg);
    // box of the block's columns, rows behind the last vertex left out
    const bool real = i0 < V;
    const float inf = __builtin_inff();
    float lo[3] = {real ? px : inf, real ? py : inf, real ? pz : inf}, hi[3] = {real ? px : -inf, real ? py : -inf, real ? pz : -inf};
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = wave_min_uniform(lo[k]); hi[k] = wave_max_uniform(hi[k]); }
    if (lane == 0) {
        float* o = colbox + ((size_t)b * blocks + qb) * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
    }
}

__global__ __launch_bounds__(64) void v2v_tree_kernel(
    const float* __restrict__ prow, int V, int Vp, const uint64_t* __restrict__ bits,
    const TreeNode* __restrict__ nodes, const int32_t* __restrict__ rows, const float* __restrict__ bounds,
    const uint64_t* __restrict__ masked, int N, const int32_t* __restrict__ frontier,
    const int32_t* __restrict__ order, uint64_t* __restrict__ keys)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int pair = __builtin_amdgcn_readfirstlane(order[blockIdx.y >> 1]);      // launch order over 128-blocks
    const int sub = pair >> 16, qb = (pair & 0xffff) * 2 + (blockIdx.y & 1);
    const float* pb = prow + (size_t)b * Vp * 3;
    const int i0 = qb * kTreeCols + lane;
    uint64_t* kb = keys + (size_t)b * Vp;
    // any value read here is the key of a real row (seed, or another walk's improvement): a valid bound
    const uint64_t init = __hip_atomic_load(kb + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Column c;
    c.px = pb[3 * i0]; c.py = pb[3 * i0 + 1]; c.pz = pb[3 * i0 + 2];
    c.best = __uint_as_float((uint32_t)(init >> 32));
    c.arg = (int)(uint32_t)init;
    const float* bb = bounds + (size_t)b * N * 8;
    const uint64_t* mk = masked + (size_t)qb * N;     // per node: the lanes with an allowed row below it
    const uint64_t* m0 = bits + (size_t)qb * V;
    int node = __builtin_amdgcn_readfirstlane(frontier[sub]);
    const int end = __builtin_amdgcn_readfirstlane(nodes[node].skip);
    while (node < end) {
        // one round of scalar loads per node: its box, with the skip pointer and the leaf's row range in the two padding
        // words (tree_inner_bounds_kernel), and the lanes that have an allowed row below it
        const float* box = bb + (size_t)node * 8;
        const uint64_t lanes = mk[node];
        const float lx = box[0], ly = box[1], lz = box[2], hx = box[4], hy = box[5], hz = box[6];
        const int skip = __float_as_int(box[3]), leaf = __float_as_int(box[7]);
        // only a lane that has an allowed row below the node AND is within reach of its box can improve; the distance
        // to the box through the nearest point of it (median of three), bit for bit box_dist2()
        const float dx = c.px - __builtin_amdgcn_fmed3f(c.px, lx, hx);
        const float dy = c.py - __builtin_amdgcn_fmed3f(c.py, ly, hy);
        const float dz = c.pz - __builtin_amdgcn_fmed3f(c.pz, lz, hz);
        const float g = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * kPruneSlack;
        const uint64_t reach = __builtin_amdgcn_ballot_w64(g <= c.best) & lanes;
        if (reach == 0) {
            node = skip;
        } else if (leaf >= 0) {
            v2v_rows(c, pb, m0, leaf & 0xfffff, leaf >> 20, reach);
            node = skip;
        } else {
            node = node + 1;
        }
        node = __builtin_amdgcn_readfirstlane(node);
    }
    const uint64_t k0 = v2v_key(c.best, c.arg);
    if (k0 < init) atomicMin((unsigned long long*)(kb + i0), (unsigned long long)k0);
}

// Second form (the default): lanes over LEAVES first.  The walk above spends most of their vector instructions on box tests that fail
// (~20 node tests of ~12 instructions per wavefront against ~80 instructions of row arithmetic; the kernel is ~70 % VALU
// issue).  Here a wavefront tests all leaves of its subtree AT ONCE, one leaf per lane, against the box of its 64 columns
// and the largest bound among them (conservative: box-to-box distance), and only the survivors get the per-column test
// and their rows.  Same rows as the walks -> the same keys.  colbox: the box of every 64-column block, left by
// v2v_seed_kernel ([B][column blocks][8]).
// Round 5: the (leaf, column) pairs of the SPARSE leaves, one per lane.  Of the ~6 leaves of a subtree whose rows a
// wavefront evaluates, five are in reach of only ~8 of its 64 columns (tools/diag/scan_counts.py): walking their rows for all
// 64 lanes costs ~90 instructions per leaf = 11 per useful (leaf, column) pair, where a leaf in reach of 40 columns costs 2.
// Such leaves are no longer walked; their pairs go to a queue in LDS and are evaluated 64 at a time, every LANE its own
// pair: the leaf's rows in groups of four through per-lane 16-byte loads (the same padded groups v2v_rows_packed reads
// through scalar loads), the column from LDS, its admissible rows of the leaf one 64-bit window of ITS row of the bit matrix
// (symmetric mask, leaves of at most 64 rows), the same distance arithmetic, the result merged into the column's key in LDS
// with a 64-bit atomic minimum (several lanes may hold the same column with different leaves).  The same rows are
// evaluated as before -> the same keys.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPairQueue = 128;
__device__ __forceinline__ void v2v_flush_pairs(int count, const uint2* s_q, const float4* s_col,
                                                unsigned long long* s_key, const float* __restrict__ pg,
                                                const uint64_t* __restrict__ bits, int V, int gi_base, int lane)
{
    const float inf = __builtin_inff();
    const bool active = lane < count;
    const uint2 e = s_q[active ? lane : 0];
    const int range = (int)e.x, col = (int)(e.y >> 24), g0 = (int)(e.y & 0xffffffu);
    const int j0 = range & 0xfffff, nrows = range >> 20;
    const float4 p = s_col[col];
    const unsigned long long key0 = s_key[col];
    const int gi = gi_base + col;
    const uint64_t* brow = bits + (size_t)(j0 >> 6) * V + gi;              // bits[w][j]: word w of row j; symmetric mask
    const int sh = j0 & 63;
    uint64_t am = (gi < V ? brow[0] : 0ull) >> sh;
    if (sh + nrows > 64) am |= (gi < V ? brow[V] : 0ull) << (64 - sh);
    if (nrows < 64) am &= (1ull << nrows) - 1ull;
    float best = active ? __uint_as_float((uint32_t)(key0 >> 32)) : -1.0f;      // (idle lanes never improve)
    int arg = (int)(uint32_t)key0;
    const int ngroups = active ? (nrows + 3) >> 2 : 0;
    const int gmax = wave_max_uniform_i32(ngroups);
    const v2f px = splat2(p.x), py = splat2(p.y), pz = splat2(p.z);
    const float* pl = pg + (size_t)g0 * 12;
    // NOT-admissible bits, four rows per trip: a row's distance becomes a quiet NaN where its bit is set (two instructions:
    // bit -> all ones, AND-OR), and v_min3 / v_min skip NaN operands.  Lanes whose leaf has no group left repeat their last
    // one with every row masked (no branch around the loads).  The next trip's rows are NOT requested ahead (they were,
    // for a while): the twelve registers that took put the kernel at 74 -- six wavefronts per SIMD; at 62 there are eight,
    // and the search's time is its wavefronts' chains of dependent steps, not its instruction count (search alone at batch
    // 64: 122 -> 112 us)
    uint32_t nam = ~(uint32_t)am, nam_hi = ~(uint32_t)(am >> 32);
    auto maskd = [&](float d, uint32_t word, int k) {
        const int x = __builtin_amdgcn_sbfe(word, k, 1);                  // -1 where the row is not admissible
        uint32_t r;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(0x7FC00000u), "v"(__float_as_uint(d)));
        return __uint_as_float(r);
    };
    for (int g = 0; g < gmax; ++g) {
        if (g == 8) nam = nam_hi;
        const bool on = g < ngroups;
        const f32x4 x = *reinterpret_cast<const f32x4*>(pl), y = *reinterpret_cast<const f32x4*>(pl + 4),
                    z = *reinterpret_cast<const f32x4*>(pl + 8);
        if (g + 1 < ngroups) pl += 12;
        const v2f dx0 = px - (v2f){x[0], x[1]}, dy0 = py - (v2f){y[0], y[1]}, dz0 = pz - (v2f){z[0], z[1]};
        const v2f dx1 = px - (v2f){x[2], x[3]}, dy1 = py - (v2f){y[2], y[3]}, dz1 = pz - (v2f){z[2], z[3]};
        const v2f d01 = fma2(dz0, dz0, fma2(dy0, dy0, dx0 * dx0)), d23 = fma2(dz1, dz1, fma2(dy1, dy1, dx1 * dx1));
        const uint32_t word = on ? nam : 0xfu;
        float d[4] = {maskd(d01.x, word, 0), maskd(d01.y, word, 1), maskd(d23.x, word, 2), maskd(d23.y, word, 3)};
        const float m = min4_raw(d[0], d[1], d[2], d[3]);
        if (__builtin_amdgcn_ballot_w64(m <= best)) {
            int firstu = 3;
#pragma unroll
            for (int u = 2; u >= 0; --u) firstu = d[u] == m ? u : firstu;
            const int j = j0 + g * 4 + firstu;
            if (m < best || (m == best && m < inf && j < arg)) { best = m; arg = j; }
        }
        nam >>= 4;
    }
    const unsigned long long k1 = v2v_key(best, arg);
    if (active && k1 < key0) atomicMin(&s_key[col], k1);
}

template <int kShared, bool kPairs>
__device__ __forceinline__ void v2v_scan_body(
    const float* __restrict__ prow, int V, int Vp, const uint64_t* __restrict__ bits,
    const float* __restrict__ leafbox, const float* __restrict__ colbox, const uint64_t* __restrict__ masked_leaf,
    const uint64_t* __restrict__ masked, int N, int L, const int32_t* __restrict__ frontier,
    const int32_t* __restrict__ sub_leaf, const int32_t* __restrict__ order, uint64_t* __restrict__ keys,
    const float* __restrict__ prow_g, const uint64_t* __restrict__ bits_g, int G,   // rows / mask words in groups of four
    int dense)                                   // kPairs: a leaf in reach of fewer columns than this is queued, not walked
{
    // (72 registers = 7 wavefronts per SIMD, 80 = 6, ...: touching the last one is what sets the kernel's register count)
    if constexpr (kShared == 7) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if constexpr (kShared == 6) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if constexpr (kShared == 5) asm volatile("v_mov_b32 v99, 0" ::: "v99");
    if constexpr (kShared == 4) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const int b = blockIdx.x, lane = threadIdx.x;
#ifdef TUCH_SCAN_CLOCKS
    const unsigned long long clk0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long clk_cands = 0;
    const unsigned long long tick0 = SCAN_TICK();
    unsigned long long t_flush = 0, t_rows = 0, t_first = 0;
    const int clk_slot = (blockIdx.y * gridDim.x + blockIdx.x) & (kScanClockSlots - 1);
#endif
    const int pair = __builtin_amdgcn_readfirstlane(order[blockIdx.y >> 1]);      // launch order over 128-blocks
    const int sub = pair >> 16, qb = (pair & 0xffff) * 2 + (blockIdx.y & 1);
    const float* pb = prow + (size_t)b * Vp * 3;
    const int i0 = qb * kTreeCols + lane;
    uint64_t* kb = keys + (size_t)b * Vp;
    const uint64_t init = __hip_atomic_load(kb + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Column c;
    c.px = pb[3 * i0]; c.py = pb[3 * i0 + 1]; c.pz = pb[3 * i0 + 2];
    c.best = __uint_as_float((uint32_t)(init >> 32));
    c.arg = (int)(uint32_t)init;
    const float* pg = prow_g + (size_t)b * G * 12;
    const uint64_t* mg = bits_g + (size_t)qb * G * 4;
    const int first = __builtin_amdgcn_readfirstlane(sub_leaf[2 * sub]), count = __builtin_amdgcn_readfirstlane(sub_leaf[2 * sub + 1]);
    // the largest bound among the columns that have an allowed row below this subtree at all
    const uint64_t alive = masked[(size_t)qb * N + __builtin_amdgcn_readfirstlane(frontier[sub])];
    SCAN_COUNT(0);
    if (alive == 0) {
        SCAN_COUNT(10);
        return;
    }
    // lower bounds are compared as g <= bound * (1 + 1e-6): the slack of kPruneSlack on the side that changes rarely
    constexpr float kBoundSlack = 1.000001f;
    float reach2 = wave_max_uniform(((alive >> lane) & 1) ? c.best : 0.0f) * kBoundSlack;
    float best_s = c.best * kBoundSlack;
    // the boxes of a trip's 64 leaves, for the per-column test of the candidates among them: a candidate's box is one
    // broadcast read of LDS (two 16-byte reads) instead of six v_readlane + three v_mov (one scalar operand per vector
    // instruction) -- the per-column test is a third of this kernel's vector instructions (tools/diag/scan_counts.py: 21
    // candidates per wavefront, 6 of them reach their rows)
    __shared__ float4 leaf_lo[64], leaf_hi[64];
    // kPairs: the wavefront's columns and their keys for the lanes that evaluate queued (leaf, column) pairs
    __shared__ float4 s_col[kPairs ? 64 : 1];
    __shared__ unsigned long long s_key[kPairs ? 64 : 1];
    __shared__ uint2 s_q[kPairs ? kPairQueue : 1];        // (row range of the leaf, its first group | column << 24)
    int queued = 0;                                       // wave-uniform
    if constexpr (kPairs) {
        s_col[lane] = make_float4(c.px, c.py, c.pz, 0.0f);
        s_key[lane] = init;
    }
    auto flush = [&](int n) {
#ifdef TUCH_SCAN_COUNTS
        if (lane == 0) { atomicAdd(&g_scan_counts[16], 1ull); atomicAdd(&g_scan_counts[17], (unsigned long long)n); }
#endif
#ifdef TUCH_SCAN_CLOCKS
        const unsigned long long tf0 = SCAN_TICK();
#endif
        s_key[lane] = v2v_key(c.best, c.arg);             // (what the rows walked on the spot have found since)
        // (a one-wavefront workgroup: no s_barrier is emitted, but the fences are needed -- without them lanes read the
        // queue / the keys before the other lanes' writes: 157 of 4806 minima wrong on the 1602-vertex fixture)
        __syncthreads();
        v2v_flush_pairs(n, s_q, s_col, s_key, pg, bits, V, qb * kTreeCols, lane);
        __syncthreads();
        const unsigned long long k = s_key[lane];
        c.best = __uint_as_float((uint32_t)(k >> 32));
        c.arg = (int)(uint32_t)k;
#ifdef TUCH_SCAN_CLOCKS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
        t_flush += SCAN_TICK() - tf0;
#endif
    };
    const float* cbx = colbox + ((size_t)b * (Vp / kTreeCols) + qb) * 8;
    const float clx = cbx[0], cly = cbx[1], clz = cbx[2], chx = cbx[4], chy = cbx[5], chz = cbx[6];
    const float* lb = leafbox + ((size_t)b * L + first) * 8;
    const uint64_t* ml = masked_leaf + (size_t)qb * L + first;
    for (int base = 0; base < count; base += 64) {
        // one leaf per lane: the gap between its box and the block's, against the largest bound -- as it is NOW: a search
        // that started from poor bounds (new bodies) has tightened them in the trips before
        if (base > 0) reach2 = wave_max_uniform(((alive >> lane) & 1) ? c.best : 0.0f) * kBoundSlack;
        const int li = base + lane;
        bool cand = false;
        SCAN_COUNT(1);
#ifdef TUCH_SCAN_CLOCKS
        const unsigned long long tl0 = SCAN_TICK();
#endif
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;   // box; lo.w = first group, hi.w = row range of the leaf
        uint64_t lanes_of = 0;
        if (li < count) {
            lo = *reinterpret_cast<const float4*>(lb + (size_t)li * 8);
            hi = *reinterpret_cast<const float4*>(lb + (size_t)li * 8 + 4);
            lanes_of = ml[li];
            const float ex = fmaxf(fmaxf(lo.x - chx, clx - hi.x), 0.0f);
            const float ey = fmaxf(fmaxf(lo.y - chy, cly - hi.y), 0.0f);
            const float ez = fmaxf(fmaxf(lo.z - chz, clz - hi.z), 0.0f);
            const float g = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
            cand = g <= reach2 && (lanes_of & alive) != 0;
#ifdef TUCH_SCAN_COUNTS
            // what a first-level test against sub-blocks of 16 / 8 columns (own box, own largest bound) would let through
            if (cand) {
                bool any16 = false, any8 = false;
                for (int w = 8; w <= 16; w += 8)
                    for (int s0 = 0; s0 < 64; s0 += w) {
                        float bl[3] = {3e38f, 3e38f, 3e38f}, bh[3] = {-3e38f, -3e38f, -3e38f}, bb = 0.0f;
                        bool live = false;
                        for (int t = s0; t < s0 + w; ++t) {
                            const float* q = pb + 3 * (size_t)(qb * kTreeCols + t);
                            for (int k = 0; k < 3; ++k) { bl[k] = fminf(bl[k], q[k]); bh[k] = fmaxf(bh[k], q[k]); }
                            if ((alive >> t) & (lanes_of >> t) & 1) {
                                live = true;
                                bb = fmaxf(bb, __uint_as_float((uint32_t)(__hip_atomic_load(kb + qb * kTreeCols + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32)));
                            }
                        }
                        const float fx = fmaxf(fmaxf(lo.x - bh[0], bl[0] - hi.x), 0.0f);
                        const float fy = fmaxf(fmaxf(lo.y - bh[1], bl[1] - hi.y), 0.0f);
                        const float fz = fmaxf(fmaxf(lo.z - bh[2], bl[2] - hi.z), 0.0f);
                        const bool pass = live && (fx * fx + fy * fy + fz * fz) * kPruneSlack <= bb;
                        if (w == 16) any16 |= pass; else any8 |= pass;
                    }
                if (any16) atomicAdd(&g_scan_counts[11], 1ull);
                if (any8) atomicAdd(&g_scan_counts[12], 1ull);
            }
#endif
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(cand);
#ifdef TUCH_SCAN_CLOCKS
        clk_cands += __builtin_popcountll(todo);
        t_first += SCAN_TICK() - tl0;
#endif
        if (todo == 0) continue;
        leaf_lo[lane] = lo;                      // one wavefront per workgroup: LDS operations of a wavefront stay in order
        leaf_hi[lane] = hi;
        while (todo) {
            const int u = __builtin_ctzll(todo);
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(u));
            SCAN_COUNT(2);
            const float4 blo = leaf_lo[u], bhi = leaf_hi[u];
            const uint64_t lanes = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(lanes_of >> 32), u) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)lanes_of, u);
            const float dx = c.px - __builtin_amdgcn_fmed3f(c.px, blo.x, bhi.x);
            const float dy = c.py - __builtin_amdgcn_fmed3f(c.py, blo.y, bhi.y);
            const float dz = c.pz - __builtin_amdgcn_fmed3f(c.pz, blo.z, bhi.z);
            const float g = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const uint64_t reach = __builtin_amdgcn_ballot_w64(g <= best_s) & lanes;
            if (reach) {
                SCAN_COUNT(3);
#ifdef TUCH_SCAN_COUNTS
                if (threadIdx.x == 0) atomicAdd(&g_scan_counts[15], (unsigned long long)__builtin_popcountll(reach));   // columns in reach
#endif
                const int nreach = __builtin_popcountll(reach);
                if (kPairs && nreach < dense) {
                    if ((reach >> lane) & 1) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(reach >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)reach, 0));
                        s_q[queued + rank] = make_uint2(__float_as_uint(bhi.w), __float_as_uint(blo.w) | ((uint32_t)lane << 24));
                    }
                    queued += nreach;
                    if (queued >= 64) {                   // a full wavefront of pairs: evaluate, keep the rest
                        flush(64);
                        best_s = c.best * kBoundSlack;
                        queued -= 64;
                        const uint2 rest = s_q[64 + (lane < queued ? lane : 0)];
                        __syncthreads();
                        if (lane < queued) s_q[lane] = rest;
                        __syncthreads();
                    }
                } else {
#ifdef TUCH_SCAN_COUNTS
                    if (lane == 0) { atomicAdd(&g_scan_counts[18], 1ull); atomicAdd(&g_scan_counts[19], (unsigned long long)nreach); }
#endif
                    const int leaf = __builtin_amdgcn_readfirstlane(__float_as_int(bhi.w));
                    const int g0 = __builtin_amdgcn_readfirstlane(__float_as_int(blo.w));
#ifdef TUCH_SCAN_CLOCKS
                    const unsigned long long tr0 = SCAN_TICK();
#endif
                    v2v_rows_packed(c, pg + (size_t)g0 * 12, mg + (size_t)g0 * 4, leaf & 0xfffff, ((leaf >> 20) + 3) >> 2, reach);
                    best_s = c.best * kBoundSlack;
#ifdef TUCH_SCAN_CLOCKS
                    t_rows += SCAN_TICK() - tr0;
#endif
                }
            }
        }
    }
    if constexpr (kPairs) {
        if (queued > 0) flush(queued);
    }
    const uint64_t k0 = v2v_key(c.best, c.arg);
    if (k0 < init) atomicMin((unsigned long long*)(kb + i0), (unsigned long long)k0);
#ifdef TUCH_SCAN_CLOCKS
    if (lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_scan_clocks[clk_slot][0] = clk0;
        g_scan_clocks[clk_slot][1] = __builtin_amdgcn_s_memrealtime();
        g_scan_clocks[clk_slot][2] = ((unsigned long long)xcc << 32) | hw;
        g_scan_clocks[clk_slot][3] = ((unsigned long long)blockIdx.y << 32) | (clk_cands << 8) | (unsigned)((count + 63) / 64);
        g_scan_clocks[clk_slot][4] = SCAN_TICK() - tick0;
        g_scan_clocks[clk_slot][5] = t_flush;
        g_scan_clocks[clk_slot][6] = t_rows;
        g_scan_clocks[clk_slot][7] = t_first;
    }
#endif
}

#define TUCH_SCAN_PARAMS                                                                                                   \
    const float* __restrict__ prow, int V, int Vp, const uint64_t* __restrict__ bits, const float* __restrict__ leafbox,    \
    const float* __restrict__ colbox, const uint64_t* __restrict__ masked_leaf, const uint64_t* __restrict__ masked, int N, \
    int L, const int32_t* __restrict__ frontier, const int32_t* __restrict__ sub_leaf, const int32_t* __restrict__ order,   \
    uint64_t* __restrict__ keys, const float* __restrict__ prow_g, const uint64_t* __restrict__ bits_g, int G, int dense
#define TUCH_SCAN_ARGS prow, V, Vp, bits, leafbox, colbox, masked_leaf, masked, N, L, frontier, sub_leaf, order, keys, prow_g, bits_g, G, dense
template <bool kPairs>
__global__ __launch_bounds__(64) void v2v_scan_kernel(TUCH_SCAN_PARAMS) { v2v_scan_body<0, kPairs>(TUCH_SCAN_ARGS); }
// The same beside the inside test's chain of small kernels (another stream): at most kSlots of a SIMD's 8 wave slots, by
// REGISTER count.  (Round 2 capped the walk with an unused LDS allocation -- 25 x 6400 B is ALL of a CU's LDS: the
// chain's kernels that need LDS themselves, ray_near and ray_tiles_fill, then waited for the search to drain.)
template <int kSlots, bool kPairs>
__global__ __launch_bounds__(64) void v2v_scan_shared_kernel(TUCH_SCAN_PARAMS)
{
    v2v_scan_body<kSlots, kPairs>(TUCH_SCAN_ARGS);
}
#undef TUCH_SCAN_PARAMS
#undef TUCH_SCAN_ARGS

// keys -> (min, argmin) in the caller's vertex numbering; all-masked column -> (inf, 0)
__global__ __launch_bounds__(kBlock) void v2v_tree_finalize_kernel(
    const uint64_t* __restrict__ keys, const int32_t* __restrict__ qperm, int V, int Vp,
    float* __restrict__ out_min, int32_t* __restrict__ out_arg, int32_t* __restrict__ hint)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= V) return;
    const uint64_t k = keys[(size_t)b * Vp + i];
    const float d = __uint_as_float((uint32_t)(k >> 32));
    const int v = qperm[i];
    if (hint) hint[(size_t)b * Vp + i] = d < __builtin_inff() ? (int)(uint32_t)k : -1;
    if (out_min) out_min[(size_t)b * V + v] = d;
    if (out_arg) out_arg[(size_t)b * V + v] = d < __builtin_inff() ? qperm[(uint32_t)k] : 0;
}

// The second half of the capped search: columns whose key still is (cap^2, hint) -- nothing within the cap -- and that the
// inside test has meanwhile found INSIDE the body need their exact partner after all (their term has no cap).  One wavefront per
// block of 64 columns; such a column is searched by all 64 lanes over ALL rows (the same d^2 expression and the same (d^2, row)
// order as the scan: the exact result, whatever the scan would have pruned).  Few columns per call in an iterative fit (the
// vertices that crossed the surface since the previous iteration); any number is handled, only slower.  Also records this
// call's flags as the next call's prediction.
constexpr int kFixWaves = 4;
__global__ __launch_bounds__(64 * kFixWaves) void v2v_fix_kernel(
    const float* __restrict__ prow, int V, int Vp, const uint64_t* __restrict__ bits, const int32_t* __restrict__ qperm,
    const uint8_t* __restrict__ exterior, uint8_t* __restrict__ prev_exterior, float cap2, uint64_t* __restrict__ keys,
    float* __restrict__ out_min, int32_t* __restrict__ out_arg, int32_t* __restrict__ hint)
{
    // every wavefront of the workgroup holds the block's 64 columns; a column to search is shared by ROWS: wavefront w takes
    // the w-th quarter, four rows per lane and trip (eight independent loads in flight), the quarters' minima meet in LDS
    __shared__ uint64_t part[kFixWaves][64];
    __shared__ unsigned long long s_todo;
    const int qb = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = qb * kTreeCols + lane;
    const bool real = i0 < V;
    // ONE wavefront reads the flags and decides which columns are searched (a caller may run this kernel beside the segment
    // filter, which re-marks `exterior` -- either value is fine: a vertex it re-marks needs no partner beyond the cap --, and
    // wavefronts that read different values must not disagree about the columns they share)
    int v = 0;
    bool need = false;
    if (wave == 0) {
        v = qperm[real ? i0 : V - 1];
        const uint8_t ext = real ? exterior[(size_t)b * V + v] : (uint8_t)1;
        const uint64_t k0 = keys[(size_t)b * Vp + i0];
        need = real && ext == 0 && (uint32_t)(k0 >> 32) == __float_as_uint(cap2);
        const unsigned long long any = __builtin_amdgcn_ballot_w64(need);
        if (lane == 0) s_todo = any;
        if (real) prev_exterior[(size_t)b * V + v] = ext;
    }
    __syncthreads();
    unsigned long long todo = s_todo;
    if (!todo) return;
    const float* pb = prow + (size_t)b * Vp * 3;
    const uint64_t* mrow = bits + (size_t)qb * V;
    const int per = (V + kFixWaves - 1) / kFixWaves, beg = wave * per, end = min(beg + per, V);
    uint64_t mine = ~0ull;
    while (todo) {
        const int c = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const float* pc = pb + 3 * (size_t)(qb * kTreeCols + c);
        const float px = pc[0], py = pc[1], pz = pc[2];
        uint64_t best = ~0ull;
        for (int j0 = beg + lane; j0 < end; j0 += 256) {
            uint64_t mw[4];
            float r[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + 64 * u, end - 1);
                mw[u] = mrow[j];
                r[u][0] = pb[3 * j]; r[u][1] = pb[3 * j + 1]; r[u][2] = pb[3 * j + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 64 * u;
                const float dx = px - r[u][0], dy = py - r[u][1], dz = pz - r[u][2];
                const uint64_t key = v2v_key(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)), j);
                const bool ok = j < end && ((mw[u] >> c) & 1ull);
                best = ok && key < best ? key : best;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(best >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, m);
            best = o < best ? o : best;
        }
        if (lane == c) mine = best;
    }
    part[wave][lane] = mine;
    __syncthreads();
    if (wave == 0 && need) {
        uint64_t k = part[0][lane];
#pragma unroll
        for (int w = 1; w < kFixWaves; ++w) k = part[w][lane] < k ? part[w][lane] : k;
        const bool found = k != ~0ull;
        const float d = found ? __uint_as_float((uint32_t)(k >> 32)) : __builtin_inff();
        keys[(size_t)b * Vp + i0] = found ? k : v2v_key(__builtin_inff(), 0);
        if (hint) hint[(size_t)b * Vp + i0] = found ? (int)(uint32_t)k : -1;
        if (out_min) out_min[(size_t)b * V + v] = d;
        if (out_arg) out_arg[(size_t)b * V + v] = found ? qperm[(uint32_t)k] : 0;
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct TreeV2VLayout { size_t prow, bounds, keys, leafbox, colbox, prow_g, total; };

static int flat_mode(const tuch_contact_model* m);
int choose_v2v_frontier(const tuch_contact_model* m, int B, bool iterative = false);

TreeV2VLayout tree_v2v_layout(const tuch_contact_model* m, int B)
{
    TreeV2VLayout l;
    size_t o = 0;
    const int Vp = m->tree_qblocks * 2 * kTreeCols;
    l.prow = tuch_ws_take(o, (size_t)B * Vp * 3 * sizeof(float) + 64);
    l.bounds = tuch_ws_take(o, (size_t)B * m->tree_nodes * 8 * sizeof(float));
    l.keys = tuch_ws_take(o, (size_t)B * Vp * sizeof(uint64_t));
    l.leafbox = tuch_ws_take(o, ((size_t)B * m->tree_leaves + 4) * 8 * sizeof(float));     // + padding
    l.colbox = tuch_ws_take(o, (size_t)B * 2 * m->tree_qblocks * 8 * sizeof(float));
    l.prow_g = tuch_ws_take(o, ((size_t)B * m->tree_groups * 12 + 16) * sizeof(float));     // (+ a trip's read-ahead)
    l.total = o;
    return l;
}

bool use_v2v_tree(const tuch_contact_model* m)
{
    if (m->tree_nodes <= 0 || !m->tree_mask_bits || !m->tree_v2v_info) return false;
    return m->opt.v2v_tree != 0;
}

static int flat_mode(const tuch_contact_model* m)
{
    // 2: lanes over a subtree's leaves first (v2v_scan_kernel, the default); 0: the stackless walk (v2v_tree_kernel: models
    // without the leaf tables, and the A/B reference).  (Rounds 2-3 also had "leaf boxes four at a time" and a matrix-core
    // form: identical keys, both slower -- DESIGN.md section 3; removed in round 4.)
    if (!(m->tree_sub_leaf && m->tree_masked_leaf)) return 0;
    return m->opt.v2v_flat >= 2 ? 2 : 0;
}

int choose_v2v_frontier(const tuch_contact_model* m, int B, bool iterative)
{
    // option v2v_waves = 0: the form's own default -- the walks want many short wavefronts (65536: 32 subtrees at batch
    // 64), the leaf scan tests up to 64 leaves per wavefront at once and is best with a quarter as many (measured at
    // batch 64, step time: 7000 / 14000 / 30000 / 65536 -> 0.56+ / 0.546 / 0.550 / 0.562 ms, round 3).  Round 5, with the
    // sparse pairs one per lane: a caller that says its bounds are near-final (an iterative fit: the hints are the previous
    // iteration's partners) gets a quarter of that again -- fewer prologues, fuller flushes: 0.412 against 0.426 ms per
    // step at batch 64 --; on NEW bodies the search is slower that way (183 against 163 us: two subtrees per block tighten
    // poor bounds later than eight do), so that stays the default
    const long target = m->opt.v2v_waves > 0 ? m->opt.v2v_waves : (flat_mode(m) >= 2 ? (iterative ? 3500L : 14000L) : 65536L);
    int f = 0;
    while (f + 1 < m->tree_num_frontiers &&
           (long)B * m->tree_qblocks * (m->tree_frontier_off_host[f + 1] - m->tree_frontier_off_host[f]) < target)
        ++f;
    return f;
}

int choose_row_splits(int B, int V)
{
    const int cblocks = ceil_div(V, kColsPerBlock);
    int s = 1;
    while (s < 16 && (long)B * cblocks * s < 2048 && V / (s * 2) >= 256) s *= 2;
    return s;
}

}  // namespace

extern "C" int tuch_geomask_words(int V) { return ((V + 63) / 64 + 1) & ~1; }

extern "C" size_t tuch_geomask_bits_bytes(int V)
{
    return V > 0 ? (size_t)tuch_geomask_words(V) * V * sizeof(uint64_t) : 0;
}

extern "C" int tuch_pack_geomask(const uint8_t* geomask, int V, uint64_t* bits, void* stream)
{
    TUCH_REQUIRE(geomask && bits && V > 0, "tuch_pack_geomask: bad arguments");
    const int W = tuch_geomask_words(V);
    hipLaunchKernelGGL(pack_mask_kernel, dim3(ceil_div(V, kBlock), W), dim3(kBlock), 0,
                       (hipStream_t)stream, geomask, V, W, bits);
    return tuch_check_launch("tuch_pack_geomask");
}

extern "C" size_t tuch_v2v_workspace_bytes(int B, int V)
{
    if (B <= 0 || V <= 0) return 0;
    return (size_t)B * choose_row_splits(B, V) * V * (sizeof(float) + sizeof(int));
}

extern "C" int tuch_v2v_min_masked(const float* verts, const uint64_t* geomask_bits, int B, int V,
                                   float* min_d2, int32_t* argmin, void* workspace,
                                   size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(verts && geomask_bits && (min_d2 || argmin), "tuch_v2v_min_masked: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && V > 0, "tuch_v2v_min_masked: bad sizes B=%d V=%d", B, V);
    const int nsplit = choose_row_splits(B, V);
    const size_t n = (size_t)B * nsplit * V;
    if (!workspace || workspace_bytes < n * (sizeof(float) + sizeof(int))) {
        tuch_set_error("tuch_v2v_min_masked: workspace %zu < %zu bytes", workspace_bytes,
                       n * (sizeof(float) + sizeof(int)));
        return TUCH_ERR_WORKSPACE;
    }
    float* pmin = (float*)workspace;
    int* parg = (int*)(pmin + n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(v2v_partial_kernel, dim3(B, nsplit, ceil_div(V, kColsPerBlock)), dim3(kBlock), 0, s,
                       verts, geomask_bits, V, ceil_div(V, nsplit), pmin, parg);
    hipLaunchKernelGGL(v2v_merge_kernel, dim3(ceil_div(V, kBlock), B), dim3(kBlock), 0, s,
                       (const float*)pmin, (const int*)parg, V, nsplit, min_d2, argmin);
    return tuch_check_launch("tuch_v2v_min_masked");
}

// Model-level form of tuch_v2v_min_masked (tuch/smplify/losses.py:76-78,92-93; tuch/train/loss.py:255-257,
// 269-270): the model's mask, and -- when the model has a cluster tree -- the tree-pruned walk.
extern "C" size_t tuch_v2v_model_workspace_bytes(const tuch_contact_model* m, int B)
{
    if (!m || B <= 0) return 0;
    const size_t flat = tuch_v2v_workspace_bytes(B, m->V);
    if (m->tree_nodes <= 0 || !m->tree_mask_bits) return flat;
    tuch_ws_scope scope(m->opt.canary != 0);
    const size_t tree = tree_v2v_layout(m, B).total;
    return tree > flat ? tree : flat;
}

extern "C" size_t tuch_v2v_hint_bytes(const tuch_contact_model* m, int B)
{
    if (!m || B <= 0 || m->tree_nodes <= 0 || !m->tree_mask_bits) return 0;
    return (size_t)B * m->tree_qblocks * 2 * kTreeCols * sizeof(int32_t);
}

extern "C" int tuch_v2v_min_model_shared_zero(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                                              int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                                              int leave_room, void* zero, size_t zero_bytes, void* stream);
extern "C" int tuch_v2v_min_model_shared(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                                         int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                                         int leave_room, void* stream)
{
    return tuch_v2v_min_model_shared_zero(m, verts, B, min_d2, argmin, hint_inout, workspace, workspace_bytes, leave_room, nullptr, 0,
                                          stream);
}

extern "C" int tuch_v2v_min_model(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                                  int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes, void* stream)
{
    return tuch_v2v_min_model_shared(m, verts, B, min_d2, argmin, hint_inout, workspace, workspace_bytes, 0, stream);
}

// zero / zero_bytes (multiple of 16, or NULL / 0): a caller buffer cleared by this call's FIRST kernel -- on the stream, before
// anything enqueued behind the call (no fill launch of the caller's own).
static int v2v_min_model_impl(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                              int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                              int leave_room, void* zero, size_t zero_bytes, void* stream, const uint8_t* prev_exterior, float cap);

extern "C" int tuch_v2v_min_model_shared_zero(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                                              int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                                              int leave_room, void* zero, size_t zero_bytes, void* stream)
{
    return v2v_min_model_impl(m, verts, B, min_d2, argmin, hint_inout, workspace, workspace_bytes, leave_room, zero, zero_bytes,
                              stream, nullptr, 0.0f);
}

// can this model's search run capped?  (the leaf scan seeded from hints: what the cap rides on)
extern "C" int tuch_v2v_min_model_can_cap(const tuch_contact_model* m)
{
    return m && m->mask_bits && use_v2v_tree(m) && flat_mode(m) >= 2 && m->opt.v2v_cap != 0 ? 1 : 0;
}

extern "C" int tuch_v2v_min_model_capped(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                                         int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                                         int leave_room, void* zero, size_t zero_bytes, const uint8_t* prev_exterior, float cap,
                                         void* stream)
{
    TUCH_REQUIRE(prev_exterior && hint_inout && cap >= 0.0f, "tuch_v2v_min_model_capped: needs hints, the previous flags and a cap >= 0");
    TUCH_REQUIRE(tuch_v2v_min_model_can_cap(m), "tuch_v2v_min_model_capped: this model's search cannot be capped");
    return v2v_min_model_impl(m, verts, B, min_d2, argmin, hint_inout, workspace, workspace_bytes, leave_room, zero, zero_bytes,
                              stream, prev_exterior, cap);
}

extern "C" int tuch_v2v_min_model_fix(const tuch_contact_model* m, int B, const uint8_t* exterior, uint8_t* prev_exterior,
                                      float cap, float* min_d2, int32_t* argmin, void* hint_inout, void* workspace,
                                      size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(m && exterior && prev_exterior && workspace, "tuch_v2v_min_model_fix: null pointer");
    TUCH_REQUIRE(tuch_v2v_min_model_can_cap(m) && B > 0 && B <= 65535, "tuch_v2v_min_model_fix: bad arguments");
    const TreeV2VLayout l = tree_v2v_layout(m, B);
    TUCH_REQUIRE(workspace_bytes >= l.total, "tuch_v2v_min_model_fix: not the workspace of the capped call");
    char* ws = (char*)workspace;
    const int Vp = m->tree_qblocks * 2 * kTreeCols;
    hipLaunchKernelGGL(v2v_fix_kernel, dim3(2 * m->tree_qblocks, B), dim3(64 * kFixWaves), 0, (hipStream_t)stream, (const float*)(ws + l.prow), m->V, Vp,
                       (const uint64_t*)m->tree_mask_bits, (const int32_t*)m->tree_qperm, exterior, prev_exterior, cap * cap,
                       (uint64_t*)(ws + l.keys), min_d2, argmin, (int32_t*)hint_inout);
    return tuch_check_launch("tuch_v2v_min_model_fix");
}

static int v2v_min_model_impl(const tuch_contact_model* m, const float* verts, int B, float* min_d2,
                              int32_t* argmin, void* hint_inout, void* workspace, size_t workspace_bytes,
                              int leave_room, void* zero, size_t zero_bytes, void* stream, const uint8_t* prev_exterior, float cap)
{
    TUCH_REQUIRE(m && verts && (min_d2 || argmin), "tuch_v2v_min_model: null pointer");
    TUCH_REQUIRE(m->mask_bits, "tuch_v2v_min_model: the model has no geodesic mask");
    const bool iterative = (leave_room & 2) != 0;      // flags: 1 leave room for kernels on another stream, 2 bounds are near-final
    leave_room &= 1;
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_v2v_min_model: bad batch %d", B);
    TUCH_REQUIRE((zero_bytes & 15) == 0 && (((uintptr_t)zero) & 15) == 0 && (zero || zero_bytes == 0),
                 "tuch_v2v_min_model: the buffer to clear must be 16-byte aligned and a multiple of 16 bytes");
    const bool fused_zero = zero && zero_bytes && use_v2v_tree(m) && flat_mode(m) >= 2 && hint_inout;
    if (zero && zero_bytes && !fused_zero && hipMemsetAsync(zero, 0, zero_bytes, (hipStream_t)stream) != hipSuccess)
        return tuch_check_launch("tuch_v2v_min_model: clearing the caller's buffer");
    if (!use_v2v_tree(m))
        return tuch_v2v_min_masked(verts, m->mask_bits, B, m->V, min_d2, argmin, workspace, workspace_bytes, stream);
    tuch_ws_scope scope(m->opt.canary != 0);
    const TreeV2VLayout l = scope.record(0, [&] { return tree_v2v_layout(m, B); });
    if (!workspace || workspace_bytes < l.total) {
        tuch_set_error("tuch_v2v_min_model: workspace %zu < %zu bytes", workspace_bytes, l.total);
        return TUCH_ERR_WORKSPACE;
    }
    scope.arm(workspace, m->canary_hits, (hipStream_t)stream);
    char* ws = (char*)workspace;
    float* prow = (float*)(ws + l.prow);
    float* bounds = (float*)(ws + l.bounds);
    uint64_t* keys = (uint64_t*)(ws + l.keys);
    float* leafbox = (float*)(ws + l.leafbox);
    float* colbox = (float*)(ws + l.colbox);
    const int scan = flat_mode(m);          // 2: lanes over leaves first, 0: the walk
    hipStream_t s = (hipStream_t)stream;
    const int V = m->V, Vp = m->tree_qblocks * 2 * kTreeCols, N = m->tree_nodes;
    const TreeNode* nodes = (const TreeNode*)m->tree_node;
    const int row_blocks = ceil_div(m->tree_leaves, kBoundsBlock / 16);
    if (scan >= 2 && hint_inout) {
        // rows + the seed from the previous call's partners in one launch (see v2v_rows_seed_kernel)
        hipLaunchKernelGGL(v2v_rows_seed_kernel, dim3(row_blocks + ceil_div(2 * m->tree_qblocks, kBoundsBlock / 64), B),
                           dim3(kBoundsBlock), 0, s, verts, V, Vp, (const int32_t*)m->tree_qperm, (const int32_t*)m->tree_rows,
                           (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, N, prow, bounds, leafbox,
                           (const int32_t*)m->tree_leaf_group, scan >= 2 ? (float*)(ws + l.prow_g) : (float*)nullptr,
                           m->tree_groups, row_blocks, (const uint64_t*)m->tree_mask_bits, (const int32_t*)hint_inout, keys,
                           colbox, (uint4*)(fused_zero ? zero : nullptr),
                           fused_zero ? zero_bytes / 16 : (size_t)0, prev_exterior, cap * cap);
    } else {
    hipLaunchKernelGGL(v2v_rows_kernel, dim3(row_blocks, B), dim3(kBoundsBlock), 0, s,
                       verts, V, Vp, (const int32_t*)m->tree_qperm, (const int32_t*)m->tree_rows,
                       (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, N, prow, bounds,
                       scan >= 2 ? leafbox : (float*)nullptr, (const int32_t*)m->tree_leaf_group,
                       scan >= 2 ? (float*)(ws + l.prow_g) : (float*)nullptr, m->tree_groups);
    if (scan < 2)      // (the leaf scans seed from the leaf boxes and never look at an inner node)
        hipLaunchKernelGGL(tree_inner_bounds_kernel<4>, dim3(B), dim3(kBoundsBlock), tree_inner_bounds_lds<4>(N), s, nodes, N,
                           (const int32_t*)m->tree_height_off, (const int32_t*)m->tree_height_nodes, m->tree_heights, bounds,
                           (const int32_t*)m->tree_v2v_info);
    hipLaunchKernelGGL(v2v_seed_kernel, dim3(B, 2 * m->tree_qblocks), dim3(64), 0, s, (const float*)prow, V, Vp,
                       (const uint64_t*)m->tree_mask_bits, nodes, (const int32_t*)m->tree_rows, (const float*)bounds,
                       (const uint64_t*)m->tree_masked, N, (const int32_t*)hint_inout, keys, scan >= 2 ? colbox : (float*)nullptr,
                       scan >= 2 ? (const float*)leafbox : (const float*)nullptr,
                       (const uint64_t*)m->tree_masked_leaf, m->tree_leaves);
    }
    const int f = choose_v2v_frontier(m, B, iterative);
    const int f0 = m->tree_frontier_off_host[f], nsub = m->tree_frontier_off_host[f + 1] - f0;
    // leave_room: an unused LDS allocation caps the walk at 25 of a CU's 32 wave slots.  The walk is one grid of 220 k
    // short one-wave workgroups; when the inside test runs beside it on another stream -- a chain of mostly small
    // kernels -- every slot is taken and those (and a 16 MB memset) queue behind the walk's workgroups: the chain only
    // got going when the walk was done (tools/graph_timeline.py: the memset took 236 us).  -2.5 % step time; alone the
    // walk is 10 % slower with the cap (0.28 -> 0.31 ms), hence a flag (TUCH_V2V_LDS: bytes, to compare).
    const int lds_pad = leave_room && m->opt.v2v_lds > 0 ? m->opt.v2v_lds : 0;
    if (scan == 2) {
        // pairs: the sparse leaves' (leaf, column) pairs one per lane (needs a symmetric mask and leaves of at most 64 rows:
        // a column's admissible rows of a leaf are one window of ITS row of the bit matrix)
        const bool pairs = m->opt.v2v_pairs > 0 && m->mask_symmetric && m->tree_leaf_rows_max <= 64 && m->tree_groups < (1 << 24);
        const int capped = leave_room && m->opt.v2v_lds < 0 ? -m->opt.v2v_lds : 0;       // wave slots by register count
        void (*kernel)(const float*, int, int, const uint64_t*, const float*, const float*, const uint64_t*, const uint64_t*, int, int,
                       const int32_t*, const int32_t*, const int32_t*, uint64_t*, const float*, const uint64_t*, int, int);
        if (pairs) kernel = capped == 4 ? v2v_scan_shared_kernel<4, true> : capped == 5 ? v2v_scan_shared_kernel<5, true> : capped == 6 ? v2v_scan_shared_kernel<6, true>
                            : capped ? v2v_scan_shared_kernel<7, true> : v2v_scan_kernel<true>;
        else kernel = capped == 4 ? v2v_scan_shared_kernel<4, false> : capped == 5 ? v2v_scan_shared_kernel<5, false> : capped == 6 ? v2v_scan_shared_kernel<6, false>
                      : capped ? v2v_scan_shared_kernel<7, false> : v2v_scan_kernel<false>;
        hipLaunchKernelGGL(kernel, dim3(B, 2 * nsub * m->tree_qblocks), dim3(64), capped ? (size_t)0 : (size_t)lds_pad, s, (const float*)prow,
                           V, Vp, (const uint64_t*)m->tree_mask_bits, (const float*)leafbox, (const float*)colbox,
                           (const uint64_t*)m->tree_masked_leaf, (const uint64_t*)m->tree_masked, N, m->tree_leaves,
                           (const int32_t*)m->tree_frontier_nodes + f0, (const int32_t*)m->tree_sub_leaf + 2 * (size_t)f0,
                           (const int32_t*)m->tree_launch_order + (size_t)f0 * m->tree_qblocks, keys,
                           (const float*)(ws + l.prow_g), (const uint64_t*)m->tree_mask_bits_g, m->tree_groups,
                           std::min(m->opt.v2v_pairs, 33));       // (the queue holds 63 + 32 pairs)
    }
    else
    hipLaunchKernelGGL(v2v_tree_kernel, dim3(B, 2 * nsub * m->tree_qblocks), dim3(64), (size_t)lds_pad, s, (const float*)prow, V, Vp,
                       (const uint64_t*)m->tree_mask_bits, nodes, (const int32_t*)m->tree_rows, (const float*)bounds,
                       (const uint64_t*)m->tree_masked, N, (const int32_t*)m->tree_frontier_nodes + f0,
                       (const int32_t*)m->tree_launch_order + (size_t)f0 * m->tree_qblocks, keys);
    hipLaunchKernelGGL(v2v_tree_finalize_kernel, dim3(ceil_div(V, kBlock), B), dim3(kBlock), 0, s,
                       (const uint64_t*)keys, (const int32_t*)m->tree_qperm, V, Vp, min_d2, argmin, (int32_t*)hint_inout);
    return tuch_check_launch("tuch_v2v_min_model");
}

extern "C" int tuch_batch_pairwise_dist(const float* x, const float* y, int B, int Nx, int Ny,
                                        int squared, float* P, void* stream)
{
    TUCH_REQUIRE(x && y && P, "tuch_batch_pairwise_dist: null pointer");
    TUCH_REQUIRE(B > 0 && Nx > 0 && Ny > 0 && Nx <= 65535 && B <= 65535,
                 "tuch_batch_pairwise_dist: bad sizes");
    hipLaunchKernelGGL(pairwise_kernel, dim3(ceil_div(Ny, kBlock), Nx, B), dim3(kBlock), 0,
                       (hipStream_t)stream, x, y, Nx, Ny, squared, P);
    return tuch_check_launch("tuch_batch_pairwise_dist");
}

extern "C" int tuch_batch_pairwise_dist_bwd(const float* x, const float* y, const float* grad_P, int B, int Nx, int Ny,
                                            int squared, float* grad_x, float* grad_y, void* stream)
{
    TUCH_REQUIRE(x && y && grad_P && (grad_x || grad_y), "tuch_batch_pairwise_dist_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && Nx > 0 && Ny > 0 && B <= 65535, "tuch_batch_pairwise_dist_bwd: bad sizes");
    if (grad_x)
        hipLaunchKernelGGL(pairwise_bwd_x_kernel, dim3(Nx, B), dim3(kBlock), 0, (hipStream_t)stream, x, y, grad_P, Nx, Ny,
                           squared, grad_x);
    if (grad_y)
        hipLaunchKernelGGL(pairwise_bwd_y_kernel, dim3(ceil_div(Ny, 64), B), dim3(kBlock), 0, (hipStream_t)stream, x, y,
                           grad_P, Nx, Ny, squared, grad_y);
    return tuch_check_launch("tuch_batch_pairwise_dist_bwd");
}

extern "C" size_t tuch_v2v_min_indexed_workspace_bytes(int B, int max_points_per_body)
{
    if (B <= 0 || max_points_per_body <= 0) return 0;
    return (size_t)B * ceil_div(max_points_per_body, kIndexedChunk) * 8 * sizeof(float);
}

// internal form with the options of the fused HD branch (hd_contact.hip): counts[b] rows per body at offsets[b],
// seeds (best, arg) per column, the row to report for columns whose rows are all masked
int tuch_v2v_min_indexed_seeded(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                                const int32_t* counts, const float* seed_best, const int32_t* seed_arg,
                                const int32_t* all_masked_arg, const uint64_t* geomask_bits, int B, int V,
                                int max_points_per_body, float* min_d2, int32_t* argmin, void* workspace, hipStream_t s)
{
    const int max_chunks = ceil_div(max_points_per_body, kIndexedChunk);
    float* boxes = (float*)workspace;
    hipLaunchKernelGGL(v2v_indexed_boxes_kernel, dim3(ceil_div(max_chunks, 4), B), dim3(256), 0, s, points, offsets, counts,
                       max_chunks, boxes);
    hipLaunchKernelGGL(v2v_indexed_kernel, dim3(ceil_div(max_points_per_body, 64), B), dim3(64 * kIndexedWaves), 0, s,
                       points, vertex_ids, offsets, counts, seed_best, seed_arg, all_masked_arg, geomask_bits, V,
                       (const float*)boxes, max_chunks, min_d2, argmin);
    return tuch_check_launch("tuch_v2v_min_indexed");
}

extern "C" int tuch_v2v_min_indexed(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                                    const uint64_t* geomask_bits, int B, int V, int max_points_per_body,
                                    float* min_d2, int32_t* argmin, void* workspace, size_t workspace_bytes,
                                    void* stream)
{
    TUCH_REQUIRE(points && vertex_ids && offsets && geomask_bits && min_d2 && argmin,
                 "tuch_v2v_min_indexed: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && V > 0 && max_points_per_body >= 0, "tuch_v2v_min_indexed: bad sizes");
    if (max_points_per_body == 0) return TUCH_OK;
    const size_t need = tuch_v2v_min_indexed_workspace_bytes(B, max_points_per_body);
    if (!workspace || workspace_bytes < need) {
        tuch_set_error("tuch_v2v_min_indexed: workspace %zu < %zu bytes", workspace_bytes, need);
        return TUCH_ERR_WORKSPACE;
    }
    return tuch_v2v_min_indexed_seeded(points, vertex_ids, offsets, nullptr, nullptr, nullptr, nullptr, geomask_bits, B, V,
                                       max_points_per_body, min_d2, argmin, workspace, (hipStream_t)stream);
}
