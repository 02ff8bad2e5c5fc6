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

// Ragged variant for resampled (HD) point sets, tuch/train/loss.py:288-291: body b owns points
// off[b]..off[b+1]; point a carries the template vertex vid[a] whose geodesic-mask row/column
// it inherits (geovec_verts, loss.py:88).  Column a: min over the body's rows r with
// geomask[vid[r]][vid[a]].  argmin is the row index relative to the body's first point.
__global__ __launch_bounds__(kBlock) void v2v_indexed_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ vid, const int32_t* __restrict__ off,
    const uint64_t* __restrict__ bits, int V, float* __restrict__ out_min, int32_t* __restrict__ out_arg)
{
    const int b = blockIdx.y;
    const int beg = off[b], n = off[b + 1] - beg;
    const int a = blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x * kBlock >= n) return;
    const int ac = min(a, n - 1);
    const float px = pts[3 * (size_t)(beg + ac)], py = pts[3 * (size_t)(beg + ac) + 1], pz = pts[3 * (size_t)(beg + ac) + 2];
    const int va = vid[beg + ac];
    const uint64_t* col = bits + (size_t)(va >> 6) * V;
    const int sh = va & 63;
    float best = __builtin_inff();
    int arg = 0;
    for (int r = 0; r < n; ++r) {
        const float* q = pts + 3 * (size_t)(beg + r);       // wave-uniform
        const int vr = vid[beg + r];
        const float dx = px - q[0], dy = py - q[1], dz = pz - q[2];
        float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        if (!((col[vr] >> sh) & 1)) d = __builtin_inff();
        if (d < best) { best = d; arg = r; }
    }
    if (a < n) { out_min[beg + a] = best; out_arg[beg + a] = arg; }
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

extern "C" int tuch_v2v_min_indexed(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                                    const uint64_t* geomask_bits, int B, int V, int max_points_per_body,
                                    float* min_d2, int32_t* argmin, void* stream)
{
    TUCH_REQUIRE(points && vertex_ids && offsets && geomask_bits && min_d2 && argmin,
                 "tuch_v2v_min_indexed: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && V > 0 && max_points_per_body >= 0, "tuch_v2v_min_indexed: bad sizes");
    if (max_points_per_body == 0) return TUCH_OK;
    hipLaunchKernelGGL(v2v_indexed_kernel, dim3(ceil_div(max_points_per_body, kBlock), B), dim3(kBlock), 0,
                       (hipStream_t)stream, points, vertex_ids, offsets, geomask_bits, V, min_d2, argmin);
    return tuch_check_launch("tuch_v2v_min_indexed");
}
