// Minimum squared distance between two contact regions (K5 of SURVEY.md §2.2).
//
// Replaces (a) the region-to-region term of tuch/smplify/losses.py:107-117, which slices
// the geodesically masked [1,V,V] matrix per annotated region pair, and (b)
// TUCH.contact_from_verts, tuch/train/train_module.py:69-91, which runs three bmm's
// per region pair in a Python loop over all P pairs ("Speed up this function will
// speed up training loop!", :74).
//
// One block per (pair, body).  The second region's vertices are staged in LDS once
// and read by all lanes at the same address (broadcast, conflict-free); every
// lane walks its share of the first region; the block minimum and its (i, j)
// are found with a wavefront shuffle reduction on (d2, flat index) keys so that
// ties resolve to the first flat index like torch.min / argmin.
// Distances are direct differences (DESIGN.md "Parity").
#include "common.h"
#include "model.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPairsPerBlock = 8;
constexpr int kTile = 1024;   // second-region vertices staged per pass (12 KB of LDS)

struct Best {
    float d;
    int idx;   // flat index a * n2 + bb in region-list order
};

__device__ __forceinline__ Best better(Best x, Best y)
{
    return (y.d < x.d || (y.d == x.d && y.idx < x.idx)) ? y : x;
}

__global__ __launch_bounds__(kBlock) void region_pair_min_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ region_off,
    const int32_t* __restrict__ region_vidx, const int32_t* __restrict__ pairs,
    const uint8_t* __restrict__ select,        // [B,P] or nullptr (= all)
    const uint32_t* __restrict__ pair_mask,    // per-pair geomask blocks or nullptr (= unmasked)
    const int64_t* __restrict__ pair_mask_off,
    int V, int P, unsigned long long* __restrict__ keys)   // [B,P], preset to all ones
{
    __shared__ float sx[kTile], sy[kTile], sz[kTile];
    __shared__ Best swave[kBlock / 64];
    // few pairs are selected in the SMPLify use: one workgroup walks kPairsPerBlock pairs so that
    // the launch does not consist of tens of thousands of empty workgroups
    const int b = blockIdx.y;
    const int per_block = select ? kPairsPerBlock : 1;
    for (int p = blockIdx.x * per_block; p < min(P, (int)(blockIdx.x + 1) * per_block); ++p) {
    const size_t o = (size_t)b * P + p;
    if (select && !select[o]) continue;
    __syncthreads();
    const int r1 = pairs[2 * p], r2 = pairs[2 * p + 1];
    const int a_beg = region_off[r1], n1 = region_off[r1 + 1] - a_beg;
    const int b_beg = region_off[r2], n2 = region_off[r2 + 1] - b_beg;
    const float* vb = verts + (size_t)b * V * 3;
    Best best = {__builtin_inff(), 0x7fffffff};
    for (int t0 = 0; t0 < n2; t0 += kTile) {
        const int tn = min(kTile, n2 - t0);
        __syncthreads();
        for (int k = threadIdx.x; k < tn; k += kBlock) {
            const int v = region_vidx[b_beg + t0 + k];
            sx[k] = vb[3 * v]; sy[k] = vb[3 * v + 1]; sz[k] = vb[3 * v + 2];
        }
        __syncthreads();
        // rows of the first region are split over blockIdx.z (more workgroups per selected pair)
        const int rows = (n1 + gridDim.z - 1) / gridDim.z;
        const int a_lo = blockIdx.z * rows, a_hi = min(n1, a_lo + rows);
        for (int a = a_lo + threadIdx.x; a < a_hi; a += kBlock) {
            const int i = region_vidx[a_beg + a];
            const float px = vb[3 * i], py = vb[3 * i + 1], pz = vb[3 * i + 2];
            const uint32_t* mrow = pair_mask ? pair_mask + pair_mask_off[p] + (size_t)a * ((n2 + 31) / 32) + (t0 >> 5)
                                             : nullptr;
            for (int g = 0; g < tn; g += 32) {
                const uint32_t word = mrow ? mrow[g >> 5] : 0xffffffffu;
                const int gn = min(32, tn - g);
                for (int kk = 0; kk < gn; ++kk) {
                    const int k = g + kk;
                    const float dx = px - sx[k], dy = py - sy[k], dz = pz - sz[k];
                    float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    if (!((word >> kk) & 1)) d = __builtin_inff();
                    const int flat = a * n2 + k + t0;
                    if (d < best.d || (d == best.d && flat < best.idx)) { best.d = d; best.idx = flat; }
                }
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        Best other = {__shfl_down(best.d, s, 64), __shfl_down(best.idx, s, 64)};
        best = better(best, other);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) swave[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        Best r = swave[0];
        for (int w = 1; w < kBlock / 64; ++w) r = better(r, swave[w]);
        // d >= 0, so the float's bit pattern orders like the value: (d, flat index) packs into one
        // 64-bit key whose minimum is independent of the arrival order -> deterministic
        if (r.idx != 0x7fffffff)
            atomicMin(&keys[o], ((unsigned long long)__float_as_uint(r.d) << 32) | (unsigned int)r.idx);
    }
    }
}

// keys -> (min d2, arg-min vertex ids); unselected / empty pairs give 0 and (-1, -1).  The keys
// live in the storage of out_ij and are overwritten in place.
__global__ __launch_bounds__(kBlock) void region_pair_finalize_kernel(
    const int32_t* __restrict__ region_off, const int32_t* __restrict__ region_vidx,
    const int32_t* __restrict__ pairs, int P, float* __restrict__ out_min, int32_t* __restrict__ out_ij)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const size_t o = (size_t)b * P + p;
    const unsigned long long key = ((const unsigned long long*)out_ij)[o];
    if (key == ~0ull) {
        out_min[o] = 0.0f;
        out_ij[2 * o] = -1;
        out_ij[2 * o + 1] = -1;
        return;
    }
    const int r1 = pairs[2 * p], r2 = pairs[2 * p + 1];
    const int n2 = region_off[r2 + 1] - region_off[r2];
    const int idx = (int)(unsigned int)key;
    out_min[o] = __uint_as_float((unsigned int)(key >> 32));
    out_ij[2 * o] = region_vidx[region_off[r1] + idx / n2];
    out_ij[2 * o + 1] = region_vidx[region_off[r2] + idx % n2];
}

// d(min d2)/dv: +2 g (v_i - v_j) to i, the negative to j
__global__ __launch_bounds__(kBlock) void region_pair_min_bwd_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ ij, const float* __restrict__ gout,
    int V, int P, float* __restrict__ grad)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const size_t o = (size_t)b * P + p;
    const int i = ij[2 * o], j = ij[2 * o + 1];
    const float g = gout[o];
    if (i < 0 || j < 0 || g == 0.0f) return;
    const float* vb = verts + (size_t)b * V * 3;
    float* gb = grad + (size_t)b * V * 3;
    for (int c = 0; c < 3; ++c) {
        const float d = 2.0f * g * (vb[3 * i + c] - vb[3 * j + c]);
        atomicAdd(gb + 3 * i + c, d);
        atomicAdd(gb + 3 * j + c, -d);
    }
}

}  // namespace

extern "C" int tuch_region_pair_min(const tuch_contact_model* m, const float* verts, int B,
                                    const uint8_t* select, int use_geomask, float* out_min,
                                    int32_t* out_ij, void* stream)
{
    TUCH_REQUIRE(m && verts && out_min && out_ij, "tuch_region_pair_min: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_region_pair_min: bad batch %d", B);
    TUCH_REQUIRE(m->num_pairs > 0, "tuch_region_pair_min: model has no region pairs");
    TUCH_REQUIRE(!use_geomask || m->pair_mask, "tuch_region_pair_min: model has no geodesic mask");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out_ij, 0xFF, (size_t)B * m->num_pairs * 2 * sizeof(int32_t), s) != hipSuccess) {
        tuch_set_error("tuch_region_pair_min: hipMemsetAsync failed");
        return TUCH_ERR_HIP;
    }
    // few pairs selected (SMPLify r2r): split their rows over 4 workgroups; all pairs: 1 is enough
    const int row_splits = select ? 4 : 1;
    hipLaunchKernelGGL(region_pair_min_kernel, dim3(ceil_div(m->num_pairs, select ? kPairsPerBlock : 1), B, row_splits), dim3(kBlock), 0, s,
                       verts, (const int32_t*)m->region_off, (const int32_t*)m->region_vidx,
                       (const int32_t*)m->pairs, select,
                       use_geomask ? (const uint32_t*)m->pair_mask : (const uint32_t*)nullptr,
                       (const int64_t*)m->pair_mask_off, m->V, m->num_pairs, (unsigned long long*)out_ij);
    hipLaunchKernelGGL(region_pair_finalize_kernel, dim3(ceil_div(m->num_pairs, kBlock), B), dim3(kBlock), 0, s,
                       (const int32_t*)m->region_off, (const int32_t*)m->region_vidx, (const int32_t*)m->pairs,
                       m->num_pairs, out_min, out_ij);
    return tuch_check_launch("tuch_region_pair_min");
}

extern "C" int tuch_region_pair_min_bwd(const tuch_contact_model* m, const float* verts, int B,
                                        const int32_t* ij, const float* grad_out, float* grad_verts,
                                        void* stream)
{
    TUCH_REQUIRE(m && verts && ij && grad_out && grad_verts, "tuch_region_pair_min_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && m->num_pairs > 0, "tuch_region_pair_min_bwd: bad arguments");
    hipLaunchKernelGGL(region_pair_min_bwd_kernel, dim3(ceil_div(m->num_pairs, kBlock), B), dim3(kBlock), 0,
                       (hipStream_t)stream, verts, ij, grad_out, m->V, m->num_pairs, grad_verts);
    return tuch_check_launch("tuch_region_pair_min_bwd");
}
