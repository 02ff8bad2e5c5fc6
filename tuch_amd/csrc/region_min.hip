// Minimum squared distance between two contact regions (K5 of SURVEY.md §2.2).
//
// Replaces (a) the region-to-region term of tuch/smplify/losses.py:107-117, which slices
// the geodesically masked [1,V,V] matrix per annotated region pair, and (b)
// TUCH.contact_from_verts, tuch/train/train_module.py:69-91, which runs three bmm's
// per region pair in a Python loop over all P pairs ("Speed up this function will
// speed up training loop!", :74).
//
// One workgroup per (body, pair).  The second region's vertices are staged in LDS and read by all lanes at the
// same address (broadcast, conflict-free); every lane owns one vertex of the first region (64-row blocks dealt to
// the wavefronts); the minimum and its (i, j) are found with a wavefront shuffle reduction on (d2, flat index)
// keys, so that ties resolve to the first flat index like torch.min / argmin.
// Distances are direct differences (DESIGN.md "Parity").
#include "common.h"
#include "model.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;   // second-region vertices staged per pass (12 KB of LDS)

struct Best {
    float d;
    int idx;   // flat index a * n2 + bb in region-list order
};

__device__ __forceinline__ Best better(Best x, Best y)
{
    return (y.d < x.d || (y.d == x.d && y.idx < x.idx)) ? y : x;
}

// FEW pairs selected (the SMPLify use: ~5 of ~180 per body): one wavefront per (body, 64 rows of the first region, share
// of the body's selected pairs) -- a pair of 500-vertex regions is ~50 k wavefront instructions: as one workgroup of four
// wavefronts it takes 30 us, as nine one-wave workgroups 13.  The wavefront finds its pairs in the body's row of `select`
// itself (a ballot per 64 pairs): with a grid over ALL pairs, 97 % of the 105 k workgroups of a batch of 64 left at once
// and the kernel's 44 us were their dispatch (~1 workgroup per clock), 113 us beside the crossing kernel.
constexpr int kPairShares = 16, kPairColSplit = 4;   // shares of a body's selected pairs; column ranges of a pair
template <bool kMasked, bool kInverted>
__global__ __launch_bounds__(64) void region_pair_rows_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ region_off,
    const int32_t* __restrict__ region_vidx, const int32_t* __restrict__ pairs,
    const uint8_t* __restrict__ select,        // [B,P]
    const uint32_t* __restrict__ pair_mask,    // per-pair geomask blocks (kMasked)
    const int64_t* __restrict__ pair_mask_off,
    int V, int P, unsigned long long* __restrict__ keys)   // [B,P], preset to all ones (kInverted: to zero, the key stored
                                                           // complemented and merged with atomicMax: 0 = no candidate)
{
    // (3 KB of LDS: a task's columns are a quarter of a region.  With the 12 KB tile of the all-pairs kernel a CU held 13 of
    // these one-wave workgroups, and the launch was rounds of workgroups waiting for their bytes of `select`: ~40 us whatever
    // was selected)
    constexpr int kRowsTile = 256;
    __shared__ __attribute__((aligned(16))) float sx[kRowsTile], sy[kRowsTile], sz[kRowsTile];
    const int b = blockIdx.x, share = blockIdx.y, lane = threadIdx.x;
    const float* vb = verts + (size_t)b * V * 3;
    const float inf = __builtin_inff();
    int seen = 0;                                                // selected pairs of the body so far (wave-uniform)
    for (int p00 = 0; p00 < P; p00 += 256) {
        uint8_t sel[4];                                          // four independent loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) sel[u] = p00 + 64 * u + lane < P ? select[(size_t)b * P + p00 + 64 * u + lane] : (uint8_t)0;
        unsigned long long todos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) todos[u] = __builtin_amdgcn_ballot_w64(sel[u] != 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const int p0 = p00 + 64 * u;
        unsigned long long todo = todos[u];
        while (todo) {
            const int p = p0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            if ((seen++ % kPairShares) != share) continue;
            const size_t o = (size_t)b * P + p;
            const int r1 = pairs[2 * p], r2 = pairs[2 * p + 1];
            const int a_beg = region_off[r1], n1 = region_off[r1 + 1] - a_beg;
            const int b_beg = region_off[r2], n2 = region_off[r2 + 1] - b_beg;
            // a task = 64 rows of the first region x a quarter of the second region's columns (whole 32-column mask words):
            // one wavefront walking 64 x 1000 pairs is ~20 us of dependent vector work, the kernel's time at any batch size
            const int zr = blockIdx.z / kPairColSplit, zc = blockIdx.z % kPairColSplit;
            const int a = zr * 64 + lane;                           // this lane's row of the first region
            if (zr * 64 >= n1) continue;
            const int words2 = (n2 + 31) / 32;
            const int c_beg = min(n2, (words2 * zc / kPairColSplit) * 32), c_end = min(n2, (words2 * (zc + 1) / kPairColSplit) * 32);
            if (c_beg >= c_end) continue;
            const int i = region_vidx[a_beg + min(a, n1 - 1)];
            const float px = vb[3 * i], py = vb[3 * i + 1], pz = vb[3 * i + 2];
            const int wpr = (n2 + 31) / 32;
            const uint32_t* mrow = kMasked ? pair_mask + pair_mask_off[p] + (size_t)min(a, n1 - 1) * wpr : nullptr;
            float best = inf;
            int best_k = -1;                                        // column of the second region
            // columns ascend, so within a lane the first minimum (strict '<') is the one with the smallest flat index
            auto column = [&](int k, float x, float y, float z, bool allowed) {
                const float dx = px - x, dy = py - y, dz = pz - z;
                float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (kMasked && !allowed) d = inf;
                if (d < best) { best = d; best_k = k; }
            };
            for (int t0 = c_beg; t0 < c_end; t0 += kRowsTile) {
                const int tn = min(kRowsTile, c_end - t0);
                __syncthreads();
                // staging is two dependent loads per vertex (index, coordinates): four vertices per lane in flight -- one
                // at a time a 1000-vertex region was sixteen round trips in a row, most of the kernel's 40 us
                for (int k = lane; k < tn; k += 256) {
                    int v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = region_vidx[b_beg + t0 + min(k + 64 * u, tn - 1)];
                    float c[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { c[u][0] = vb[3 * v[u]]; c[u][1] = vb[3 * v[u] + 1]; c[u][2] = vb[3 * v[u] + 2]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (k + 64 * u < tn) { sx[k + 64 * u] = c[u][0]; sy[k + 64 * u] = c[u][1]; sz[k + 64 * u] = c[u][2]; }
                }
                __syncthreads();
                if (a < n1) {
                    uint32_t next_word = kMasked ? mrow[t0 >> 5] : 0xffffffffu;
                    for (int g = 0; g < tn; g += 32) {
                        const uint32_t word = next_word;             // the next 32 columns' word is on its way meanwhile
                        if (kMasked && g + 32 < tn) next_word = mrow[(t0 + g + 32) >> 5];
                        const int gn = min(32, tn - g);
                        int kk = 0;
                        for (; kk + 4 <= gn; kk += 4) {               // four columns per LDS read (b128 broadcasts)
                            const float4 x4 = *(const float4*)&sx[g + kk];
                            const float4 y4 = *(const float4*)&sy[g + kk];
                            const float4 z4 = *(const float4*)&sz[g + kk];
                            column(t0 + g + kk + 0, x4.x, y4.x, z4.x, (word >> (kk + 0)) & 1);
                            column(t0 + g + kk + 1, x4.y, y4.y, z4.y, (word >> (kk + 1)) & 1);
                            column(t0 + g + kk + 2, x4.z, y4.z, z4.z, (word >> (kk + 2)) & 1);
                            column(t0 + g + kk + 3, x4.w, y4.w, z4.w, (word >> (kk + 3)) & 1);
                        }
                        for (; kk < gn; ++kk) column(t0 + g + kk, sx[g + kk], sy[g + kk], sz[g + kk], (word >> kk) & 1);
                    }
                }
            }
            Best r = {best, best_k >= 0 ? a * n2 + best_k : 0x7fffffff};
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                Best other = {__shfl_down(r.d, s, 64), __shfl_down(r.idx, s, 64)};
                r = better(r, other);
            }
            // d >= 0, so the float's bit pattern orders like the value: (d, flat index) packs into one
            // 64-bit key whose minimum is independent of the arrival order -> deterministic
            if (lane == 0 && r.idx != 0x7fffffff) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(r.d) << 32) | (unsigned int)r.idx;
                if (kInverted) atomicMax(&keys[o], ~key); else atomicMin(&keys[o], key);
            }
        }
        }
    }
}

// ALL pairs (contact_from_verts, train_module.py:69-91: select == NULL): one workgroup of four wavefronts per (body, pair).  The second region is staged in
// LDS ONCE per pair, the wavefronts take the 64-row blocks of the first region in turn, their minima meet in LDS: one
// owner per key, no atomic (251 -> 188 us for the 182 x 64 pairs of the bench's batch: the row-block form stages the second
// region once per 64 rows).
constexpr int kPairWaves = 4;
template <bool kMasked, bool kInverted>
__global__ __launch_bounds__(64 * kPairWaves) void region_pair_min_kernel(
    const float* __restrict__ verts, const int32_t* __restrict__ region_off,
    const int32_t* __restrict__ region_vidx, const int32_t* __restrict__ pairs,
    const uint8_t* __restrict__ select,        // [B,P] or nullptr (= all)
    const uint32_t* __restrict__ pair_mask,    // per-pair geomask blocks (kMasked)
    const int64_t* __restrict__ pair_mask_off,
    int V, int P, unsigned long long* __restrict__ keys)   // [B,P], preset to all ones (kInverted: to zero, the key stored
                                                           // complemented: 0 = no candidate)
{
    __shared__ __attribute__((aligned(16))) float sx[kTile], sy[kTile], sz[kTile];
    __shared__ float s_best[kPairWaves];
    __shared__ int s_idx[kPairWaves];
    const int b = blockIdx.x, p = blockIdx.y;
    const size_t o = (size_t)b * P + p;
    if (select && !select[o]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r1 = pairs[2 * p], r2 = pairs[2 * p + 1];
    const int a_beg = region_off[r1], n1 = region_off[r1 + 1] - a_beg;
    const int b_beg = region_off[r2], n2 = region_off[r2 + 1] - b_beg;
    const float* vb = verts + (size_t)b * V * 3;
    const int wpr = (n2 + 31) / 32;
    const float inf = __builtin_inff();
    Best r = {inf, 0x7fffffff};
    // the rows of the first region: block z = wave, wave + 4, ... ; rows and columns ascend, so within a lane the first
    // minimum (strict '<') is the one with the smallest flat index
    const int zblocks = (n1 + 63) / 64;
    for (int z0 = 0; z0 < zblocks; z0 += kPairWaves) {
        const int z = z0 + wave;
        const bool have = z < zblocks;
        const int a = z * 64 + lane;                               // this lane's row of the first region
        const int i = region_vidx[a_beg + min(have ? a : 0, n1 - 1)];
        const float px = vb[3 * i], py = vb[3 * i + 1], pz = vb[3 * i + 2];
        const uint32_t* mrow = kMasked ? pair_mask + pair_mask_off[p] + (size_t)min(have ? a : 0, n1 - 1) * wpr : nullptr;
        // the row's mask words, all requested before the first column is looked at (one per 32 columns inside the loop was a
        // dependent global load each time round: ~1 us x 9 per row block)
        constexpr int kWords = 16;
        uint32_t mw[kWords];
        if (kMasked && wpr <= kWords) {
#pragma unroll
            for (int u = 0; u < kWords; ++u) mw[u] = u < wpr ? mrow[u] : 0u;
        }
        float best = inf;
        int best_k = -1;                                            // column of the second region
        auto column = [&](int k, float x, float y, float z2, bool allowed) {
            const float dx = px - x, dy = py - y, dz = pz - z2;
            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (kMasked && !allowed) d = inf;
            if (d < best) { best = d; best_k = k; }
        };
        for (int t0 = 0; t0 < n2; t0 += kTile) {
            const int tn = min(kTile, n2 - t0);
            if (z0 == 0 || n2 > kTile) {                            // (a region that fits one tile is staged once)
                __syncthreads();
                for (int k = threadIdx.x; k < tn; k += 64 * kPairWaves) {
                    const int v = region_vidx[b_beg + t0 + k];
                    sx[k] = vb[3 * v]; sy[k] = vb[3 * v + 1]; sz[k] = vb[3 * v + 2];
                }
                __syncthreads();
            }
            if (have && a < n1)
                for (int g = 0; g < tn; g += 32) {
                    uint32_t word = 0xffffffffu;
                    if (kMasked) {
                        if (wpr <= kWords) {
                            const int wi = (t0 + g) >> 5;
#pragma unroll
                            for (int u = 0; u < kWords; ++u) word = u == wi ? mw[u] : word;      // (no dynamic register index)
                        } else {
                            word = mrow[(t0 + g) >> 5];
                        }
                    }
                    const int gn = min(32, tn - g);
                    int kk = 0;
                    for (; kk + 4 <= gn; kk += 4) {               // four columns per LDS read (b128 broadcasts)
                        const float4 x4 = *(const float4*)&sx[g + kk];
                        const float4 y4 = *(const float4*)&sy[g + kk];
                        const float4 z4 = *(const float4*)&sz[g + kk];
                        column(t0 + g + kk + 0, x4.x, y4.x, z4.x, (word >> (kk + 0)) & 1);
                        column(t0 + g + kk + 1, x4.y, y4.y, z4.y, (word >> (kk + 1)) & 1);
                        column(t0 + g + kk + 2, x4.z, y4.z, z4.z, (word >> (kk + 2)) & 1);
                        column(t0 + g + kk + 3, x4.w, y4.w, z4.w, (word >> (kk + 3)) & 1);
                    }
                    for (; kk < gn; ++kk) column(t0 + g + kk, sx[g + kk], sy[g + kk], sz[g + kk], (word >> kk) & 1);
                }
        }
        const Best mine = {best, best_k >= 0 ? a * n2 + best_k : 0x7fffffff};
        r = better(r, mine);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        Best other = {__shfl_down(r.d, s, 64), __shfl_down(r.idx, s, 64)};
        r = better(r, other);
    }
    if (lane == 0) { s_best[wave] = r.d; s_idx[wave] = r.idx; }
    __syncthreads();
    // d >= 0, so the float's bit pattern orders like the value: (d, flat index) packs into one 64-bit key; ties resolve
    // to the first flat index like torch.min / argmin
    if (threadIdx.x == 0) {
        Best t = {s_best[0], s_idx[0]};
#pragma unroll
        for (int w = 1; w < kPairWaves; ++w) t = better(t, Best{s_best[w], s_idx[w]});
        if (t.idx != 0x7fffffff) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(t.d) << 32) | (unsigned int)t.idx;
            keys[o] = kInverted ? ~key : key;
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
    if (select) {       // few pairs: a wavefront per 64 rows
        const dim3 rgrid(B, kPairShares, ceil_div(m->region_max, 64) * kPairColSplit);
        if (use_geomask)
            hipLaunchKernelGGL((region_pair_rows_kernel<true, false>), rgrid, dim3(64), 0, s, verts, (const int32_t*)m->region_off,
                               (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)m->pair_mask,
                               (const int64_t*)m->pair_mask_off, m->V, m->num_pairs, (unsigned long long*)out_ij);
        else
            hipLaunchKernelGGL((region_pair_rows_kernel<false, false>), rgrid, dim3(64), 0, s, verts, (const int32_t*)m->region_off,
                               (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)nullptr,
                               (const int64_t*)nullptr, m->V, m->num_pairs, (unsigned long long*)out_ij);
    } else {
    const dim3 grid(B, m->num_pairs);
    if (use_geomask)
        hipLaunchKernelGGL((region_pair_min_kernel<true, false>), grid, dim3(64 * kPairWaves), 0, s, verts, (const int32_t*)m->region_off,
                           (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)m->pair_mask,
                           (const int64_t*)m->pair_mask_off, m->V, m->num_pairs, (unsigned long long*)out_ij);
    else
        hipLaunchKernelGGL((region_pair_min_kernel<false, false>), grid, dim3(64 * kPairWaves), 0, s, verts, (const int32_t*)m->region_off,
                           (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)nullptr,
                           (const int64_t*)nullptr, m->V, m->num_pairs, (unsigned long long*)out_ij);
    }
    hipLaunchKernelGGL(region_pair_finalize_kernel, dim3(ceil_div(m->num_pairs, kBlock), B), dim3(kBlock), 0, s,
                       (const int32_t*)m->region_off, (const int32_t*)m->region_vidx, (const int32_t*)m->pairs,
                       m->num_pairs, out_min, out_ij);
    return tuch_check_launch("tuch_region_pair_min");
}

// The search alone: keys [B,P] (64-bit) must be ZERO on entry; on return a pair's key is the complement of
// (bits of min d2) << 32 | flat index (first-region row * n2 + second-region column), 0 for an unselected / empty pair.
// No clearing launch, no finalize launch: the consumer (tuch_smplify_stage2_fused) decodes the keys itself, and the
// caller clears them together with whatever else it has to clear.
extern "C" int tuch_region_pair_keys(const tuch_contact_model* m, const float* verts, int B, const uint8_t* select,
                                     int use_geomask, void* keys_zeroed, void* stream)
{
    TUCH_REQUIRE(m && verts && keys_zeroed, "tuch_region_pair_keys: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535, "tuch_region_pair_keys: bad batch %d", B);
    TUCH_REQUIRE(m->num_pairs > 0, "tuch_region_pair_keys: model has no region pairs");
    TUCH_REQUIRE(!use_geomask || m->pair_mask, "tuch_region_pair_keys: model has no geodesic mask");
    hipStream_t s = (hipStream_t)stream;
    if (select) {
        const dim3 rgrid(B, kPairShares, ceil_div(m->region_max, 64) * kPairColSplit);
        if (use_geomask)
            hipLaunchKernelGGL((region_pair_rows_kernel<true, true>), rgrid, dim3(64), 0, s, verts, (const int32_t*)m->region_off,
                               (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)m->pair_mask,
                               (const int64_t*)m->pair_mask_off, m->V, m->num_pairs, (unsigned long long*)keys_zeroed);
        else
            hipLaunchKernelGGL((region_pair_rows_kernel<false, true>), rgrid, dim3(64), 0, s, verts, (const int32_t*)m->region_off,
                               (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)nullptr,
                               (const int64_t*)nullptr, m->V, m->num_pairs, (unsigned long long*)keys_zeroed);
        return tuch_check_launch("tuch_region_pair_keys");
    }
    const dim3 grid(B, m->num_pairs);
    if (use_geomask)
        hipLaunchKernelGGL((region_pair_min_kernel<true, true>), grid, dim3(64 * kPairWaves), 0, s, verts, (const int32_t*)m->region_off,
                           (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)m->pair_mask,
                           (const int64_t*)m->pair_mask_off, m->V, m->num_pairs, (unsigned long long*)keys_zeroed);
    else
        hipLaunchKernelGGL((region_pair_min_kernel<false, true>), grid, dim3(64 * kPairWaves), 0, s, verts, (const int32_t*)m->region_off,
                           (const int32_t*)m->region_vidx, (const int32_t*)m->pairs, select, (const uint32_t*)nullptr,
                           (const int64_t*)nullptr, m->V, m->num_pairs, (unsigned long long*)keys_zeroed);
    return tuch_check_launch("tuch_region_pair_keys");
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
