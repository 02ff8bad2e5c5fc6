// Nearest admissible point within ragged point sets on the matrix cores: the search of the HD branch
// (tuch/train/loss.py:288-291: an n x n masked distance matrix per body, torch.min over its rows).
//
// v2v_indexed_kernel (v2v.hip) evaluates a row against 64 columns with ~10 vector instructions; most of them are not the
// distance but the mask and the running (minimum, row).  Here a wavefront takes 32 rows x 64 columns at a time:
//   * distances from v_mfma_f32_32x32x2_f32 (exact f32, an fmaf chain): |q'|^2 + R^2 - 2 p'.q' with coordinates relative
//     to the centre of the column block (p' small), K = 4: (x', y', z', |q'|^2 + R^2) . (-2px', -2py', -2pz', 1);
//   * THE MASK FROM THE MATRIX CORE AS WELL: the rows of a tile that inherit the same template vertex form a run (points
//     are kept sorted by patch: ~7 runs per tile); a column gathers ONE mask word per run, turns it into a bf16 penalty
//     (0 or 2^127) and one v_mfma_f32_32x32x16_bf16 adds  onehot(run of row i) . penalty(run, column j)  to the tile: an
//     inadmissible pair comes out of the accumulator as 1.7e38, an admissible one unchanged (+0);
//   * what is left per accumulator value is one v_and_or (the row's place in the low four mantissa bits) and half a
//     v_min3_i32: keys of positive floats order like integers.
// The winner of a column is the row with the smallest KEY -- the distance in the centred expansion, rounded to 20 mantissa
// bits: it differs from the direct-difference float32 minimum (what v2v_indexed_kernel returns, bit for bit) only between
// rows whose squared distances tie within ~1e-6 relative + a few ulp of R^2 + |q'|^2 absolute (the reference's own
// |x|^2+|y|^2-2x.y matrix is 20 times noisier); the distance reported for it is recomputed by direct
// differences.  Deterministic: every reduction has a fixed order.
// Pass 1 evaluates one representative row per tile (bounds for every column), pass 2 the tiles whose box some column
// with an admissible row in it can still reach.
#include "common.h"
#include "model.h"
#include "workspace.h"

namespace {

constexpr int kTile = 32;                 // rows per tile
constexpr float kBigNorm = 1e30f;         // "norm" of a row that does not exist
constexpr float kNoKey = 1e29f;           // keys at or above: no admissible row
constexpr uint32_t kPenalty = 0x7F00u;    // bf16 2^127
constexpr float kSlack = 0.999999f;       // lower bounds are deflated by 1e-6

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct TileMeta {                         // 192 bytes per tile of 32 consecutive points of a body
    float lo[3]; int32_t runs;            // bounding box; number of runs (rows with the same mask vertex, consecutive)
    float hi[3]; int32_t rows;            // rows that exist (32 except in the body's last tile)
    int32_t tv[32];                       // mask vertex of run k (0 behind the last run)
    uint8_t ridx[32];                     // run of row i (255: the row does not exist)
};
static_assert(sizeof(TileMeta) == 192, "TileMeta layout");

// one 32-lane group per tile
__global__ __launch_bounds__(256) void hd_tiles_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ vid, const int32_t* __restrict__ off,
    const int32_t* __restrict__ counts, int max_tiles, int rep_stride, TileMeta* __restrict__ meta,
    int32_t* __restrict__ rep_tv, float4* __restrict__ rep_xyz)
{
    const int b = blockIdx.y;
    const int tile = blockIdx.x * 8 + ((int)threadIdx.x >> 5), i = threadIdx.x & 31;
    const int beg = off[b], n = counts ? counts[b] : off[b + 1] - beg;
    const int tiles = (n + kTile - 1) / kTile;
    if (tile >= tiles) {
        // the representatives are read in groups of 32: pad the last group
        if (tile < ((tiles + 31) & ~31) && tile < rep_stride && i == 0) {
            rep_tv[(size_t)b * rep_stride + tile] = 0;
            rep_xyz[(size_t)b * rep_stride + tile] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int r = tile * kTile + i;
    const bool valid = r < n;
    const size_t at = (size_t)beg + (valid ? r : n - 1);
    const float x = pts[3 * at], y = pts[3 * at + 1], z = pts[3 * at + 2];
    const int v = vid[at];
    float lo[3] = {x, y, z}, hi[3] = {x, y, z};
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], m));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m));
        }
    const int prev = __shfl_up(v, 1, 32);
    const bool start = valid && (i == 0 || v != prev);
    const uint32_t sb = (uint32_t)(__builtin_amdgcn_ballot_w64(start) >> (threadIdx.x & 32));
    const int run = __builtin_popcount(sb & ((2u << i) - 1u)) - 1;
    const int runs = __builtin_popcount(sb);
    TileMeta* m = meta + (size_t)b * max_tiles + tile;
    if (i >= runs) m->tv[i] = 0;
    if (start) m->tv[run] = v;
    m->ridx[i] = valid ? (uint8_t)run : (uint8_t)255;
    if (i == 0) {
        m->lo[0] = lo[0]; m->lo[1] = lo[1]; m->lo[2] = lo[2]; m->runs = runs;
        m->hi[0] = hi[0]; m->hi[1] = hi[1]; m->hi[2] = hi[2]; m->rows = min(kTile, n - tile * kTile);
        rep_tv[(size_t)b * rep_stride + tile] = v;
        rep_xyz[(size_t)b * rep_stride + tile] = make_float4(x, y, z, 0.f);
    }
}

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }

// lanes 32-63 of x trade places with lanes 0-31 of y
__device__ __forceinline__ void swap_halves(uint32_t& x, uint32_t& y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}

// Lane layouts.  "Own column": lane l <-> column 64 cb + l (box tests, mask words).  Matrix layout of sub-tile s (columns
// 32 s ... 32 s + 31): lane l <-> column 32 s + (l & 31); accumulator register a of lane l is row (a & 3) + 8 (a >> 2) +
// 4 (l >> 5) of the tile; the A operand of lane l belongs to row l & 31, the k index of both operands is chosen by l >> 5.
template <int kWaves>                     // wavefronts per block of 64 columns (they share out its row tiles)
__global__ __launch_bounds__(64 * kWaves) void hd_search_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ vid, const int32_t* __restrict__ off,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ all_masked_arg, const uint64_t* __restrict__ bits, int V,
    const TileMeta* __restrict__ meta, const int32_t* __restrict__ rep_tv, const float4* __restrict__ rep_xyz, int max_tiles,
    int rep_stride, float* __restrict__ out_min, int32_t* __restrict__ out_arg)
{
    __shared__ u32x4 onehot[9];
    __shared__ float s_f[kWaves][2][64];
    __shared__ int s_i[kWaves][2][64];
    __shared__ float s_bound[64];
    const int b = blockIdx.y, cb = blockIdx.x;
    const int beg = __builtin_amdgcn_readfirstlane(off[b]);
    const int n = counts ? __builtin_amdgcn_readfirstlane(counts[b]) : __builtin_amdgcn_readfirstlane(off[b + 1]) - beg;
    if (cb * 64 >= n) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    if (threadIdx.x < 9) {
        // entry e < 8: bf16 1.0 in element e; entry 8: nothing
        u32x4 t = {0u, 0u, 0u, 0u};
        const uint32_t one = 0x3F80u << (16 * (threadIdx.x & 1));
        if (threadIdx.x < 8) t[threadIdx.x >> 1] = one;
        onehot[threadIdx.x] = t;
    }
    const int tiles = (n + kTile - 1) / kTile;
    const TileMeta* mb = meta + (size_t)b * max_tiles;
    // centre and radius of the column block from the boxes of its one or two tiles
    float lo[3], hi[3];
    {
        const TileMeta* m0 = mb + 2 * cb;
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = m0->lo[k]; hi[k] = m0->hi[k]; }
        if (2 * cb + 1 < tiles) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], m0[1].lo[k]); hi[k] = fmaxf(hi[k], m0[1].hi[k]); }
        }
    }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    const float rx = hi[0] - cx, ry = hi[1] - cy, rz = hi[2] - cz;
    const float R2 = (rx * rx + ry * ry + rz * rz) * 1.0001f + 1e-12f;
    const float* bp = pts + 3 * (size_t)beg;
    const int32_t* bv = vid + beg;
    auto offset_of = [&](float x, float y, float z) {           // R^2 - |p'|^2 >= 0: key = d^2 + this
        const float ux = x - cx, uy = y - cy, uz = z - cz;
        return R2 - __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    };
    // own column
    const int ac = min(cb * 64 + lane, n - 1);
    const float px = bp[3 * ac], py = bp[3 * ac + 1], pz = bp[3 * ac + 2];
    const int va = bv[ac];
    const uint32_t lane_off = (uint32_t)(((va >> 6) * V * 2 + ((va >> 5) & 1)) * 4);
    const float own_o = offset_of(px, py, pz);
    // columns of the two sub-tiles
    float B1[2], B2[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int c = min(cb * 64 + 32 * s + j, n - 1);
        const float ux = bp[3 * c] - cx, uy = bp[3 * c + 1] - cy, uz = bp[3 * c + 2] - cz;
        B1[s] = h ? -2.0f * uy : -2.0f * ux;
        B2[s] = h ? 1.0f : -2.0f * uz;
    }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // penalties of runs [0, K) for the own column (R[k >> 1], 16 bits per run), and whether any run is admissible.  The
    // mask word of (column's 32-vertex block, run's vertex) through a buffer load: the run's part of the address is the
    // scalar offset, the column's part the lane offset -- no address arithmetic on the vector unit
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(bits), 0, 0x7ffffffc, 0x00020000);
    const uint32_t shift = va & 31;
    constexpr uint32_t kPenaltyPair = kPenalty | (kPenalty << 16);
    auto build = [&](const int32_t* tvp, int K, uint32_t (&R)[16], uint32_t& any) {
        uint32_t A[16];                                                      // admissible runs: 0xFFFF per run
#pragma unroll
        for (int m = 0; m < 16; ++m) A[m] = 0u;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (4 * g >= K) break;                                           // wave-uniform
            const int4 tv = *reinterpret_cast<const int4*>(tvp + 4 * g);     // scalar load
            int w[4];
            w[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.x * 8, 0);
            w[1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.y * 8, 0);
            w[2] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.z * 8, 0);
            w[3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.w * 8, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * g + u;
                const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe(w[u], shift, 1);        // admissible: all ones
                A[k >> 1] = (t & (0xFFFFu << (16 * (k & 1)))) | A[k >> 1];
            }
            any |= A[2 * g] | A[2 * g + 1];
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) R[m] = ~A[m] & kPenaltyPair;
    };
    // the tile: distances + penalties for both sub-tiles.  kMany: more than 16 runs (two penalty products)
    auto product = [&](float qx, float qy, float qz, bool exists, int ridx, uint32_t (&R)[16], bool many, f32x16 (&acc)[2]) {
        const float ux = qx - cx, uy = qy - cy, uz = qz - cz;
        const float nrm = exists ? __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux)) + R2 : kBigNorm;
        const float A1 = h ? uy : ux, A2 = h ? nrm : uz;
        const unsigned e0 = (unsigned)(ridx - 8 * h);
        const u32x4 oh0 = onehot[e0 < 8u ? e0 : 8u];
        // own-column registers -> matrix layout: sub-tile 0 = {own R[0..3] | R[4..7] of lane - 32}, sub-tile 1 = {R[0..3] of
        // lane + 32 | own R[4..7]}
#pragma unroll
        for (int m = 0; m < 4; ++m) swap_halves(R[m], R[4 + m]);
        const u32x4 P0 = {R[0], R[1], R[2], R[3]}, P1 = {R[4], R[5], R[6], R[7]};
        acc[0] = mfma_f32(A1, B1[0], zero);
        acc[1] = mfma_f32(A1, B1[1], zero);
        acc[0] = mfma_f32(A2, B2[0], acc[0]);
        acc[1] = mfma_f32(A2, B2[1], acc[1]);
        acc[0] = mfma_bf16(oh0, P0, acc[0]);
        acc[1] = mfma_bf16(oh0, P1, acc[1]);
        if (many) {
            const unsigned e1 = (unsigned)(ridx - 16 - 8 * h);
            const u32x4 oh1 = onehot[e1 < 8u ? e1 : 8u];
#pragma unroll
            for (int m = 0; m < 4; ++m) swap_halves(R[8 + m], R[12 + m]);
            const u32x4 Q0 = {R[8], R[9], R[10], R[11]}, Q1 = {R[12], R[13], R[14], R[15]};
            acc[0] = mfma_bf16(oh1, Q0, acc[0]);
            acc[1] = mfma_bf16(oh1, Q1, acc[1]);
        }
    };
    __syncthreads();                                                          // onehot

    // ---- pass 1: one representative per tile -> an upper bound for every column ------------------------------------
    float bestf[2] = {kNoKey, kNoKey};
    const int rep_tiles = (tiles + 31) >> 5;
    for (int u = wave; u < rep_tiles; u += kWaves) {
        const size_t rb = (size_t)b * rep_stride + 32 * u;
        uint32_t R[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t any = 0u;
        const int K = min(32, tiles - 32 * u);
        build(rep_tv + rb, K, R, any);
        const float4 q = rep_xyz[rb + j];
        f32x16 acc[2];
        product(q.x, q.y, q.z, j < K, j, R, true, acc);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float m = bestf[s];
#pragma unroll
            for (int a = 0; a < 16; a += 2) m = fminf(fminf(m, acc[s][a]), acc[s][a + 1]);
            bestf[s] = m;
        }
    }
    s_f[wave][0][lane] = bestf[0];
    s_f[wave][1][lane] = bestf[1];
    __syncthreads();
    float bnd;                                                                // own column, squared distance
    {
        float kb = kNoKey;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
            kb = fminf(kb, fminf(s_f[w][lane >> 5][lane & 31], s_f[w][lane >> 5][(lane & 31) + 32]));
        // the winner's key is at most the key of the representative + the rounding of both
        const bool found = kb < kNoKey;
        kb = found ? __builtin_fmaf(kb, 4e-6f, kb) + 4e-6f * R2 + 1e-12f : kNoKey;
        bnd = found ? kb - own_o : __builtin_inff();
        s_bound[lane] = kb;
    }
    __syncthreads();
    int bestkey[2], btile[2] = {-1, -1};
#pragma unroll
    for (int s = 0; s < 2; ++s) bestkey[s] = __float_as_int(s_bound[32 * s + j]) | 15;

    // ---- pass 2 ------------------------------------------------------------------------------------------------------
    // 64 tiles at a time, lane <-> tile: the tile's box against the box of the column block and the largest bound in
    // it (a lower bound of what any single column could find); only the survivors get the test per column
    for (int t0 = wave; t0 < tiles; t0 += 64 * kWaves) {
    unsigned long long todo;
    {
        float bmax = bnd;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, m));
        const int tl = t0 + kWaves * lane;
        bool cand = false;
        if (tl < tiles) {
            const float4 l4 = *reinterpret_cast<const float4*>(mb[tl].lo), h4 = *reinterpret_cast<const float4*>(mb[tl].hi);
            const float gx = fmaxf(fmaxf(l4.x - hi[0], lo[0] - h4.x), 0.0f);
            const float gy = fmaxf(fmaxf(l4.y - hi[1], lo[1] - h4.y), 0.0f);
            const float gz = fmaxf(fmaxf(l4.z - hi[2], lo[2] - h4.z), 0.0f);
            cand = __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx)) * kSlack <= bmax;
        }
        todo = __builtin_amdgcn_ballot_w64(cand);
    }
    while (todo) {
        const int t = t0 + kWaves * (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const TileMeta* mt = mb + t;
        const float ex = fmaxf(fmaxf(mt->lo[0] - px, px - mt->hi[0]), 0.0f);
        const float ey = fmaxf(fmaxf(mt->lo[1] - py, py - mt->hi[1]), 0.0f);
        const float ez = fmaxf(fmaxf(mt->lo[2] - pz, pz - mt->hi[2]), 0.0f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) * kSlack;
        const bool near = lb <= bnd;
        if (__builtin_amdgcn_ballot_w64(near) == 0) continue;
        const int K = mt->runs, rows = mt->rows;
        uint32_t R[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t any = 0u;
        build(mt->tv, K, R, any);
        // a column that is near but has no admissible row here does not make the tile worth a product
        if (__builtin_amdgcn_ballot_w64(near && any != 0u) == 0) continue;
        const int slot = t * kTile + j;
        const int rc = min(slot, n - 1);
        const float qx = bp[3 * rc], qy = bp[3 * rc + 1], qz = bp[3 * rc + 2];
        const int ridx = mt->ridx[j];
        f32x16 acc[2];
        product(qx, qy, qz, j < rows, ridx, R, K > 16, acc);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int key[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) key[a] = (__float_as_int(acc[s][a]) & ~15) | a;
            int m = imin3(key[0], key[1], key[2]);
#pragma unroll
            for (int a = 3; a < 15; a += 2) m = imin3(m, key[a], key[a + 1]);
            m = min(m, key[15]);
            const bool better = m < bestkey[s];
            bestkey[s] = better ? m : bestkey[s];
            btile[s] = better ? t : btile[s];
        }
        // the own columns' bounds: smaller of the two halves' keys
        {
            uint32_t k0 = (uint32_t)bestkey[0], k1 = (uint32_t)bestkey[1];
            swap_halves(k0, k1);
            const float f = __int_as_float(min((int)k0, (int)k1) & ~15);
            bnd = fminf(bnd, (f - own_o) + 4e-6f * (f + R2));
        }
    }
    }
    // ---- the candidates' distances by direct differences, lexicographic (distance, row) minimum per column -----------
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float d = __builtin_inff();
        int slot = 0;
        if (btile[s] >= 0) {
            const int a = bestkey[s] & 15;
            slot = btile[s] * kTile + 4 * h + 8 * (a >> 2) + (a & 3);
            const int c = min(cb * 64 + 32 * s + j, n - 1);
            const float dx = bp[3 * c] - bp[3 * slot], dy = bp[3 * c + 1] - bp[3 * slot + 1], dz = bp[3 * c + 2] - bp[3 * slot + 2];
            d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        }
        s_f[wave][s][lane] = d;
        s_i[wave][s][lane] = slot;
    }
    __syncthreads();
    const int a = cb * 64 + lane;
    if (wave == 0 && a < n) {
        float best = __builtin_inff();
        int arg = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float d = s_f[w][lane >> 5][(lane & 31) + 32 * hh];
                const int r = s_i[w][lane >> 5][(lane & 31) + 32 * hh];
                if (d < best || (d == best && d < __builtin_inff() && r < arg)) { best = d; arg = r; }
            }
        out_min[beg + a] = best;
        out_arg[beg + a] = best < __builtin_inff() ? arg : (all_masked_arg ? all_masked_arg[b] : 0);
    }
}

struct SearchLayout { size_t meta, rep_tv, rep_xyz, total; int tiles, rep_stride; };
SearchLayout search_layout(int B, int N)
{
    SearchLayout l;
    l.tiles = ceil_div(N, kTile);
    l.rep_stride = (l.tiles + 31) & ~31;
    size_t o = 0;
    l.meta = tuch_ws_take(o, (size_t)B * l.tiles * sizeof(TileMeta));
    l.rep_tv = tuch_ws_take(o, (size_t)B * l.rep_stride * sizeof(int32_t));
    l.rep_xyz = tuch_ws_take(o, (size_t)B * l.rep_stride * sizeof(float4));
    l.total = o;
    return l;
}

}  // namespace

size_t tuch_hd_search_workspace_bytes(int B, int max_points_per_body)
{
    if (B <= 0 || max_points_per_body <= 0) return 0;
    return search_layout(B, max_points_per_body).total;
}

// counts[b] points of body b at offsets[b] (counts == nullptr: offsets[b + 1] - offsets[b]); vertex_ids = the row /
// column of the geodesic mask each point inherits; all_masked_arg as in tuch_v2v_min_indexed_seeded
int tuch_hd_search(const float* points, const int32_t* vertex_ids, const int32_t* offsets, const int32_t* counts,
                   const int32_t* all_masked_arg, const uint64_t* geomask_bits, int B, int V, int max_points_per_body,
                   float* min_d2, int32_t* argmin, void* workspace, hipStream_t s, int waves)
{
    const SearchLayout l = search_layout(B, max_points_per_body);
    char* ws = (char*)workspace;
    TileMeta* meta = (TileMeta*)(ws + l.meta);
    int32_t* rep_tv = (int32_t*)(ws + l.rep_tv);
    float4* rep_xyz = (float4*)(ws + l.rep_xyz);
    hipLaunchKernelGGL(hd_tiles_kernel, dim3(l.rep_stride / 8, B), dim3(256), 0, s, points, vertex_ids, offsets, counts, l.tiles,
                       l.rep_stride, meta, rep_tv, rep_xyz);
#define TUCH_LAUNCH_SEARCH(W)                                                                                                \
    hipLaunchKernelGGL(hd_search_kernel<W>, dim3(ceil_div(max_points_per_body, 64), B), dim3(64 * W), 0, s, points, vertex_ids, \
                       offsets, counts, all_masked_arg, geomask_bits, V, (const TileMeta*)meta, (const int32_t*)rep_tv,        \
                       (const float4*)rep_xyz, l.tiles, l.rep_stride, min_d2, argmin)
    if (waves >= 4) TUCH_LAUNCH_SEARCH(4);
    else if (waves >= 2) TUCH_LAUNCH_SEARCH(2);
    else TUCH_LAUNCH_SEARCH(1);
#undef TUCH_LAUNCH_SEARCH
    return tuch_check_launch("tuch_hd_search");
}

extern "C" size_t tuch_v2v_min_indexed_mfma_workspace_bytes(int B, int max_points_per_body)
{
    return tuch_hd_search_workspace_bytes(B, max_points_per_body);
}

// the matrix-core form of tuch_v2v_min_indexed (same arguments): the winner of a column may differ from the float32
// direct-difference minimum between rows that tie within ~1e-6 relative (see the head of this file)
extern "C" int tuch_v2v_min_indexed_mfma(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                                         const uint64_t* geomask_bits, int B, int V, int max_points_per_body, float* min_d2,
                                         int32_t* argmin, void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(points && vertex_ids && offsets && geomask_bits && min_d2 && argmin, "tuch_v2v_min_indexed_mfma: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && V > 0 && max_points_per_body >= 0, "tuch_v2v_min_indexed_mfma: bad sizes");
    TUCH_REQUIRE((long)((V + 63) / 64) * V * 8 < 0x7fffffffL, "tuch_v2v_min_indexed_mfma: mask too large for 32-bit lane offsets");
    if (max_points_per_body == 0) return TUCH_OK;
    const size_t need = tuch_hd_search_workspace_bytes(B, max_points_per_body);
    if (!workspace || workspace_bytes < need) {
        tuch_set_error("tuch_v2v_min_indexed_mfma: workspace %zu < %zu bytes", workspace_bytes, need);
        return TUCH_ERR_WORKSPACE;
    }
    return tuch_hd_search(points, vertex_ids, offsets, nullptr, nullptr, geomask_bits, B, V, max_points_per_body, min_d2,
                          argmin, workspace, (hipStream_t)stream, 4);
}
