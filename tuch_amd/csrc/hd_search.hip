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
// Pass 1 evaluates one representative row per tile (all 32 mask words of a group of representatives requested at once):
// an upper bound for every column.  Pass 2, per wavefront and 64 of its tiles at a time with lane <-> tile: the tile's box
// against the box of the column block and the largest bound in it, the survivors against every column's own bound (the
// columns in reach kept as a 64-bit mask); what survives both gets a record in LDS (box, runs, the first eight mask
// vertices, the reach mask).  The loop over the records requests the NEXT tile's mask words and rows before it works on
// the current one (unconditionally: with the same number of loads in flight on every path the compiler waits for exactly
// the older set); a tile none of whose reachable columns has an admissible run ends before the products.
// Measured at batch 64 (5 600 column blocks of 192 tiles): 105 tiles per block pass the block filter, 65 the per-column
// test, 40 reach the products; 215 us against 502 us for v2v_indexed_kernel, 7.0e7 instead of 2.0e8 vector instructions,
// vector unit 0.58 busy.  (The mask words of the run slots as buffer loads with the column's part of the address in the
// lane offset: no address arithmetic; -mllvm -amdgpu-mfma-vgpr-form keeps the accumulators out of the AGPRs, whose 32
// v_accvgpr_read per tile were 12 % of the kernel.)
#include "common.h"
#include "model.h"
#include "workspace.h"

namespace {

#ifdef TUCH_SCAN_COUNTS
__device__ unsigned long long g_hd_counts[16];
extern "C" int tuch_debug_hd_counts(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hd_counts), sizeof(unsigned long long) * 16) != hipSuccess) return TUCH_ERR_HIP;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_hd_counts), z, sizeof(z)) != hipSuccess) return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}
#endif
constexpr int kTile = 32;                 // rows per tile
constexpr float kBigNorm = 1e30f;         // "norm" of a row that does not exist
constexpr float kNoKey = 1e29f;           // keys at or above: no admissible row
constexpr uint32_t kPenalty = 0x7F00u;    // bf16 2^127
constexpr float kSlack = 0.999999f;       // lower bounds are deflated by 1e-6
constexpr uint32_t kPenaltyPair = kPenalty | (kPenalty << 16);
constexpr int kRepStep = 1;               // pass 1: one representative row per this many tiles (2: same time, 4: slower)
// runs of a tile whose mask words are requested one tile ahead (patch-sorted HD points: 7.6 runs per tile on average,
// never more than 12; with 12 requested ahead the extra gathers cost more than the late groups of the few longer tiles)
constexpr int kAhead = 8;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct TileMeta {                         // 192 bytes per tile of 32 consecutive points of a body
    float lo[3]; int32_t runs;            // bounding box; number of runs (rows with the same mask vertex, consecutive)
    float hi[3]; int32_t rows;            // rows that exist (32 except in the body's last tile)
    int32_t tv[32];                       // mask vertex of run k (behind the last run: the last run's again)
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
    if (tile >= tiles) return;
    const int reps = (tiles + kRepStep - 1) / kRepStep;
    // the representatives are read in groups of 32: pad the last group
    if (tile == 0 && i > 0 && reps + i - 1 < ((reps + 31) & ~31)) {
        rep_tv[(size_t)b * rep_stride + reps + i - 1] = 0;
        rep_xyz[(size_t)b * rep_stride + reps + i - 1] = make_float4(0.f, 0.f, 0.f, 0.f);
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
    // unused run slots repeat the last run's vertex: the search may fetch their mask words, they change nothing
    const int last_v = __shfl(v, min(kTile, n - tile * kTile) - 1, 32);
    if (i >= runs) m->tv[i] = last_v;
    if (start) m->tv[run] = v;
    m->ridx[i] = valid ? (uint8_t)run : (uint8_t)255;
    if (i == 0) {
        m->lo[0] = lo[0]; m->lo[1] = lo[1]; m->lo[2] = lo[2]; m->runs = runs;
        m->hi[0] = hi[0]; m->hi[1] = hi[1]; m->hi[2] = hi[2]; m->rows = min(kTile, n - tile * kTile);
        if (tile % kRepStep == 0) {
            rep_tv[(size_t)b * rep_stride + tile / kRepStep] = v;
            rep_xyz[(size_t)b * rep_stride + tile / kRepStep] = make_float4(x, y, z, 0.f);
        }
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

// mask words of N runs (mask vertices tvp[0..N), a scalar pointer) for the lane's column: all requested before any is
// used.  Buffer loads: the run's part of the address is the scalar offset, the column's part the lane offset -- no
// address arithmetic on the vector unit.
template <int N>
__device__ __forceinline__ void request_words(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off, const int32_t* tvp, int (&w)[N])
{
    static_assert(N % 4 == 0, "whole groups");
#pragma unroll
    for (int g = 0; g < N / 4; ++g) {
        const int4 tv = *reinterpret_cast<const int4*>(tvp + 4 * g);         // scalar load
        w[4 * g + 0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.x * 8, 0);
        w[4 * g + 1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.y * 8, 0);
        w[4 * g + 2] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.z * 8, 0);
        w[4 * g + 3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, tv.w * 8, 0);
    }
}
// A[(K0 + u) >> 1], 16 bits per run K0 + u (runs counted from the array's first): all ones where the column may use it
template <int N, int K0>
__device__ __forceinline__ void digest_words(const int (&w)[N], uint32_t shift, uint32_t (&A)[8])
{
    static_assert(K0 + N <= 16, "eight registers hold sixteen runs");
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int k = K0 + u;
        const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe(w[u], shift, 1);
        A[k >> 1] = (t & (0xFFFFu << (16 * (k & 1)))) | A[k >> 1];
    }
}
// groups [G0, 8) of four runs each, as far as the tile has runs (K, wave-uniform; kBase = first run of the array)
template <int G0, int kBase>
__device__ __forceinline__ void late_groups(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off, const int32_t* tvp, int K, uint32_t shift,
                                            uint32_t (&A)[8])
{
    if constexpr (G0 < 4) {
        if (kBase + 4 * G0 < K) {
            int w[4];
            request_words<4>(rsrc, lane_off, tvp + kBase + 4 * G0, w);
            digest_words<4, 4 * G0>(w, shift, A);
            late_groups<G0 + 1, kBase>(rsrc, lane_off, tvp, K, shift, A);
        }
    }
}

constexpr int kRecord = 8 + kAhead + 2;   // dwords of a candidate's record in LDS: lo, tile << 8 | runs, hi, rows, tv[kAhead], columns in reach

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
    __shared__ int s_rec[kWaves][64][kRecord];
    __shared__ float4 s_box[kWaves][64][2];
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

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(bits), 0, 0x7ffffffc, 0x00020000);
    const uint32_t shift = va & 31;
    // the tile: distances + the penalties of runs 0..15 for both sub-tiles
    auto product = [&](float qx, float qy, float qz, bool exists, int ridx, uint32_t (&R)[8], f32x16 (&acc)[2]) {
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
    };
    // the penalties of runs 16..31 on top
    auto product_high = [&](int ridx, uint32_t (&R)[8], f32x16 (&acc)[2]) {
        const unsigned e1 = (unsigned)(ridx - 16 - 8 * h);
        const u32x4 oh1 = onehot[e1 < 8u ? e1 : 8u];
#pragma unroll
        for (int m = 0; m < 4; ++m) swap_halves(R[m], R[4 + m]);
        const u32x4 Q0 = {R[0], R[1], R[2], R[3]}, Q1 = {R[4], R[5], R[6], R[7]};
        acc[0] = mfma_bf16(oh1, Q0, acc[0]);
        acc[1] = mfma_bf16(oh1, Q1, acc[1]);
    };
    __syncthreads();                                                          // onehot

    // ---- pass 1: one representative per tile -> an upper bound for every column ------------------------------------
    float bestf[2] = {kNoKey, kNoKey};
    const int reps = (tiles + kRepStep - 1) / kRepStep, rep_tiles = (reps + 31) >> 5;
    for (int u = wave; u < rep_tiles; u += kWaves) {
        const size_t rb = (size_t)b * rep_stride + 32 * u;
        const int K = min(32, reps - 32 * u);
        int w[32];
        request_words<32>(rsrc, lane_off, rep_tv + rb, w);
        const float4 q = rep_xyz[rb + j];
        uint32_t R[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, Rh[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        {
            int wl[16], wh[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { wl[k] = w[k]; wh[k] = w[16 + k]; }
            digest_words<16, 0>(wl, shift, R);
            digest_words<16, 0>(wh, shift, Rh);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) { R[m] = ~R[m] & kPenaltyPair; Rh[m] = ~Rh[m] & kPenaltyPair; }
        f32x16 acc[2];
        product(q.x, q.y, q.z, j < K, j, R, acc);
        product_high(j, Rh, acc);
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
    // Candidates, 64 tiles per wavefront at a time with lane <-> tile: the tile's box against the box of the column block
    // and the largest bound in it (a lower bound of what any single column could find).  A survivor's record (box, runs,
    // the first kAhead mask vertices) goes to LDS: the loop over the survivors then finds everything a tile needs one ds_read
    // away and requests the NEXT tile's mask words and rows before it works on the current one.
    auto request_tile = [&](int i, int (&w)[kAhead], float& qx, float& qy, float& qz, int& ridx) {
        const int* rec = s_rec[wave][i];
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
            w[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off + 8u * (uint32_t)rec[8 + k], 0, 0);
        const int t = __builtin_amdgcn_readfirstlane(rec[3]) >> 8;
        const int rc = min(t * kTile + j, n - 1);
        qx = bp[3 * rc]; qy = bp[3 * rc + 1]; qz = bp[3 * rc + 2];
        ridx = mb[t].ridx[j];
    };
    auto tile_step = [&](int i, const int (&w)[kAhead], float qx, float qy, float qz, int ridx) {
        const int* rec = s_rec[wave][i];
        // (the columns in reach were found when the candidates were listed; testing again against the bounds as they are
        // now prunes a few tiles more and costs more than it saves)
        const bool near = (rec[(lane >> 5) + 8 + kAhead] >> (lane & 31)) & 1;
        const int tk = __builtin_amdgcn_readfirstlane(rec[3]), t = tk >> 8, K = tk & 255;
        const int rows = __builtin_amdgcn_readfirstlane(rec[7]);
        uint32_t A[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, Ah[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        digest_words<kAhead, 0>(w, shift, A);
        if (K > kAhead) {                                                    // wave-uniform, rare for patch-sorted points
            late_groups<kAhead / 4, 0>(rsrc, lane_off, mb[t].tv, K, shift, A);
            if (K > 16) late_groups<0, 16>(rsrc, lane_off, mb[t].tv, K, shift, Ah);
        }
        uint32_t any = 0u;
#pragma unroll
        for (int m = 0; m < 8; ++m) { any |= A[m]; A[m] = ~A[m] & kPenaltyPair; }
        if (K > 16) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { any |= Ah[m]; Ah[m] = ~Ah[m] & kPenaltyPair; }
        }
        // a column that is near but has no admissible row here does not make the tile worth a product
        if (__builtin_amdgcn_ballot_w64(near && any != 0u) == 0) return;
#ifdef TUCH_SCAN_COUNTS
        {   // diagnostic build (tools/diag/hd_counts.py): tiles that reach the products, their columns in reach with an admissible run
            const unsigned long long live_ = __builtin_amdgcn_ballot_w64(near && any != 0u), near_ = __builtin_amdgcn_ballot_w64(near);
            if (lane == 0) {
                atomicAdd(&g_hd_counts[0], 1ull);
                atomicAdd(&g_hd_counts[1], (unsigned long long)__builtin_popcountll(live_));
                atomicAdd(&g_hd_counts[2], (unsigned long long)__builtin_popcountll(near_));
                atomicAdd(&g_hd_counts[3 + min(__builtin_popcountll(live_) / 8, 8)], 1ull);
            }
        }
#endif
        f32x16 acc[2];
        product(qx, qy, qz, j < rows, ridx, A, acc);
        if (K > 16) product_high(ridx, Ah, acc);
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
        uint32_t k0 = (uint32_t)bestkey[0], k1 = (uint32_t)bestkey[1];
        swap_halves(k0, k1);
        const float f = __int_as_float(min((int)k0, (int)k1) & ~15);
        bnd = fminf(bnd, (f - own_o) + 4e-6f * (f + R2));
    };
    for (int t0 = wave; t0 < tiles; t0 += 64 * kWaves) {
        int nc;
        {
            const float bmax = wave_max_uniform(bnd);           // (bnd: a squared distance or +inf, never a NaN)
            const int tl = t0 + kWaves * lane;
            bool cand = false;
            float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = l4;
            if (tl < tiles) {
                l4 = *reinterpret_cast<const float4*>(mb[tl].lo);
                h4 = *reinterpret_cast<const float4*>(mb[tl].hi);
                const float gx = fmaxf(fmaxf(l4.x - hi[0], lo[0] - h4.x), 0.0f);
                const float gy = fmaxf(fmaxf(l4.y - hi[1], lo[1] - h4.y), 0.0f);
                const float gz = fmaxf(fmaxf(l4.z - hi[2], lo[2] - h4.z), 0.0f);
                cand = __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx)) * kSlack <= bmax;
            }
            // the survivors against every column's own bound (tile boxes from the lanes that hold them): mask words and
            // rows are requested only for tiles some column can reach
            // (the candidate's box as two broadcast reads of LDS instead of six v_readlane + the moves their scalar operands
            // need -- as in v2v.hip's scan; s_box: this wavefront's 64 tile boxes of the trip)
            unsigned long long m = 0ull, my_reach = 0ull;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(cand);
            if (todo) {
                s_box[wave][lane][0] = l4;
                s_box[wave][lane][1] = h4;
            }
            const float bnd_s = bnd * (1.0f / kSlack);
            while (todo) {
                const int pos = (int)__builtin_ctzll(todo);
                asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(pos));
                const float4 bl = s_box[wave][pos][0], bh = s_box[wave][pos][1];
                const float dx = px - __builtin_amdgcn_fmed3f(px, bl.x, bh.x);
                const float dy = py - __builtin_amdgcn_fmed3f(py, bl.y, bh.y);
                const float dz = pz - __builtin_amdgcn_fmed3f(pz, bl.z, bh.z);
                const unsigned long long reach =
                    __builtin_amdgcn_ballot_w64(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= bnd_s);
                if (reach) m |= 1ull << pos;
                if (lane == pos) my_reach = reach;
            }
            cand = (m >> lane) & 1ull;
            nc = __builtin_popcountll(m);
            if (cand) {
                int* rec = s_rec[wave][__builtin_popcountll(m & ((1ull << lane) - 1ull))];
                rec[0] = __float_as_int(l4.x); rec[1] = __float_as_int(l4.y); rec[2] = __float_as_int(l4.z);
                rec[3] = (tl << 8) | __float_as_int(l4.w);
                rec[4] = __float_as_int(h4.x); rec[5] = __float_as_int(h4.y); rec[6] = __float_as_int(h4.z);
                rec[7] = __float_as_int(h4.w);
                rec[8 + kAhead] = (int)(uint32_t)my_reach; rec[9 + kAhead] = (int)(uint32_t)(my_reach >> 32);
#pragma unroll
                for (int g = 0; g < kAhead / 4; ++g) {
                    const int4 tv = *reinterpret_cast<const int4*>(mb[tl].tv + 4 * g);
                    rec[8 + 4 * g] = tv.x; rec[9 + 4 * g] = tv.y; rec[10 + 4 * g] = tv.z; rec[11 + 4 * g] = tv.w;
                }
            }
        }
        if (nc == 0) continue;
        int wA[kAhead], wB[kAhead], rA = 0, rB = 0;
        float qA[3] = {0.f, 0.f, 0.f}, qB[3] = {0.f, 0.f, 0.f};
        request_tile(0, wA, qA[0], qA[1], qA[2], rA);
        for (int i = 0; i < nc; i += 2) {
            // tile i lives in A; tile i + 1 is requested into B before A is worked on, and the other way round.  The
            // requests are unconditional (behind the last tile: the last tile again): with the same number of loads in
            // flight on every path the compiler can wait for exactly the older set
            request_tile(min(i + 1, nc - 1), wB, qB[0], qB[1], qB[2], rB);
            tile_step(i, wA, qA[0], qA[1], qA[2], rA);
            request_tile(min(i + 2, nc - 1), wA, qA[0], qA[1], qA[2], rA);
            if (i + 1 < nc) tile_step(i + 1, wB, qB[0], qB[1], qB[2], rB);
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
