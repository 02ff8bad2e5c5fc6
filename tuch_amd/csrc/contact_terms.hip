// Pull/push contact terms and their gradient (K4 of SURVEY.md §2.2).
//
// For every point i with partner p(i) (the masked nearest neighbour found by
// tuch_v2v_min_masked):  d_i = |x_i - x_p(i)|  and
//   SMPLify (tuch/smplify/losses.py:96-105):
//       interior:                    1.0   * tanh(d/0.04)^2
//       exterior and d < euclthres:  0.005 * tanh(d/0.005)^2
//   regressor (tuch/train/loss.py:303-315), also used on HD points (:299):
//       exterior:                    0.005 * tanh(d/0.005)^2   (no distance gate)
//       interior:                    1.0   * tanh(d/0.04)^2
// Forward: one block per body, fixed-order tree reduction => deterministic sums.
// Backward: d(term)/dd * (x_i - x_p)/d goes to BOTH endpoints (losses.py:98); the
// partner side is a scatter, done with float atomics (order-dependent only in
// the last ulp of a gradient, never in a loss value).  torch.norm's backward at
// d == 0 is 0, reproduced explicitly.
#include "common.h"
#include "model.h"

namespace {

constexpr int kBlock = 1024;

struct Term {
    float value;   // weight * tanh(d/scale)^2
    float dd;      // d value / d d
};

__device__ __forceinline__ Term contact_term(float d, bool exterior, int mode, float euclthres)
{
    Term t = {0.0f, 0.0f};
    float weight, scale;
    if (exterior) {
        if (mode == 0 && !(d < euclthres)) {
            // a NaN distance (a non-finite vertex or partner) poisons the body's sum whatever the flag says: the reference's
            // winding numbers are NaN for EVERY query of such a body (contact.py:79-109), `.le(0.99)` is false, the vertex
            // counts as interior and its tanh^2(NaN) reaches the loss (losses.py:96-105)
            if (d != d) t.value = d;
            return t;
        }
        weight = 0.005f; scale = 0.005f;
    } else {
        weight = 1.0f; scale = 0.04f;
    }
    const float th = tanhf(d / scale);
    t.value = weight * th * th;
    t.dd = 2.0f * weight * th * (1.0f - th * th) / scale;
    return t;
}

__device__ __forceinline__ float block_sum(float v, float* smem)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float r = 0.0f;
    if (wave == 0) {
        r = lane < (kBlock / 64) ? smem[lane] : 0.0f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    }
    __syncthreads();
    return r;   // valid in thread 0
}

// terms[b] = {interior sum, exterior sum}
__global__ __launch_bounds__(kBlock) void contact_terms_fwd_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner,
    const uint8_t* __restrict__ exterior, const uint8_t* __restrict__ body_valid,
    int N, int mode, float euclthres, float* __restrict__ terms)
{
    __shared__ float smem[kBlock / 64];
    const int b = blockIdx.x;
    float in_sum = 0.0f, ex_sum = 0.0f;
    if (!body_valid || body_valid[b]) {
        const float* pb = pts + (size_t)b * N * 3;
        for (int i = threadIdx.x; i < N; i += kBlock) {
            const int p = partner[(size_t)b * N + i];
            const float dx = pb[3 * i] - pb[3 * p], dy = pb[3 * i + 1] - pb[3 * p + 1],
                        dz = pb[3 * i + 2] - pb[3 * p + 2];
            const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
            const bool ext = exterior[(size_t)b * N + i] != 0;
            const Term t = contact_term(d, ext, mode, euclthres);
            if (ext) ex_sum += t.value; else in_sum += t.value;
        }
    }
    const float a = block_sum(in_sum, smem);
    const float c = block_sum(ex_sum, smem);
    if (threadIdx.x == 0) { terms[2 * b] = a; terms[2 * b + 1] = c; }
}

// Ragged form (HD points, loss.py:299-315): body b owns points off[b]..off[b+1] of ONE
// concatenated set; partner indices are global.  terms[b] = {interior sum, exterior sum}.
__global__ __launch_bounds__(kBlock) void contact_terms_ragged_fwd_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner,
    const uint8_t* __restrict__ exterior, const int32_t* __restrict__ off, int mode, float euclthres,
    float* __restrict__ terms)
{
    __shared__ float smem[kBlock / 64];
    const int b = blockIdx.x;
    float in_sum = 0.0f, ex_sum = 0.0f;
    for (int i = off[b] + threadIdx.x; i < off[b + 1]; i += kBlock) {
        const int p = partner[i];
        const float dx = pts[3 * (size_t)i] - pts[3 * (size_t)p], dy = pts[3 * (size_t)i + 1] - pts[3 * (size_t)p + 1],
                    dz = pts[3 * (size_t)i + 2] - pts[3 * (size_t)p + 2];
        const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
        const bool ext = exterior[i] != 0;
        const Term t = contact_term(d, ext, mode, euclthres);
        if (ext) ex_sum += t.value; else in_sum += t.value;
    }
    const float a = block_sum(in_sum, smem);
    const float c = block_sum(ex_sum, smem);
    if (threadIdx.x == 0) { terms[2 * b] = a; terms[2 * b + 1] = c; }
}

// grad[b][i] += g_b * dd * (x_i - x_p)/d ; grad[b][p] -= the same.  grad pre-zeroed by the caller.
// Fixed: deterministic mode -- the same scatter into zeroed 64-bit fixed-point accumulators (common.h: fixed_add; integer
// atomics are associative, the sums do not depend on the order of arrival), converted by fixed_to_float_kernel.
template <bool Fixed>
__global__ __launch_bounds__(256) void contact_terms_bwd_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner,
    const uint8_t* __restrict__ exterior, const float* __restrict__ gscale,
    int N, int mode, float euclthres, float* __restrict__ grad, long long* __restrict__ grad_fixed)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool ext = exterior[(size_t)b * N + i] != 0;
    const float g = gscale[2 * b + (ext ? 1 : 0)];   // upstream gradient of the interior / exterior sum
    if (g == 0.0f) return;
    const float* pb = pts + (size_t)b * N * 3;
    const int p = partner[(size_t)b * N + i];
    const float dx = pb[3 * i] - pb[3 * p], dy = pb[3 * i + 1] - pb[3 * p + 1],
                dz = pb[3 * i + 2] - pb[3 * p + 2];
    const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d > 0.0f)) return;
    const Term t = contact_term(d, ext, mode, euclthres);
    if (t.dd == 0.0f) return;
    const float c = g * t.dd / d;
    if (Fixed) {
        long long* gi = grad_fixed + ((size_t)b * N + i) * 3;
        long long* gp = grad_fixed + ((size_t)b * N + p) * 3;
        fixed_add(gi + 0, c * dx); fixed_add(gi + 1, c * dy); fixed_add(gi + 2, c * dz);
        fixed_add(gp + 0, -c * dx); fixed_add(gp + 1, -c * dy); fixed_add(gp + 2, -c * dz);
    } else {
        float* gi = grad + ((size_t)b * N + i) * 3;
        float* gp = grad + ((size_t)b * N + p) * 3;
        atomicAdd(gi + 0, c * dx); atomicAdd(gi + 1, c * dy); atomicAdd(gi + 2, c * dz);
        atomicAdd(gp + 0, -c * dx); atomicAdd(gp + 1, -c * dy); atomicAdd(gp + 2, -c * dz);
    }
}

// ragged backward: point i belongs to body body_of[i]; grad [N,3] pre-zeroed by the caller
__global__ __launch_bounds__(256) void contact_terms_ragged_bwd_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner, const uint8_t* __restrict__ exterior,
    const int32_t* __restrict__ body_of, const float* __restrict__ gscale, int N, int mode, float euclthres,
    float* __restrict__ grad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool ext = exterior[i] != 0;
    const float g = gscale[2 * body_of[i] + (ext ? 1 : 0)];
    if (g == 0.0f) return;
    const int p = partner[i];
    const float dx = pts[3 * (size_t)i] - pts[3 * (size_t)p], dy = pts[3 * (size_t)i + 1] - pts[3 * (size_t)p + 1],
                dz = pts[3 * (size_t)i + 2] - pts[3 * (size_t)p + 2];
    const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d > 0.0f)) return;
    const Term t = contact_term(d, ext, mode, euclthres);
    if (t.dd == 0.0f) return;
    const float c = g * t.dd / d;
    float* gi = grad + (size_t)i * 3;
    float* gp = grad + (size_t)p * 3;
    atomicAdd(gi + 0, c * dx); atomicAdd(gi + 1, c * dy); atomicAdd(gi + 2, c * dz);
    atomicAdd(gp + 0, -c * dx); atomicAdd(gp + 1, -c * dy); atomicAdd(gp + 2, -c * dz);
}


// ---- the tail of the SMPLify-DC stage-2 objective in two launches (tuch/smplify/losses.py:96-123) -------------------
// Forward: one block per body adds up its contact terms (as contact_terms_fwd_kernel) and its row of region minima and
// leaves the body's share of the objective; the block that finishes last adds the shares up in a fixed order
//   total = sum_b [ reprojection_b + prior_b + contact_scale (interior_b + exterior_b) + r2r_scale sum_p r2r[b,p] ]
// (objective_kernel's sum; a launch of its own for one block of additions cost 5 us of the serial tail).
__global__ __launch_bounds__(kBlock) void stage2_finish_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner, const uint8_t* __restrict__ exterior,
    const uint8_t* __restrict__ body_valid, int N, int mode, float euclthres, const float* __restrict__ small,
    const float* __restrict__ r2r, int P, float contact_scale, float r2r_scale, float* __restrict__ share,
    int* __restrict__ ticket, float* __restrict__ terms, float* __restrict__ out)
{
    __shared__ float smem[kBlock / 64];
    __shared__ bool last;
    const int b = blockIdx.x;
    float in_sum = 0.0f, ex_sum = 0.0f, r_sum = 0.0f;
    if (!body_valid || body_valid[b]) {
        const float* pb = pts + (size_t)b * N * 3;
        // eight points per thread and pass (one pass at SMPL size): their partners first, then both endpoints of all of
        // them -- two rounds of loads per pass instead of two per point (the sums are added in point order all the same)
        constexpr int kPer = 8;
        for (int i0 = threadIdx.x; i0 < N; i0 += kBlock * kPer) {
            int pr[kPer];
            uint8_t ex[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = min(i0 + u * kBlock, N - 1);
                pr[u] = partner[(size_t)b * N + i];
                ex[u] = exterior[(size_t)b * N + i];
            }
            float xi[kPer][3], xp[kPer][3];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = min(i0 + u * kBlock, N - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) { xi[u][c] = pb[3 * i + c]; xp[u][c] = pb[3 * pr[u] + c]; }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                if (i0 + u * kBlock >= N) break;
                const float dx = xi[u][0] - xp[u][0], dy = xi[u][1] - xp[u][1], dz = xi[u][2] - xp[u][2];
                const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
                const bool ext = ex[u] != 0;
                const Term t = contact_term(d, ext, mode, euclthres);
                if (ext) ex_sum += t.value; else in_sum += t.value;
            }
        }
    }
    if (r2r)
        for (int p = threadIdx.x; p < P; p += kBlock) r_sum += r2r[(size_t)b * P + p];
    const float a = block_sum(in_sum, smem);
    const float c = block_sum(ex_sum, smem);
    const float r = block_sum(r_sum, smem);
    if (threadIdx.x == 0) {
        if (terms) { terms[2 * b] = a; terms[2 * b + 1] = c; }
        const float mine = (small[2 * b] + small[2 * b + 1]) + contact_scale * (a + c) + r2r_scale * r;
        __hip_atomic_store(share + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock)
        acc += __hip_atomic_load(share + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float total = block_sum(acc, smem);
    if (threadIdx.x == 0) {
        out[0] = total;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
    }
}

// Backward: the upstream scalar g times the objective's constants, straight into the vertex gradient (pre-zeroed):
// contact terms (contact_terms_bwd_kernel with the weight g contact_scale [body valid]) in the vertex blocks; the last
// block of every body scatters the region minima (region_pair_min_bwd_kernel with weight g r2r_scale) and scales the
// unit gradients small_terms_kernel left for the joints, the camera and the pose.  Three launches before.
__global__ __launch_bounds__(256) void stage2_bwd_kernel(
    const float* __restrict__ gout, const uint8_t* __restrict__ body_valid, const float* __restrict__ pts,
    const int32_t* __restrict__ partner, const uint8_t* __restrict__ exterior, int N, int mode, float euclthres,
    float contact_scale, const int32_t* __restrict__ ij, int P, float r2r_scale, const float* __restrict__ gj,
    const float* __restrict__ gc, const float* __restrict__ gp, int NJ, float* __restrict__ grad,
    float* __restrict__ gj_out, float* __restrict__ gc_out, float* __restrict__ gp_out)
{
    const int b = blockIdx.y;
    const float g = gout[0];
    const float* pb = pts + (size_t)b * N * 3;
    float* gb = grad + (size_t)b * N * 3;
    if (blockIdx.x == gridDim.x - 1) {
        const float gr = r2r_scale * g;
        if (ij && gr != 0.0f)
            for (int p = threadIdx.x; p < P; p += 256) {
                const size_t o = (size_t)b * P + p;
                const int i = ij[2 * o], j = ij[2 * o + 1];
                if (i < 0 || j < 0) continue;
                for (int c = 0; c < 3; ++c) {
                    const float d = 2.0f * gr * (pb[3 * i + c] - pb[3 * j + c]);
                    atomicAdd(gb + 3 * i + c, d);
                    atomicAdd(gb + 3 * j + c, -d);
                }
            }
        for (int i = threadIdx.x; i < NJ * 3; i += 256) gj_out[(size_t)b * NJ * 3 + i] = g * gj[(size_t)b * NJ * 3 + i];
        if (threadIdx.x < 3) gc_out[b * 3 + threadIdx.x] = g * gc[b * 3 + threadIdx.x];
        if (gp && threadIdx.x < 69) gp_out[b * 69 + threadIdx.x] = g * gp[b * 69 + threadIdx.x];
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    // the point's own data is requested before the upstream gradient and the body's flag are looked at (one round of
    // loads less on a kernel that is nothing but its load latencies)
    const bool ext = exterior[(size_t)b * N + i] != 0;
    const int p = partner[(size_t)b * N + i];
    const float xi = pb[3 * i], yi = pb[3 * i + 1], zi = pb[3 * i + 2];
    const float gs = (!body_valid || body_valid[b]) ? contact_scale * g : 0.0f;
    if (gs == 0.0f) return;
    const float dx = xi - pb[3 * p], dy = yi - pb[3 * p + 1], dz = zi - pb[3 * p + 2];
    const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d > 0.0f)) return;
    const Term t = contact_term(d, ext, mode, euclthres);
    if (t.dd == 0.0f) return;
    const float c = gs * t.dd / d;
    float* gi = gb + 3 * (size_t)i;
    float* gq = gb + 3 * (size_t)p;
    atomicAdd(gi + 0, c * dx); atomicAdd(gi + 1, c * dy); atomicAdd(gi + 2, c * dz);
    atomicAdd(gq + 0, -c * dx); atomicAdd(gq + 1, -c * dy); atomicAdd(gq + 2, -c * dz);
}

#ifdef TUCH_STAGE2_CLOCKS
// diagnostic build only (tools/diag/stage2_clocks.py): s_memrealtime stamps (100 MHz) of block (0, 0) and of the block that
// arrives last: [0] start, [1] loads done, [2] atomics issued, [3] block sums, [4] ticket taken; last block: [5] start of the
// final reduction, [6] end
__device__ unsigned long long g_stage2_clocks[16];
#define STAGE2_CLOCK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_stage2_clocks[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAGE2_CLOCK_LAST(i) do { if (threadIdx.x == 0) g_stage2_clocks[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int tuch_debug_stage2_clocks(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stage2_clocks), sizeof(unsigned long long) * 16) == hipSuccess ? TUCH_OK : TUCH_ERR_HIP;
}
#else
#define STAGE2_CLOCK(i) do { } while (0)
#define STAGE2_CLOCK_LAST(i) do { } while (0)
#endif

// Forward AND the unit-seed backward of the tail in ONE launch (stage2_finish + stage2_bwd with g = 1): the objective is
// the root of the fit's autograd graph, its upstream gradient is the constant 1 (loss.backward()), so the vertex
// gradient can be written while the sums are formed -- one kernel and one dependent launch less in the serial tail of
// every iteration.  A caller whose upstream gradient is not 1 scales the outputs (ops._Stage2Tail.backward).
// grid (kFusedSplits, B) x 256 threads: a body's points are shared out over kFusedSplits blocks (the scatter's atomics
// want more than B workgroups); partial sums per (body, split), added up in a fixed order by the block that arrives last.
constexpr int kFusedSplits = 8;
constexpr int kFusedBlock = 256;

__device__ __forceinline__ float block_sum_256(float v, float* smem)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    const float r = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return r;   // same value in every thread
}

__global__ __launch_bounds__(kFusedBlock) void stage2_fused_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner, const uint8_t* __restrict__ exterior,
    const uint8_t* __restrict__ body_valid, int N, int mode, float euclthres, const float* __restrict__ small,
    const float* __restrict__ r2r, const int32_t* __restrict__ ij, int P, float contact_scale, float r2r_scale,
    float* __restrict__ share,        // [B][kFusedSplits][3] scratch
    int* __restrict__ ticket, float* __restrict__ terms, float* __restrict__ out,
    float* __restrict__ grad,         // [B,N,3] pre-zeroed, or nullptr (value only)
    // instead of (r2r, ij): the raw keys of tuch_region_pair_keys and the model's region tables
    const unsigned long long* __restrict__ pair_keys, const int32_t* __restrict__ region_off,
    const int32_t* __restrict__ region_vidx, const int32_t* __restrict__ pairs,
    long long* __restrict__ grad_fixed)     // deterministic mode: [B,N,3] 64-bit fixed-point accumulators (zeroed) instead of grad
{
    __shared__ bool last;
    const int s = blockIdx.x, b = blockIdx.y;
    const bool valid = !body_valid || body_valid[b];
    const float* pb = pts + (size_t)b * N * 3;
    float* gb = grad ? grad + (size_t)b * N * 3 : nullptr;
    long long* fb = grad_fixed ? grad_fixed + (size_t)b * N * 3 : nullptr;
    const bool want = gb || fb;
    auto add3 = [&](int at, float x, float y, float z) {
        if (fb) { fixed_add(fb + 3 * (size_t)at, x); fixed_add(fb + 3 * (size_t)at + 1, y); fixed_add(fb + 3 * (size_t)at + 2, z); }
        else { atomicAdd(gb + 3 * (size_t)at, x); atomicAdd(gb + 3 * (size_t)at + 1, y); atomicAdd(gb + 3 * (size_t)at + 2, z); }
    };
    // What the PARTNERS receive meets in LDS first: a penetrating region's vertices share few partners (20 to 40 vertices
    // of a body scatter into the same one: tools/diag/partner_degree.py), and atomics on one address queue behind each
    // other in the L2 -- the block that arrived last was 19 us behind the first (tools/diag/stage2_clocks.py).  A
    // direct-mapped table of kPartnerSlots partners per block; a partner whose slot is taken goes to memory as before.
    constexpr int kPartnerSlots = 256;
    // (deterministic mode: the same table with 64-bit fixed-point cells -- integer sums, so which contributions meet in LDS
    // and which go to memory does not change the result; without it the mode paid the queueing again, +10 us per step)
    __shared__ int p_tag[kPartnerSlots];
    __shared__ long long p_acc[kPartnerSlots][3];       // float mode: the low words hold the float sums
    const bool in_lds = want;
    if (in_lds) {
        for (int k = threadIdx.x; k < kPartnerSlots; k += kFusedBlock) { p_tag[k] = -1; p_acc[k][0] = p_acc[k][1] = p_acc[k][2] = 0; }
        __syncthreads();
    }
    auto add3_partner = [&](int at, float x, float y, float z) {
        if (in_lds) {
            const int slot = at & (kPartnerSlots - 1);
            const int old = atomicCAS(&p_tag[slot], -1, at);
            if (old == -1 || old == at) {
                if (fb) { fixed_add(&p_acc[slot][0], x); fixed_add(&p_acc[slot][1], y); fixed_add(&p_acc[slot][2], z); }
                else {
                    atomicAdd(reinterpret_cast<float*>(&p_acc[slot][0]), x); atomicAdd(reinterpret_cast<float*>(&p_acc[slot][1]), y);
                    atomicAdd(reinterpret_cast<float*>(&p_acc[slot][2]), z);
                }
                return;
            }
        }
        add3(at, x, y, z);
    };
    STAGE2_CLOCK(0);
    float in_sum = 0.0f, ex_sum = 0.0f, r_sum = 0.0f;
    const int per = (N + kFusedSplits - 1) / kFusedSplits;
    const int beg = s * per, end = min(beg + per, N);
    if (valid) {
        constexpr int kPer = 4;       // partners first, then both endpoints of all four: two rounds of loads per pass
        for (int i0 = beg + threadIdx.x; i0 < end; i0 += kFusedBlock * kPer) {
            int pr[kPer];
            uint8_t ex[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = min(i0 + u * kFusedBlock, end - 1);
                pr[u] = partner[(size_t)b * N + i];
                ex[u] = exterior[(size_t)b * N + i];
            }
            float xi[kPer][3], xp[kPer][3];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = min(i0 + u * kFusedBlock, end - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) { xi[u][c] = pb[3 * i + c]; xp[u][c] = pb[3 * pr[u] + c]; }
            }
#ifdef TUCH_STAGE2_CLOCKS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAGE2_CLOCK(1);
#endif
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = i0 + u * kFusedBlock;
                if (i >= end) break;
                const float dx = xi[u][0] - xp[u][0], dy = xi[u][1] - xp[u][1], dz = xi[u][2] - xp[u][2];
                const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
                const bool ext = ex[u] != 0;
                const Term t = contact_term(d, ext, mode, euclthres);
                if (ext) ex_sum += t.value; else in_sum += t.value;
                if (want && d > 0.0f && t.dd != 0.0f) {
                    const float c = contact_scale * t.dd / d;
                    add3(i, c * dx, c * dy, c * dz);
                    add3_partner(pr[u], -c * dx, -c * dy, -c * dz);
                }
            }
        }
    }
    STAGE2_CLOCK(2);
    if (in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < kPartnerSlots; k += kFusedBlock) {
            if (p_tag[k] < 0) continue;
            if (fb) {
                long long* dst = fb + 3 * (size_t)p_tag[k];
                for (int c = 0; c < 3; ++c) atomicAdd((unsigned long long*)(dst + c), (unsigned long long)p_acc[k][c]);
            } else
                add3(p_tag[k], *reinterpret_cast<float*>(&p_acc[k][0]), *reinterpret_cast<float*>(&p_acc[k][1]),
                     *reinterpret_cast<float*>(&p_acc[k][2]));
        }
    }
    if (s == 0 && (r2r || pair_keys)) {
        for (int p = threadIdx.x; p < P; p += kFusedBlock) {
            const size_t o = (size_t)b * P + p;
            int i = -1, j = -1;
            if (pair_keys) {
                const unsigned long long inv = pair_keys[o];
                if (inv != 0ull) {                       // (d2 bits << 32 | flat index), complemented
                    const unsigned long long key = ~inv;
                    const int r1 = pairs[2 * p], r2 = pairs[2 * p + 1];
                    const int n2 = region_off[r2 + 1] - region_off[r2];
                    const int idx = (int)(unsigned int)key;
                    r_sum += __uint_as_float((unsigned int)(key >> 32));
                    i = region_vidx[region_off[r1] + idx / n2];
                    j = region_vidx[region_off[r2] + idx % n2];
                }
            } else {
                r_sum += r2r[o];
                if (ij) { i = ij[2 * o]; j = ij[2 * o + 1]; }
            }
            if (want && r2r_scale != 0.0f) {
                if (i >= 0 && j >= 0) {
                    const float k2 = 2.0f * r2r_scale;
                    const float ex2 = k2 * (pb[3 * i] - pb[3 * j]), ey2 = k2 * (pb[3 * i + 1] - pb[3 * j + 1]),
                                ez2 = k2 * (pb[3 * i + 2] - pb[3 * j + 2]);
                    add3(i, ex2, ey2, ez2);
                    add3(j, -ex2, -ey2, -ez2);
                }
            }
        }
    }
    // the three sums together: one butterfly (the three values' steps interleave), one exchange through LDS -- one after
    // the other they were three times six dependent shuffles and six barriers on the step's serial tail
    STAGE2_CLOCK(7);
    float a = in_sum, c = ex_sum, r = r_sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); r += __shfl_down(r, o, 64); }
    __shared__ float sums3[kFusedBlock / 64][3];
    if ((threadIdx.x & 63) == 0) { float* q = sums3[threadIdx.x >> 6]; q[0] = a; q[1] = c; q[2] = r; }
    __syncthreads();
    a = sums3[0][0] + sums3[1][0] + sums3[2][0] + sums3[3][0];
    c = sums3[0][1] + sums3[1][1] + sums3[2][1] + sums3[3][1];
    r = sums3[0][2] + sums3[1][2] + sums3[2][2] + sums3[3][2];
    STAGE2_CLOCK(3);
    if (threadIdx.x == 0) {
        float* mine = share + ((size_t)b * kFusedSplits + s) * 3;
        __hip_atomic_store(mine + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (two levels of tickets -- the last block of a body takes a ticket of the launch -- were slower, 20.3 against 17.1 us:
        // it is not the 512 arrivals on one word that cost but the release in front of each, ~4.5 us per block on this
        // multi-XCD part; the body's last block then pays it twice)
        last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * gridDim.y) - 1;
    }
    __syncthreads();
    STAGE2_CLOCK(4);
    if (!last) return;
    STAGE2_CLOCK_LAST(5);
    // bodies in order, a body's splits in order: the total does not depend on which block came last
    float acc = 0.0f;
    const int B = gridDim.y;
    for (int bb = threadIdx.x; bb < B; bb += kFusedBlock) {
        float ia = 0.0f, ea = 0.0f, ra = 0.0f;
        for (int k = 0; k < kFusedSplits; ++k) {
            const float* q = share + ((size_t)bb * kFusedSplits + k) * 3;
            ia += __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ea += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ra += __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (terms) { terms[2 * bb] = ia; terms[2 * bb + 1] = ea; }
        acc += (small[2 * bb] + small[2 * bb + 1]) + contact_scale * (ia + ea) + r2r_scale * ra;
    }
    // fixed-order sum over the block (B <= 256: one body per thread; more: a thread's bodies in order first)
    __shared__ float part[kFusedBlock];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.0f;
        const int n = B < kFusedBlock ? B : kFusedBlock;
        for (int k = 0; k < n; ++k) total += part[k];
        out[0] = total;
        STAGE2_CLOCK_LAST(6);
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// deterministic mode: the fixed-point accumulators -> the float gradient
__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fixed_value(acc[i]);
}

}  // namespace

extern "C" size_t tuch_smplify_stage2_fused_scratch_floats(int B) { return (size_t)(B > 0 ? B : 0) * kFusedSplits * 3; }

// tuch_smplify_stage2_finish and, when grad_points is given, tuch_smplify_stage2_bwd for a unit upstream gradient, in one
// launch.  share: tuch_smplify_stage2_fused_scratch_floats(B) floats; ticket: one int, zero before the call (left zero);
// grad_points [B,N,3] pre-zeroed or NULL; ij [B,P,2] from tuch_region_pair_min (needed for the gradient of the region
// term) or NULL.  The unit gradients of the small terms are tuch_smplify_small_terms' own outputs.
extern "C" int tuch_smplify_stage2_fused(const float* points, const int32_t* partner, const uint8_t* exterior,
                                         const uint8_t* body_valid, int B, int N, int mode, float euclthres,
                                         const float* small_terms, const float* r2r, const int32_t* ij, int P,
                                         float contact_scale, float r2r_scale, float* share, int* ticket, float* terms,
                                         float* out, float* grad_points, const tuch_contact_model* model,
                                         const void* pair_keys, void* grad_fixed_zeroed, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && small_terms && share && ticket && out,
                 "tuch_smplify_stage2_fused: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && N > 0 && P >= 0 && (mode == 0 || mode == 1), "tuch_smplify_stage2_fused: bad arguments");
    TUCH_REQUIRE(!pair_keys || (model && model->num_pairs == P && P > 0),
                 "tuch_smplify_stage2_fused: pair keys need the model they were computed with (P = its number of pairs)");
    const bool raw = pair_keys != nullptr;
    // deterministic mode (tuch_deterministic(): TUCH_DETERMINISTIC=1 / tuch_set_deterministic): the caller passes B*N*3
    // zeroed 64-bit words; the scatter accumulates fixed-point integers there, a second launch converts to grad_points
    // grad_points NULL with grad_fixed_zeroed given: the accumulators are the result (no conversion launch -- the consumer
    // reads them: tuch_smpl_backward_split_add's g_verts_fixed; tuch_fixed_to_float converts on demand)
    const bool fixed = grad_fixed_zeroed != nullptr;
    hipLaunchKernelGGL(stage2_fused_kernel, dim3(kFusedSplits, B), dim3(kFusedBlock), 0, (hipStream_t)stream, points, partner,
                       exterior, body_valid, N, mode, euclthres, small_terms, (P > 0 && !raw ? r2r : (const float*)nullptr),
                       (P > 0 && !raw ? ij : (const int32_t*)nullptr), P, contact_scale, r2r_scale, share, ticket, terms, out,
                       fixed ? (float*)nullptr : grad_points, (const unsigned long long*)pair_keys,
                       raw ? (const int32_t*)model->region_off : nullptr, raw ? (const int32_t*)model->region_vidx : nullptr,
                       raw ? (const int32_t*)model->pairs : nullptr, fixed ? (long long*)grad_fixed_zeroed : (long long*)nullptr);
    if (fixed && grad_points) {
        const size_t n = (size_t)B * N * 3;
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const long long*)grad_fixed_zeroed, grad_points, n);
    }
    return tuch_check_launch("tuch_smplify_stage2_fused");
}

extern "C" int tuch_fixed_to_float(const void* fixed, size_t n, float* out, void* stream)
{
    TUCH_REQUIRE(fixed && out, "tuch_fixed_to_float: null pointer");
    if (n == 0) return TUCH_OK;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)fixed, out, n);
    return tuch_check_launch("tuch_fixed_to_float");
}

// loss.py:317 (and :272): the per-body terms summed over the VALID bodies and divided by their number, as one launch
// (torch: a count, two sums, a division and their conversions -- seven small kernels forward, five backward in a chain
// where every launch is 3 - 5 us).  out[0] = sum / n (0 / 0 = NaN like the reference's), out[1] = 1 / n.  One workgroup.
__global__ __launch_bounds__(256) void valid_mean_fwd_kernel(const float* __restrict__ terms, const uint8_t* __restrict__ valid,
                                                            int B, int K, float* __restrict__ out)
{
    __shared__ float s_sum[256];
    __shared__ int s_cnt[256];
    float sum = 0.0f;
    int cnt = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        if (valid[b]) {
            ++cnt;
            for (int k = 0; k < K; ++k) sum += terms[(size_t)b * K + k];
        }
    }
    s_sum[threadIdx.x] = sum;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {              // fixed order: reproducible
        if ((int)threadIdx.x < d) { s_sum[threadIdx.x] += s_sum[threadIdx.x + d]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = (float)s_cnt[0];
        out[0] = s_sum[0] / n;
        out[1] = 1.0f / n;
    }
}

// grad_terms[b][k] = upstream * (valid[b] ? 1 / n : 0)
__global__ __launch_bounds__(256) void valid_mean_bwd_kernel(const float* __restrict__ upstream, const float* __restrict__ fwd_out,
                                                            const uint8_t* __restrict__ valid, int B, int K,
                                                            float* __restrict__ grad_terms)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * K) return;
    grad_terms[i] = valid[i / K] ? upstream[0] * fwd_out[1] : 0.0f;
}

extern "C" int tuch_valid_mean_fwd(const float* terms, const uint8_t* valid, int B, int K, float* out, void* stream)
{
    TUCH_REQUIRE(terms && valid && out, "tuch_valid_mean_fwd: null pointer");
    TUCH_REQUIRE(B > 0 && K > 0, "tuch_valid_mean_fwd: bad sizes");
    hipLaunchKernelGGL(valid_mean_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, terms, valid, B, K, out);
    return tuch_check_launch("tuch_valid_mean_fwd");
}

extern "C" int tuch_valid_mean_bwd(const float* upstream, const float* fwd_out, const uint8_t* valid, int B, int K,
                                   float* grad_terms, void* stream)
{
    TUCH_REQUIRE(upstream && fwd_out && valid && grad_terms, "tuch_valid_mean_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && K > 0, "tuch_valid_mean_bwd: bad sizes");
    hipLaunchKernelGGL(valid_mean_bwd_kernel, dim3(ceil_div(B * K, 256)), dim3(256), 0, (hipStream_t)stream, upstream, fwd_out,
                       valid, B, K, grad_terms);
    return tuch_check_launch("tuch_valid_mean_bwd");
}

extern "C" int tuch_contact_terms_fwd(const float* points, const int32_t* partner,
                                      const uint8_t* exterior, const uint8_t* body_valid, int B, int N,
                                      int mode, float euclthres, float* terms, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && terms, "tuch_contact_terms_fwd: null pointer");
    TUCH_REQUIRE(B > 0 && N > 0 && (mode == 0 || mode == 1), "tuch_contact_terms_fwd: bad arguments");
    hipLaunchKernelGGL(contact_terms_fwd_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, points,
                       partner, exterior, body_valid, N, mode, euclthres, terms);
    return tuch_check_launch("tuch_contact_terms_fwd");
}

extern "C" int tuch_contact_terms_bwd(const float* points, const int32_t* partner,
                                      const uint8_t* exterior, const float* grad_scale, int B, int N,
                                      int mode, float euclthres, float* grad_points, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && grad_scale && grad_points,
                 "tuch_contact_terms_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && N > 0 && (mode == 0 || mode == 1), "tuch_contact_terms_bwd: bad arguments");
    hipLaunchKernelGGL(contact_terms_bwd_kernel<false>, dim3(ceil_div(N, 256), B), dim3(256), 0,
                       (hipStream_t)stream, points, partner, exterior, grad_scale, N, mode, euclthres,
                       grad_points, (long long*)nullptr);
    return tuch_check_launch("tuch_contact_terms_bwd");
}

extern "C" int tuch_contact_terms_bwd_fixed(const float* points, const int32_t* partner,
                                            const uint8_t* exterior, const float* grad_scale, int B, int N,
                                            int mode, float euclthres, void* grad_fixed_zeroed, float* grad_points, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && grad_scale && grad_points && grad_fixed_zeroed,
                 "tuch_contact_terms_bwd_fixed: null pointer");
    TUCH_REQUIRE(B > 0 && N > 0 && (mode == 0 || mode == 1), "tuch_contact_terms_bwd_fixed: bad arguments");
    hipLaunchKernelGGL(contact_terms_bwd_kernel<true>, dim3(ceil_div(N, 256), B), dim3(256), 0,
                       (hipStream_t)stream, points, partner, exterior, grad_scale, N, mode, euclthres,
                       (float*)nullptr, (long long*)grad_fixed_zeroed);
    const size_t n = (size_t)B * N * 3;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)grad_fixed_zeroed, grad_points, n);
    return tuch_check_launch("tuch_contact_terms_bwd_fixed");
}

extern "C" int tuch_contact_terms_ragged_fwd(const float* points, const int32_t* partner,
                                             const uint8_t* exterior, const int32_t* offsets, int B, int mode,
                                             float euclthres, float* terms, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && offsets && terms, "tuch_contact_terms_ragged_fwd: null pointer");
    TUCH_REQUIRE(B > 0 && (mode == 0 || mode == 1), "tuch_contact_terms_ragged_fwd: bad arguments");
    hipLaunchKernelGGL(contact_terms_ragged_fwd_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, points,
                       partner, exterior, offsets, mode, euclthres, terms);
    return tuch_check_launch("tuch_contact_terms_ragged_fwd");
}

extern "C" int tuch_contact_terms_ragged_bwd(const float* points, const int32_t* partner,
                                             const uint8_t* exterior, const int32_t* body_of_point,
                                             const float* grad_scale, int N, int mode, float euclthres,
                                             float* grad_points, void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && body_of_point && grad_scale && grad_points,
                 "tuch_contact_terms_ragged_bwd: null pointer");
    TUCH_REQUIRE(N >= 0 && (mode == 0 || mode == 1), "tuch_contact_terms_ragged_bwd: bad arguments");
    if (N == 0) return TUCH_OK;
    hipLaunchKernelGGL(contact_terms_ragged_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       points, partner, exterior, body_of_point, grad_scale, N, mode, euclthres, grad_points);
    return tuch_check_launch("tuch_contact_terms_ragged_bwd");
}

// The stage-2 objective behind the inside test and the nearest-vertex search, and its backward pass, as one launch
// each (see stage2_finish_kernel / stage2_bwd_kernel).  share: [B] floats of scratch; ticket: one int, zero before the
// first call and left zero by every call (one per stream in flight); terms: [B,2] or NULL.
extern "C" int tuch_smplify_stage2_finish(const float* points, const int32_t* partner, const uint8_t* exterior,
                                          const uint8_t* body_valid, int B, int N, int mode, float euclthres,
                                          const float* small_terms, const float* r2r, int P, float contact_scale,
                                          float r2r_scale, float* share, int* ticket, float* terms, float* out,
                                          void* stream)
{
    TUCH_REQUIRE(points && partner && exterior && small_terms && share && ticket && out,
                 "tuch_smplify_stage2_finish: null pointer");
    TUCH_REQUIRE(B > 0 && N > 0 && P >= 0 && (mode == 0 || mode == 1), "tuch_smplify_stage2_finish: bad arguments");
    hipLaunchKernelGGL(stage2_finish_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, points, partner, exterior,
                       body_valid, N, mode, euclthres, small_terms, (P > 0 ? r2r : (const float*)nullptr), P,
                       contact_scale, r2r_scale, share, ticket, terms, out);
    return tuch_check_launch("tuch_smplify_stage2_finish");
}

extern "C" int tuch_smplify_stage2_bwd(const float* grad_out, const uint8_t* body_valid, const float* points,
                                       const int32_t* partner, const uint8_t* exterior, int B, int N, int mode,
                                       float euclthres, float contact_scale, const int32_t* ij, int P, float r2r_scale,
                                       const float* gj, const float* gc, const float* gp, int NJ, float* grad_points,
                                       float* gj_out, float* gc_out, float* gp_out, void* stream)
{
    TUCH_REQUIRE(grad_out && points && partner && exterior && gj && gc && grad_points && gj_out && gc_out &&
                 (!gp || gp_out), "tuch_smplify_stage2_bwd: null pointer");
    TUCH_REQUIRE(B > 0 && B <= 65535 && N > 0 && NJ > 0 && P >= 0 && (mode == 0 || mode == 1),
                 "tuch_smplify_stage2_bwd: bad arguments");
    hipLaunchKernelGGL(stage2_bwd_kernel, dim3(ceil_div(N, 256) + 1, B), dim3(256), 0, (hipStream_t)stream, grad_out,
                       body_valid, points, partner, exterior, N, mode, euclthres, contact_scale,
                       (P > 0 ? ij : (const int32_t*)nullptr), P, r2r_scale, gj, gc, gp, NJ, grad_points, gj_out, gc_out,
                       gp_out);
    return tuch_check_launch("tuch_smplify_stage2_bwd");
}
