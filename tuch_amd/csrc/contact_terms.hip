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
        if (mode == 0 && !(d < euclthres)) return t;
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
__global__ __launch_bounds__(256) void contact_terms_bwd_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ partner,
    const uint8_t* __restrict__ exterior, const float* __restrict__ gscale,
    int N, int mode, float euclthres, float* __restrict__ grad)
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
    float* gi = grad + ((size_t)b * N + i) * 3;
    float* gp = grad + ((size_t)b * N + p) * 3;
    atomicAdd(gi + 0, c * dx); atomicAdd(gi + 1, c * dy); atomicAdd(gi + 2, c * dz);
    atomicAdd(gp + 0, -c * dx); atomicAdd(gp + 1, -c * dy); atomicAdd(gp + 2, -c * dz);
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

}  // namespace

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
    hipLaunchKernelGGL(contact_terms_bwd_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0,
                       (hipStream_t)stream, points, partner, exterior, grad_scale, N, mode, euclthres,
                       grad_points);
    return tuch_check_launch("tuch_contact_terms_bwd");
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
