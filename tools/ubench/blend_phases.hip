// Where does blend_kernel's time go at batch 64?  The kernel of csrc/smpl_lbs.hip (one wavefront = kTiles body tiles x 16
// columns, full K) with s_memtime stamps: start, operands arrived, MFMAs done, stores issued -- per wavefront.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/blend_phases.hip -o tools/ubench/blend_phases
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kSteps = 56;
template <int kTilesIn>
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ feat, int fpad, const float* __restrict__ blend,
                                                    int B, int N3p, float* __restrict__ v_posed, unsigned long long* stamps)
{
    constexpr int kTiles = kTilesIn == 0 ? 1 : kTilesIn;
    typedef float avec __attribute__((ext_vector_type(kTiles)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lq = lane >> 4;
    const int m_base = blockIdx.y * (16 * kTiles);
    const int col = blockIdx.x * 64 + wave * 16 + lm;
    const float* b_ptr = blend + (size_t)lq * N3p + col;
    const float* a_ptr = feat + (size_t)lq * fpad + m_base + kTiles * lm;
    float bv[kSteps];
    avec av[kSteps];
#pragma unroll
    for (int i = 0; i < kSteps; ++i) { bv[i] = b_ptr[(size_t)(4 * i) * N3p]; av[i] = *(const avec*)(a_ptr + (size_t)(4 * i) * fpad); }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kSteps; ++i)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t], bv[i], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float keep = 0;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) keep += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    asm volatile("" :: "v"(keep));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + (lq * 4 + r) * kTiles + t;
            if (m < B) v_posed[(size_t)m * N3p + col] = acc[t][r];
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
    }
}
// the round-4 kernel: 64 bodies x 32 columns per workgroup, K split over four wavefronts, partial tiles through LDS
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__global__ __launch_bounds__(256) void old_blend_kernel(const float* __restrict__ feat, int fpad, const float* __restrict__ blend, int B, int N3,
                                                        float* __restrict__ v_posed, unsigned long long* stamps)
{
    constexpr int kBlendSteps = 55, kBlendWaveSteps = 14, kBlendJ = 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __shared__ float red[4][3][kBlendJ * 4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    const int m_base = blockIdx.y * 64;
    const int cbase = blockIdx.x * (16 * kBlendJ) + lm * kBlendJ;
    const float* a_ptr = feat + m_base + lm * 4;
    const float* b_ptr = blend + (cbase < N3 ? cbase : 0);
    f32x4 a[kBlendWaveSteps];
    f32x2u bv[kBlendWaveSteps];
#pragma unroll
    for (int i = 0; i < kBlendWaveSteps; ++i) {
        const int k = min(wave * kBlendWaveSteps + i, kBlendSteps - 1) * 4 + lq;
        a[i] = *(const f32x4*)(a_ptr + (size_t)k * fpad);
        bv[i] = *(const f32x2u*)(b_ptr + (size_t)k * N3);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4][kBlendJ];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kBlendWaveSteps; ++i) {
        const bool live = wave * kBlendWaveSteps + i < kBlendSteps;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float av = (i < kBlendSteps - 3 * kBlendWaveSteps || live) ? a[i][t] : 0.f;
#pragma unroll
            for (int j = 0; j < kBlendJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i][j], acc[t][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float keep = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) keep += acc[t][0][0] + acc[t][1][3];
    asm volatile("" :: "v"(keep));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t == wave) continue;
        const int slot = wave < t ? wave : wave - 1;
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[t][slot][j * 4 + r][lane] = acc[t][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m_base + (lq * 4 + r) * 4 + wave;
#pragma unroll
        for (int j = 0; j < kBlendJ; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) sum = (t == wave) ? acc[t][j][r] : sum;
#pragma unroll
            for (int slot = 0; slot < 3; ++slot) sum += red[wave][slot][j * 4 + r][lane];
            if (m < B && cbase + j < N3) v_posed[(size_t)m * N3 + cbase + j] = sum;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
    }
}
template <int kTiles> void run(int B)
{
    const int N3p = 20736, fpad = 64, rows = 224;
    float *feat, *blend, *vp; unsigned long long* st;
    const bool old = kTiles == 0;
    const int gy = old ? 1 : (B + 16 * (old ? 4 : kTiles) - 1) / (16 * (old ? 4 : kTiles)), gx = old ? N3p / 32 : N3p / 64;
    hipMalloc(&feat, rows * fpad * 4); hipMalloc(&blend, (size_t)rows * N3p * 4); hipMalloc(&vp, (size_t)B * N3p * 4);
    hipMalloc(&st, (size_t)gx * gy * 16 * 8);
    hipMemset(feat, 0, rows * fpad * 4); hipMemset(blend, 0, (size_t)rows * N3p * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        if constexpr (kTiles == 0) hipLaunchKernelGGL(old_blend_kernel, dim3(gx, gy), dim3(256), 0, 0, feat, fpad, blend, B, N3p, vp, st);
        else hipLaunchKernelGGL(blend_kernel<kTiles>, dim3(gx, gy), dim3(256), 0, 0, feat, fpad, blend, B, N3p, vp, st);
    };
    for (int it = 0; it < 5; ++it) launch();
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)gx * gy * 16);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = 0; i < h.size(); i += 4) { lo = std::min(lo, h[i]); hi = std::max(hi, h[i + 3]); }
    std::vector<double> start, loads, mfma, store;
    for (size_t i = 0; i < h.size(); i += 4) {
        start.push_back((h[i] - lo) * 1.0); loads.push_back((h[i + 1] - h[i]) * 1.0); mfma.push_back((h[i + 2] - h[i + 1]) * 1.0);
        store.push_back((h[i + 3] - h[i + 2]) * 1.0);
    }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("tiles %d batch %d: %.1f us per launch (20 back to back); (ticks of s_memtime below)\n", kTiles, B, ms / 20 * 1e3);
    printf("   operands arrive: median %.1f p90 %.1f max %.1f | MFMAs: median %.1f p90 %.1f max %.1f | stores: median %.1f p90 %.1f max %.1f ticks\n",
           pct(loads, .5), pct(loads, .9), pct(loads, 1), pct(mfma, .5), pct(mfma, .9), pct(mfma, 1), pct(store, .5), pct(store, .9), pct(store, 1));
}
int main() { run<0>(64); run<1>(64); run<2>(64); run<4>(64); run<1>(8); run<0>(8); return 0; }
