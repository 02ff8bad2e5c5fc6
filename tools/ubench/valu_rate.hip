// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for plain / packed FP32 fma,
// sqrt, rcp, with SGPR operands, at 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITER 2000
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {        // 8 independent plain fma (VGPR operands + 2 SGPR via a, b)
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 1) { // 8 independent packed fma
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            p4 = __builtin_elementwise_fma(p4, pa, pb); p5 = __builtin_elementwise_fma(p5, pa, pb);
            p6 = __builtin_elementwise_fma(p6, pa, pb); p7 = __builtin_elementwise_fma(p7, pa, pb);
        } else if (MODE == 2) { // 8 sqrt
            x0 = __builtin_amdgcn_sqrtf(x0); x1 = __builtin_amdgcn_sqrtf(x1); x2 = __builtin_amdgcn_sqrtf(x2); x3 = __builtin_amdgcn_sqrtf(x3);
            x4 = __builtin_amdgcn_sqrtf(x4); x5 = __builtin_amdgcn_sqrtf(x5); x6 = __builtin_amdgcn_sqrtf(x6); x7 = __builtin_amdgcn_sqrtf(x7);
        } else if (MODE == 3) { // 8 rcp
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (MODE == 4) { // 8 plain add
            x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
        } else if (MODE == 5) { // 8 packed mul
            p0 *= pa; p1 *= pa; p2 *= pa; p3 *= pa; p4 *= pa; p5 *= pa; p6 *= pa; p7 *= pa;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}
template <int MODE>
void run(const char* name, float* out)
{
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int grid = 256 * 4 * waves;     // one-wave workgroups: `waves` per SIMD if spread evenly
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 1.0001f, 1e-6f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 1.0001f, 1e-6f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double instr_per_simd = (double)waves * N_ITER * 8;
        printf("%-12s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.2f cycles)\n", name, waves, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    run<0>("fma", out); run<1>("pk_fma", out); run<2>("sqrt", out); run<3>("rcp", out); run<4>("add", out); run<5>("pk_mul", out);
    return 0;
}
