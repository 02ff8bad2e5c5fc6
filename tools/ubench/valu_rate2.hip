// More issue-rate probes for gfx950 (see valu_rate.hip): min3 / max3, compare-to-mask + scalar OR + ballot branch, readlane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, int sel)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int cnt = 0;
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {        // 8 min3
            x0 = __builtin_fminf(__builtin_fminf(x0, x1), a); x1 = __builtin_fminf(__builtin_fminf(x1, x2), a);
            x2 = __builtin_fminf(__builtin_fminf(x2, x3), a); x3 = __builtin_fminf(__builtin_fminf(x3, x4), a);
            x4 = __builtin_fminf(__builtin_fminf(x4, x5), a); x5 = __builtin_fminf(__builtin_fminf(x5, x6), a);
            x6 = __builtin_fminf(__builtin_fminf(x6, x7), a); x7 = __builtin_fminf(__builtin_fminf(x7, x0), a);
        } else if (MODE == 1) { // 4 x (2 compares to masks + OR + ballot branch): the candidate test of the ray walk
            x0 += a; x1 += a; x2 += a; x3 += a;
            if (__builtin_amdgcn_ballot_w64((x0 >= b) | (x1 <= -b))) cnt += 1;
            if (__builtin_amdgcn_ballot_w64((x1 >= b) | (x2 <= -b))) cnt += 1;
            if (__builtin_amdgcn_ballot_w64((x2 >= b) | (x3 <= -b))) cnt += 1;
            if (__builtin_amdgcn_ballot_w64((x3 >= b) | (x0 <= -b))) cnt += 1;
        } else if (MODE == 2) { // 4 x (mul + 1 compare + ballot branch)
            x0 += a; x1 += a; x2 += a; x3 += a;
            if (__builtin_amdgcn_ballot_w64(x0 * x1 >= b)) cnt += 1;
            if (__builtin_amdgcn_ballot_w64(x1 * x2 >= b)) cnt += 1;
            if (__builtin_amdgcn_ballot_w64(x2 * x3 >= b)) cnt += 1;
            if (__builtin_amdgcn_ballot_w64(x3 * x0 >= b)) cnt += 1;
        } else if (MODE == 3) { // 8 readlane + add (broadcast of a lane's value, wave-uniform lane index)
            const int l = (i + sel) & 63;
            x0 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), l));
            x1 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x2), l));
            x2 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x3), l));
            x3 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), l));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + cnt;
}
template <int MODE>
void run(const char* name, float* out, int per_iter)
{
    const int waves = 8, grid = 256 * 4 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 1.0001f, 1e30f, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 1.0001f, 1e30f, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s 8 waves/SIMD: %.3f ms -> %.1f cycles per loop trip per SIMD-wave-slot (%d items per trip: %.2f cycles each at 2.4 GHz)\n",
           name, ms, ms * 1e6 / (waves * N_ITER) * 2.4, per_iter, ms * 1e6 / (waves * N_ITER) * 2.4 / per_iter);
}
int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    run<0>("min3 x8", out, 8); run<1>("add + 2cmp|or + branch x4", out, 4); run<2>("add + mul cmp + branch x4", out, 4);
    run<3>("readlane + add x4", out, 4);
    return 0;
}
