// How fast can 18.6 MB (the SMPL blend matrix) be streamed once?  rows x 20736 floats, read with 16-byte loads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_read.hip -o tools/ubench/stream_read && tools/ubench/stream_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// variant A: flat -- thread t reads float4 t, t + stride, ... (kPer in flight), sums, writes one float per thread (rarely)
template <int kPer>
__global__ __launch_bounds__(256) void flat_kernel(const f32x4* __restrict__ p, size_t n4, float* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) v[u] = i + u * stride < n4 ? p[i + u * stride] : (f32x4){0, 0, 0, 0};
    float s = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    if (s == 12345.678f) out[i] = s;
}
// variant B: the blend kernel's pattern -- workgroup = 64 columns, wave w rows 56w .. 56w+55, lane (lm, lq): row 4i + lq, cols 4 lm
__global__ __launch_bounds__(256) void slab_kernel(const float* __restrict__ p, int N3p, float* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lq = lane >> 4;
    const float* b = p + (size_t)(wave * 56 + lq) * N3p + blockIdx.x * 64 + 4 * lm;
    f32x4 v[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) v[i] = *(const f32x4*)(b + (size_t)(4 * i) * N3p);
    float s = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// variant C: rows-major slabs -- workgroup = 256 columns (1 KB per row), wave w rows w, w+4, ...: lane reads 16 B: a wave reads one contiguous KB per load
template <int kPer>
__global__ __launch_bounds__(256) void rowslab_kernel(const float* __restrict__ p, int N3p, int rows, float* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.y * (4 * kPer) + wave;
    const float* b = p + (size_t)r0 * N3p + blockIdx.x * 256 + 4 * lane;
    f32x4 v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = r0 + 4 * i < rows ? *(const f32x4*)(b + (size_t)(4 * i) * N3p) : (f32x4){0, 0, 0, 0};
    float s = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ void touch_kernel(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] += 1.0f; }

template <typename F> float timeit(F f, int n, bool flush, float* big, size_t bign)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float total = 0;
    for (int it = 0; it < n; ++it) {
        if (flush) { hipLaunchKernelGGL(touch_kernel, dim3(4096), dim3(256), 0, 0, big, bign); }
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); total += ms;
    }
    return total / n * 1e3f;
}
int main()
{
    const int rows = 224, N3p = 20736;
    const size_t n = (size_t)rows * N3p;
    float *p, *out, *big;
    const size_t bign = (size_t)192 << 20;     // 768 MB of floats: evicts the 256 MB Infinity Cache
    hipMalloc(&p, n * 4); hipMalloc(&out, 1 << 20); hipMalloc(&big, bign * 4);
    hipMemset(p, 0, n * 4); hipMemset(big, 0, bign * 4);
    printf("%.1f MB\n", n * 4 / 1e6);
    for (int flush = 0; flush < 2; ++flush) {
        printf("%s\n", flush ? "after evicting the caches (768 MB touched in between):" : "back to back (data may sit in L2 / Infinity Cache):");
        printf("  empty launch             %6.1f us\n", timeit([&] { hipLaunchKernelGGL(flat_kernel<1>, dim3(1), dim3(256), 0, 0, (const f32x4*)p, (size_t)0, out); }, 20, flush, big, bign));
        printf("  flat  1 load / thread    %6.1f us\n", timeit([&] { hipLaunchKernelGGL(flat_kernel<1>, dim3(n / 4 / 256), dim3(256), 0, 0, (const f32x4*)p, n / 4, out); }, 20, flush, big, bign));
        printf("  flat  4 loads / thread   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(flat_kernel<4>, dim3(n / 4 / 256 / 4), dim3(256), 0, 0, (const f32x4*)p, n / 4, out); }, 20, flush, big, bign));
        printf("  flat  8 loads / thread   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(flat_kernel<8>, dim3(n / 4 / 256 / 8), dim3(256), 0, 0, (const f32x4*)p, n / 4, out); }, 20, flush, big, bign));
        printf("  flat 16 loads / thread   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(flat_kernel<16>, dim3(n / 4 / 256 / 16 + 1), dim3(256), 0, 0, (const f32x4*)p, n / 4, out); }, 20, flush, big, bign));
        printf("  blend pattern (324 wg)   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(slab_kernel, dim3(N3p / 64), dim3(256), 0, 0, (const float*)p, N3p, out); }, 20, flush, big, bign));
        printf("  row slabs  4 rows/wave   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(rowslab_kernel<4>, dim3(N3p / 256, rows / 16), dim3(256), 0, 0, (const float*)p, N3p, rows, out); }, 20, flush, big, bign));
        printf("  row slabs  7 rows/wave   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(rowslab_kernel<7>, dim3(N3p / 256, rows / 28), dim3(256), 0, 0, (const float*)p, N3p, rows, out); }, 20, flush, big, bign));
        printf("  row slabs 14 rows/wave   %6.1f us\n", timeit([&] { hipLaunchKernelGGL(rowslab_kernel<14>, dim3(N3p / 256, rows / 56), dim3(256), 0, 0, (const float*)p, N3p, rows, out); }, 20, flush, big, bign));
    }
    return 0;
}
