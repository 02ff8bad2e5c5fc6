// Shared host/device helpers for libtuch_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define TUCH_OK 0
#define TUCH_ERR_ARG -1
#define TUCH_ERR_HIP -2
#define TUCH_ERR_WORKSPACE -3

void tuch_set_error(const char* fmt, ...);
int tuch_check_launch(const char* what);

#define TUCH_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            tuch_set_error(__VA_ARGS__);        \
            return TUCH_ERR_ARG;                \
        }                                       \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Model constants: device memory, or host memory under TUCH_HOST_TABLES=1 (api.hip: sanitizer runs of the table builders)
int tuch_host_tables();
int tuch_table_upload(void** dst, const void* src, size_t bytes);
int tuch_table_download(void* dst_host, const void* src, size_t bytes);
void tuch_table_free(void* p);

// Deterministic mode (TUCH_DETERMINISTIC=1, read once when the library is loaded, or tuch_set_deterministic): the gradient
// scatters that are float atomics otherwise (order-dependent in the last ulp) accumulate 64-bit fixed-point numbers with
// INTEGER atomics -- associative, so the sums do not depend on the order of arrival -- and are converted once at the end:
// an optimisation then reproduces bit for bit.  kFixedScale: 2^36 (range +-1.3e8, step 1.5e-11).
int tuch_deterministic();
constexpr double kFixedScale = 68719476736.0;
static __device__ __forceinline__ void fixed_add(long long* acc, float x)
{
    atomicAdd((unsigned long long*)acc, (unsigned long long)__double2ll_rn((double)x * kFixedScale));
}
static __device__ __forceinline__ float fixed_value(long long acc) { return (float)((double)acc * (1.0 / kFixedScale)); }
