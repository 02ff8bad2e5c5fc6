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
