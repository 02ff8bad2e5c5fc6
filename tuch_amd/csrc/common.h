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

// Maximum / minimum over the 64 lanes of a wavefront, wave-uniform (a scalar register): six v_max/min_f32_dpp (quad swaps,
// the two row mirrors, then row_bcast15 / row_bcast31 into the later rows) and one v_readlane of lane 63 -- no trip
// through the LDS crossbar (a __shfl_xor butterfly is six dependent ds_bpermute round trips and ~45 instructions).  No
// operand may be a NaN.  The s_nop are the two wait states a DPP read of a just-written register needs (inline assembly
// is not seen by the hazard recogniser).
#define TUCH_WAVE_REDUCE(op)                                                                           \
    asm("s_nop 1\n\t"                                                                                 \
        op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
        op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
        op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                    \
        op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                         \
        op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                       \
        op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"                            \
        : "+v"(v))
static __device__ __forceinline__ float wave_max_uniform(float v)
{
    TUCH_WAVE_REDUCE("v_max_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
static __device__ __forceinline__ int wave_max_uniform_i32(int v)
{
    TUCH_WAVE_REDUCE("v_max_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ float wave_min_uniform(float v)
{
    TUCH_WAVE_REDUCE("v_min_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Model constants: device memory, or host memory under TUCH_HOST_TABLES=1 (api.hip: sanitizer runs of the table builders)
int tuch_host_tables();
int tuch_table_upload(void** dst, const void* src, size_t bytes);
int tuch_table_download(void* dst_host, const void* src, size_t bytes);
void tuch_table_free(void* p);

// Deterministic mode (the default; TUCH_DETERMINISTIC=0, read once when the library is loaded, or tuch_set_deterministic(0),
// selects float atomics): the gradient
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

// torch.optim.Adam's update of one element (no weight decay / amsgrad), shared by adam_kernel (adam.hip) and the body
// model's last backward kernel (smpl_lbs.hip: PoseAdam) so that the two give the same BITS: every operation is spelled
// out (no contraction left to the compiler, which may group a * b + c * d either way in different kernels).
//   m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= step_size m / (sqrt(v) inv_sqrt_bc2 + eps)
struct AdamScalars { float b1, b2, step_size, inv_sqrt_bc2, eps; };
static __device__ __forceinline__ AdamScalars adam_scalars(float lr, float beta1, float beta2, float eps, float t)
{
    const float bc1 = 1.0f - __builtin_powf(beta1, t), bc2 = 1.0f - __builtin_powf(beta2, t);
    return AdamScalars{beta1, beta2, lr / bc1, 1.0f / __builtin_sqrtf(bc2), eps};
}
#pragma clang fp contract(off)
static __device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const AdamScalars& a)
{
    const float mn = __builtin_fmaf(g - m, 1.0f - a.b1, m);                       // lerp, as torch
    const float vn = __builtin_fmaf((1.0f - a.b2) * g, g, v * a.b2);
    const float denom = __builtin_fmaf(__builtin_sqrtf(vn), a.inv_sqrt_bc2, a.eps);
    m = mn;
    v = vn;
    p = p - a.step_size * mn / denom;
}
#pragma clang fp contract(fast)
