// Adam update of the few small parameter tensors an SMPLify-DC loop optimises (tuch/smplify/smplifydc.py:117,150:
// torch.optim.Adam over body pose [B,69], global orientation [B,3], betas [B,10], camera translation [B,3]) in ONE
// launch: torch's fused / foreach implementations take two or three (step counters, the update) and sit at the very end
// of every iteration's serial chain.  Same arithmetic as torch.optim.Adam without weight decay / amsgrad:
//   t += 1;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One workgroup: the step counter lives on the device (capturable) and is read by every thread before any writes it.
#include "common.h"

namespace {

constexpr int kAdamBlock = 1024;
constexpr int kAdamMaxTensors = 8;

struct AdamTensors {
    float* p[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    float beta1[kAdamMaxTensors], beta2[kAdamMaxTensors];
    int n[kAdamMaxTensors];
    int count;
};

__global__ __launch_bounds__(kAdamBlock) void adam_kernel(AdamTensors a, float* __restrict__ step, float lr, float eps)
{
    const float t = step[0] + 1.0f;
    for (int k = 0; k < a.count; ++k) {
        const AdamScalars sc = adam_scalars(lr, a.beta1[k], a.beta2[k], eps, t);
        for (int i = threadIdx.x; i < a.n[k]; i += kAdamBlock) {
            float p = a.p[k][i], m = a.m[k][i], v = a.v[k][i];
            adam_update(p, m, v, a.g[k][i], sc);                                  // common.h: the same bits as the fused form
            a.m[k][i] = m;
            a.v[k][i] = v;
            a.p[k][i] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) step[0] = t;
}

}  // namespace

// params / grads / exp_avg / exp_avg_sq: `count` (<= 8) device pointers each, sizes[k] floats; step: one device float
// (the number of updates so far, incremented here); betas as [count][2].
extern "C" int tuch_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int* sizes, const float* betas, float* step, float lr,
                              float eps, void* stream)
{
    TUCH_REQUIRE(count > 0 && count <= kAdamMaxTensors, "tuch_adam_step: %d tensors (1..%d)", count, kAdamMaxTensors);
    TUCH_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && betas && step, "tuch_adam_step: null pointer");
    AdamTensors a;
    a.count = count;
    for (int k = 0; k < count; ++k) {
        TUCH_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k] && sizes[k] >= 0, "tuch_adam_step: bad tensor %d", k);
        a.p[k] = params[k]; a.g[k] = grads[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
        a.n[k] = sizes[k]; a.beta1[k] = betas[2 * k]; a.beta2[k] = betas[2 * k + 1];
    }
    hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(kAdamBlock), 0, (hipStream_t)stream, a, step, lr, eps);
    return tuch_check_launch("tuch_adam_step");
}
