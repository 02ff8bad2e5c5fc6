// Guard words between workspace regions (workspace.h): the two kernels and the per-thread plan.
#include "workspace.h"

thread_local tuch_ws_plan* tuch_ws_active = nullptr;

namespace {

constexpr uint32_t kPattern = 0xDEADBEEFu;
struct GuardList { int n; uint32_t at256[kMaxGuards]; };        // offsets in units of 256 bytes (regions are 256-aligned)

__global__ __launch_bounds__(64) void ws_arm_kernel(uint32_t* ws, GuardList g)
{
    ws[(size_t)g.at256[blockIdx.x] * 64 + threadIdx.x] = kPattern;
}

__global__ __launch_bounds__(64) void ws_check_kernel(const uint32_t* ws, GuardList g, int32_t* hits)
{
    if (ws[(size_t)g.at256[blockIdx.x] * 64 + threadIdx.x] != kPattern) atomicAdd(hits, 1);
}

GuardList list_of(const tuch_ws_plan& p)
{
    GuardList g;
    g.n = p.n;
    for (int i = 0; i < p.n; ++i) g.at256[i] = (uint32_t)(p.at[i] / 256);
    return g;
}

}  // namespace

void tuch_ws_arm(const tuch_ws_plan& p, void* workspace, hipStream_t s)
{
    if (p.n > 0) hipLaunchKernelGGL(ws_arm_kernel, dim3(p.n), dim3(64), 0, s, (uint32_t*)workspace, list_of(p));
}

void tuch_ws_check(const tuch_ws_plan& p, const void* workspace, int32_t* hits, hipStream_t s)
{
    if (p.n > 0) hipLaunchKernelGGL(ws_check_kernel, dim3(p.n), dim3(64), 0, s, (const uint32_t*)workspace, list_of(p), hits);
}

// Self-test of the mechanism (tests): three regions in `workspace`, guards armed, one word written just past the second
// region by a kernel -> the check must count it.  Returns the number of guard words found changed (expected: 1), or a
// negative error code.
#include "model.h"
namespace {
__global__ void ws_overrun_kernel(uint32_t* region, size_t words) { region[words] = 0; }
}

extern "C" int tuch_contact_model_canary_selftest(const tuch_contact_model* m, void* workspace, size_t workspace_bytes, void* stream)
{
    TUCH_REQUIRE(m && workspace, "tuch_contact_model_canary_selftest: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int32_t before = 0, after = 0;
    if (hipMemcpy(&before, m->canary_hits, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return TUCH_ERR_HIP;
    size_t second = 0;
    {
        tuch_ws_scope scope(true);
        const size_t total = scope.record(0, [&] {
            size_t o = 0;
            (void)tuch_ws_take(o, 1000);
            second = tuch_ws_take(o, 512);
            (void)tuch_ws_take(o, 64);
            return o;
        });
        TUCH_REQUIRE(workspace_bytes >= total, "tuch_contact_model_canary_selftest: workspace %zu < %zu", workspace_bytes, total);
        scope.arm(workspace, m->canary_hits, s);
        hipLaunchKernelGGL(ws_overrun_kernel, dim3(1), dim3(1), 0, s, (uint32_t*)((char*)workspace + second), (size_t)128);
    }   // <- check
    if (hipStreamSynchronize(s) != hipSuccess ||
        hipMemcpy(&after, m->canary_hits, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(m->canary_hits, &before, sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
        return TUCH_ERR_HIP;
    return after - before;
}
