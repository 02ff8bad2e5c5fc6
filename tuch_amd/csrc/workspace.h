// Workspace regions and their guard words (debug option "canary", TUCH_CANARY=1 at model creation).
//
// Every hot call carves its caller-supplied workspace into regions with tuch_ws_take(o, bytes).  With the option on,
// each region is followed by kGuardBytes that no kernel may touch: the entry point writes 0xDEADBEEF words there before
// it enqueues its kernels (arm) and compares them after the last one (check, in the scope's destructor); a changed
// word adds one to the model's device counter, read with tuch_contact_model_canary_hits.  Both steps are kernels on the
// call's stream: no host synchronisation, capturable.  With the option off none of this exists (no bytes, no launches).
#pragma once
#include "common.h"

constexpr int kGuardBytes = 256;
constexpr int kMaxGuards = 96;

struct tuch_ws_plan {
    int guard = 0;              // bytes behind every region: 0 or kGuardBytes
    bool recording = false;     // layout code run by the entry point itself records the guard positions
    size_t base = 0;            // offset of the layout being recorded within the call's workspace
    int n = 0;
    size_t at[kMaxGuards];
    tuch_ws_plan* outer = nullptr;
};

extern thread_local tuch_ws_plan* tuch_ws_active;

inline size_t tuch_ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

// one region: returns its offset, advances o past it (and past its guard)
inline size_t tuch_ws_take(size_t& o, size_t bytes, bool guarded = true)
{
    const size_t at = o;
    o += tuch_ws_align(bytes);
    tuch_ws_plan* p = tuch_ws_active;
    if (p && p->guard && guarded) {
        if (p->recording && p->n < kMaxGuards) p->at[p->n++] = p->base + o;
        o += p->guard;
    }
    return at;
}

// sizing a nested layout from inside a recorded one must not record (its guards are recorded with their own base)
struct tuch_ws_pause {
    bool was = false;
    tuch_ws_pause() { if (tuch_ws_active) { was = tuch_ws_active->recording; tuch_ws_active->recording = false; } }
    ~tuch_ws_pause() { if (tuch_ws_active) tuch_ws_active->recording = was; }
};

void tuch_ws_arm(const tuch_ws_plan& p, void* workspace, hipStream_t s);
void tuch_ws_check(const tuch_ws_plan& p, const void* workspace, int32_t* hits, hipStream_t s);

// One per entry point: switches the guards on for every layout computed inside, records, arms, and checks on exit.
struct tuch_ws_scope {
    tuch_ws_plan plan;
    void* ws = nullptr;
    int32_t* hits = nullptr;
    hipStream_t stream = nullptr;
    explicit tuch_ws_scope(bool on)
    {
        plan.guard = on ? kGuardBytes : 0;
        plan.outer = tuch_ws_active;
        tuch_ws_active = &plan;
    }
    // run `layout()` (a layout function) with its guards recorded at workspace offset `base`
    template <class F> auto record(size_t base, F layout) -> decltype(layout())
    {
        tuch_ws_plan* inner = tuch_ws_active;       // (another scope of the same entry point may be the innermost one)
        tuch_ws_active = &plan;
        plan.base = base;
        plan.recording = plan.guard != 0;
        auto l = layout();
        plan.recording = false;
        tuch_ws_active = inner;
        return l;
    }
    // the recorded guards live in `workspace`: write them now, compare when the scope ends
    void arm(void* workspace, int32_t* hit_counter, hipStream_t s)
    {
        if (!plan.guard || !plan.n || !hit_counter) return;
        ws = workspace; hits = hit_counter; stream = s;
        tuch_ws_arm(plan, ws, s);
    }
    ~tuch_ws_scope()
    {
        if (ws) tuch_ws_check(plan, ws, hits, stream);
        tuch_ws_active = plan.outer;
    }
};
