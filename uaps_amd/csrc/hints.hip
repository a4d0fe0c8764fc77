#include "hints.hpp"
#include <string.h>

namespace {
thread_local uaps_call_hints g_hints;
thread_local bool g_have = false;
thread_local uaps::LaunchEvents g_launch;
}

namespace uaps {
uaps_call_hints take_hints() {
    uaps_call_hints h;
    if (g_have) { h = g_hints; g_have = false; }
    else memset(&h, 0, sizeof h);
    return h;
}
LaunchEvents& launch_events() { return g_launch; }
}  // namespace uaps

extern "C" int uaps_next_launch_events(void* start, void* stop) {
    const int used = g_launch.used ? 1 : 0;
    g_launch.start = (hipEvent_t)start; g_launch.stop = (hipEvent_t)stop;
    g_launch.armed = start != nullptr && stop != nullptr;
    g_launch.used = false;
    return used;
}

extern "C" int uaps_next_call_hints(const uaps_call_hints* h) {
    if (!h) { g_have = false; return UAPS_OK; }
    for (int i = 0; i < 3; ++i)
        if (h->bound[i] && !(h->mul[i] > 0.f && h->mul[i] < 3.0e38f)) return UAPS_EINVAL;
    g_hints = *h; g_have = true;
    return UAPS_OK;
}
