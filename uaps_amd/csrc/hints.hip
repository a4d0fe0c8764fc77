#include "hints.hpp"
#include <string.h>

namespace {
thread_local uaps_call_hints g_hints;
thread_local bool g_have = false;
}

namespace uaps {
uaps_call_hints take_hints() {
    uaps_call_hints h;
    if (g_have) { h = g_hints; g_have = false; }
    else memset(&h, 0, sizeof h);
    return h;
}
}  // namespace uaps

extern "C" int uaps_next_call_hints(const uaps_call_hints* h) {
    if (!h) { g_have = false; return UAPS_OK; }
    for (int i = 0; i < 3; ++i)
        if (h->bound[i] && !(h->mul[i] > 0.f && h->mul[i] < 3.0e38f)) return UAPS_EINVAL;
    g_hints = *h; g_have = true;
    return UAPS_OK;
}
