// brute.hip -- placeholder until the brute-force kernel lands (next commit).
#include "common.h"

extern "C" int annchor_brute_force(annchor_ctx *c, int32_t, int64_t *, double *)
{
    if (!c) return ANNCHOR_EINVAL;
    ann_set_err(c, "brute force not built into this library yet");
    return ANNCHOR_EINVAL;
}
