// emd.hip -- placeholder until the exact-OT kernel lands (next commit).
#include "common.h"

int ann_emd_launch(annchor_ctx *c, const PairSource &, double *, double *, uint8_t *)
{
    ann_set_err(c, "wasserstein kernel not built into this library yet");
    return ANNCHOR_EINVAL;
}
