// Entry points declared in include/uavppo.h whose kernels are not written yet: they fail
// loudly (non-zero + message), never fall back.  Each moves to its own file when implemented.
#include "common.h"
#define NOT_YET(name) do { uav_set_error(name ": not implemented yet"); return 99; } while (0)
extern "C" {
int uav_rollout(uav_ctx*, void*, int, const uav_env_cfg*, int, const float*, int, int, uint64_t, float*, float*, float*,
                float*, int32_t*, float*, float*, float*, float*, uint8_t*, float*, float*, const int32_t*,
                const double*, int32_t*, uav_stream) { NOT_YET("uav_rollout"); }
}
