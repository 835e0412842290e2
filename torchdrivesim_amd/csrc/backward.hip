// Backward kernels of K2a / K2b (config 5).  Placeholder until the analytic gradients land: the entry points exist,
// fail loudly, and never fall back to anything else.
#include "tds_common.h"

TDS_EXPORT int tds_collision_bwd_f32(const float *, const float *, const uint8_t *, const float *, float *, float *, int64_t, int64_t,
                                     int64_t, int, void *) {
    tds::set_error("tds_collision_bwd_f32: not implemented yet");
    return TDS_EINVAL;
}

TDS_EXPORT int tds_offroad_bwd_f32(const tds_map_t *, const float *, const float *, const float *, const uint8_t *, const float *, float *,
                                   float *, float *, int64_t, float, void *) {
    tds::set_error("tds_offroad_bwd_f32: not implemented yet");
    return TDS_EINVAL;
}
