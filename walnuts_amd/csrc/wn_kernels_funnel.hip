// the funnel device model: kernels for every launch geometry + registry entry (wn_kernels.inc)
#include "wn_models.h"
#define WN_MODEL_ID 2
#define WN_MODEL_TAG funnel
#define WN_MODEL_TYPE wn::FunnelModel
#include "wn_kernels.inc"
