// the std_normal device model: kernels for every launch geometry + registry entry (wn_kernels.inc)
#include "wn_models.h"
#define WN_MODEL_ID 0
#define WN_MODEL_TAG std_normal
#define WN_MODEL_TYPE wn::StdNormalModel
#include "wn_kernels.inc"
