// the diag_normal device model: kernels for every launch geometry + registry entry (wn_kernels.inc)
#include "wn_models.h"
#define WN_MODEL_ID 1
#define WN_MODEL_TAG diag_normal
#define WN_MODEL_TYPE wn::DiagNormalModel
#include "wn_kernels.inc"
