// the rw1 device model (models/rw1.h): kernels for every launch geometry + registry entry (wn_kernels.inc)
#include "models/rw1.h"
#define WN_MODEL_ID 3
#define WN_MODEL_TAG rw1
#define WN_MODEL_TYPE wn::Rw1Model
#include "wn_kernels.inc"
