// transition + init kernels of the diag_normal device model, all launch geometries
#include <string>
#define WN_MODEL_TYPE wn::DiagNormalModel
#define WN_MODEL_TAG diag_normal
#include "wn_kernels.inc"
