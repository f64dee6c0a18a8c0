// transition + init kernels of the std_normal device model, all launch geometries
#include <string>
#define WN_MODEL_TYPE wn::StdNormalModel
#define WN_MODEL_TAG std_normal
#include "wn_kernels.inc"
