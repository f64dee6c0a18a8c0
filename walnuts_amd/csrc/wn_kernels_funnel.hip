// transition + init kernels of the funnel device model, all launch geometries
#include <string>
#define WN_MODEL_TYPE wn::FunnelModel
#define WN_MODEL_TAG funnel
#include "wn_kernels.inc"
