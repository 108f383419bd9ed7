// shared_mlp_fwd2.hip -- mlp_layer_fwd_kernel's instances with 64 output channels per workgroup (shared_mlp_fwd.h)
#include "shared_mlp_fwd.h"

SIG3D_MLP_FWD_INSTANCES(2, 0)
SIG3D_MLP_FWD_INSTANCES(2, 1)
