// shared_mlp_fwd1.hip -- mlp_layer_fwd_kernel's instances with 32 output channels per workgroup (shared_mlp_fwd.h)
#include "shared_mlp_fwd.h"

SIG3D_MLP_FWD_INSTANCES(1, 0)
SIG3D_MLP_FWD_INSTANCES(1, 1)
