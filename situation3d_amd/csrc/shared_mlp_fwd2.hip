// shared_mlp_fwd2.hip -- mlp_layer_fwd_kernel's instances with 64 output channels per workgroup (shared_mlp_fwd.h)
#include "shared_mlp_fwd.h"

#ifndef SIG3D_MLP_TIMING   // (with the phase timing compiled in, shared_mlp.hip holds every instance itself)
SIG3D_MLP_FWD_INSTANCES(2, 0)
SIG3D_MLP_FWD_INSTANCES(2, 1)
#endif
