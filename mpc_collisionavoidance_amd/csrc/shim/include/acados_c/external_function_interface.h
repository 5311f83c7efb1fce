/* forwarding header of the usvmpc acados shim: the node includes this path (nmpc_guidance_ca1.cpp:18-26) */
#include "acados_solver_usv_model_guidance_ca1.h"
