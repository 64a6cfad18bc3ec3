/* stand-in for <cuda_fp16.h> on the host, see ngp_cuda_on_host.h (TEST INFRASTRUCTURE) */
#include "ngp_cuda_on_host.h"
