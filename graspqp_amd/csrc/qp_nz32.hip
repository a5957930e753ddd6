// NZ = 32 instantiation of the box-QP kernels.
#include "qp_kernels.h"
GQ_DEFINE_QP_NZ(32)
