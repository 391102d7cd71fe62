// k_pw_bf16.hip - the pointwise GEMM kernels of k_pw_impl.h for precision 1 (0 fp32, 1 bf16, 2 fp8 operands).
#include "k_pw_impl.h"
template void launch_pw_prec<1>(const PwArgs&, hipStream_t);
