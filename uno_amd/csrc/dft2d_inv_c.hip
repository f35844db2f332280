// K3 instantiations with 9..12 k-steps (modes2 33..48)
#include "dft2d_inv_kernel.h"

namespace uno {
int launch_dft2d_inv_c(const Dft2dParams& p, hipStream_t s) { return dispatch_inv_range<9, 12>(p, s); }
}  // namespace uno
