// K3 instantiations with 5..8 k-steps (modes2 17..32)
#include "dft2d_inv_kernel.h"

namespace uno {
int launch_dft2d_inv_b(const Dft2dParams& p, hipStream_t s) { return dispatch_inv_range<5, 8>(p, s); }
}  // namespace uno
