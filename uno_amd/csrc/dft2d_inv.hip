// K3 dispatcher + the instantiations with 1..4 k-steps (modes2 <= 16); the kernel lives in dft2d_inv_kernel.h
#include "dft2d_inv_kernel.h"

namespace uno {

int launch_dft2d_inv_b(const Dft2dParams& p, hipStream_t s);      // dft2d_inv_b.hip: modes2 17..32
int launch_dft2d_inv_c(const Dft2dParams& p, hipStream_t s);      // dft2d_inv_c.hip: modes2 33..48

int launch_dft2d_inv(const Dft2dParams& p, hipStream_t s) {
    const int KS = (p.m2 + 3) / 4, JT = (2 * p.m1 + 15) / 16;
    if (KS > 12 || JT > 5) {
        set_error("dft2d_inv: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
        return -2;
    }
    if (KS <= 4) return dispatch_inv_range<1, 4>(p, s);
    return KS <= 8 ? launch_dft2d_inv_b(p, s) : launch_dft2d_inv_c(p, s);
}

}  // namespace uno
