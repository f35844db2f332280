// K1 dispatcher + the 16x16x4-only instantiations; the kernel lives in dft2d_fwd_kernel.h
#include "dft2d_fwd_kernel.h"

namespace uno {

int launch_dft2d_fwd_r4(const Dft2dParams& p, hipStream_t s, int NT, int MT, int R4);      // dft2d_fwd_r4.hip

int launch_dft2d_fwd(const Dft2dParams& p, hipStream_t s) {
    const int NT = (p.m2 + 15) / 16, MT = (2 * p.m1 + 15) / 16;
    const bool vec = ((p.W - 1) >> 1) >= 16;       // at least one full chunk of 16 column pairs
    const int rem = p.m2 - 16 * (NT - 1);          // modes in the last 16-wide tile
    if (vec && rem <= 8 && NT <= 3 && MT <= 5) return launch_dft2d_fwd_r4(p, s, NT, MT, (rem + 3) / 4);
#define UNO_CASE(nt, mt) if (NT == nt && MT == mt) return vec ? launch_fwd_t<nt, mt, true, 0>(p, s) : launch_fwd_t<nt, mt, false, 0>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4) UNO_CASE(2, 5)
    UNO_CASE(3, 1) UNO_CASE(3, 2) UNO_CASE(3, 3) UNO_CASE(3, 4) UNO_CASE(3, 5)
#undef UNO_CASE
    set_error("dft2d_fwd: modes (%d, %d) exceed the compiled range (modes1 <= 40, modes2 <= 48)", p.m1, p.m2);
    return -2;
}

}  // namespace uno
