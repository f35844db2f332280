// K1 instantiations whose last mode tile runs on 4x4x1 MFMAs (R4 = 1 or 2 groups of four modes)
#include "dft2d_fwd_kernel.h"

namespace uno {

int launch_dft2d_fwd_r4(const Dft2dParams& p, hipStream_t s, int NT, int MT, int R4) {
#define UNO_CASE(nt, mt) \
    if (NT == nt && MT == mt) return R4 == 1 ? launch_fwd_t<nt, mt, true, 1>(p, s) : launch_fwd_t<nt, mt, true, 2>(p, s);
    UNO_CASE(1, 1) UNO_CASE(1, 2) UNO_CASE(1, 3) UNO_CASE(1, 4) UNO_CASE(1, 5)
    UNO_CASE(2, 1) UNO_CASE(2, 2) UNO_CASE(2, 3) UNO_CASE(2, 4) UNO_CASE(2, 5)
    UNO_CASE(3, 1) UNO_CASE(3, 2) UNO_CASE(3, 3) UNO_CASE(3, 4) UNO_CASE(3, 5)
#undef UNO_CASE
    set_error("dft2d_fwd_r4: unsupported tile configuration (%d, %d)", NT, MT);
    return -2;
}

}  // namespace uno
