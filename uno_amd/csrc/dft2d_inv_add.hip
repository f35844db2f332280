// K3-A dispatcher: K3 full-tile form + up-sampled addend (dft2d_inv_add_kernel.h), k-step counts 1..6 of the row stage (modes2 <= 24) x 1..7 of the column stage (modes1 <= 27)
#include "dft2d_inv_add_kernel.h"

namespace uno {

bool dft2d_inv_add_applies(const Dft2dParams& p) {
    if (!inv_add_shape_ok(p)) return false;
    InvGeometry ft;
    return inv_ft_geometry(p, (p.m2 + 3) / 4, &ft);
}

int launch_dft2d_inv_add(const Dft2dParams& p, hipStream_t s) {
    if (!p.add_src || !p.add_p0 || !p.add_rowop || !p.add_v0 || !p.add_colop) { set_error("dft2d_inv_add: missing addend tables"); return -1; }
    if (!inv_add_shape_ok(p)) { set_error("dft2d_inv_add: shape outside the compiled range (%dx%d, modes %d, %d)", p.H, p.W, p.m1, p.m2); return -2; }
    const int KS = (p.m2 + 3) / 4, KSK = (p.m1 + 4) >> 2;
#define UNO_CASE(ks, kc) if (KS == ks && KSK == kc) return launch_inv_add_t<ks, kc>(p, s);
#define UNO_ROW(ks) UNO_CASE(ks, 1) UNO_CASE(ks, 2) UNO_CASE(ks, 3) UNO_CASE(ks, 4) UNO_CASE(ks, 5) UNO_CASE(ks, 6) UNO_CASE(ks, 7)
    UNO_ROW(1) UNO_ROW(2) UNO_ROW(3) UNO_ROW(4) UNO_ROW(5) UNO_ROW(6)
#undef UNO_ROW
#undef UNO_CASE
    set_error("dft2d_inv_add: modes (%d, %d) outside the compiled range", p.m1, p.m2);
    return -2;
}

}  // namespace uno
