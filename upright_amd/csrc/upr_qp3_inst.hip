// upr_qp3_inst.hip -- the instantiations of the production QP kernel, one PART of upr_qp3_list.h per translation unit:
//     hipcc -c -DUPR_QP3_PART=k upr_qp3_inst.hip -o qp3_part_k.o        (k = 0 .. UPR_QP3_NPARTS - 1; __graft_entry__.build())
// Compiled without UPR_QP3_PART it is empty.
#define UPR_QP3_LAUNCH_IMPL
#include "upr_common.h"
#include "upr_kin.h"
#include "upr_qp.h"
#include "upr_qp2.h"
#include "upr_qp3.h"
#include "upr_qp3_launch.h"

#ifdef UPR_QP3_PART
#define UPR_X_(a, b, c, e, n, rows, sf, dense) template int upr_qp3_launch<upr_qp3_cfg<a, b, c, e, n, 256, rows, sf, dense>>(hipStream_t, int, const upr_qp_args&);
#define UPR_Y_(nt, rows) template int upr_qp3_launch<upr_qp3_cfg<9, 1, 4, 3, 20, nt, rows>>(hipStream_t, int, const upr_qp_args&);
#if UPR_QP3_PART == 0
UPR_QP3_PART0(UPR_Y_)
#elif UPR_QP3_PART == 1
UPR_QP3_PART1(UPR_X_)
#elif UPR_QP3_PART == 2
UPR_QP3_PART2(UPR_X_)
#elif UPR_QP3_PART == 3
UPR_QP3_PART3(UPR_X_)
#elif UPR_QP3_PART == 4
UPR_QP3_PART4(UPR_X_)
#elif UPR_QP3_PART == 5
UPR_QP3_PART5(UPR_X_)
#elif UPR_QP3_PART == 6
UPR_QP3_PART6(UPR_X_)
#else
#error "UPR_QP3_PART out of range (upr_qp3_list.h: UPR_QP3_NPARTS)"
#endif
#endif
