// upr_qp3_launch.h -- launcher of one instantiation of the production QP kernel.  Declared for upr_api.hip; DEFINED (with
// the kernel) where UPR_QP3_LAUNCH_IMPL is set: upr_qp3_inst.hip, one part of upr_qp3_list.h per translation unit, or upr_api.hip
// itself in -DUPR_MONOLITHIC (experiment) builds.
// Returns 0, -1 (the working set exceeds the 160 KiB of LDS of a CU) or the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>

#include "upr_qp3.h"
#include "upr_qp3_list.h"

template <class C>
int upr_qp3_launch(hipStream_t stream, int B, const upr_qp_args& A);

#if defined(UPR_QP3_LAUNCH_IMPL) || defined(UPR_MONOLITHIC)
template <class C>
int upr_qp3_launch(hipStream_t stream, int B, const upr_qp_args& A) {
    const size_t lds = (size_t)upr_qp3_lds<C>::total * sizeof(double);
    if (lds > 160 * 1024) return -1;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)upr_qp3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((upr_qp3_kernel<C>), dim3(B), dim3(C::NT), lds, stream, A);
    return (int)hipGetLastError();
}
#endif

#ifndef UPR_MONOLITHIC
// every listed instantiation lives in another translation unit
#define UPR_X_(a, b, c, e, n, rows, sf, dense) extern template int upr_qp3_launch<upr_qp3_cfg<a, b, c, e, n, 256, rows, sf, dense>>(hipStream_t, int, const upr_qp_args&);
UPR_QP3_EXTRA(UPR_X_)
#undef UPR_X_
#define UPR_Y_(nt, rows) extern template int upr_qp3_launch<upr_qp3_cfg<9, 1, 4, 3, 20, nt, rows>>(hipStream_t, int, const upr_qp_args&);
UPR_QP3_HEADLINE(UPR_Y_)
#undef UPR_Y_
#endif
