/*
 * ppp_window.hip -- the kernels of the window path (ppp_window.h) as a translation unit of their own: the engine
 * (ppp_engine.hip) sees their declarations (ppp_window_decl.h) and launches them; the instantiations it launches are listed
 * here.  Compiled with the engine's flags (-ffp-contract=off: one rounding per written operation).
 */
#include <hip/hip_runtime.h>
#define PPP_KERNELS_FOREIGN /* ppp_kernels.h: types and device helpers only -- its kernels belong to the engine's translation unit */
#include "ppp_window.h"

template __global__ void k_win_scatter<2, false>(WinArgs);
template __global__ void k_win_scatter<4, false>(WinArgs);
template __global__ void k_win_scatter<8, false>(WinArgs);
template __global__ void k_win_scatter<4, true>(WinArgs);
template __global__ void k_win_scatter<8, true>(WinArgs);
template __global__ void k_win_scatter_b<4, false>(const WinArgs *__restrict__);
template __global__ void k_win_scatter_b<8, false>(const WinArgs *__restrict__);
template __global__ void k_win_scatter_b<4, true>(const WinArgs *__restrict__);
template __global__ void k_win_scatter_b<8, true>(const WinArgs *__restrict__);
template __global__ void k_win_slice<256>(WinArgs);
template __global__ void k_win_slice<512>(WinArgs);
template __global__ void k_win_slice<768>(WinArgs);
template __global__ void k_win_slice<1024>(WinArgs);
template __global__ void k_win_slice_b<256>(const WinArgs *__restrict__);
template __global__ void k_win_slice_b<512>(const WinArgs *__restrict__);
template __global__ void k_win_slice_b<768>(const WinArgs *__restrict__);
template __global__ void k_win_slice_b<1024>(const WinArgs *__restrict__);
template __global__ void k_win_census<true>(const float *__restrict__, int, const float *__restrict__, int, float, float, float, int *, int *, int *);
template __global__ void k_win_census<false>(const float *__restrict__, int, const float *__restrict__, int, float, float, float, int *, int *, int *);
