/*
 * ppp_window_decl.h -- what the host side of the engine needs of the window path (ppp_window.h): the launch arguments, the
 * plan's constants and LDS sizes, and the kernels' declarations.  The kernels themselves are compiled in their own translation
 * unit (ppp_window.hip), so that the engine and the window kernels build side by side (the slice kernel alone is 8 x 100 KB of
 * ISA) and a change on one side does not recompile the other.
 */
#pragma once
#include "ppp_kernels.h"

/* DevMeta.win_flag: why this pass must be repeated on the slab-index path (not an error of the input) */
enum { WIN_FLAG_OVERFLOW = 1,  /* a window or its left side holds more points than the plan's LDS capacity      */
       WIN_FLAG_REACH = 2,     /* a nearest-neighbour ball or a normal neighbourhood reaches beyond its window  */
       WIN_FLAG_STALE = 4 };   /* bounds or slice walk of this pass differ from the plan the launches were sized by */
__device__ inline void win_flag(DevMeta *m, int why) { atomicOr(&m->win_flag, why); }

#define WIN_CLASSES 5
#ifndef WIN_CNT_STRIDE
#define WIN_CNT_STRIDE 1 /* ints between two windows' counters (a 128-byte line each, stride 32, changed nothing: 10 M points / 1024
                            windows 98 .. 107 us packed, 116 padded -- the scatter's bill is its 16-byte stores, not the atomics) */
#endif
#ifndef WIN_EMAX
#define WIN_EMAX 8 /* staged points per thread at most (capw <= WIN_EMAX * blockDim) */
#endif
#ifndef WIN_CE
#define WIN_CE 4 /* left points (pairing candidates) per thread at most (cap_el <= WIN_CE * blockDim) */
#endif

struct WinArgs {
    DevMeta *m;
    DevParams P;
    const float *X, *Y, *Z;
    const int *idmap;
    int n;
    const float *plan_px; /* the plan's slice positions (host walk over the cached bounds) */
    int S, sb, se, first_kept, nkept;
    float pad, px0, inv_step, y0, yscale; /* bucket(y) = (int)((y - y0) * yscale), clamped to [0, NBc) */
    float plan_mn[3], plan_mx[3];
    int plan_nvalid;
    int capw, cap_el, NB, NBc, stride, W_cap, node_cap;
    int rec_lds; /* waypoint records in the slice workgroup's LDS (the pairing scratch the knots leave free): waypoints per word row; 0: in the waypoints' global slots (wps_rec) */
    int g_scatter, g_slice, g_finish; /* workgroups of this workpiece per launch */
    int finish; /* 0: a slice-range handle stops after HandEyeTransform (the list is compacted only) */
    int *win_cnt;
    float4 *win_pts;
    MinMaxPart *win_part;
    float *px, *lo, *hi;
    float *node_x, *node_y, *node_z;
    int *node_start, *node_cnt, *band_cnt;
    int *wp_cnt, *wp_off, *tail;
    float4 *wps_xyz, *wps_normal;
    float4 *wps_rec; /* per waypoint slot, 4 x float4: what the searches of a waypoint leave for its pose (covariance sums, count, nearest point, sample) */
    int *wps_nn;
    float *wps_pre;
    float *wp_pre, *wp_smooth, *wp_out, *out2;
    int out2_cap;
    DevMeta *meta_host; /* pinned host memory: the last workgroup of the finish launch leaves the meta block there (no copy command behind a pass) */
    int *fin_ticket;    /* arrivals of that launch's workgroups (cleared by the last) */
    const PlanAuto *plan_rec; /* device record of the cloud's conversion pass (bounds, walk length, pad), or NULL: what the checker compares the pass with instead of plan_mn / plan_mx / plan_nvalid */
};

__host__ __device__ inline size_t win_slice_lds_bytes(int capw, int cap_el, int NB)
{
    return (size_t)capw * 16 + (((size_t)NB + 1) * 4 + 15) / 16 * 16 + (size_t)cap_el * 16 + 16;
}

#ifndef WSC_T
#define WSC_T 1024
#endif
/* STAGED (large clouds): the workgroup's kept points leave through LDS in window order, so that a wave stores runs of
   consecutive 16-byte pieces instead of 64 pieces in 64 different lines.  With thousands of workgroups' partial lines in
   flight the L2 no longer merges the pieces of a line before it evicts it: 10 M points wrote 229 MB for 107 MB of points
   (2 M points: 42 for 21), against 1.09 x at 1 M points, where the plain form stays. */
__host__ __device__ inline size_t win_scatter_lds_bytes(int S, int ppt, int threads, bool staged)
{
    return staged ? (size_t)12 * S + 16 + (size_t)18 * ppt * threads : (size_t)8 * S;
}
#ifndef WSL_T
#define WSL_T 1024
#endif
#define WIN_FIN_GROUPS 32 /* first-level arrival counters of the finish launch (win_publish_meta) */
#define WIN_S_MAX 8192 /* slices of a plan on this path (plane table and counters of the scatter, offsets of the finish live in LDS) */
#define WIN_AUTO_SCAP 4096 /* slices the LDS counters of this form have room for */

/* ---- the launches (defined in ppp_window.h, instantiated in ppp_window.hip) ---- */
template <int PPT, bool STAGED> __global__ void k_win_scatter(WinArgs A);
template <int PPT, bool STAGED> __global__ void k_win_scatter_b(const WinArgs *__restrict__ mem);
template <int TMAX> __global__ void k_win_slice(WinArgs A);
template <int TMAX> __global__ void k_win_slice_b(const WinArgs *__restrict__ mem);
__global__ void k_win_finish(WinArgs A);
__global__ void k_win_finish_b(const WinArgs *__restrict__ mem);
__global__ void k_collect_meta_win(const WinArgs *__restrict__ mem, int count, DevMeta *out);
__global__ void k_win_gather_stage(WinArgs A, float4 *wp_xyz, int *wp_nn, float4 *wp_normal);
template <bool IN_LDS>
__global__ void k_win_census(const float *__restrict__ X, int n, const float *__restrict__ px, int S, float px0, float inv_step, float pad,
                             int *cnt_win, int *cnt_el, int *cnt_er);
__global__ void k_win_census_auto(const float *__restrict__ X, int n, const float *px, const PlanAuto *plan, float inv_step, int *cnt, int *ticket,
                                  PlanAuto *plan_host, float *px_host, int *census_host);
