/*
 * ppp_kernels.h -- gfx950 kernels of the polishing-path hot path.
 *
 * Data layout in HBM (DESIGN.md "Data layout"):
 *   X,Y,Z        float[N]   cloud in original index order (after the x1000), NaN = dropped point
 *   sorted4      float4[N]  {x,y,z,bits(idx)} grouped by x-slab, ascending (y, idx) inside a slab
 *   slab_start   int[B+1]   CSR offsets of the slabs; slab_xmin/xmax float[B] exact x bounds
 *   px,lo,hi     float[S]   plane x and PassThrough limits of every slice
 *   node_y,z     float[..]  spline knots of every slice (bump-allocated segments)
 *   wp_*         per-waypoint stage buffers, W x 6 float output list
 *
 * No MFMA anywhere: every kernel is a scan / bin / gather / sort bounded by HBM or
 * LDS bandwidth and by dependent-load latency.
 */
#pragma once
/* The kernels of this header belong to the engine's translation unit.  A translation unit that needs the header's types and device
   helpers only (ppp_window.hip) defines PPP_KERNELS_FOREIGN: the non-template kernels are templates there and, never launched,
   never instantiated (a static kernel would still be compiled and emitted: 47 functions in that translation unit's code object). */
#ifdef PPP_KERNELS_FOREIGN
#define PPP_KERNEL template <typename PPP_NEVER_ = void> __global__
#else
#define PPP_KERNEL __global__
#endif
#include "ppp_device.h"

struct DevMeta {
    u32 mn_ord[3], mx_ord[3];
    float mn[3], mx[3];
    int n_valid;
    int S, first_kept, nkept, W;
    int err, err_slice;
    int sweeps, any_short, rpy_oob;
    int node_cursor;
    int smooth_done;
    int emit_ticket;         /* tiles that have emitted (only counted when the list is finished in order, App. B.6) */
    int big_slabs, big_slices;   /* work lists of the LDS-overflow fallback kernels */
    unsigned long long arena_cursor; /* bump allocator of the global arena those kernels use */
    int B;
    float slab_x0, slab_invw;
    int api_cnt, api_flag;
    int sb, se;              /* slice range of this handle: [sb, se) */
    float incl_lo, incl_hi;  /* x interval of the points this handle indexes */
    int n_sorted;            /* points in the slab index */
    float ytab_scale;        /* y-bucket table of the slabs: bucket(y) = (int)((y - mn[1]) * ytab_scale), clamped to [0, YTB) */
    int win_flag;            /* window path (ppp_window.h): non-zero = this pass must be repeated on the slab-index path */
};

struct DevParams {
    double tool_radius, path_resolution, rpy_resolution, trim;
    float ee_length, normal_radius;
    float handeye[6];
    float viewpoint[3];
    int change_range, pairing, walk, drop_ends, smooth, smooth_max_sweeps;
    int slice_begin, slice_end, ranged;
    float incl_lo, incl_hi;
    float nn_hint2; /* (a few mean point spacings)^2: first search bound of the waypoints' 1-NN queries */
    int knots_on_plane; /* every knot's x is its plane's x (no dynamic adjustment): the y -> x spline is that constant */
    /* a slice-range handle streams only its part of the cloud: the whole cloud's bounds and point count (what getMinMax3D
       gives the reference, and what the walk and the slab grid are built from) come with the plan */
    int bounds_given, g_nvalid;
    float g_mn[3], g_mx[3];
    /* the slab index is being built on demand (an API mirror) behind a finished window-path pass: the set-up keeps that
       pass's run state (W, errors, knot cursor) in the meta block instead of resetting it */
    int keep_run_state;
};

#define SCAT_COARSE_SHIFT 6 /* two-pass scatter of large clouds: 64 neighbouring slabs form a coarse bin */
enum { DERR_NONE = 0, DERR_SLICE = 1, DERR_CAPACITY = 2, DERR_DOMAIN = 3, DERR_QUERY = 4, DERR_MARGIN = 5 };

__device__ inline void set_err(DevMeta *m, int code, int slice)
{
    atomicCAS(&m->err, 0, code);
    if (slice >= 0) atomicMin(&m->err_slice, slice);
}

__device__ inline int slab_of(const DevMeta *m, float x)
{
    int b = (int)((x - m->slab_x0) * m->slab_invw);
    b = b < 0 ? 0 : b;
    return b >= m->B ? m->B - 1 : b;
}
__device__ inline int idx_of(const float4 &p) { return __float_as_int(p.w); }

/* Per-slab y-bucket table (k_slab_sort writes it, the waypoint searches of k_pose read it): entry q of slab b is the number
   of the slab's points -- sorted by (y, index) -- whose bucket is below q, for YTB uniform buckets over the cloud's y range;
   entry YTB is the slab's population.  bucket() is monotone in y, so the lower bound of any y lies inside its own bucket:
   a search is one table look-up plus a binary search over that bucket's few points instead of over the whole slab
   (ten dependent reads become three).  Exactness never depends on the table: it only narrows the first interval. */
#ifndef YTB
#define YTB 512
#endif
__device__ inline int ytab_bucket(const DevMeta *m, float y)
{
    int q = (int)((y - m->mn[1]) * m->ytab_scale);
    q = q < 0 ? 0 : q;
    return q >= YTB ? YTB - 1 : q;
}

/* ------------------------------------------------------------------ */
/* a1: constructor scaling (path_slicing_alg.cpp:14-24).  Non-finite   */
/* points are canonicalised to NaN: PassThrough, getMinMax3D and the   */
/* kd-tree all skip them.                 */
/* ------------------------------------------------------------------ */
PPP_KERNEL void k_ingest(const char *raw, size_t stride, int n, int scale, float *X, float *Y, float *Z)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = (const float *)(raw + (size_t)i * stride);
    float x = p[0], y = p[1], z = p[2];
    if (scale) { x *= 1000; y *= 1000; z *= 1000; }
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) { x = y = z = __int_as_float(0x7fc00000); }
    X[i] = x; Y[i] = y; Z[i] = z;
}

/* sort keys: [63] side, [62:31] order-preserving bits of y, [30:0] position */
#define YK_MAKE(y, pos) (((u64)f2ord(y) << 31) | (u64)(u32)(pos))
#define YK_Y(k) ((u32)(((k) >> 31) & 0xffffffffu))
#define YK_POS(k) ((int)((k) & 0x7fffffffu))

struct MinMaxPart { float mn[3], mx[3]; int cnt, pad; };

/* a2: pcl::getMinMax3D (path_slicing_alg.cpp:303, path_dynamic_alg.cpp:345).  One partial per
   workgroup, no atomics (same-address atomics serialise at ~11 ns each on this part). */
/* HIST: the same pass also builds the x-slab histogram (LDS-privatised, flushed with one global
   atomic per non-empty slab and workgroup).  The slab grid (x0, invw, B) comes from the bounds the
   host cached when the cloud was set; k_setup stores the same grid in DevMeta for every later
   kernel, so the index is consistent whatever this launch's own min/max turn out to be. */
__device__ inline int slab_of_grid(float x, float x0, float invw, int B)
{
    int b = (int)((x - x0) * invw);
    b = b < 0 ? 0 : b;
    return b >= B ? B - 1 : b;
}
#ifndef MM_T
#define MM_T 512
#endif
#ifndef MM_UNROLL
#define MM_UNROLL 2
#endif
/* bx / gx: this workgroup's index and the number of workgroups working on THIS cloud (the batched launch runs the
   workgroups of many clouds side by side: blockIdx.y = cloud) */
template <bool HIST>
__device__ __forceinline__ void minmax_body(const float *__restrict__ X, const float *__restrict__ Y,
                                            const float *__restrict__ Z, int n, MinMaxPart *part, float x0, float invw,
                                            int B, int *slab_cnt, float xlo, float xhi, int *cursor, const int bx, const int gx)
{
    extern __shared__ __attribute__((aligned(16))) int s_hist[];
    if (HIST) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) s_hist[b] = 0;
        /* the one-level scatter reserves its runs on zero-based per-slab cursors: cleared here, a launch ahead */
        if (cursor) for (int b = bx * (int)blockDim.x + (int)threadIdx.x; b < B; b += gx * (int)blockDim.x) cursor[b] = 0;
        __syncthreads();
    }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    const int n4 = n >> 2;
    const float4 *X4 = (const float4 *)X, *Y4 = (const float4 *)Y, *Z4 = (const float4 *)Z;
    const int stride = gx * (int)blockDim.x;
    for (int i = bx * blockDim.x + threadIdx.x; i < n4; i += MM_UNROLL * stride) { /* MM_UNROLL groups of four points: their reads travel together */
        float4 xq[MM_UNROLL], yq[MM_UNROLL], zq[MM_UNROLL];
#pragma unroll
        for (int u = 0; u < MM_UNROLL; ++u) {
            const int j = i + u * stride;
            if (j < n4) { xq[u] = X4[j]; yq[u] = Y4[j]; zq[u] = Z4[j]; }
            else xq[u] = yq[u] = zq[u] = make_float4(NAN, NAN, NAN, NAN);
        }
#pragma unroll
        for (int u = 0; u < MM_UNROLL; ++u) {
            const float xs[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w}, ys[4] = {yq[u].x, yq[u].y, yq[u].z, yq[u].w}, zs[4] = {zq[u].x, zq[u].y, zq[u].z, zq[u].w};
            for (int k = 0; k < 4; ++k) {
                if (xs[k] == xs[k]) {
                    mn[0] = fminf(mn[0], xs[k]); mx[0] = fmaxf(mx[0], xs[k]);
                    mn[1] = fminf(mn[1], ys[k]); mx[1] = fmaxf(mx[1], ys[k]);
                    mn[2] = fminf(mn[2], zs[k]); mx[2] = fmaxf(mx[2], zs[k]);
                    cnt++;
                    if (HIST && xs[k] >= xlo && xs[k] <= xhi) atomicAdd(&s_hist[slab_of_grid(xs[k], x0, invw, B)], 1);
                }
            }
        }
    }
    if (bx == 0 && threadIdx.x < (n & 3)) {
        int i = (n4 << 2) + threadIdx.x;
        float x = X[i];
        if (x == x) {
            float y = Y[i], z = Z[i];
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
            cnt++;
            if (HIST && x >= xlo && x <= xhi) atomicAdd(&s_hist[slab_of_grid(x, x0, invw, B)], 1);
        }
    }
    if (HIST) {
        __syncthreads();
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int c = s_hist[b];
            if (c) atomicAdd(&slab_cnt[b], c);
        }
    }
    __shared__ float s_mn[3][MM_T / 64], s_mx[3][MM_T / 64];
    __shared__ int s_cnt[MM_T / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_cnt[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPart r;
        r.cnt = 0; r.pad = 0;
        for (int w = 0; w < MM_T / 64; ++w) r.cnt += s_cnt[w];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < MM_T / 64; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            r.mn[d] = a; r.mx[d] = b;
        }
        part[bx] = r;
    }
}
template <bool HIST>
__global__ void __launch_bounds__(MM_T) k_minmax(const float *__restrict__ X, const float *__restrict__ Y,
                                                const float *__restrict__ Z, int n, MinMaxPart *part, float x0, float invw,
                                                int B, int *slab_cnt, float xlo, float xhi, int *cursor)
{
    minmax_body<HIST>(X, Y, Z, n, part, x0, invw, B, slab_cnt, xlo, xhi, cursor, blockIdx.x, gridDim.x);
}

/* Device form of ppp_slice_walk for one thread: identical values, but without a data dependent
   branch per element (a float compare feeding a scalar branch costs ~60 cycles per iteration on
   one lane).  Integer walks are closed forms; float walks keep the reference's sequential
   float accumulation (each add rounds) in a predicated loop whose trip count comes from a
   double estimate, with the literal loop as the tail.  front[] is LDS scratch. */
__device__ inline int slice_walk_device(int walk, float min_x, float max_x, double toolRadius, float *px, int cap, float *front,
                                        int front_cap)
{
    const int step = (int)(toolRadius * 2);
    if (step <= 0 || !(min_x <= max_x)) return 0;
    const float fstep = (float)step;
    auto up_arm = [&](float loc, int k0) { /* while (loc < max_x) { px[k++] = loc; loc += step; } */
        int k = k0;
        double est = ((double)max_x - (double)loc) / (double)step;
        int n_est = est > 0 ? (int)fmin(est + 2.0, 4.0e6) : 1;
        for (int it = 0; it < n_est; ++it) {
            const bool v = loc < max_x;
            if (v && k < cap) px[k] = loc;
            k += v ? 1 : 0;
            loc += fstep;
        }
        while (loc < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = loc; k++; loc += fstep; }
        return k;
    };
    switch (walk) {
    case 0: {
        float loc = (min_x + max_x) / 2 - fstep;
        int nfront = 0;
        double est = ((double)loc - (double)min_x) / (double)step;
        int n_est = est > 0 ? (int)fmin(est + 2.0, 4.0e6) : 1;
        for (int it = 0; it < n_est; ++it) {
            const bool v = loc > min_x;
            if (v && nfront < front_cap) front[nfront] = loc;
            nfront += v ? 1 : 0;
            loc -= fstep;
        }
        while (loc > min_x && nfront < PPP_WALK_HARD_MAX) { if (nfront < front_cap) front[nfront] = loc; nfront++; loc -= fstep; }
        if (nfront <= front_cap) {
            for (int i = 0; i < nfront; ++i) if (nfront - 1 - i < cap) px[nfront - 1 - i] = front[i];
        } else { /* LDS scratch too small: redo the arm writing straight to its final place */
            loc = (min_x + max_x) / 2 - fstep;
            for (int i = nfront - 1; i >= 0 && loc > min_x; --i) { if (i < cap) px[i] = loc; loc -= fstep; }
        }
        return up_arm((min_x + max_x) / 2, nfront);
    }
    case 1: {
        const int imin = (int)min_x, imax = (int)max_x;
        const int c = (imax + imin) / 2;
        int nfront = 0, nback = 0;
        if (imax > c - step && c - step > imin) nfront = (c - imin - 1) / step;
        if (imax > c + step && c + step > imin) nback = (imax - c - 1) / step;
        for (int i = 0; i < nfront; ++i) if (i < cap) px[i] = (float)(c - (nfront - i) * step);
        if (nfront < cap) px[nfront] = (min_x + max_x) / 2;
        for (int j = 0; j < nback; ++j) if (nfront + 1 + j < cap) px[nfront + 1 + j] = (float)(c + (j + 1) * step);
        return nfront + 1 + nback;
    }
    case 2: {
        int loc = (int)(min_x + toolRadius);
        int k = 0;
        if (k < cap) px[k] = (float)loc;
        k++;
        loc += step;
        while (loc < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = (float)loc; k++; loc += step; }
        return k;
    }
    case 3: return up_arm((float)(min_x + toolRadius), 0);
    case 4: { float x = min_x; x += (float)(step / 2); return up_arm(x, 0); }
    }
    return 0;
}

/* k_ingest and k_minmax<false> as one pass over a new cloud: the converted coordinates are written and reduced in the same
   registers (ppp_set_cloud*: a planner fed with a new cloud every time pays this launch on its critical path) */
/* What the last workgroup of k_ingest_minmax leaves for the plan of a new cloud (device copy + pinned host copy): the reduced
   bounds and, when the window path may apply, the slice walk from them (written to the plan's plane table) with the window pad --
   so that the census of the windows can follow in the same stream without the host in between */
struct PlanAuto {
    MinMaxPart fin;
    int S;       /* slices of the walk (-1: no walk asked for) */
    float pad;   /* plan_window's pad from these bounds          */
    int census;  /* k_win_census_auto: 1 = the counters behind this record are the census of (S, pad), 0 = it did not run */
    int reserved;
};
struct PlanAutoArgs {
    int *ticket;          /* zero between launches */
    PlanAuto *dev, *host; /* host: pinned            */
    int walk;             /* -1: bounds only         */
    double tool_radius;
    float normal_radius;
    float *px;            /* the plan's plane table (device) */
    int px_cap;
    float *px_host;       /* pinned: a copy of that table for the host (nullptr: the census that follows hands it over) */
};
PPP_KERNEL void __launch_bounds__(MM_T) k_ingest_minmax(const char *__restrict__ raw, size_t stride, int n, int scale, float *__restrict__ X,
                                                       float *__restrict__ Y, float *__restrict__ Z, MinMaxPart *part, PlanAutoArgs PA)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    const int step = (int)(gridDim.x * blockDim.x);
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * step) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) { /* four points' reads in flight */
            const int i = i0 + u * step;
            if (i < n) { const float *p = (const float *)(raw + (size_t)i * stride); v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * step;
            if (i >= n) break;
            float x = v[u][0], y = v[u][1], z = v[u][2];
            if (scale) { x *= 1000; y *= 1000; z *= 1000; }
            if (!(isfinite(x) && isfinite(y) && isfinite(z))) { x = y = z = __int_as_float(0x7fc00000); }
            X[i] = x; Y[i] = y; Z[i] = z;
            if (x == x) {
                mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
                mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
                mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
                cnt++;
            }
        }
    }
    __shared__ float s_mn[3][MM_T / 64], s_mx[3][MM_T / 64];
    __shared__ int s_cnt[MM_T / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_cnt[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPart r;
        r.cnt = 0; r.pad = 0;
        for (int w = 0; w < MM_T / 64; ++w) r.cnt += s_cnt[w];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < MM_T / 64; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            r.mn[d] = a; r.mx[d] = b;
        }
        /* published word by word with agent-scope stores (write-through), the ticket drawn once they are acknowledged: a release
           fence here would write back the whole L2 -- the 12 bytes a point this pass has just converted -- once per workgroup
           (measured: 31 us for 1 M points against 16 us for the two separate kernels) */
        int *dst = (int *)&part[blockIdx.x];
        const int *src = (const int *)&r;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(MinMaxPart) / 4); ++k) __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!PA.ticket) return;
    /* the last workgroup to get here reduces the partials (the host's own sequential min / max / sum give the same values) */
    __shared__ int s_last, s_nfront, s_cc, s_S;
    __shared__ float s_mid;
    __shared__ float s_front[1024];
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0); /* vmcnt(0): the stores above have been acknowledged */
        s_last = __hip_atomic_fetch_add(PA.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    for (int d = 0; d < 3; ++d) { mn[d] = INFINITY; mx[d] = -INFINITY; }
    cnt = 0;
    for (int q = threadIdx.x; q < (int)gridDim.x; q += blockDim.x) {
        MinMaxPart r;
        {
            const int *src = (const int *)&part[q];
            int *dst = (int *)&r;
#pragma unroll
            for (int k = 0; k < (int)(sizeof(MinMaxPart) / 4); ++k) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cnt += r.cnt;
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], r.mn[d]); mx[d] = fmaxf(mx[d], r.mx[d]); }
    }
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    __syncthreads();
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_cnt[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        PlanAuto r;
        r.fin.cnt = 0; r.fin.pad = 0;
        for (int w = 0; w < MM_T / 64; ++w) r.fin.cnt += s_cnt[w];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < MM_T / 64; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            r.fin.mn[d] = a; r.fin.mx[d] = b;
        }
        if (!r.fin.cnt) for (int d = 0; d < 3; ++d) { r.fin.mn[d] = 3.402823466e+38f; r.fin.mx[d] = -3.402823466e+38f; }
        r.S = -1; r.pad = 0.f; r.census = 0; r.reserved = 0;
        s_nfront = -1;
        const int istep = (int)(PA.tool_radius * 2);
        if (PA.walk >= 0 && r.fin.cnt > 0) {
            if (PA.walk == 1 && istep > 0 && r.fin.mn[0] <= r.fin.mx[0]) { /* closed form per index: every thread fills its entries below (as setup_body) */
                const int imin = (int)r.fin.mn[0], imax = (int)r.fin.mx[0];
                const int cc = (imax + imin) / 2;
                int nfront = 0, nback = 0;
                if (imax > cc - istep && cc - istep > imin) nfront = (cc - imin - 1) / istep;
                if (imax > cc + istep && cc + istep > imin) nback = (imax - cc - 1) / istep;
                r.S = nfront + 1 + nback;
                s_nfront = nfront; s_cc = cc; s_S = r.S; s_mid = (r.fin.mn[0] + r.fin.mx[0]) / 2;
            } else r.S = slice_walk_device(PA.walk, r.fin.mn[0], r.fin.mx[0], PA.tool_radius, PA.px, PA.px_cap, s_front, 1024);
            /* plan_window's pad, in its arithmetic */
            const double rx = (double)r.fin.mx[0] - r.fin.mn[0], ry = (double)r.fin.mx[1] - r.fin.mn[1];
            const double area = rx * ry;
            const double spacing = area > 0 ? sqrt(area / r.fin.cnt) : 1.0;
            r.pad = (float)fmax(3.0, (double)PA.normal_radius + fmax(1.5, spacing));
        }
        *PA.dev = r;
        *PA.host = r;
        *PA.ticket = 0;
    }
    __syncthreads();
    if (s_nfront >= 0) {
        const int istep = (int)(PA.tool_radius * 2);
        for (int i = threadIdx.x; i < s_S && i < PA.px_cap; i += blockDim.x) {
            const float v = i < s_nfront ? (float)(s_cc - (s_nfront - i) * istep) : (i == s_nfront ? s_mid : (float)(s_cc + (i - s_nfront) * istep));
            PA.px[i] = v;
            if (PA.px_host) PA.px_host[i] = v;
        }
    } else if (PA.px_host && PA.walk >= 0) { /* (the sequential walks: one thread wrote the table above) */
        const int Sw = PA.dev->S;
        for (int i = threadIdx.x; i < Sw && i < PA.px_cap; i += blockDim.x) PA.px_host[i] = PA.px[i];
    }
}

/* Resets the per-run state, finishes a2, runs a3 (slice walk + the PassThrough limits of
   rangedX_index(int), path_slicing_alg.cpp:152-158,247) and clears the slab histogram. */
#ifndef SETUP_T
#define SETUP_T 1024
#endif
__device__ __forceinline__ void setup_body(DevMeta *m, const DevParams &P, const MinMaxPart *__restrict__ part, int nparts,
                                           float *px, float *lo, float *hi, int S_cap, int B, int *slab_cnt, float slab_x0,
                                           float slab_invw, int *slab_start, int *slab_cursor, int *coarse_cursor, const bool own_launch = true)
{   /* own_launch = false: this is the extra workgroup of the one-level scatter launch -- the scatter workgroups beside it read
       the histogram too (each scans it for itself), so it is left alone (k_slab_sort clears it), and they keep their own cursors */
    __shared__ int s_scan[17];
    __shared__ float s_mn[3][SETUP_T / 64], s_mx[3][SETUP_T / 64];
    __shared__ int s_cnt[SETUP_T / 64];
    __shared__ int s_S, s_total, s_nfront, s_c;
    __shared__ float s_mid;
    __shared__ float s_front[4096];
    __shared__ int s_cnts[8192]; /* the slab histogram (make_plan caps B at 8192) */
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    STAMP_BEGIN();
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        MinMaxPart r = part[i];
        cnt += r.cnt;
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], r.mn[d]); mx[d] = fmaxf(mx[d], r.mx[d]); }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_cnt[wid] = cnt; }
    STAMP(5, 0); /* partials */
    /* exclusive scan of the slab histogram (k_minmax<true> filled it) -> CSR offsets + scatter
       cursors; the histogram is cleared for the next run.  The counts come into LDS with coalesced loads (B <= 8192), are
       scanned there, and go out coalesced: a thread reading its own run of counts from global one after the other was 80 %
       of this kernel at 8192 slabs. */
    {
        for (int i = threadIdx.x; i < B; i += blockDim.x) s_cnts[i] = slab_cnt[i]; /* (loads only: they pipeline) */
        if (own_launch) for (int i = threadIdx.x; i < B; i += blockDim.x) slab_cnt[i] = 0;
        __syncthreads();
        const int per = (B + blockDim.x - 1) / blockDim.x;
        const int b0 = threadIdx.x * per;
        int sum = 0;
        for (int k = 0; k < per; ++k) if (b0 + k < B) sum += s_cnts[b0 + k];
        int total;
        int pre = block_exscan(sum, s_scan, &total);
        for (int k = 0; k < per; ++k) {
            if (b0 + k < B) { const int c = s_cnts[b0 + k]; s_cnts[b0 + k] = pre; pre += c; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < B; i += blockDim.x) {
            const int v = s_cnts[i];
            slab_start[i] = v;
            if (own_launch) slab_cursor[i] = v;
            if (coarse_cursor && (i & ((1 << SCAT_COARSE_SHIFT) - 1)) == 0) coarse_cursor[i >> SCAT_COARSE_SHIFT] = v;
        }
        if (threadIdx.x == 0) { slab_start[B] = total; s_total = total; }
    }
    __syncthreads();
    STAMP(5, 1); /* slab scan */
    if (threadIdx.x == 0) {
        int c = 0;
        for (int w = 0; w < SETUP_T / 64; ++w) c += s_cnt[w];
        DevMeta r; /* built in registers, stored once: no dependent global round trips */
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < SETUP_T / 64; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            /* getMinMax3D starts from +-FLT_MAX */
            r.mn[d] = c ? a : 3.402823466e+38f;
            r.mx[d] = c ? b : -3.402823466e+38f;
            if (P.bounds_given) { r.mn[d] = P.g_mn[d]; r.mx[d] = P.g_mx[d]; } /* this launch saw the handle's part only */
            r.mn_ord[d] = f2ord(r.mn[d]); r.mx_ord[d] = f2ord(r.mx[d]);
        }
        if (P.bounds_given) c = P.g_nvalid;
        r.n_valid = c;
        r.W = 0; r.err = 0; r.err_slice = 0x7fffffff; r.sweeps = 0; r.any_short = 0; r.rpy_oob = 0;
        r.node_cursor = 0; r.api_cnt = 0; r.api_flag = 0; r.smooth_done = -1; r.emit_ticket = 0;
        r.big_slabs = 0; r.big_slices = 0; r.arena_cursor = 0; r.win_flag = 0;
        if (P.keep_run_state) {
            r.W = m->W; r.err = m->err; r.err_slice = m->err_slice; r.any_short = m->any_short; r.rpy_oob = m->rpy_oob;
            r.node_cursor = m->node_cursor; r.smooth_done = m->smooth_done; r.win_flag = m->win_flag;
        }
        int S = 0;
        s_nfront = -1;
        const int istep = (int)(P.tool_radius * 2);
        if (c && P.walk == 1 /* centre-out integer walk */ && istep > 0 && r.mn[0] <= r.mx[0]) {
            /* the centre-out integer walk is a closed form per index: only the counts here, every thread fills its
               own entries below (one thread writing S values was two thirds of this kernel) */
            const int imin = (int)r.mn[0], imax = (int)r.mx[0];
            const int cc = (imax + imin) / 2;
            int nfront = 0, nback = 0;
            if (imax > cc - istep && cc - istep > imin) nfront = (cc - imin - 1) / istep;
            if (imax > cc + istep && cc + istep > imin) nback = (imax - cc - 1) / istep;
            S = nfront + 1 + nback;
            s_nfront = nfront; s_c = cc; s_mid = (r.mn[0] + r.mx[0]) / 2;
        } else if (c) S = slice_walk_device(P.walk, r.mn[0], r.mx[0], P.tool_radius, px, S_cap, s_front, 4096);
        if (S > S_cap) { r.err = DERR_CAPACITY; S = S_cap; }
        r.S = S;
        r.first_kept = P.drop_ends ? 1 : 0;
        int nk = P.drop_ends ? S - 2 : S;
        r.nkept = nk < 0 ? 0 : nk;
        r.B = B;
        r.slab_x0 = slab_x0;     /* the grid k_minmax<true> binned with */
        r.slab_invw = slab_invw;
        r.sb = P.slice_begin < 0 ? 0 : (P.slice_begin > S ? S : P.slice_begin);
        r.se = (P.slice_end <= 0 || P.slice_end > S) ? S : P.slice_end;
        r.incl_lo = P.incl_lo; r.incl_hi = P.incl_hi;
        r.n_sorted = s_total;
        { const float yr = r.mx[1] - r.mn[1]; r.ytab_scale = (c && yr > 0.f) ? (float)YTB / yr : 0.f; }
        *m = r;
        s_S = S;
    }
    __syncthreads();
    STAMP(5, 2); /* bounds + slice walk (one thread) */
    const int S = s_S;
    const int nfront = s_nfront, istep = (int)(P.tool_radius * 2);
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        if (nfront >= 0) px[s] = s < nfront ? (float)(s_c - (nfront - s) * istep) : (s == nfront ? s_mid : (float)(s_c + (s - nfront) * istep));
        int position = (int)px[s];
        lo[s] = (float)(-2 + position);
        hi[s] = (float)(2 + position);
    }
    STAMP(5, 3); /* band limits */
}

/* ------------------------------------------------------------------ */
/* Slice binning, generalised: every point goes to one x-slab (the      */
/* replacement for kdtree.setInputCloud + the S PassThrough scans).     */
/* ------------------------------------------------------------------ */
#ifndef SCAT_T
#define SCAT_T 1024
#endif
/* LEVEL 0: the cloud straight into its slabs (one pass; the writes of a workgroup are runs of chunk / B points).
   Large clouds have thousands of slabs and those runs shrink to a point or two -- 16-byte writes scattered over the
   whole array.  They go in two passes instead: LEVEL 1 bins into COARSE groups of 64 neighbouring slabs (runs of
   hundreds of points, written to `out4` = the sorted4 buffer used as scratch), LEVEL 2 reads that -- each chunk now
   touches a few dozen slabs -- and bins into the slabs proper.  Same result as LEVEL 0 up to the order inside a slab,
   which k_slab_sort fixes anyway. */
/* Every thread keeps its PPT points in registers between the counting and the writing pass: the input is read once. */
template <int LEVEL, int PPT>
__device__ __forceinline__ void slab_scatter_body(const float *__restrict__ X, const float *__restrict__ Y,
                                                  const float *__restrict__ Z, const float4 *__restrict__ in4, int n,
                                                  const DevMeta *m, int *cursor, float4 *out4, const int *__restrict__ idmap, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) int s_hist[];
    const int B = LEVEL == 1 ? ((m->B + (1 << SCAT_COARSE_SHIFT) - 1) >> SCAT_COARSE_SHIFT) : m->B;
    const float xlo = m->incl_lo, xhi = m->incl_hi;
    if (LEVEL == 2) n = m->n_sorted; /* the scratch holds the kept points only */
    auto bin = [&](float x) { return LEVEL == 1 ? (slab_of(m, x) >> SCAT_COARSE_SHIFT) : slab_of(m, x); };
    STAMP_BEGIN();
    /* this thread's points: i0 + threadIdx.x + k * blockDim.x (coalesced), requested before anything else */
    const int i0 = bx * (PPT * (int)blockDim.x);
    float4 p[PPT];
    int pb[PPT]; /* bin, or -1: not mine / dropped */
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = i0 + threadIdx.x + k * (int)blockDim.x;
        pb[k] = -1;
        if (i < n) {
            if (LEVEL == 2) p[k] = in4[i];
            else p[k] = make_float4(X[i], Y[i], Z[i], __int_as_float(idmap ? idmap[i] : i)); /* idmap: cloud indices of a part */
        } else p[k] = make_float4(NAN, 0.f, 0.f, 0.f);
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    STAMP(2, 0); /* zero */
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const float x = p[k].x;
        const bool keep = LEVEL == 2 ? (i0 + (int)threadIdx.x + k * (int)blockDim.x < n) : (x >= xlo && x <= xhi); /* NaN fails both */
        if (keep) { pb[k] = bin(x); atomicAdd(&s_hist[pb[k]], 1); }
    }
    __syncthreads();
    STAMP(2, 1); /* count */
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        int c = s_hist[b];
        if (c) s_hist[b] = atomicAdd(&cursor[b], c);
    }
    __syncthreads();
    STAMP(2, 2); /* reserve (global atomics) */
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (pb[k] >= 0) out4[atomicAdd(&s_hist[pb[k]], 1)] = p[k];
    STAMP(2, 3); /* scatter */
}

/* One workgroup per slab: exact bucket sort on (y, cloud index) -- buckets are uniform in y over
   the cloud's y range --, then the exact x bounds of the slab.  ARENA = false: keys and histogram
   in LDS, slabs larger than `cap` are appended to a work list; ARENA = true: second pass over that
   list with the same code on a bump-allocated global arena (any slab size, slower). */
#ifndef SORT_T
#define SORT_T 512
#endif
template <bool ARENA>
__device__ __forceinline__ void slab_sort_body(const float4 *__restrict__ unsorted4, const int *__restrict__ slab_start,
                                               float4 *sorted4, float *slab_xmin, float *slab_xmax, DevMeta *m, int cap,
                                               int *big_list, char *arena, unsigned long long arena_cap, int *ytab, int *slab_cnt, const int bx)
{   /* (bx: slab number; a slice-range handle launches only the slabs of its interval, offset by the first one) */
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ float s_mn[SORT_T / 64], s_mx[SORT_T / 64];
    __shared__ int s_scr[17];
    __shared__ unsigned long long s_off;
    int b = bx;
    if (ARENA) {
        if (b >= m->big_slabs) return;
        b = big_list[b];
    }
    const int s0 = slab_start[b], c = slab_start[b + 1] - s0;
    if (!ARENA && slab_cnt && threadIdx.x == 0) slab_cnt[b] = 0; /* the histogram is used up (k_setup clears it itself when it has its own launch) */
    u64 *key;
    int *hist;
    int NB;
    if (ARENA) {
        NB = next_pow2(c);
        const unsigned long long need = (unsigned long long)c * 8 + (unsigned long long)(NB + 1) * 4 + 16;
        if (threadIdx.x == 0) s_off = atomicAdd(&m->arena_cursor, (need + 15) & ~15ull);
        __syncthreads();
        if (s_off + need > arena_cap) { if (threadIdx.x == 0) set_err(m, DERR_CAPACITY, -1); return; }
        key = (u64 *)(arena + s_off);
        hist = (int *)(key + c);
    } else {
        if (c > cap) {
            if (threadIdx.x == 0) big_list[atomicAdd(&m->big_slabs, 1)] = b;
            /* Until the arena pass has sorted it the slab holds its points in arrival order, an empty y-bucket row and no x
               bounds.  The kernels behind this launch in the stream (whole-cloud normals, the Area2Cloud searches of the dynamic
               adjustment) walk the whole index before the host knows that it is incomplete and runs the pass again: whatever a
               slab's places and row held from an earlier plan -- cloud indices, offsets -- sent them outside their buffers
               (found by a randomised case: an aligned 123 x 124 plate, brute pairing, dynamic adjustment). */
            for (int i = threadIdx.x; i < c; i += blockDim.x) sorted4[s0 + i] = unsorted4[s0 + i];
            if (ytab) for (int q = threadIdx.x; q <= YTB; q += blockDim.x) ytab[(size_t)b * (YTB + 1) + q] = 0;
            if (threadIdx.x == 0) { slab_xmin[b] = -INFINITY; slab_xmax[b] = INFINITY; }
            return;
        }
        NB = min(cap, next_pow2(c));
        key = (u64 *)s_raw;
        hist = (int *)(key + cap);
    }
    float mn = INFINITY, mx = -INFINITY;
    if (c > 0) {
        const float y0 = m->mn[1];
        const float yr = m->mx[1] - y0;
        const float scale = yr > 0.f ? (float)NB / yr : 0.f;
        const float4 *src = unsorted4 + s0;
        auto gen = [&](int i) { return YK_MAKE(src[i].y, i); };
        auto bucket = [&](u64 k) {
            int q = (int)((ord2f(YK_Y(k)) - y0) * scale);
            return q < 0 ? 0 : (q >= NB ? NB - 1 : q);
        };
        auto less = [&](u64 a, u64 bb) {
            u32 ya = YK_Y(a), yb = YK_Y(bb);
            if (ya != yb) return ya < yb;
            return idx_of(src[YK_POS(a)]) < idx_of(src[YK_POS(bb)]); /* equal y: cloud index */
        };
        if (!ARENA && c <= 8 * (int)blockDim.x) block_bucket_sort_cached<8>(key, c, hist, NB, s_scr, gen, bucket, less); /* y read once */
        else block_bucket_sort(key, c, hist, NB, s_scr, gen, bucket, less);
        int *tab = ytab ? ytab + (size_t)b * (YTB + 1) : nullptr;
        for (int i = threadIdx.x; i < c; i += blockDim.x) {
            float4 p = src[YK_POS(key[i])];
            sorted4[s0 + i] = p;
            mn = fminf(mn, p.x); mx = fmaxf(mx, p.x);
            if (tab) { /* the table entries between the previous point's bucket and this point's: i points lie below them */
                const int bi = ytab_bucket(m, p.y);
                const int bp = i > 0 ? ytab_bucket(m, ord2f(YK_Y(key[i - 1]))) : -1;
                for (int q = bp + 1; q <= bi; ++q) tab[q] = i;
                if (i == c - 1) for (int q = bi + 1; q <= YTB; ++q) tab[q] = c;
            }
        }
    } else if (ytab) {
        for (int q = threadIdx.x; q <= YTB; q += blockDim.x) ytab[(size_t)b * (YTB + 1) + q] = 0;
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); } /* launched with 256 or SORT_T threads */
        slab_xmin[b] = mn; slab_xmax[b] = mx;
    }
}

/* ------------------------------------------------------------------ */
/* Per-slice kernel: rangedX_index + insert_point + map flattening.     */
/* One workgroup per slice, the band lives in LDS.                      */
/* ------------------------------------------------------------------ */
struct SliceLds {
    float4 *a4;   /* band points                                  */
    u64 *keys;    /* sort keys                                    */
    float *candz; /* z of node candidates / pair scratch (brute)  */
    u16 *elpos, *erpos, *rstar, *lstar;
};
__host__ __device__ inline size_t slice_lds_bytes(int capb) { return (size_t)capb * (16 + 8 + 4 + 2 * 4); }
__device__ inline SliceLds carve_slice_lds(char *raw, int capb)
{
    SliceLds L;
    L.a4 = (float4 *)raw;
    L.keys = (u64 *)(L.a4 + capb);
    L.candz = (float *)(L.keys + capb);
    L.elpos = (u16 *)(L.candz + capb);
    L.erpos = L.elpos + capb;
    L.rstar = L.erpos + capb;
    L.lstar = L.rstar + capb;
    return L;
}

/* Gathers the PassThrough band [lo,hi] from the slabs into L.a4 and orders it by ascending
   cloud index (L.keys[j] & 0xffff = position in a4 of the j-th index).  Returns the count, or
   -1 when the band does not fit. */
__device__ inline int band_gather_sorted(const SliceLds &L, int capb, const float4 *__restrict__ sorted4,
                                         const int *__restrict__ slab_start, const DevMeta *m, float lo, float hi,
                                         int *s_n)
{
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    int n = 0;
    if (lo <= hi && m->n_valid > 0) {
        const int b0 = slab_of(m, lo), b1 = slab_of(m, hi);
        const int i0 = slab_start[b0], i1 = slab_start[b1 + 1];
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            float4 p = sorted4[i];
            if (!(p.x < lo || p.x > hi)) {
                int slot = atomicAdd(s_n, 1);
                if (slot < capb) L.a4[slot] = p;
            }
        }
    }
    __syncthreads();
    n = *s_n;
    if (n > capb) return -1;
    /* ascending cloud index: a bucket sort on the index (the candidate-z area is free here and takes the histogram) -- the 64-bit
       bitonic network this replaces was ~60 us of the kernel for a 1300-point band */
    __shared__ int s_imax, s_bscr[17];
    if (threadIdx.x == 0) s_imax = 0;
    __syncthreads();
    {
        int mx = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) mx = max(mx, idx_of(L.a4[i]));
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx) atomicMax(&s_imax, mx);
    }
    __syncthreads();
    {
        int NB = next_pow2(max(n, 64));
        while (NB >= capb) NB >>= 1;
        const u64 span = (u64)(u32)s_imax + 1ull;
        auto gen = [&](int i) { return ((u64)(u32)idx_of(L.a4[i]) << 32) | (u64)(u32)i; };
        auto bucket = [&](u64 k) { return (int)(((k >> 32) * (u64)NB) / span); };
        auto less = [&](u64 a, u64 b) { return a < b; };
        block_bucket_sort(L.keys, n, (int *)L.candz, NB, s_bscr, gen, bucket, less);
    }
    return n;
}

/* insert_point on the band held in LDS, in the order given by L.keys (low 16 bits = position).
   kd flavour: path_slicing_alg.cpp:164-237.  brute flavour: Path_Generation.cpp:107-206.
   Output: *out_m nodes (ascending y, duplicates resolved last-writer-wins) in L.keys/L.candz
   order: node k has y = ord2f(keys[k] >> 32), z = candz[keys[k] & 0xffffffff].
   Returns m >= 0, or -1 (empty right side with a non-empty left side: the reference crashes). */
__device__ inline int insert_point_lds(const SliceLds &L, int n, float Px, int pairing, int *s_scr)
{
    /* --- side split, order preserving (path_slicing_alg.cpp:174-182) --- */
    __shared__ int s_nl, s_nr, s_np;
    if (threadIdx.x == 0) { s_nl = 0; s_nr = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += blockDim.x) {
        int j = base + threadIdx.x;
        int fl = 0, fr = 0;
        u16 pos = 0;
        if (j < n) {
            pos = (u16)(L.keys[j] & 0xffffu);
            float4 p = L.a4[pos];
            float distance2plane = (p.x - Px) * 1.f + (p.y - 0.f) * 0.f + (p.z - 0.f) * 0.f;
            fl = distance2plane > 0;
            fr = distance2plane < 0;
        }
        int tl, tr;
        int pl = block_exscan(fl, s_scr, &tl);
        int pr = block_exscan(fr, s_scr, &tr);
        int ol = s_nl, orr = s_nr;
        if (fl) L.elpos[ol + pl] = pos;
        if (fr) L.erpos[orr + pr] = pos;
        __syncthreads();
        if (threadIdx.x == 0) { s_nl = ol + tl; s_nr = orr + tr; }
        __syncthreads();
    }
    const int nEl = s_nl, nEr = s_nr;
    if (nEl == 0) return 0;
    if (nEr == 0) return -1;

    int ncand = 0;
    if (pairing == 0) {
        /* --- kd flavour: 1-NN across the plane and back (path_slicing_alg.cpp:194-210) --- */
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) {
            const float4 q = L.a4[L.elpos[i]];
            float best = INFINITY; int br = 0;
            for (int j = 0; j < nEr; ++j) {
                const float4 c = L.a4[L.erpos[j]];
                float d = dist2_flann(q.x, q.y, q.z, c.x, c.y, c.z);
                if (d < best) { best = d; br = j; } /* ties: lowest index */
            }
            const float4 R = L.a4[L.erpos[br]];
            best = INFINITY; int bl = 0;
            for (int k = 0; k < nEl; ++k) {
                const float4 c = L.a4[L.elpos[k]];
                float d = dist2_flann(R.x, R.y, R.z, c.x, c.y, c.z);
                if (d < best) { best = d; bl = k; }
            }
            L.rstar[i] = (u16)br;
            L.lstar[i] = (u16)bl;
        }
        __syncthreads();
        ncand = nEl;
        /* --- lerp onto the plane (path_slicing_alg.cpp:220-232) --- */
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) {
            const float4 R = L.a4[L.erpos[L.rstar[i]]];
            const float4 Lp = L.a4[L.elpos[L.lstar[i]]];
            float t = (Px - R.x) / (Lp.x - R.x);
            float y = R.y + t * (Lp.y - R.y);
            float z = R.z + t * (Lp.z - R.z);
            if (y == 0.f) y = 0.f; /* -0.0 and +0.0 are one std::map key */
            L.candz[i] = z;
            L.keys[i] = ((u64)f2ord(y) << 32) | (u32)i;
        }
    } else {
        /* --- brute flavour: the two argmins do not depend on the flags, so they run in
               parallel; the greedy flag walk stays sequential (Path_Generation.cpp:137-179) --- */
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) {
            const float4 q = L.a4[L.elpos[i]];
            float best = INFINITY; int bj = 0;
            for (int j = 0; j < nEr; ++j) {
                const float4 c = L.a4[L.erpos[j]];
                float d = norm_eigen3(q.x - c.x, q.y - c.y, q.z - c.z);
                if (d <= best) { best = d; bj = j; } /* compare[norm] = j : last j wins a tie */
            }
            L.rstar[i] = (u16)bj;
        }
        for (int j = threadIdx.x; j < nEr; j += blockDim.x) {
            const float4 q = L.a4[L.erpos[j]];
            float best = INFINITY; int bk = 0;
            for (int k = 0; k < nEl; ++k) {
                const float4 c = L.a4[L.elpos[k]];
                float d = norm_eigen3(q.x - c.x, q.y - c.y, q.z - c.z);
                if (d <= best) { best = d; bk = k; }
            }
            L.lstar[j] = (u16)bk;
        }
        __syncthreads();
        u16 *pairL = (u16 *)L.candz;        /* left_pair  (as El positions) */
        u16 *pairR = pairL + nEl;            /* right_pair (as Er positions); |right| <= nEl */
        /* the walk's records made beforehand, in parallel -- step i: j*(i) | k*(j*) << 16 -- and the two flag arrays as bytes: a step
           is one record (requested a step ahead) and three flag bytes instead of a chain of table look-ups and 64-bit
           read-modify-writes (1 M points / 256 slices: 650 of them a slice, the kernel 409 -> see DESIGN.md A.2) */
        u32 *trip = (u32 *)L.keys;           /* keys: 8 bytes per band point; the first half */
        unsigned char *lf = (unsigned char *)(trip + nEl), *rf = lf + nEl; /* El_flag / Er_flag: nEl + nEr <= n bytes of the rest */
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) { const int j = L.rstar[i]; trip[i] = (u32)j | ((u32)L.lstar[j] << 16); }
        for (int i = threadIdx.x; i < (n + 3) / 4; i += blockDim.x) ((u32 *)lf)[i] = 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            int nl = 0, nr = 0;
            u32 nx = trip[0];
            for (int i = 0; i < nEl; ++i) {
                const u32 tr = nx;
                if (i + 1 < nEl) nx = trip[i + 1];
                const int j = (int)(tr & 0xffffu), k = (int)(tr >> 16);
                const int fi = lf[i], fj = rf[j], fk = lf[k];
                if (fi) continue;                   /* if (El_flag[i] == 0) */
                if (fj) continue;                   /* Er_flag[compare.begin()->second] != 0: continue */
                pairR[nr++] = (u16)j; rf[j] = 1;
                if (!fk) { pairL[nl++] = (u16)k; lf[k] = 1; }
            }
            s_np = nl; /* the reference loops i < left_pair.size() (Path_Generation.cpp:189) */
        }
        __syncthreads();
        ncand = s_np;
        /* read pairs to registers, then overwrite the scratch with keys / z */
        float ys[16], zs[16]; /* ncand <= capb <= 4096, blockDim >= 256 (k_insert_api's) -> <= 16 per thread */
        int cntl = 0;
        for (int i = threadIdx.x; i < ncand; i += blockDim.x) {
            const float4 R = L.a4[L.erpos[pairR[i]]];
            const float4 Lp = L.a4[L.elpos[pairL[i]]];
            float t = (Px - R.x) / (Lp.x - R.x);
            float y = R.y + t * (Lp.y - R.y);
            float z = R.z + t * (Lp.z - R.z);
            if (y == 0.f) y = 0.f;
            ys[cntl] = y; zs[cntl] = z; cntl++;
        }
        __syncthreads();
        cntl = 0;
        for (int i = threadIdx.x; i < ncand; i += blockDim.x) {
            L.candz[i] = zs[cntl];
            L.keys[i] = ((u64)f2ord(ys[cntl]) << 32) | (u32)i;
            cntl++;
        }
    }
    __syncthreads();
    /* --- std::map semantics: ascending key, last writer wins: a bucket sort over the candidates' own y range (keys made once, held in
           registers between its two passes; the side tables are used up and take the histogram) --- */
    {
        __shared__ u32 s_ylo, s_yhi;
        if (threadIdx.x == 0) { s_ylo = 0xffffffffu; s_yhi = 0u; }
        __syncthreads();
        u32 lo = 0xffffffffu, hi = 0u;
        for (int i = threadIdx.x; i < ncand; i += blockDim.x) { const u32 y = (u32)(L.keys[i] >> 32); lo = min(lo, y); hi = max(hi, y); }
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (u32)__shfl_xor((int)lo, o, 64)); hi = max(hi, (u32)__shfl_xor((int)hi, o, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_ylo, lo); atomicMax(&s_yhi, hi); }
        __syncthreads();
        int NB = next_pow2(max(ncand, 64));
        const int room = (int)(((char *)(L.lstar) - (char *)(L.elpos)) / 3); /* the four u16 side tables = 2 ints per band point: lstar - elpos is three of them, 6 bytes a point */
        while (NB >= room) NB >>= 1;
        const float y0 = ord2f(s_ylo), y1 = ord2f(s_yhi);
        const float scale = (ncand > 0 && y1 > y0) ? (float)NB / (y1 - y0) : 0.f;
        auto gen = [&](int i) { return L.keys[i]; };
        auto bucket = [&](u64 k) { const int q = (int)((ord2f((u32)(k >> 32)) - y0) * scale); return q < 0 ? 0 : (q >= NB ? NB - 1 : q); };
        auto less = [&](u64 a, u64 b) { return a < b; };
        block_bucket_sort_cached<16>(L.keys, ncand, (int *)L.elpos, NB, s_scr, gen, bucket, less);
    }
    return ncand;
}

/* keeps the last entry of every equal-y run; writes node_y/node_z; returns m */
__device__ inline int flatten_nodes(const SliceLds &L, int ncand, float *out_y, float *out_z, int out_cap, int *s_scr)
{
    __shared__ int s_m;
    if (threadIdx.x == 0) s_m = 0;
    __syncthreads();
    for (int base = 0; base < ncand; base += blockDim.x) {
        int j = base + threadIdx.x;
        int keep = 0;
        u64 k = 0;
        if (j < ncand) {
            k = L.keys[j];
            keep = (j == ncand - 1) || ((u32)(L.keys[j + 1] >> 32) != (u32)(k >> 32));
        }
        int tot;
        int pre = block_exscan(keep, s_scr, &tot);
        int o = s_m;
        if (keep && o + pre < out_cap) {
            out_y[o + pre] = ord2f((u32)(k >> 32));
            out_z[o + pre] = L.candz[(u32)k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_m = o + tot;
        __syncthreads();
    }
    return s_m;
}

#ifndef K_SLICE_T
#define K_SLICE_T 512 /* threads of the generic insert_point kernel (brute flavour, API mirror): 1 M points / 256 slices 663 (256 threads) -> 409 us; 1024: 404 */
#endif
PPP_KERNEL void __launch_bounds__(K_SLICE_T) k_slice(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                               DevMeta *m, const float *__restrict__ px, const float *__restrict__ lo,
                                               const float *__restrict__ hi, int pairing, int capb, float *node_x, float *node_y,
                                               float *node_z, int node_cap, int *node_start, int *node_cnt, int *band_cnt, int *big_list)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_n, s_base;
    const int s = blockIdx.x;
    if (s >= m->S) return;
    if (s < m->sb || s >= m->se) { /* another handle's slice */
        if (threadIdx.x == 0) { node_start[s] = 0; node_cnt[s] = 0; band_cnt[s] = 0; }
        return;
    }
    SliceLds L = carve_slice_lds(s_raw, capb);
    const float Px = px[s];
    int n = band_gather_sorted(L, capb, sorted4, slab_start, m, lo[s], hi[s], &s_n);
    if (n < 0) { /* does not fit LDS: leave it to the arena pass (k_slice_brute_arena) */
        if (threadIdx.x == 0) { big_list[atomicAdd(&m->big_slices, 1)] = s; node_start[s] = 0; node_cnt[s] = 0; band_cnt[s] = s_n; }
        return;
    }
    if (threadIdx.x == 0) band_cnt[s] = n;
    int ncand = insert_point_lds(L, n, Px, pairing, s_scr);
    if (ncand < 0) {
        if (threadIdx.x == 0) { set_err(m, DERR_SLICE, s); node_start[s] = 0; node_cnt[s] = 0; }
        return;
    }
    /* count distinct keys first to reserve the segment, then write */
    int mcount = 0;
    for (int j = threadIdx.x; j < ncand; j += blockDim.x)
        mcount += (j == ncand - 1) || ((u32)(L.keys[j + 1] >> 32) != (u32)(L.keys[j] >> 32));
    int tot;
    block_exscan(mcount, s_scr, &tot);
    if (threadIdx.x == 0) {
        int base = atomicAdd(&m->node_cursor, tot);
        if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, s); base = 0; tot = 0; }
        s_base = base;
        node_start[s] = base;
        node_cnt[s] = tot;
        if (tot < 3) set_err(m, DERR_SLICE, s); /* gsl_spline_alloc needs >= 3 knots */
    }
    __syncthreads();
    if (node_cnt[s] == 0 && tot != 0) return;
    flatten_nodes(L, ncand, node_y + s_base, node_z + s_base, tot, s_scr);
    for (int i = threadIdx.x; i < tot; i += blockDim.x) node_x[s_base + i] = Px; /* insert_cloud.points[i].x = PlanePoint[0] */
}

/* insert_point of v1 (Path_Generation.cpp:107-206, either pairing) for a band that does not fit LDS: the same steps as
   insert_point_lds on a segment of the arena, 32-bit positions, one workgroup per listed slice.  The greedy flag walk is
   one thread over global memory -- this is the rare path of a very dense band, exact before fast. */
__host__ __device__ inline size_t slice_brute_bytes(size_t n)
{   /* a4 16 + keys 8 + el er rstar lstar pairL pairR 24 + hist 4 (NB <= n) + flags */
    return n * 52 + (n / 4) + 256;
}
PPP_KERNEL void __launch_bounds__(1024) k_slice_brute_arena(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                                            DevMeta *m, const float *__restrict__ px, const float *__restrict__ lo,
                                                            const float *__restrict__ hi, int pairing, int ncloud, float *node_x, float *node_y,
                                                            float *node_z, int node_cap, int *node_start, int *node_cnt, const int *band_cnt,
                                                            const int *big_list, char *arena, unsigned long long arena_cap)
{
    __shared__ int s_scr[17];
    __shared__ int s_n, s_nl, s_nr, s_np, s_base, s_m;
    __shared__ unsigned long long s_off;
    if ((int)blockIdx.x >= m->big_slices) return;
    const int s = big_list[blockIdx.x];
    const size_t cap = (size_t)band_cnt[s];
    const unsigned long long need = slice_brute_bytes(cap);
    if (threadIdx.x == 0) { s_off = atomicAdd(&m->arena_cursor, (need + 15) & ~15ull); s_n = 0; s_nl = 0; s_nr = 0; s_m = 0; }
    __syncthreads();
    if (s_off + need > arena_cap) {
        if (threadIdx.x == 0) { set_err(m, DERR_CAPACITY, s); node_start[s] = 0; node_cnt[s] = 0; }
        return;
    }
    char *mem = arena + s_off;
    float4 *a4 = (float4 *)mem;
    u64 *keys = (u64 *)(a4 + cap);
    int *el = (int *)(keys + cap), *er = el + cap, *rstar = er + cap, *lstar = rstar + cap, *pairL = lstar + cap, *pairR = pairL + cap;
    int *hist = pairR + cap;                    /* cap + 1 ints at most (NB <= cap / 2 .. cap) */
    u64 *flags = (u64 *)(((uintptr_t)(hist + cap + 2) + 7) & ~(uintptr_t)7);
    const float Px = px[s], blo = lo[s], bhi = hi[s];
    /* rangedX_index: the PassThrough band, every point of it */
    if (blo <= bhi && m->n_valid > 0) {
        const int b0 = slab_of(m, blo), b1 = slab_of(m, bhi);
        const int i0 = slab_start[b0], i1 = slab_start[b1 + 1];
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            const float4 p = sorted4[i];
            if (!(p.x < blo || p.x > bhi)) {
                const int slot = atomicAdd(&s_n, 1);
                if ((size_t)slot < cap) a4[slot] = p;
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if ((size_t)n > cap) { if (threadIdx.x == 0) { set_err(m, DERR_CAPACITY, s); node_start[s] = 0; node_cnt[s] = 0; } return; }
    int NB = next_pow2(max(n, 64)) >> 1;
    {   /* ascending cloud index, as PassThrough returns the indices */
        const double scale = (double)NB / (double)max(ncloud, 1);
        auto gen = [&](int i) { return ((u64)(u32)idx_of(a4[i]) << 32) | (u32)i; };
        auto bucket = [&](u64 k) { int q = (int)((double)(u32)(k >> 32) * scale); return q < 0 ? 0 : (q >= NB ? NB - 1 : q); };
        auto less = [&](u64 a, u64 b) { return a < b; };
        block_bucket_sort(keys, n, hist, NB, s_scr, gen, bucket, less);
    }
    /* side split, order preserving (Path_Generation.cpp:115-127) */
    for (int base = 0; base < n; base += blockDim.x) {
        const int j = base + threadIdx.x;
        int fl = 0, fr = 0, pos = 0;
        if (j < n) {
            pos = (int)(u32)keys[j];
            const float4 p = a4[pos];
            const float distance2plane = (p.x - Px) * 1.f + (p.y - 0.f) * 0.f + (p.z - 0.f) * 0.f;
            fl = distance2plane > 0;
            fr = distance2plane < 0;
        }
        int tl, tr;
        const int pl = block_exscan(fl, s_scr, &tl);
        const int pr = block_exscan(fr, s_scr, &tr);
        const int ol = s_nl, orr = s_nr;
        if (fl) el[ol + pl] = pos;
        if (fr) er[orr + pr] = pos;
        __syncthreads();
        if (threadIdx.x == 0) { s_nl = ol + tl; s_nr = orr + tr; }
        __syncthreads();
    }
    const int nEl = s_nl, nEr = s_nr;
    if (nEl == 0 || nEr == 0) { /* empty map -> fewer than 3 knots; empty right side: the reference crashes */
        if (threadIdx.x == 0) { set_err(m, DERR_SLICE, s); node_start[s] = 0; node_cnt[s] = 0; }
        return;
    }
    int ncand = 0;
    float *cy = (float *)rstar, *cz = (float *)lstar; /* reused once the pairs are read */
    if (pairing == 0) {
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) {
            const float4 q = a4[el[i]];
            float best = INFINITY; int br = 0;
            for (int j = 0; j < nEr; ++j) {
                const float4 c = a4[er[j]];
                const float d = dist2_flann(q.x, q.y, q.z, c.x, c.y, c.z);
                if (d < best) { best = d; br = j; }
            }
            const float4 R = a4[er[br]];
            best = INFINITY; int bl = 0;
            for (int k = 0; k < nEl; ++k) {
                const float4 c = a4[el[k]];
                const float d = dist2_flann(R.x, R.y, R.z, c.x, c.y, c.z);
                if (d < best) { best = d; bl = k; }
            }
            pairR[i] = br; pairL[i] = bl;
        }
        ncand = nEl;
    } else {
        for (int i = threadIdx.x; i < nEl; i += blockDim.x) {
            const float4 q = a4[el[i]];
            float best = INFINITY; int bj = 0;
            for (int j = 0; j < nEr; ++j) {
                const float4 c = a4[er[j]];
                const float d = norm_eigen3(q.x - c.x, q.y - c.y, q.z - c.z);
                if (d <= best) { best = d; bj = j; } /* compare[norm] = j : last j wins a tie */
            }
            rstar[i] = bj;
        }
        for (int j = threadIdx.x; j < nEr; j += blockDim.x) {
            const float4 q = a4[er[j]];
            float best = INFINITY; int bk = 0;
            for (int k = 0; k < nEl; ++k) {
                const float4 c = a4[el[k]];
                const float d = norm_eigen3(q.x - c.x, q.y - c.y, q.z - c.z);
                if (d <= best) { best = d; bk = k; }
            }
            lstar[j] = bk;
        }
        const int wl = (nEl + 63) >> 6, wr = (nEr + 63) >> 6;
        for (int i = threadIdx.x; i < wl + wr; i += blockDim.x) flags[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) { /* the greedy flag walk (Path_Generation.cpp:137-179) */
            int nl = 0, nr = 0;
            for (int i = 0; i < nEl; ++i) {
                if ((flags[i >> 6] >> (i & 63)) & 1) continue;
                const int j = rstar[i];
                if ((flags[wl + (j >> 6)] >> (j & 63)) & 1) continue;
                pairR[nr++] = j;
                flags[wl + (j >> 6)] |= 1ull << (j & 63);
                const int k = lstar[j];
                if (!((flags[k >> 6] >> (k & 63)) & 1)) {
                    pairL[nl++] = k;
                    flags[k >> 6] |= 1ull << (k & 63);
                }
            }
            s_np = nl; /* the reference loops i < left_pair.size() (Path_Generation.cpp:189) */
        }
        __syncthreads();
        ncand = s_np;
    }
    __syncthreads();
    /* lerp onto the plane; the pairs are read before rstar / lstar become cy / cz */
    for (int base = 0; base < ncand; base += blockDim.x) {
        const int i = base + threadIdx.x;
        float y = 0.f, z = 0.f;
        if (i < ncand) {
            const float4 R = a4[er[pairR[i]]];
            const float4 Lp = a4[el[pairL[i]]];
            const float t = (Px - R.x) / (Lp.x - R.x);
            y = R.y + t * (Lp.y - R.y);
            z = R.z + t * (Lp.z - R.z);
            if (y == 0.f) y = 0.f; /* -0.0 and +0.0 are one std::map key */
        }
        __syncthreads(); /* (pairing 0 wrote pairR / pairL only; rstar / lstar are free either way once the walk is over) */
        if (i < ncand) { cy[i] = y; cz[i] = z; }
    }
    __syncthreads();
    {   /* std::map: ascending y, the last writer (highest i) of equal keys is the one kept */
        NB = next_pow2(max(ncand, 64)) >> 1;
        const float y0 = m->mn[1], yr = m->mx[1] - y0;
        const float scale = yr > 0.f ? (float)NB / yr : 0.f;
        auto gen = [&](int i) { return ((u64)f2ord(cy[i]) << 32) | (u32)i; };
        auto bucket = [&](u64 k) { int q = (int)((ord2f((u32)(k >> 32)) - y0) * scale); return q < 0 ? 0 : (q >= NB ? NB - 1 : q); };
        auto less = [&](u64 a, u64 b) { return a < b; };
        block_bucket_sort(keys, ncand, hist, NB, s_scr, gen, bucket, less);
    }
    int mcount = 0;
    for (int j = threadIdx.x; j < ncand; j += blockDim.x) mcount += (j == ncand - 1) || ((u32)(keys[j + 1] >> 32) != (u32)(keys[j] >> 32));
    int tot;
    block_exscan(mcount, s_scr, &tot);
    if (threadIdx.x == 0) {
        int base = atomicAdd(&m->node_cursor, tot);
        if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, s); base = 0; tot = 0; }
        s_base = base; s_np = tot;
        node_start[s] = base;
        node_cnt[s] = tot;
        if (tot < 3) set_err(m, DERR_SLICE, s);
    }
    __syncthreads();
    const int nk = s_np;
    if (nk == 0) return;
    for (int i = threadIdx.x; i < nk; i += blockDim.x) node_x[s_base + i] = Px;
    for (int base = 0; base < ncand; base += blockDim.x) {
        const int j = base + threadIdx.x;
        int keep = 0;
        u64 k = 0;
        if (j < ncand) { k = keys[j]; keep = (j == ncand - 1) || ((u32)(keys[j + 1] >> 32) != (u32)(k >> 32)); }
        int t2;
        const int pre = block_exscan(keep, s_scr, &t2);
        const int o = s_m;
        if (keep) { node_y[s_base + o + pre] = ord2f((u32)(k >> 32)); node_z[s_base + o + pre] = cz[(u32)k]; }
        __syncthreads();
        if (threadIdx.x == 0) s_m = o + t2;
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ */
/* kd flavour, fast path (path_slicing_alg.cpp:164-237).  The band is   */
/* sorted once by (side, y) in LDS; both nearest-neighbour queries of   */
/* every left point are then windowed scans around a binary search      */
/* (exact: a candidate is skipped only when dy*dy alone exceeds the     */
/* best distance).  El order only matters through the map's last-writer */
/* rule, which is carried as the query's cloud index in the sort key.   */
/* ------------------------------------------------------------------ */
/* waypoints of one path: the reference's `dy = miny + trim; while (dy < bigy - trim) { ...; dy += res; }`
   (path_translation_alg.cpp:158-166).  Closed form when the accumulated sums are exact, the loop otherwise. */
__device__ inline int sample_count(double miny, double bigy, double trim, double res, int cap)
{
    double dy = miny + trim;
    const double lim = bigy - trim;
    if (!(dy < lim)) return 0;
    const double est = (lim - dy) / res;
    if (est < 1.0e6 && sums_exact(dy, res, est + 2.0)) {
        int j = (int)floor(est) - 1; /* never above the count; dy + j * res is exact for these j */
        if (j < 0) j = 0;
        while (dy + (double)j * res < lim) ++j;
        return j > cap ? cap + 1 : j;
    }
    int cnt = 0;
    while (dy < lim && cnt <= cap) { cnt++; dy += res; }
    return cnt;
}

struct SliceKdMem {
    float4 *a4;  /* band points                                              (dead after the NN phase) */
    u64 *keys;   /* band sorted by (side, y): (side<<63) | YK_MAKE(y, slot)  (dead after the NN phase) */
    float *cy, *cz; /* node candidate of the i-th left point: y, z */
    int *cidx;      /* ... and the cloud index of that left point  */
    /* aliases: */
    int *hist_band; /* over cy..cidx while the band is sorted        */
    u64 *ckeys;     /* over a4 once the candidates exist             */
    int *hist_cand; /* behind ckeys                                   */
};
__host__ __device__ inline size_t slice_kd_bytes(size_t capb) { return capb * (16 + 8 + 12) + 64; }
__device__ inline SliceKdMem carve_slice_kd(char *raw, size_t capb)
{
    SliceKdMem L;
    L.a4 = (float4 *)raw;
    L.keys = (u64 *)(L.a4 + capb);
    L.cy = (float *)(L.keys + capb);
    L.cz = L.cy + capb;
    L.cidx = (int *)(L.cz + capb);
    L.hist_band = (int *)L.cy;            /* <= capb + 1 ints of the 3 capb available */
    L.ckeys = (u64 *)L.a4;                /* capb keys = half of a4                    */
    L.hist_cand = (int *)(L.ckeys + capb); /* <= capb + 1 ints, second half of a4 (+ pad) */
    return L;
}
#define KD_SIDE (1ull << 63)

/* nearest point of keys[a..b) (one side, ascending y) to q; ties -> lowest cloud index */
__device__ inline int nn_sorted_side(const u64 *keys, const float4 *a4, int a, int b, const float4 q)
{
    const u32 ty = f2ord(q.y);
    int lo = a, hi = b;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (YK_Y(keys[mid]) < ty) lo = mid + 1; else hi = mid;
    }
    float best = INFINITY;
    int bidx = 0x7fffffff, bj = a;
    /* candidates in order, four at a time: key -> point is two dependent LDS reads per candidate, eight such chains in a row
       per walk when taken one by one; the four keys and then the four points are independent reads (one past the candidate
       that ends the walk is harmless) */
    auto visit = [&](const float4 &c, int i) {
        const float dy = q.y - c.y;
        if (dy * dy > best) return false;
        const float d = dist2_flann(q.x, q.y, q.z, c.x, c.y, c.z);
        const int id = idx_of(c);
        if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bj = i; }
        return true;
    };
    for (int i = lo; i < b; i += 4) {
        const int e = b - 1;
        const u64 k0 = keys[i], k1 = keys[min(i + 1, e)], k2 = keys[min(i + 2, e)], k3 = keys[min(i + 3, e)];
        const float4 c0 = a4[YK_POS(k0)], c1 = a4[YK_POS(k1)], c2 = a4[YK_POS(k2)], c3 = a4[YK_POS(k3)];
        if (!visit(c0, i)) break;
        if (i + 1 > e || !visit(c1, i + 1)) break;
        if (i + 2 > e || !visit(c2, i + 2)) break;
        if (i + 3 > e || !visit(c3, i + 3)) break;
    }
    for (int i = lo - 1; i >= a; i -= 4) {
        const u64 k0 = keys[i], k1 = keys[max(i - 1, a)], k2 = keys[max(i - 2, a)], k3 = keys[max(i - 3, a)];
        const float4 c0 = a4[YK_POS(k0)], c1 = a4[YK_POS(k1)], c2 = a4[YK_POS(k2)], c3 = a4[YK_POS(k3)];
        if (!visit(c0, i)) break;
        if (i - 1 < a || !visit(c1, i - 1)) break;
        if (i - 2 < a || !visit(c2, i - 2)) break;
        if (i - 3 < a || !visit(c3, i - 3)) break;
    }
    return bj;
}

/* ARENA = false: the band lives in LDS (capb points); a slice whose band does not fit is appended
   to a work list.  ARENA = true: second pass over that list, same code on a global arena. */
#ifndef SLICE_KD_T
#define SLICE_KD_T 1024
#endif
template <bool ARENA>
__device__ __forceinline__ void slice_kd_body(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                              DevMeta *m, const float *__restrict__ px, const float *__restrict__ lo,
                                              const float *__restrict__ hi, int capb_lds, float *node_x, float *node_y,
                                              float *node_z, int node_cap, int *node_start, int *node_cnt, int *band_cnt, int *big_list,
                                              char *arena, unsigned long long arena_cap, double trim, double res, int W_cap, int *slice_wpcnt,
                                              const int bx)
{   /* slice_wpcnt (optional): the slice's waypoint count -- getPath's sampling loop over [first knot + trim, last knot - trim) --
       left for k_pose, whose every workgroup needs the counts of ALL slices for its offset in the list */
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_n, s_plane, s_ner, s_base, s_m;
    __shared__ unsigned long long s_off;
    int s = bx;
    if (ARENA) {
        if (s >= m->big_slices) return;
        s = big_list[s];
    } else {
        if (s >= m->S) return;
        if (s < m->sb || s >= m->se) { /* another handle's slice */
            if (threadIdx.x == 0) { node_start[s] = 0; node_cnt[s] = 0; band_cnt[s] = 0; if (slice_wpcnt) slice_wpcnt[s] = 0; }
            return;
        }
    }
    STAMP_BEGIN();
    const float Px = px[s], blo = lo[s], bhi = hi[s];
    const int b0 = slab_of(m, blo), b1 = slab_of(m, bhi);
    const int i0 = slab_start[b0], i1 = slab_start[b1 + 1];
    size_t capb = (size_t)capb_lds;
    char *mem = s_raw;
    if (ARENA) { /* size the arena segment from the band count of the first pass */
        capb = (size_t)band_cnt[s];
        const unsigned long long need = slice_kd_bytes(capb);
        if (threadIdx.x == 0) s_off = atomicAdd(&m->arena_cursor, (need + 15) & ~15ull);
        __syncthreads();
        if (s_off + need > arena_cap) {
            if (threadIdx.x == 0) { set_err(m, DERR_CAPACITY, s); node_start[s] = 0; node_cnt[s] = 0; if (slice_wpcnt) slice_wpcnt[s] = 0; }
            return;
        }
        mem = arena + s_off;
    }
    SliceKdMem L = carve_slice_kd(mem, capb);
    if (threadIdx.x == 0) { s_n = 0; s_plane = 0; s_ner = 0; s_m = 0; }
    __syncthreads();
    /* rangedX_index: the PassThrough band, minus points exactly on the plane (neither side) */
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        float4 p = sorted4[i];
        if (!(p.x < blo || p.x > bhi)) {
            float distance2plane = (p.x - Px) * 1.f + (p.y - 0.f) * 0.f + (p.z - 0.f) * 0.f;
            if (distance2plane > 0 || distance2plane < 0) {
                int slot = atomicAdd(&s_n, 1);
                if ((size_t)slot < capb) L.a4[slot] = p;
            } else atomicAdd(&s_plane, 1);
        }
    }
    __syncthreads();
    STAMP(0, 0); /* band gather */
    const int n = s_n;
    if (!ARENA) {
        if (threadIdx.x == 0) band_cnt[s] = n + s_plane;
        if ((size_t)n > capb) { /* does not fit LDS: leave it to the arena pass */
            if (threadIdx.x == 0) { big_list[atomicAdd(&m->big_slices, 1)] = s; node_start[s] = 0; node_cnt[s] = 0; if (slice_wpcnt) slice_wpcnt[s] = 0; }
            return;
        }
    }
    /* sort the band by (side, y): Er first, then El, ascending y inside each */
    const float y0 = m->mn[1];
    const float yr = m->mx[1] - y0;
    {
        const int NBh = next_pow2(max(n, 64)) >> 1; /* buckets per side (about one point each) */
        const float scale = yr > 0.f ? (float)NBh / yr : 0.f;
        auto gen = [&](int i) {
            float4 p = L.a4[i];
            return ((p.x - Px) > 0 ? KD_SIDE : 0ull) | YK_MAKE(p.y, i); /* side 1 = El (left, x > Px), 0 = Er */
        };
        auto bucket = [&](u64 k) {
            int q = (int)((ord2f(YK_Y(k)) - y0) * scale);
            q = q < 0 ? 0 : (q >= NBh ? NBh - 1 : q);
            return q + ((k & KD_SIDE) ? NBh : 0);
        };
        auto less = [&](u64 a, u64 bb) { return a < bb; };
        int ner = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) ner += (L.a4[i].x - Px) > 0 ? 0 : 1;
        ner = wave_sum(ner);
        if ((threadIdx.x & 63) == 0 && ner) atomicAdd(&s_ner, ner);
        block_bucket_sort(L.keys, n, L.hist_band, 2 * NBh, s_scr, gen, bucket, less);
    }
    STAMP(0, 1); /* band sort */
    const int nEr = s_ner, nEl = n - nEr;
    if (nEl == 0 || nEr == 0) {
        /* empty left side: empty map -> < 3 knots; empty right side: empty FLANN tree */
        if (threadIdx.x == 0) { set_err(m, DERR_SLICE, s); node_start[s] = 0; node_cnt[s] = 0; if (slice_wpcnt) slice_wpcnt[s] = 0; }
        return;
    }
    for (int i0q = 0; i0q < nEl; i0q += blockDim.x) {
        const int i = i0q + threadIdx.x;
        float y = 0.f, z = 0.f;
        int qi = 0;
        if (i < nEl) {
            const float4 q = L.a4[YK_POS(L.keys[nEr + i])];
            const int jr = nn_sorted_side(L.keys, L.a4, 0, nEr, q);
            const float4 R = L.a4[YK_POS(L.keys[jr])];
            const int jl = nn_sorted_side(L.keys, L.a4, nEr, n, R);
            const float4 Lp = L.a4[YK_POS(L.keys[jl])];
            float t = (Px - R.x) / (Lp.x - R.x);
            y = R.y + t * (Lp.y - R.y);
            z = R.z + t * (Lp.z - R.z);
            if (y == 0.f) y = 0.f; /* -0.0 and +0.0 are one std::map key */
            qi = idx_of(q);
        }
        if (i < nEl) { L.cy[i] = y; L.cz[i] = z; L.cidx[i] = qi; }
    }
    __syncthreads(); /* the band (a4, keys) is dead from here: ckeys / hist_cand reuse it */
    STAMP(0, 2); /* nearest neighbours + lerp */
    {   /* sort the node candidates by y */
        const int NBc = next_pow2(max(nEl, 64));
        const float scale = yr > 0.f ? (float)NBc / yr : 0.f;
        auto gen = [&](int i) { return YK_MAKE(L.cy[i], i); };
        auto bucket = [&](u64 k) {
            int q = (int)((ord2f(YK_Y(k)) - y0) * scale);
            return q < 0 ? 0 : (q >= NBc ? NBc - 1 : q);
        };
        auto less = [&](u64 a, u64 bb) { return a < bb; };
        block_bucket_sort(L.ckeys, nEl, L.hist_cand, NBc, s_scr, gen, bucket, less);
    }
    STAMP(0, 3); /* candidate sort */
    /* std::map semantics: one node per distinct y.  Node[y] = ... is overwritten by every later
       writer and El is walked in ascending cloud index, so the value kept is the one written by
       the candidate with the highest cloud index inside the run of equal keys.  One scan over the "last of its run" flags
       gives every kept candidate its place; the slice's segment of the knot arrays is reserved once the total is known. */
    if (threadIdx.x == 0) s_m = 0;
    __syncthreads();
    for (int base = 0; base < nEl; base += blockDim.x) {
        const int j = base + threadIdx.x;
        int keep = 0;
        u64 k = 0;
        if (j < nEl) {
            k = L.ckeys[j];
            keep = (j == nEl - 1) || (YK_Y(L.ckeys[j + 1]) != YK_Y(k));
        }
        int t2;
        const int pre = block_exscan(keep, s_scr, &t2);
        const int o = s_m;
        if (keep) { /* parked in LDS until the segment is known: which candidate supplies the knot (every member of the run has its y) */
            int best_i = YK_POS(k);
            int best_idx = L.cidx[best_i];
            for (int q = j - 1; q >= 0 && YK_Y(L.ckeys[q]) == YK_Y(k); --q) {
                const int ci = YK_POS(L.ckeys[q]);
                if (L.cidx[ci] > best_idx) { best_idx = L.cidx[ci]; best_i = ci; }
            }
            L.hist_cand[o + pre] = best_i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_m = o + t2;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int tot = s_m;
        int base = atomicAdd(&m->node_cursor, tot);
        if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, s); base = 0; tot = 0; }
        s_base = base;
        s_plane = tot; /* (s_plane is free again: the knot count, for the workgroup, without re-reading node_cnt from memory) */
        node_start[s] = base;
        node_cnt[s] = tot;
        if (tot < 3) set_err(m, DERR_SLICE, s);
    }
    __syncthreads();
    const int nknots = s_plane;
    if (slice_wpcnt && threadIdx.x == 0)
        slice_wpcnt[s] = nknots >= 1 ? sample_count((double)L.cy[L.hist_cand[0]], (double)L.cy[L.hist_cand[nknots - 1]], trim, res, W_cap) : 0;
    if (nknots == 0) return;
    for (int i = threadIdx.x; i < nknots; i += blockDim.x) {
        const int ci = L.hist_cand[i];
        node_x[s_base + i] = Px; /* insert_cloud.points[i].x = PlanePoint[0] */
        node_y[s_base + i] = L.cy[ci];
        node_z[s_base + i] = L.cz[ci];
    }
    STAMP(0, 4); /* map flattening + node write */
}

/* API mirrors: rangedX_index(position) and insert_point(indices, plane) on one workgroup */
PPP_KERNEL void __launch_bounds__(256) k_band_indices(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                                      DevMeta *m, float lo, float hi, int capb, int *out, int out_cap)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_n;
    SliceLds L = carve_slice_lds(s_raw, capb);
    int n = band_gather_sorted(L, capb, sorted4, slab_start, m, lo, hi, &s_n);
    if (n < 0) { if (threadIdx.x == 0) { m->api_cnt = s_n; m->api_flag = DERR_CAPACITY; } return; }
    for (int j = threadIdx.x; j < n && j < out_cap; j += blockDim.x) out[j] = (int)(L.keys[j] >> 32);
    if (threadIdx.x == 0) { m->api_cnt = n; m->api_flag = 0; }
}

/* rangedX_index for a band that does not fit LDS: same result from global scratch
   (tmp, sorted: n keys each; hist: NB + 1 ints), one workgroup. */
PPP_KERNEL void __launch_bounds__(256) k_band_indices_big(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                                          DevMeta *m, float lo, float hi, int npts, u64 *tmp, u64 *sorted,
                                                          int *hist, int NB, int cap, int *out)
{
    __shared__ int s_n;
    __shared__ int s_scr[17];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (lo <= hi && m->n_valid > 0) {
        const int b0 = slab_of(m, lo), b1 = slab_of(m, hi);
        const int i0 = slab_start[b0], i1 = slab_start[b1 + 1];
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            float4 p = sorted4[i];
            if (!(p.x < lo || p.x > hi)) {
                int slot = atomicAdd(&s_n, 1);
                if (slot < cap) tmp[slot] = (u64)(u32)idx_of(p);
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > cap) { if (threadIdx.x == 0) { m->api_cnt = n; m->api_flag = DERR_CAPACITY; } return; }
    auto gen = [&](int i) { return tmp[i]; };
    auto bucket = [&](u64 k) { int q = (int)((k * (u64)NB) / (u64)max(npts, 1)); return q >= NB ? NB - 1 : q; };
    auto less = [&](u64 a, u64 b) { return a < b; };
    block_bucket_sort(sorted, n, hist, NB, s_scr, gen, bucket, less);
    for (int j = threadIdx.x; j < n; j += blockDim.x) out[j] = (int)sorted[j];
    if (threadIdx.x == 0) { m->api_cnt = n; m->api_flag = 0; }
}

PPP_KERNEL void __launch_bounds__(256) k_insert_api(const float *__restrict__ X, const float *__restrict__ Y,
                                                    const float *__restrict__ Z, int npts, const int *__restrict__ indices,
                                                    int n, float Px, int pairing, int capb, DevMeta *m, float *out_y,
                                                    float *out_z, int out_cap)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    SliceLds L = carve_slice_lds(s_raw, capb);
    if (n > capb) { if (threadIdx.x == 0) { m->api_cnt = 0; m->api_flag = DERR_CAPACITY; } return; }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int id = indices[i];
        float4 p;
        if (id < 0 || id >= npts) p = make_float4(NAN, NAN, NAN, __int_as_float(id));
        else p = make_float4(X[id], Y[id], Z[id], __int_as_float(id));
        L.a4[i] = p;
        L.keys[i] = (u64)(u32)i; /* the caller's order is the El/Er order */
    }
    __syncthreads();
    int ncand = insert_point_lds(L, n, Px, pairing, s_scr);
    if (ncand < 0) { if (threadIdx.x == 0) { m->api_cnt = 0; m->api_flag = DERR_SLICE; } return; }
    int mm = flatten_nodes(L, ncand, out_y, out_z, out_cap, s_scr);
    if (threadIdx.x == 0) { m->api_cnt = mm; m->api_flag = 0; }
}

/* ppp_get_nodes: the knots of every slice, packed (slice s at off[s], x | y | z planes of `total` floats each), for ONE copy to the
   host -- the slices' segments lie cap_el apart in the knot arrays (window path) and are a quarter full */
PPP_KERNEL void __launch_bounds__(256) k_nodes_pack(const int *__restrict__ start, const int *__restrict__ off, int S, int total,
                                                    const float *__restrict__ nx, const float *__restrict__ ny, const float *__restrict__ nz,
                                                    float *__restrict__ out)
{
    const int s = (int)blockIdx.x;
    if (s >= S) return;
    const int st = start[s], o = off[s], cnt = off[s + 1] - o;
    for (int i = (int)threadIdx.x; i < cnt; i += (int)blockDim.x) {
        out[o + i] = nx[st + i];
        out[total + o + i] = ny[st + i];
        out[2 * total + o + i] = nz[st + i];
    }
}

/* ------------------------------------------------------------------ */
/* a9: getPath sampling (path_translation_alg.cpp:149-169)              */
/* ------------------------------------------------------------------ */
/* ppp_finish_path_async: the list was sampled elsewhere (slice-range handles); rebuild the per-run state
   getPath's second half needs from the per-slice counts: offsets, TailIndex, the B.6 flag, W. */
PPP_KERNEL void __launch_bounds__(1024) k_count_given(DevMeta *m, DevParams P, int nk, int W_given, int *wp_cnt, int *wp_off, int *tail,
                                                      int W_cap)
{
    __shared__ int scratch[17];
    __shared__ int s_run, s_short;
    if (threadIdx.x == 0) { s_run = 0; s_short = 0; }
    __syncthreads();
    const int res_i = (int)P.rpy_resolution;
    for (int base = 0; base < nk; base += blockDim.x) {
        const int k = base + threadIdx.x;
        const int cnt = k < nk ? wp_cnt[k] : 0;
        int tot;
        const int pre = block_exscan(cnt, scratch, &tot);
        const int run = s_run;
        if (k < nk) {
            wp_off[k] = run + pre;
            tail[k] = run + pre + cnt - 1;
            if (P.rpy_resolution > 2 && cnt <= res_i) s_short = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m->err = 0; m->err_slice = 0x7fffffff; m->sweeps = 0; m->rpy_oob = 0; m->smooth_done = -1; m->emit_ticket = 0;
        m->any_short = s_short;
        m->nkept = nk;
        m->big_slabs = 0; m->big_slices = 0; /* the index of this handle plays no part in what follows */
        m->win_flag = 0;                     /* ... and neither does a window pass: nothing here may ask for a re-run on the slab path */
        int W = s_run;
        if (W != W_given || W > W_cap) { m->err = DERR_CAPACITY; W = 0; }
        m->W = W;
        wp_off[nk] = W;
    }
}

/* ... and the list itself into the buffer k_pose leaves it in */
PPP_KERNEL void __launch_bounds__(256) k_load_pre(const DevMeta *m, const float *__restrict__ pre6, float *wp_pre)
{
    const int W = m->W;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (m->err || w >= W) return;
    for (int d = 0; d < 6; ++d) wp_pre[6 * (size_t)w + d] = pre6[6 * (size_t)w + d];
}

PPP_KERNEL void k_eval_api(DevMeta *m, const float *__restrict__ node_x, const float *__restrict__ node_y,
                           const float *__restrict__ node_z, const int *__restrict__ node_start,
                           const int *__restrict__ node_cnt, int s, const double *__restrict__ yq, int kq, double *out)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kq) return;
    const int st = node_start[s], mm = node_cnt[s];
    const float *nx = node_x + st, *ny = node_y + st, *nz = node_z + st;
    auto Yf = [&](int i) { return (double)ny[i]; };
    auto Zf = [&](int i) { return (double)nz[i]; };
    auto Xf = [&](int i) { return (double)nx[i]; };
    double y = yq[t];
    if (mm < 3 || y < (double)ny[0] || y > (double)ny[mm - 1] || !(y == y)) {
        out[3 * t] = out[3 * t + 1] = out[3 * t + 2] = NAN;
        m->api_flag = DERR_DOMAIN;
        return;
    }
    int i = gsl_bsearch(mm, y, Yf);
    out[3 * t] = steffen_eval_at(i, mm, y, Yf, Xf);
    out[3 * t + 1] = y;
    out[3 * t + 2] = steffen_eval_at(i, mm, y, Yf, Zf);
}

/* Spline::point(y) on caller-supplied knots (include/Spline.h:10-25: two gsl_interp_steffen splines y->x, y->z over the
   same strictly increasing y): the same steffen.c restatement as above, on double knots.  Outside [y_0, y_{m-1}] GSL
   raises GSL_EDOM (and its default handler aborts): NaN + the flag here. */
PPP_KERNEL void k_spline_eval_d(const double *__restrict__ ky, const double *__restrict__ kx, const double *__restrict__ kz, int mm,
                                const double *__restrict__ yq, int kq, double *out, int *flag)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kq) return;
    auto Yf = [&](int i) { return ky[i]; };
    auto Zf = [&](int i) { return kz[i]; };
    auto Xf = [&](int i) { return kx[i]; };
    const double y = yq[t];
    if (mm < 3 || y < ky[0] || y > ky[mm - 1] || !(y == y)) {
        out[3 * t] = out[3 * t + 1] = out[3 * t + 2] = NAN;
        *flag = DERR_DOMAIN;
        return;
    }
    const int i = gsl_bsearch(mm, y, Yf);
    out[3 * t] = steffen_eval_at(i, mm, y, Yf, Xf);
    out[3 * t + 1] = y;
    out[3 * t + 2] = steffen_eval_at(i, mm, y, Yf, Zf);
}

/* ------------------------------------------------------------------ */
/* a10/a11/a12: nearest cloud point, its normal, pose, hand-eye          */
/* ------------------------------------------------------------------ */
struct SlabView {
    const float4 *sorted4;
    const int *slab_start;
    const float *slab_xmin, *slab_xmax;
    const DevMeta *m;
    /* optional LDS copy of sorted4[lds_lo, lds_hi) (the slabs around one slice) */
    const float4 *lds;
    int lds_lo, lds_hi;
    /* optional y-bucket table of the slabs (YTB + 1 ints per slab) and an LDS copy of the rows of slabs [tab_lo, tab_hi) */
    const int *ytab = nullptr;
    const int *lds_tab = nullptr;
    int tab_lo = 0, tab_hi = 0;
    __device__ inline float4 at(int i) const { return (i >= lds_lo && i < lds_hi) ? lds[i - lds_lo] : sorted4[i]; }
    /* [lo, hi) inside slab bb = [s0, s1) that holds the lower bound of qy */
    __device__ inline void narrow(int bb, int s0, int s1, float qy, int &lo, int &hi) const
    {
        lo = s0; hi = s1;
        if (ytab) {
            const int *T = (bb >= tab_lo && bb < tab_hi) ? lds_tab + (bb - tab_lo) * (YTB + 1) : ytab + (size_t)bb * (YTB + 1);
            const int q = ytab_bucket(m, qy);
            lo = s0 + T[q]; hi = s0 + T[q + 1];
        }
    }
};

/* first position in [s0,s1) whose y >= qy */
__device__ inline int lower_bound_y(const SlabView &V, int s0, int s1, float qy)
{
    while (s0 < s1) {
        int mid = (s0 + s1) >> 1;
        if (V.at(mid).y < qy) s0 = mid + 1; else s1 = mid;
    }
    return s0;
}

/* kdtree.nearestKSearch(q, 1): exact, ties -> lowest cloud index */
__device__ inline int nearest_in_slabs(const SlabView &V, float qx, float qy, float qz, float4 *found)
{
    const int B = V.m->B;
    float best = INFINITY;
    int bidx = 0x7fffffff;
    float4 bp = make_float4(NAN, NAN, NAN, 0.f);
    auto scan_slab = [&](int b) {
        const int s0 = V.slab_start[b], s1 = V.slab_start[b + 1];
        if (s0 >= s1) return;
        const int p = lower_bound_y(V, s0, s1, qy);
        for (int i = p; i < s1; ++i) {
            const float4 c = V.at(i);
            float dy = qy - c.y;
            if (dy * dy > best) break;
            float d = dist2_flann(qx, qy, qz, c.x, c.y, c.z);
            int id = idx_of(c);
            if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bp = c; }
        }
        for (int i = p - 1; i >= s0; --i) {
            const float4 c = V.at(i);
            float dy = qy - c.y;
            if (dy * dy > best) break;
            float d = dist2_flann(qx, qy, qz, c.x, c.y, c.z);
            int id = idx_of(c);
            if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bp = c; }
        }
    };
    const int b = slab_of(V.m, qx);
    scan_slab(b);
    for (int bb = b + 1; bb < B; ++bb) {
        if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
        float dx = V.slab_xmin[bb] - qx;
        if (dx > 0.f && dx * dx > best) break;
        scan_slab(bb);
    }
    for (int bb = b - 1; bb >= 0; --bb) {
        if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
        float dx = qx - V.slab_xmax[bb];
        if (dx > 0.f && dx * dx > best) break;
        scan_slab(bb);
    }
    *found = bp;
    return bidx == 0x7fffffff ? -1 : bidx;
}

/* pcl::NormalEstimation::computeFeature for one cloud point p:
   radius search, computeMeanAndCovarianceMatrix shifted by the nearest neighbour (p itself),
   eigen33, flipNormalTowardsViewpoint.
   The neighbours are summed in the order the radius search returns them -- ascending (distance, cloud index) -- so the
   float covariance, and with it the normal, carries the same bits as the reference's; the principal curvatures of the
   dynamic adjustment are built on this field and feed discontinuous decisions.  (k_pose's per-waypoint normals use
   the faster slab-order sum of normal_at_point_group: they only feed continuous outputs.) */
#define NRM_CAP 48
__device__ inline void normal_at_point(const SlabView &V, const float4 p, float radius, const float vp[3], float out[4])
{
    const int B = V.m->B;
    const float r2 = radius * radius;
    int cpos[NRM_CAP], cidx[NRM_CAP];
    float cd[NRM_CAP];
    int count = 0;
    /* visit(i, c, d) for every indexed point within the radius */
    auto for_each_neighbour = [&](auto visit) {
        auto scan_slab = [&](int b) {
            const int s0 = V.slab_start[b], s1 = V.slab_start[b + 1];
            if (s0 >= s1) return;
            const int q0 = lower_bound_y(V, s0, s1, p.y);
            for (int i = q0; i < s1; ++i) {
                const float4 c = V.at(i);
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                const float d = dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z);
                if (d <= r2) visit(i, c, d);
            }
            for (int i = q0 - 1; i >= s0; --i) {
                const float4 c = V.at(i);
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                const float d = dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z);
                if (d <= r2) visit(i, c, d);
            }
        };
        const int b = slab_of(V.m, p.x);
        scan_slab(b);
        for (int bb = b + 1; bb < B; ++bb) {
            if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
            const float dx = V.slab_xmin[bb] - p.x;
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
        for (int bb = b - 1; bb >= 0; --bb) {
            if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
            const float dx = p.x - V.slab_xmax[bb];
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
    };
    for_each_neighbour([&](int i, const float4 &c, float d) {
        if (count < NRM_CAP) { cpos[count] = i; cidx[count] = idx_of(c); cd[count] = d; }
        count++;
    });
    if (count < 3) { out[0] = out[1] = out[2] = out[3] = NAN; return; }
    float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const float4 &c) {
        const float x = c.x - p.x, y = c.y - p.y, z = c.z - p.z;
        accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
        accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
        accu[6] += x; accu[7] += y; accu[8] += z;
    };
    if (count <= NRM_CAP) { /* the usual case: a few dozen neighbours, selected in order from the thread's own list */
        for (int r = 0; r < count; ++r) {
            int bj = 0;
            float bd = INFINITY;
            int bi = 0x7fffffff;
            for (int j = 0; j < count; ++j) {
                const float dj = cd[j];
                if (dj < bd || (dj == bd && cidx[j] < bi)) { bd = dj; bi = cidx[j]; bj = j; }
            }
            add(V.at(cpos[bj]));
            cd[bj] = INFINITY; cidx[bj] = 0x7fffffff; /* taken */
        }
    } else { /* a very dense neighbourhood: no list, the next neighbour in order is found by scanning again */
        float ld = -1.f;
        int li = -1;
        for (int r = 0; r < count; ++r) {
            float bd = INFINITY;
            int bi = 0x7fffffff;
            float4 bc = p;
            for_each_neighbour([&](int, const float4 &c, float d) {
                const int id = idx_of(c);
                const bool after = d > ld || (d == ld && id > li);
                if (after && (d < bd || (d == bd && id < bi))) { bd = d; bi = id; bc = c; }
            });
            add(bc);
            ld = bd; li = bi;
        }
    }
    float cnt = (float)count;
    for (int i = 0; i < 9; ++i) accu[i] /= cnt;
    float cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, n[3];
    pcl_eigen33_smallest(cov, &ev, n);
    float eig_sum = cov[0] + cov[4] + cov[8];
    float curv = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
    float vx = vp[0] - p.x, vy = vp[1] - p.y, vz = vp[2] - p.z;
    float cos_theta = vx * n[0] + vy * n[1] + vz * n[2];
    if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2]; out[3] = curv;
}

/* ---- the same two queries with G lanes per query (G = 1, 2, 4 or 8 consecutive lanes of a wave) ----
   The cost of a query is its chain of dependent binary-search reads, one chain per slab it touches; a group walks G
   slabs side by side: round r gives lane g the slab at offset 0, +1, -1, +2, -2, ... (position r*G + g of that list)
   from the query's own slab.  A side is closed by the first slab that lies outside the grid or, being non-empty, is
   farther in x than the current bound -- every slab beyond it is farther still.  All lanes of a group call these
   together (inactive queries included: the shuffles are executed by whole groups) and all receive the result. */
__device__ inline int group_signed_offset(int o) { return o == 0 ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2)); }

/* (G is a run-time value: one copy of these two searches in the kernel instead of one per group size -- k_pose was 25 000
   instructions with three, far beyond the instruction cache) */
__device__ inline int nearest_in_slabs_group(const int G, const SlabView &V, bool active, float qx, float qy, float qz, float4 *found, int g, float bound2)
{
    const int B = V.m->B;
    /* bound2: only points closer than this are looked for (INFINITY = all).  A lane that starts in a far slab has no
       candidate of its own yet, and without a bound it would scan that slab's whole y-window; with the caller's guess
       (a few point spacings) far slabs are closed by their x gap at once.  Nothing found within the guess -> the
       caller repeats the query unbounded, so the result is exact either way. */
    float best = bound2;
    int bidx = 0x7fffffff;
    float4 bp = make_float4(NAN, NAN, NAN, 0.f);
    const int b = slab_of(V.m, qx);
    int right_closed = active ? 0 : 1, left_closed = right_closed;
    for (int round = 0; !(right_closed && left_closed); ++round) {
        const int so = group_signed_offset(round * G + g);
        const int bb = b + so;
        int rc = 0, lc = 0;
        const bool side_closed = (so > 0 && right_closed) || (so < 0 && left_closed);
        if (!side_closed) {
            if (bb < 0) lc = 1;
            else if (bb >= B) rc = 1;
            else {
                const int s0 = V.slab_start[bb], s1 = V.slab_start[bb + 1];
                if (s0 < s1) {
                    const float dx = so > 0 ? V.slab_xmin[bb] - qx : (so < 0 ? qx - V.slab_xmax[bb] : 0.f);
                    if (dx > 0.f && dx * dx > best) { if (so > 0) rc = 1; else lc = 1; }
                    else {
                        /* the staged window covers this slab almost always: then every read is an LDS read and the
                           per-access "LDS or global?" test of SlabView::at leaves the loops */
                        auto scan = [&](auto at) {
                            int lo, hi;
                            V.narrow(bb, s0, s1, qy, lo, hi);
                            while (lo < hi) { const int mid = (lo + hi) >> 1; if (at(mid).y < qy) lo = mid + 1; else hi = mid; }
                            const int p = lo;
                            /* candidates in the reference's order, read four at a time (the reads are independent; a read past
                               the one that ends the walk is harmless): the walk is a chain of dependent LDS round trips otherwise */
                            auto visit = [&](const float4 &c) {
                                const float dy = qy - c.y;
                                if (dy * dy > best) return false;
                                const float d = dist2_flann(qx, qy, qz, c.x, c.y, c.z);
                                const int id = idx_of(c);
                                if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bp = c; }
                                return true;
                            };
                            for (int i = p; i < s1; i += 4) {
                                const int e = s1 - 1;
                                const float4 c0 = at(i), c1 = at(min(i + 1, e)), c2 = at(min(i + 2, e)), c3 = at(min(i + 3, e));
                                if (!visit(c0)) break;
                                if (i + 1 > e || !visit(c1)) break;
                                if (i + 2 > e || !visit(c2)) break;
                                if (i + 3 > e || !visit(c3)) break;
                            }
                            for (int i = p - 1; i >= s0; i -= 4) {
                                const float4 c0 = at(i), c1 = at(max(i - 1, s0)), c2 = at(max(i - 2, s0)), c3 = at(max(i - 3, s0));
                                if (!visit(c0)) break;
                                if (i - 1 < s0 || !visit(c1)) break;
                                if (i - 2 < s0 || !visit(c2)) break;
                                if (i - 3 < s0 || !visit(c3)) break;
                            }
                        };
                        if (s0 >= V.lds_lo && s1 <= V.lds_hi) { const float4 *L = V.lds - V.lds_lo; scan([&](int i) { return L[i]; }); }
                        else scan([&](int i) { return V.at(i); });
                    }
                }
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) { /* group minimum on (distance, cloud index); closings are OR-ed */
            const float od = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            const float ox = __shfl_xor(bp.x, o, 64), oy = __shfl_xor(bp.y, o, 64), oz = __shfl_xor(bp.z, o, 64), ow = __shfl_xor(bp.w, o, 64);
            if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; bp = make_float4(ox, oy, oz, ow); }
            rc |= __shfl_xor(rc, o, 64);
            lc |= __shfl_xor(lc, o, 64);
        }
        right_closed |= rc; left_closed |= lc;
    }
    *found = bp;
    return bidx == 0x7fffffff ? -1 : bidx;
}

/* normal_at_point with G lanes: every lane sums the neighbours of its slabs, the partial sums are added pairwise across
   the group (a fixed tree, so the result depends on the slab contents only -- identical for a slice-range handle and a
   whole-cloud handle) */
__device__ inline void normal_at_point_group(const int G, const SlabView &V, bool active, const float4 p, float radius, const float vp[3], float out[4], int g)
{
    const int B = V.m->B;
    const float r2 = radius * radius;
    float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int count = 0;
    const int b = slab_of(V.m, p.x);
    int right_closed = active ? 0 : 1, left_closed = right_closed;
    for (int round = 0; !(right_closed && left_closed); ++round) {
        const int so = group_signed_offset(round * G + g);
        const int bb = b + so;
        int rc = 0, lc = 0;
        const bool side_closed = (so > 0 && right_closed) || (so < 0 && left_closed);
        if (!side_closed) {
            if (bb < 0) lc = 1;
            else if (bb >= B) rc = 1;
            else {
                const int s0 = V.slab_start[bb], s1 = V.slab_start[bb + 1];
                if (s0 < s1) {
                    const float dx = so > 0 ? V.slab_xmin[bb] - p.x : (so < 0 ? p.x - V.slab_xmax[bb] : 0.f);
                    if (dx > 0.f && dx * dx > r2) { if (so > 0) rc = 1; else lc = 1; }
                    else {
                        auto scan = [&](auto at) { /* LDS-only reads when the staged window covers the slab (see the NN scan) */
                            int lo, hi;
                            V.narrow(bb, s0, s1, p.y, lo, hi);
                            while (lo < hi) { const int mid = (lo + hi) >> 1; if (at(mid).y < p.y) lo = mid + 1; else hi = mid; }
                            const int q0 = lo;
                            auto visit = [&](const float4 &c) { /* same order of additions as one point after the other */
                                const float dy = p.y - c.y;
                                if (dy * dy > r2) return false;
                                if (dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z) <= r2) {
                                    const float x = c.x - p.x, y = c.y - p.y, z = c.z - p.z;
                                    accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
                                    accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
                                    accu[6] += x; accu[7] += y; accu[8] += z;
                                    count++;
                                }
                                return true;
                            };
                            for (int i = q0; i < s1; i += 4) { /* four independent reads in flight (see the NN scan) */
                                const int e = s1 - 1;
                                const float4 c0 = at(i), c1 = at(min(i + 1, e)), c2 = at(min(i + 2, e)), c3 = at(min(i + 3, e));
                                if (!visit(c0)) break;
                                if (i + 1 > e || !visit(c1)) break;
                                if (i + 2 > e || !visit(c2)) break;
                                if (i + 3 > e || !visit(c3)) break;
                            }
                            for (int i = q0 - 1; i >= s0; i -= 4) {
                                const float4 c0 = at(i), c1 = at(max(i - 1, s0)), c2 = at(max(i - 2, s0)), c3 = at(max(i - 3, s0));
                                if (!visit(c0)) break;
                                if (i - 1 < s0 || !visit(c1)) break;
                                if (i - 2 < s0 || !visit(c2)) break;
                                if (i - 3 < s0 || !visit(c3)) break;
                            }
                        };
                        if (s0 >= V.lds_lo && s1 <= V.lds_hi) { const float4 *L = V.lds - V.lds_lo; scan([&](int i) { return L[i]; }); }
                        else scan([&](int i) { return V.at(i); });
                    }
                }
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) { rc |= __shfl_xor(rc, o, 64); lc |= __shfl_xor(lc, o, 64); }
        right_closed |= rc; left_closed |= lc;
    }
    for (int o = G >> 1; o > 0; o >>= 1) {
        for (int i = 0; i < 9; ++i) accu[i] += __shfl_xor(accu[i], o, 64);
        count += __shfl_xor(count, o, 64);
    }
    if (count < 3) { out[0] = out[1] = out[2] = out[3] = NAN; return; }
    float cnt = (float)count;
    for (int i = 0; i < 9; ++i) accu[i] /= cnt;
    float cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, n[3];
    pcl_eigen33_smallest<false>(cov, &ev, n); /* continuous outputs only: the device float trig is enough */
    float eig_sum = cov[0] + cov[4] + cov[8];
    float curv = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
    float vx = vp[0] - p.x, vy = vp[1] - p.y, vz = vp[2] - p.z;
    float cos_theta = vx * n[0] + vy * n[1] + vz * n[2];
    if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2]; out[3] = curv;
}

/* One workgroup per kept slice: a9 (sampling, path_translation_alg.cpp:156-169), then for every
   waypoint the nearest cloud point, its PCL normal, the tool frame and the hand-eye transform
   (:178-211).  The slabs within POSE_PAD mm of the plane and the slice's spline knots are staged
   in LDS, so the binary searches and window scans are LDS reads instead of dependent HBM/L2
   round trips; anything outside the staged window falls back to the global copy (exactness
   never depends on the window). */
#ifndef POSE_STAGE_CAP
#define POSE_STAGE_CAP 7680 /* most points a workgroup stages (120 KiB of the CU's 160); the plan asks for what the slab grid and the density need */
#endif
#define POSE_PRE 6        /* staged points a thread requests before the bookkeeping (registers; none in the 1024-thread form, which is short of them) */
/* staged points | knots (y, z, x) | y-bucket rows of the staged slabs */
__host__ __device__ inline size_t pose_lds_bytes(int knot_cap, int stage_cap, int tab_slabs)
{
    return (size_t)stage_cap * 16 + (size_t)knot_cap * 12 + (size_t)tab_slabs * (YTB + 1) * 4;
}
/* lanes per waypoint (the two searches walk G slabs side by side): a function of the slice's waypoint count ONLY, so the
   partial sums of the normals -- and with them the last bits of the list -- do not depend on the launch geometry */
#ifndef POSE_G2_MAX
#define POSE_G2_MAX 384 /* 2 lanes x 384 waypoints = 768 threads, the largest form that keeps its registers (cfg 5, 258 waypoints per slice: 145 -> 135 us) */
#endif
#ifndef POSE_G4_MAX
#define POSE_G4_MAX 128 /* (cfg 2, 103 waypoints a slice: 4 lanes 26.5 us, 2 lanes 29.6, 1 lane 29.3 -- the searches are a third of the kernel, the rest is the same for any lane count) */
#endif
__host__ __device__ inline int pose_lanes(int cnt) { return cnt <= POSE_G4_MAX ? 4 : (cnt <= POSE_G2_MAX ? 2 : 1); }

#ifndef POSE_T
#define POSE_T 1024
#endif
#ifndef POSE_GMAX
#define POSE_GMAX 4
#endif
/* ALIGNED (trans2center ran): path_translation_alg.cpp:146-174 -- every sampled point goes through invTransAlign, and
   the nearest point and its normal are looked up in the cloud carried back by invTransAlign (`back`: that cloud's own
   slab index, built once by ppp_trans2center; same point indices) */
struct PoseBack {
    const float4 *sorted4;
    const int *slab_start;
    const float *slab_xmin, *slab_xmax;
    const DevMeta *m;
    float inv[3][4];
    const int *ytab;
};
template <bool ALIGNED, int PRE>
__device__ __forceinline__ void pose_body(DevMeta *m, const DevParams &P, const float4 *__restrict__ sorted4,
                                          const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
                                          const float *__restrict__ slab_xmax, const float *__restrict__ px,
                                          const float *__restrict__ node_x, const float *__restrict__ node_y,
                                          const float *__restrict__ node_z,
                                          const int *__restrict__ node_start, const int *__restrict__ node_cnt,
                                          int *wp_cnt, int *wp_off, int *tail, int W_cap, int arena_ran, int knot_cap, int stage_cap,
                                          int tab_slabs, float pad, float4 *wp_xyz, int *wp_nn, float4 *wp_normal, float *wp_pre, const PoseBack &back,
                                          const int *__restrict__ ytab, const int *__restrict__ slice_wpcnt, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    float4 *s_pts = (float4 *)s_raw;
    float *s_ny = (float *)(s_pts + stage_cap);
    float *s_nz = s_ny + knot_cap;
    float *s_nx = s_nz + knot_cap;
    int *s_tab = (int *)(s_nx + knot_cap); /* tab_slabs rows of YTB + 1 */
    __shared__ int s_scan[17];
    __shared__ int s_mycnt, s_myoff, s_run;
    const int k = bx;
    const int nk = m->nkept;
    if (m->err || k >= nk) return;
    /* work was left for the arena passes but they were not launched: report, the host re-runs */
    if (!arena_ran && (m->big_slabs > 0 || m->big_slices > 0)) { if (threadIdx.x == 0) atomicCAS(&m->err, 0, DERR_CAPACITY); return; }
    /* Everything this workgroup will stage -- the slabs around its plane (widest symmetric range that fits), their y-bucket
       rows, the slice's knots -- is REQUESTED here, into registers, in as few dependent rounds as the data allow (plane and
       knot segment; slab offsets; then points, rows and knots together with the slices' waypoint counts): the trips from
       memory run beside each other and beside the bookkeeping below instead of one after the other. */
    const int s = k + m->first_kept;
    const float Px = px[s];
    const int st = node_start[s], mm = node_cnt[s];
    const int first_kept = m->first_kept, sb = m->sb, se = m->se;
    int c2_first = 0; /* waypoint count of slice threadIdx.x (the first chunk of the scan below) */
    if (slice_wpcnt && (int)threadIdx.x < nk) {
        const int s2 = (int)threadIdx.x + first_kept;
        if (s2 >= sb && s2 < se) c2_first = slice_wpcnt[s2];
    }
    int bL = slab_of(m, Px - pad), bR = slab_of(m, Px + pad);
    while (slab_start[bR + 1] - slab_start[bL] > stage_cap && bL < bR) {
        const int bc = slab_of(m, Px);
        if (bR - bc >= bc - bL) --bR; else ++bL;
    }
    int lds_lo = slab_start[bL], lds_hi = slab_start[bR + 1];
    if (lds_hi - lds_lo > stage_cap || ALIGNED) lds_hi = lds_lo; /* one over-full slab: no staging (nor for the other frame's index) */
    float4 pre4[PRE > 0 ? PRE : 1];
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int i = lds_lo + (int)threadIdx.x + q * (int)blockDim.x;
        pre4[q] = i < lds_hi ? sorted4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int *yt = ALIGNED ? back.ytab : ytab;
    int tab_lo = 0, tab_hi = 0;
    if (yt && lds_hi > lds_lo && bR - bL + 1 <= tab_slabs) { tab_lo = bL; tab_hi = bR + 1; }
    const int tab_n = (tab_hi - tab_lo) * (YTB + 1);
    constexpr int TPRE = PRE > 0 ? 2 : 0, KPRE = PRE > 0 ? 2 : 0;
    int tpre[TPRE > 0 ? TPRE : 1];
#pragma unroll
    for (int q = 0; q < TPRE; ++q) {
        const int i = (int)threadIdx.x + q * (int)blockDim.x;
        tpre[q] = i < tab_n ? yt[(size_t)tab_lo * (YTB + 1) + i] : 0;
    }
    const bool nodes_in_lds = mm <= knot_cap;
    float kpre[KPRE > 0 ? KPRE : 1][3];
#pragma unroll
    for (int q = 0; q < KPRE; ++q) {
        const int i = (int)threadIdx.x + q * (int)blockDim.x;
        const bool in = nodes_in_lds && i < mm;
        kpre[q][0] = in ? node_y[st + i] : 0.f; kpre[q][1] = in ? node_z[st + i] : 0.f; kpre[q][2] = in ? node_x[st + i] : 0.f;
    }
    /* a9 bookkeeping (the former k_count launch): every workgroup scans the waypoint counts of ALL kept slices -- left by
       k_slice_kd, or recomputed here from two knots and a closed form each -- to know its own offset in the list and W */
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nk; base += blockDim.x) {
        const int k2 = base + threadIdx.x;
        int c2 = 0;
        if (k2 < nk) {
            const int s2 = k2 + first_kept;
            if (s2 >= sb && s2 < se) {
                if (slice_wpcnt) c2 = base == 0 ? c2_first : slice_wpcnt[s2];
                else {
                    const int st2 = node_start[s2], mm2 = node_cnt[s2];
                    if (mm2 >= 1) c2 = sample_count((double)node_y[st2], (double)node_y[st2 + mm2 - 1], P.trim, P.path_resolution, W_cap);
                }
            }
        }
        int tot;
        const int pre = block_exscan(c2, s_scan, &tot);
        const int run = s_run;
        if (k2 == k) { s_mycnt = c2; s_myoff = run + pre; }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
    const int W = s_run, cnt = s_mycnt, off = s_myoff;
    if (W > W_cap) { if (threadIdx.x == 0) { set_err(m, DERR_CAPACITY, -1); if (k == 0) m->W = 0; } return; }
    if (threadIdx.x == 0) {
        wp_cnt[k] = cnt; wp_off[k] = off;
        tail[k] = off + cnt - 1; /* TailIndex.push_back(WayPointsList.size()-1) */
        if (P.rpy_resolution > 2 && cnt <= (int)P.rpy_resolution) m->any_short = 1;
        if (k == 0) { m->W = W; wp_off[nk] = W; }
    }
    if (W == 0 || cnt == 0) return;
    STAMP_BEGIN();
    /* what has arrived goes to LDS; what did not fit the registers is fetched now */
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int i = lds_lo + (int)threadIdx.x + q * (int)blockDim.x;
        if (i < lds_hi) s_pts[i - lds_lo] = pre4[q];
    }
    for (int i = lds_lo + (int)threadIdx.x + PRE * (int)blockDim.x; i < lds_hi; i += blockDim.x) s_pts[i - lds_lo] = sorted4[i];
#pragma unroll
    for (int q = 0; q < TPRE; ++q) {
        const int i = (int)threadIdx.x + q * (int)blockDim.x;
        if (i < tab_n) s_tab[i] = tpre[q];
    }
    for (int i = (int)threadIdx.x + TPRE * (int)blockDim.x; i < tab_n; i += blockDim.x) s_tab[i] = yt[(size_t)tab_lo * (YTB + 1) + i];
    if (nodes_in_lds) {
#pragma unroll
        for (int q = 0; q < KPRE; ++q) {
            const int i = (int)threadIdx.x + q * (int)blockDim.x;
            if (i < mm) { s_ny[i] = kpre[q][0]; s_nz[i] = kpre[q][1]; s_nx[i] = kpre[q][2]; }
        }
        for (int i = (int)threadIdx.x + KPRE * (int)blockDim.x; i < mm; i += blockDim.x) { s_ny[i] = node_y[st + i]; s_nz[i] = node_z[st + i]; s_nx[i] = node_x[st + i]; }
    }
    __syncthreads();
    STAMP(1, 0); /* staging */
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, s_pts, lds_lo, lds_hi};
    V.ytab = yt; V.lds_tab = s_tab; V.tab_lo = tab_lo; V.tab_hi = tab_hi;
    if (ALIGNED) { V.sorted4 = back.sorted4; V.slab_start = back.slab_start; V.slab_xmin = back.slab_xmin; V.slab_xmax = back.slab_xmax; V.m = back.m; }
    const float *ny = nodes_in_lds ? s_ny : node_y + st, *nz = nodes_in_lds ? s_nz : node_z + st;
    const float *nx = nodes_in_lds ? s_nx : node_x + st; /* the plane x, or cloud x after the dynamic adjustment */
    auto Yf = [&](int i) { return (double)ny[i]; };
    auto Zf = [&](int i) { return (double)nz[i]; };
    auto Xf = [&](int i) { return (double)nx[i]; };
    const double start = (double)ny[0] + P.trim;
    /* G lanes per waypoint: as many as the workgroup has to spare (the two searches walk G slabs side by side) */
    const bool dy_closed_form = sums_exact(start, P.path_resolution, (double)cnt);
    const int G = pose_lanes(cnt);
    float HE[3][3];
    handeye_rotation(P.handeye, HE); /* the calibration's rotation: once per thread, not once per waypoint */
    {
        const int g = threadIdx.x & (G - 1), per = blockDim.x / G;
        for (int base = 0; base < cnt; base += per) {
            const int t = base + threadIdx.x / G;
            const bool act = t < cnt; /* groups without a waypoint run along (the group shuffles need whole groups) and store nothing */
            double dy = start;
            if (dy_closed_form) dy = start + (double)(act ? t : 0) * P.path_resolution; /* every partial sum is exact: same bits */
            else for (int r = 0; r < (act ? t : 0); ++r) dy += P.path_resolution;       /* the reference accumulates */
            const int iv = gsl_bsearch(mm, dy, Yf);
            /* x: with every knot on the plane all slopes are zero and Steffen's cubic is d + delta * (0 + delta * (0 + delta * 0))
               = the knot value, exactly -- the evaluation (a dozen f64 divisions) is only needed after a dynamic adjustment */
            const double xd = P.knots_on_plane ? Xf(iv) : steffen_eval_at(iv, mm, dy, Yf, Xf);
            const double zd = steffen_eval_at(iv, mm, dy, Yf, Zf);
            /* Vector4f(point) then invTransAlign (identity: Alignment=false); std::reverse on every second slice */
            const int w = off + ((k & 1) ? (cnt - 1 - t) : t);
            float4 q = make_float4((float)xd, (float)dy, (float)zd, 1.f);
            if (ALIGNED) { /* wayPointXYZ = invTransAlign * wayPointXYZ: Matrix4f * Vector4f, the products added left to right */
                const float ax = q.x, ay = q.y, az = q.z;
                q.x = ((back.inv[0][0] * ax + back.inv[0][1] * ay) + back.inv[0][2] * az) + back.inv[0][3] * 1.f;
                q.y = ((back.inv[1][0] * ax + back.inv[1][1] * ay) + back.inv[1][2] * az) + back.inv[1][3] * 1.f;
                q.z = ((back.inv[2][0] * ax + back.inv[2][1] * ay) + back.inv[2][2] * az) + back.inv[2][3] * 1.f;
            }
            STAMP(1, 1); /* dy accumulation + spline */
            float n4[4] = {NAN, NAN, NAN, NAN};
            const bool finite = q.x == q.x && q.y == q.y && q.z == q.z;
            float4 p = make_float4(NAN, NAN, NAN, 0.f);
            int id = -1;
            for (int pass = 0; pass < 2; ++pass) { /* second pass, unbounded: a hole in the cloud (all lanes of a group agree on id) */
                id = nearest_in_slabs_group(G, V, act && finite, q.x, q.y, q.z, &p, g, pass == 0 ? P.nn_hint2 : INFINITY);
                if (__all(id >= 0 || !(act && finite))) break;
            }
            if (P.ranged && act && finite && id >= 0 && g == 0) {
                /* only part of the cloud is indexed: the answer is the whole cloud's as long as the ball that
                   proves the nearest neighbour and the normal's radius search stay inside the indexed interval */
                const float dq = sqrtf(dist2_flann(q.x, q.y, q.z, p.x, p.y, p.z)) * 1.0001f;
                const float need_lo = fminf(q.x - dq, p.x - P.normal_radius * 1.0001f), need_hi = fmaxf(q.x + dq, p.x + P.normal_radius * 1.0001f);
                if ((need_lo < m->incl_lo && m->incl_lo > m->mn[0]) || (need_hi > m->incl_hi && m->incl_hi < m->mx[0])) set_err(m, DERR_MARGIN, s);
            }
            STAMP(1, 2); /* nearest */
            normal_at_point_group(G, V, act && finite && id >= 0, p, P.normal_radius, P.viewpoint, n4, g);
            STAMP(1, 3); /* normal */
            if (act && g == 0) {
                if (id < 0) { n4[0] = n4[1] = n4[2] = n4[3] = NAN; set_err(m, DERR_QUERY, -1); }
                float wp[6], rpy[3];
                pose_from_normal(n4, rpy);
                if (P.change_range) { wp[0] = q.x / 1000; wp[1] = q.y / 1000; wp[2] = q.z / 1000; }
                else { wp[0] = q.x; wp[1] = q.y; wp[2] = q.z; }
                wp[3] = rpy[0]; wp[4] = rpy[1]; wp[5] = rpy[2];
                handeye_apply(HE, P.handeye, wp);
                wp_xyz[w] = q;
                wp_nn[w] = id;
                wp_normal[w] = make_float4(n4[0], n4[1], n4[2], n4[3]);
                for (int d = 0; d < 6; ++d) wp_pre[6 * (size_t)w + d] = wp[d];
            }
            STAMP(1, 4); /* pose + hand-eye */
        }
    }
}

PPP_KERNEL void k_normals_api(DevMeta *m, DevParams P, const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                              const float *__restrict__ slab_xmin, const float *__restrict__ slab_xmax,
                              const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
                              int npts, const int *__restrict__ idx, int k, float *out4)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    int id = idx[t];
    float n4[4] = {NAN, NAN, NAN, NAN};
    if (id >= 0 && id < npts && X[id] == X[id]) {
        float4 p = make_float4(X[id], Y[id], Z[id], __int_as_float(id));
        normal_at_point(V, p, P.normal_radius, P.viewpoint, n4);
    }
    for (int d = 0; d < 4; ++d) out4[4 * (size_t)t + d] = n4[d];
}

/* whole-cloud normal field: one thread per point, walked in slab order so that neighbouring
   threads search the same slabs (L1/L2 friendly); result scattered to cloud index order */
/* The same for the point at position `at` of the slab index -- the whole-cloud field: one thread per point, its neighbours'
   (distance, index) keys and positions in a column of LDS of its own instead of per-thread arrays (which the compiler
   keeps in scratch memory: 592 bytes per lane, every access a trip to L1), its own slab walked from its own position
   outwards, the neighbouring slabs entered through their y-bucket rows.  A neighbourhood beyond NRM_LDS_CAP points is
   summed by scanning again for each next neighbour.  Same sums in the same order as normal_at_point: same bits. */
#ifndef NRM_LDS_CAP
#define NRM_LDS_CAP 12 /* 36 KiB per workgroup: four of them per CU (16: 167 us for a million points, 12: 148, 20: 205) */
#endif
struct NrmLds { u64 key[NRM_LDS_CAP][256]; int pos[NRM_LDS_CAP][256]; };
__device__ inline void normal_at_indexed_point(const SlabView &V, NrmLds &L, const int at, const float4 p, float radius, const float vp[3], float out[4])
{
    const int B = V.m->B, tid = threadIdx.x;
    const float r2 = radius * radius;
    int count = 0;
    auto for_each_neighbour = [&](auto visit) {
        auto scan_from = [&](int s0, int s1, int q0) {
            for (int i = q0; i < s1; ++i) {
                const float4 c = V.at(i);
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                const float d = dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z);
                if (d <= r2) visit(i, c, d);
            }
            for (int i = q0 - 1; i >= s0; --i) {
                const float4 c = V.at(i);
                const float dy = p.y - c.y;
                if (dy * dy > r2) break;
                const float d = dist2_flann(p.x, p.y, p.z, c.x, c.y, c.z);
                if (d <= r2) visit(i, c, d);
            }
        };
        auto scan_slab = [&](int bb) {
            const int s0 = V.slab_start[bb], s1 = V.slab_start[bb + 1];
            if (s0 >= s1) return;
            if (at >= s0 && at < s1) { scan_from(s0, s1, at); return; } /* the point's own slab: any split inside its window will do */
            int lo, hi;
            V.narrow(bb, s0, s1, p.y, lo, hi);
            scan_from(s0, s1, lower_bound_y(V, lo, hi, p.y));
        };
        const int b = slab_of(V.m, p.x);
        scan_slab(b);
        for (int bb = b + 1; bb < B; ++bb) {
            if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
            const float dx = V.slab_xmin[bb] - p.x;
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
        for (int bb = b - 1; bb >= 0; --bb) {
            if (V.slab_start[bb] == V.slab_start[bb + 1]) continue;
            const float dx = p.x - V.slab_xmax[bb];
            if (dx > 0.f && dx * dx > r2) break;
            scan_slab(bb);
        }
    };
    for_each_neighbour([&](int i, const float4 &c, float d) {
        if (count < NRM_LDS_CAP) { L.key[count][tid] = ((u64)__float_as_uint(d) << 32) | (u32)idx_of(c); L.pos[count][tid] = i; }
        count++;
    });
    if (count < 3) { out[0] = out[1] = out[2] = out[3] = NAN; return; }
    float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const float4 &c) {
        const float x = c.x - p.x, y = c.y - p.y, z = c.z - p.z;
        accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
        accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
        accu[6] += x; accu[7] += y; accu[8] += z;
    };
    if (count <= NRM_LDS_CAP) { /* d >= 0: its bit pattern orders like its value; (d, index) keys are all different */
        u64 last = 0;
        for (int r = 0; r < count; ++r) {
            u64 best = ~0ull;
            int bj = 0;
            for (int j = 0; j < count; ++j) {
                const u64 kj = L.key[j][tid];
                if ((r == 0 || kj > last) && kj < best) { best = kj; bj = j; }
            }
            add(V.at(L.pos[bj][tid]));
            last = best;
        }
    } else {
        float ld = -1.f;
        int li = -1;
        for (int r = 0; r < count; ++r) {
            float bd = INFINITY;
            int bi = 0x7fffffff;
            float4 bc = p;
            for_each_neighbour([&](int, const float4 &c, float d) {
                const int id = idx_of(c);
                const bool after = d > ld || (d == ld && id > li);
                if (after && (d < bd || (d == bd && id < bi))) { bd = d; bi = id; bc = c; }
            });
            add(bc);
            ld = bd; li = bi;
        }
    }
    float cnt = (float)count;
    for (int i = 0; i < 9; ++i) accu[i] /= cnt;
    float cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, n[3];
    pcl_eigen33_smallest(cov, &ev, n);
    float eig_sum = cov[0] + cov[4] + cov[8];
    float curv = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
    float vx = vp[0] - p.x, vy = vp[1] - p.y, vz = vp[2] - p.z;
    float cos_theta = vx * n[0] + vy * n[1] + vz * n[2];
    if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2]; out[3] = curv;
}

PPP_KERNEL void __launch_bounds__(256) k_normals_all(DevMeta *m, DevParams P, const float4 *__restrict__ sorted4,
                                                     const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
                                                     const float *__restrict__ slab_xmax, const int *__restrict__ ytab, int nsorted, float4 *out4)
{
    __shared__ NrmLds s_l;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (nsorted < 0 ? m->n_sorted : nsorted)) return;
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0, ytab};
    const float4 p = sorted4[i];
    float n4[4];
    normal_at_indexed_point(V, s_l, i, p, P.normal_radius, P.viewpoint, n4);
    out4[idx_of(p)] = make_float4(n4[0], n4[1], n4[2], n4[3]);
}

PPP_KERNEL void k_nearest_api(DevMeta *m, const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                              const float *__restrict__ slab_xmin, const float *__restrict__ slab_xmax,
                              const float *__restrict__ q, int k, int *out)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    float4 p;
    float qx = q[3 * t], qy = q[3 * t + 1], qz = q[3 * t + 2];
    out[t] = (qx == qx && qy == qy && qz == qz) ? nearest_in_slabs(V, qx, qy, qz, &p) : -1;
}

/* ------------------------------------------------------------------ */
/* a14: reduceRPY (path_translation_alg.cpp:37-86)                      */
/* ------------------------------------------------------------------ */
__device__ inline void rpy_segment(float *W6, int n, int &preId, int tailId, int res, DevMeta *m)
{
    int lastId;
    do {
        double dr[3];
        lastId = preId + res;
        if (lastId >= n) { m->rpy_oob = 1; break; } /* the reference reads past the list here (App. B.6) */
        for (int D = 3; D < 6; D++) {
            float a = W6[6 * (size_t)lastId + D], b = W6[6 * (size_t)preId + D];
            if (a * b >= 0) {
                dr[D - 3] = (double)((a - b) / res);
            } else {
                double no1, no2;
                if (a < 0) { no2 = b; no1 = 2 * M_PI + a; }
                else { no2 = 2 * M_PI + b; no1 = a; }
                dr[D - 3] = (double)fabsf(a - b) < fabs(no1 - no2) ? (double)(a - b) : (no1 - no2);
                dr[D - 3] /= res;
            }
        }
        for (int wi = 1; wi < res; wi++)
            for (int D = 3; D < 6; ++D)
                W6[6 * (size_t)(preId + wi) + D] = (float)(dr[D - 3] + W6[6 * (size_t)(preId + wi - 1) + D]);
        preId = lastId;
    } while ((preId + res) <= tailId);
    if (preId != tailId)
        for (int i = preId + 1; i <= tailId; i++)
            for (int D = 3; D < 6; ++D) W6[6 * (size_t)i + D] = W6[6 * (size_t)preId + D];
}

/* reduceRPY (path_translation_alg.cpp:37-86) for ONE waypoint -- the key waypoints (every RPYres-th of a slice) are
   never modified, so every interval is independent --, then the -180..180 limit (:81-85) and TransFlangeposition
   (:89-112).  p: the waypoint after smoothing (xyz smoothed, rpy as computed); src: any list holding the computed rpy
   of every waypoint (reduceRPY reads the key waypoints' angles).  Valid unless a slice is shorter than RPYres + 1. */
__device__ __forceinline__ void finish_one_waypoint(const DevMeta *m, const DevParams &P, const int *__restrict__ tail,
                                           const float *__restrict__ src, int w, float p[6])
{
#pragma clang fp contract(on) /* continuous output only; the same text runs inside win_finish_body's contracted region */
    const bool reduce = P.rpy_resolution > 2;
    if (reduce) {
        const int res = (int)P.rpy_resolution;
        /* segment of w: first tail >= w */
        int lo = 0, hi = m->nkept - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (tail[mid] < w) lo = mid + 1; else hi = mid; }
        const int tl = tail[lo], p0 = lo == 0 ? 0 : tail[lo - 1] + 1;
        const int nfull = (tl - p0) / res;       /* passes of the do-while */
        const int lastkey = p0 + nfull * res;
        const int r = w - p0;
        if (w > lastkey) {
            for (int D = 3; D < 6; ++D) p[D] = src[6 * (size_t)lastkey + D];
        } else if (r % res != 0) {
            const int pre = p0 + (r / res) * res, last = pre + res, wi = w - pre;
            for (int D = 3; D < 6; D++) {
                float a = src[6 * (size_t)last + D], bb = src[6 * (size_t)pre + D];
                double dr;
                if (a * bb >= 0) {
                    dr = (double)((a - bb) / res);
                } else {
                    double no1, no2;
                    if (a < 0) { no2 = bb; no1 = 2 * M_PI + a; }
                    else { no2 = 2 * M_PI + bb; no1 = a; }
                    dr = (double)fabsf(a - bb) < fabs(no1 - no2) ? (double)(a - bb) : (no1 - no2);
                    dr /= res;
                }
                float v = bb;
                for (int q = 1; q <= wi; ++q) v = (float)(dr + v); /* the reference accumulates in float */
                p[D] = v;
            }
        }
        for (int D = 3; D < 6; ++D) p[D] = (double)p[D] > M_PI ? (float)((double)p[D] - 2 * M_PI) : p[D];
    }
    float R[3][3];
    rot_zyx(p[3], p[4], p[5], R);
    const float ee[3] = {0.f, 0.f, -P.ee_length};
    float t[3];
    for (int i = 0; i < 3; ++i) t[i] = R[i][0] * ee[0] + R[i][1] * ee[1] + R[i][2] * ee[2] + p[i] * 1.f;
    p[0] = t[0]; p[1] = t[1]; p[2] = t[2];
}

/* The same three steps over the whole list in order, by one thread: only when a slice is shorter than RPYres + 1
   waypoints do the reference's segments overlap (App. B.6).  out holds the smoothed list and becomes the final one. */
__device__ inline void finish_list_in_order(DevMeta *m, const DevParams &P, const int *__restrict__ tail, float *out)
{
    const int W = m->W;
    const int res = (int)P.rpy_resolution;
    int preId = 0;
    for (int id = 0; id < m->nkept; ++id) { rpy_segment(out, W, preId, tail[id], res, m); preId = tail[id] + 1; }
    for (int q = 0; q < W; ++q) {
        float p[6];
        for (int d = 0; d < 6; ++d) p[d] = out[6 * (size_t)q + d];
        for (int D = 3; D < 6; ++D) p[D] = (double)p[D] > M_PI ? (float)((double)p[D] - 2 * M_PI) : p[D];
        float R[3][3];
        rot_zyx(p[3], p[4], p[5], R);
        const float ee[3] = {0.f, 0.f, -P.ee_length};
        for (int i = 0; i < 3; ++i) out[6 * (size_t)q + i] = R[i][0] * ee[0] + R[i][1] * ee[1] + R[i][2] * ee[2] + p[i] * 1.f;
        for (int D = 3; D < 6; ++D) out[6 * (size_t)q + D] = p[D];
    }
}

/* ------------------------------------------------------------------ */
/* a13: postion_smooth (path_translation_alg.cpp:114-141), solved directly.                          */
/* The reference sweeps  y_i += 0.65 (x_i - y_i) + 0.35 (y_{i+1} + y_{i-1} - 2 y_i)  (Gauss-Seidel, ends fixed)   */
/* until the list is stationary; what it converges to is, per coordinate, the solution of the        */
/* tridiagonal system  1.35 y_i - 0.35 y_{i-1} - 0.35 y_{i+1} = 0.65 x_i,  y_0 = x_0, y_{W-1} = x_{W-1}          */
/* (SURVEY.md App. A.7).  The matrix is Toeplitz and strongly diagonally dominant, so its inverse is  */
/* a two-sided geometric kernel: with r = (1.35 - sqrt(1.35^2 - 0.7^2)) / 0.7 = 0.2795... the root of  */
/* 0.35 r^2 - 1.35 r + 0.35 = 0 and c = 0.65 / (1.35 - 0.7 r),                                         */
/*   p_i = c * sum_j r^|i-j| x_j   (j over the interior 1..W-2)                                       */
/* satisfies every interior equation, and  y_i = p_i + A r^i + B r^(W-1-i)  with A, B from the two    */
/* fixed ends is THE solution.  r^33 < 1e-18: truncated to 32 neighbours either side the sum is exact */
/* in double precision, so every waypoint is an independent 65-tap filter over its neighbours --      */
/* one launch, no sweeps, no stop rule, no snapshots -- evaluated outside-in (Horner in r), always in */
/* the same order, so the result does not depend on how the list is tiled or sharded.                 */
/* Against the oracle's sequential float sweep (which stops when the float state is stationary,       */
/* DESIGN.md B.12) the difference is the float storage floor: <= 1.2e-7 m measured for W >= 50,        */
/* <= 6e-7 m for the 3..5-waypoint lists where the reference's own 1e-5 test ends the sweeps early.   */
/* The same launch finishes the list per waypoint: reduceRPY, the +-pi limit, the flange offset and,  */
/* in the batched form, the copy into the caller's gather buffer.                                     */
/* ------------------------------------------------------------------ */
#define SM_MAXS 512          /* ppp_params.smooth_max_sweeps is validated against it (the oracle's sweep cap) */
#ifndef SMF_T
#define SMF_T 256            /* waypoints (= threads) per tile */
#endif
#define SMF_K 32             /* filter half-width: r^33 = 5e-19 */
#define SMF_END 128          /* r^128 = 1e-71: beyond that many waypoints from an end its homogeneous term is exactly absorbed */
#define SMF_M (SMF_T + 2 * SMF_K)
__host__ __device__ inline int smooth_tiles(int W) { return W > 0 ? (W + SMF_T - 1) / SMF_T : 1; }

/* r^k by repeated squaring (k < 2^10) */
__device__ inline double smooth_rpow(double r, int k)
{
    double v = 1.0, b = r;
#pragma unroll
    for (int q = 0; q < 10; ++q) { if (k & (1 << q)) v *= b; b *= b; }
    return v;
}

__device__ __forceinline__ void smooth_solve_body(DevMeta *m, const DevParams &P, int W_cap, const float *__restrict__ wp_pre,
                                                  float *wp_smooth, float *wp_out, const int *__restrict__ tail,
                                                  float *dst2, int cap2, const int bx)
{
    __shared__ float s_x[3][SMF_M];
    __shared__ int s_tail[4096];
    __shared__ int s_last;
    __shared__ double s_rp[SMF_END]; /* r^k */
    __shared__ double s_e[2][3];     /* x - p at the two fixed ends */
    __shared__ int s_err;
    const int W = m->W;
    /* the error state is read ONCE per workgroup (thread 0, then a barrier): tile 0 may raise DERR_CAPACITY below while the
       waves of this workgroup are still arriving here, and a wave that saw it would leave before the barriers its siblings wait at */
    if (threadIdx.x == 0) s_err = m->err;
    __syncthreads();
    if (s_err || W == 0) return;
    const int ntiles = smooth_tiles(W);
    const int tile = bx;
    if (tile >= ntiles) return;
    /* weight_data = 0.65, weight_smooth = 1 - weight_data (path_translation_alg.cpp:118): r = 0.27951480..., c = 0.56309250... */
    const double wd = 0.65, ws = 1 - wd, dg = wd + 2 * ws;
    const double r = (dg - sqrt(dg * dg - 4 * ws * ws)) / (2 * ws), c = wd / (dg - 2 * ws * r);
    const int t0 = tile * SMF_T;
    const int base = t0 - SMF_K; /* LDS slot l <-> list index base + l */
    const bool solve = P.smooth && W > 2;
    /* tile + halo; the sources are the INTERIOR waypoints: the two fixed ends enter through A and B */
    for (int l = threadIdx.x; l < SMF_M; l += blockDim.x) {
        const int g = base + l;
        const bool src = g >= 1 && g <= W - 2;
        for (int j = 0; j < 3; ++j) s_x[j][l] = src ? wp_pre[6 * (size_t)g + j] : 0.f;
    }
    const bool in_order = P.rpy_resolution > 2 && m->any_short; /* App. B.6: overlapping segments, finished by one thread below */
    const bool copy2 = dst2 != nullptr && W <= cap2;
    if (dst2 != nullptr && W > cap2 && tile == 0 && threadIdx.x == 0) set_err(m, DERR_CAPACITY, -1);
    /* TailIndex in LDS: every waypoint's segment search is then eight LDS reads instead of eight dependent trips to L2 */
    const int *tl = tail;
    if (!in_order && P.rpy_resolution > 2 && m->nkept <= 4096) {
        for (int i = threadIdx.x; i < m->nkept; i += blockDim.x) s_tail[i] = tail[i];
        tl = s_tail;
    }
    /* tiles within SMF_END waypoints of an end also need that end's homogeneous term: p at the end is a one-sided sum over
       its SMF_K interior neighbours, one lane per neighbour (wave j = coordinate j, lanes 0..31 the front end, 32..63 the back
       end), added by a fixed shuffle tree */
    const bool near_end = solve && (t0 < SMF_END || t0 + SMF_T - 1 > W - 1 - SMF_END);
    if (near_end) {
        if (threadIdx.x < SMF_END) s_rp[threadIdx.x] = smooth_rpow(r, threadIdx.x);
        const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (j < 3) {
            const int k = (lane & 31) + 1;
            const int gq = lane < 32 ? k : W - 1 - k;
            double v = (gq >= 1 && gq <= W - 2) ? (double)wp_pre[6 * (size_t)gq + j] * smooth_rpow(r, k) : 0.0;
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((lane & 31) == 0) s_e[lane >> 5][j] = (double)wp_pre[6 * (size_t)(lane < 32 ? 0 : W - 1) + j] - c * v;
        }
    }
    __syncthreads();
    const int g = t0 + threadIdx.x;
    if (g < W) {
#pragma clang fp contract(on) /* the 65-tap filter and the flange offset: continuous output only (same text as win_finish_body, so a list finished from gathered blocks carries the same bits) */
        float p[6];
        for (int d = 0; d < 6; ++d) p[d] = wp_pre[6 * (size_t)g + d];
        if (solve) {
            const int l = g - base;
            double y[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double acc = (double)s_x[j][l - SMF_K] + (double)s_x[j][l + SMF_K];
#pragma unroll 8
                for (int k = SMF_K - 1; k >= 1; --k) acc = ((double)s_x[j][l - k] + (double)s_x[j][l + k]) + r * acc;
                y[j] = c * ((double)s_x[j][l] + r * acc);
            }
            if (near_end) {
                const double D = W - 1 < SMF_END ? s_rp[W - 1] : 0.0; /* r^(W-1): couples the two ends of a short list */
                const double r0 = g < SMF_END ? s_rp[g] : 0.0, r1 = W - 1 - g < SMF_END ? s_rp[W - 1 - g] : 0.0;
                const double det = 1.0 - D * D;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double e0 = s_e[0][j], e1 = s_e[1][j];
                    const double A = (e0 - D * e1) / det, B = (e1 - D * e0) / det;
                    y[j] = y[j] + (A * r0 + B * r1);
                }
            }
            if (g >= 1 && g <= W - 2) { p[0] = (float)y[0]; p[1] = (float)y[1]; p[2] = (float)y[2]; } /* the ends stay what they are */
        }
        for (int d = 0; d < 6; ++d) wp_smooth[6 * (size_t)g + d] = p[d];
        if (!in_order) finish_one_waypoint(m, P, tl, wp_pre, g, p);
        for (int d = 0; d < 6; ++d) wp_out[6 * (size_t)g + d] = p[d];
        if (copy2 && !in_order) for (int d = 0; d < 6; ++d) dst2[6 * (size_t)g + d] = p[d];
    }
    if (in_order) { /* the last tile to arrive finishes the whole list */
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&m->emit_ticket, 1) == ntiles - 1;
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (threadIdx.x == 0) { finish_list_in_order(m, P, tail, wp_out); m->emit_ticket = 0; }
            __threadfence();
            __syncthreads();
            if (copy2) for (size_t i = threadIdx.x; i < 6 * (size_t)W; i += blockDim.x) dst2[i] = wp_out[i];
        }
    }
    if (tile == 0 && threadIdx.x == 0) { m->sweeps = 0; m->smooth_done = 0; }
}

/* ------------------------------------------------------------------ */
/* Launch forms of the pipeline kernels.                                 */
/* Single: one cloud per launch, blockIdx.x = workgroup.                 */
/* Batched (BASELINE config 3: many small workpieces on ONE GPU): the    */
/* same bodies over ALL members of a batch in one launch per stage --    */
/* blockIdx.y = member, blockIdx.x = that member's workgroup; a member   */
/* with fewer workgroups than the widest one leaves its surplus at once. */
/* Every member is described by a BatchMember record in device memory    */
/* (its handle's buffers, sizes and per-stage grids), read through a     */
/* uniform address (scalar loads).                                       */
/* ------------------------------------------------------------------ */
struct BatchMember {
    DevMeta *m;
    DevParams P;
    const float *X, *Y, *Z;
    int n;
    MinMaxPart *mm_part;
    float slab_x0, slab_invw, incl_lo, incl_hi;
    int B, S_cap, slab_cap, capb, node_cap, W_cap, out2_cap, knot_cap, stage_cap, tab_slabs;
    float pose_pad;
    int g_minmax, g_scatter, g_sort, g_slice, g_pose, g_smooth; /* workgroups of this member per stage */
    int *slab_cnt, *slab_start, *slab_cursor, *coarse_cursor;
    float *px, *lo, *hi;
    float4 *unsorted4, *sorted4;
    float *slab_xmin, *slab_xmax;
    int *big_slabs, *big_slices;
    float *node_x, *node_y, *node_z;
    int *node_start, *node_cnt, *band_cnt;
    int *wp_cnt, *wp_off, *tail;
    float4 *wp_xyz, *wp_normal;
    int *wp_nn;
    float *wp_pre, *wp_smooth, *wp_out, *out2;
    int *ytab, *slice_wpcnt;
};

PPP_KERNEL void __launch_bounds__(SETUP_T) k_setup(DevMeta *m, DevParams P, const MinMaxPart *__restrict__ part, int nparts,
                                               float *px, float *lo, float *hi, int S_cap, int B, int *slab_cnt, float slab_x0,
                                               float slab_invw, int *slab_start, int *slab_cursor, int *coarse_cursor)
{
    setup_body(m, P, part, nparts, px, lo, hi, S_cap, B, slab_cnt, slab_x0, slab_invw, slab_start, slab_cursor, coarse_cursor);
}
/* The one-level scatter with k_setup folded into its launch (clouds below the two-pass size: one launch and ~7 us less per
   pass).  Workgroup `nscat` (one past the scatter workgroups) is the set-up workgroup: bounds, walk, band limits, CSR offsets
   and the meta block for the kernels that follow.  The scatter workgroups do not wait for it: each scans the slab histogram
   for itself in LDS (B <= 4096 counts: a microsecond) and reserves its runs on zero-based per-slab cursors (cleared by
   k_minmax); the slab grid and the kept interval come as arguments instead of from the meta block. */
struct ScatGrid { float x0, invw, xlo, xhi; int B; };
template <int PPT>
__device__ __forceinline__ void slab_scatter_fused_body(const float *__restrict__ X, const float *__restrict__ Y,
                                                        const float *__restrict__ Z, int n, const ScatGrid &G,
                                                        const int *__restrict__ slab_cnt, int *cursor, float4 *out4,
                                                        const int *__restrict__ idmap, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) int s_hist[];
    __shared__ int s_scan[17];
    const int B = G.B;
    int *s_start = s_hist + B;
    const int i0 = bx * (PPT * (int)blockDim.x);
    float4 p[PPT];
    int pb[PPT]; /* slab, or -1: not mine / dropped */
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = i0 + threadIdx.x + k * (int)blockDim.x;
        pb[k] = -1;
        if (i < n) p[k] = make_float4(X[i], Y[i], Z[i], __int_as_float(idmap ? idmap[i] : i));
        else p[k] = make_float4(NAN, 0.f, 0.f, 0.f);
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) { s_hist[b] = 0; s_start[b] = slab_cnt[b]; }
    __syncthreads();
    {   /* exclusive scan of the histogram: this workgroup's own copy of slab_start */
        const int per = (B + blockDim.x - 1) / blockDim.x;
        const int b0 = threadIdx.x * per;
        int sum = 0;
        for (int k = 0; k < per; ++k) if (b0 + k < B) sum += s_start[b0 + k];
        int total;
        int pre = block_exscan(sum, s_scan, &total);
        for (int k = 0; k < per; ++k) if (b0 + k < B) { const int c = s_start[b0 + k]; s_start[b0 + k] = pre; pre += c; }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const float x = p[k].x;
        if (x >= G.xlo && x <= G.xhi) { pb[k] = slab_of_grid(x, G.x0, G.invw, B); atomicAdd(&s_hist[pb[k]], 1); } /* NaN fails both */
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int c = s_hist[b];
        if (c) s_hist[b] = s_start[b] + atomicAdd(&cursor[b], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (pb[k] >= 0) out4[atomicAdd(&s_hist[pb[k]], 1)] = p[k];
}
template <int PPT>
__global__ void __launch_bounds__(SCAT_T) k_scatter_setup(const float *__restrict__ X, const float *__restrict__ Y,
                                                          const float *__restrict__ Z, int n, ScatGrid G, int *slab_cnt, int *cursor,
                                                          float4 *out4, const int *idmap, int nscat,
                                                          DevMeta *m, DevParams P, const MinMaxPart *__restrict__ part, int nparts, float *px,
                                                          float *lo, float *hi, int S_cap, int *slab_start)
{
    if ((int)blockIdx.x == nscat) {
        setup_body(m, P, part, nparts, px, lo, hi, S_cap, G.B, slab_cnt, G.x0, G.invw, slab_start, nullptr, nullptr, false);
        return;
    }
    slab_scatter_fused_body<PPT>(X, Y, Z, n, G, slab_cnt, cursor, out4, idmap, blockIdx.x);
}
template <int LEVEL, int PPT>
__global__ void __launch_bounds__(SCAT_T) k_slab_scatter(const float *__restrict__ X, const float *__restrict__ Y,
                                                      const float *__restrict__ Z, const float4 *__restrict__ in4, int n,
                                                      const DevMeta *m, int *cursor, float4 *out4, const int *idmap)
{
    slab_scatter_body<LEVEL, PPT>(X, Y, Z, in4, n, m, cursor, out4, idmap, blockIdx.x);
}
template <bool ARENA>
__global__ void __launch_bounds__(SORT_T) k_slab_sort(const float4 *__restrict__ unsorted4, const int *__restrict__ slab_start,
                                                   float4 *sorted4, float *slab_xmin, float *slab_xmax, DevMeta *m, int cap,
                                                   int *big_list, char *arena, unsigned long long arena_cap, int *ytab, int first_slab,
                                                   int *slab_cnt)
{
    slab_sort_body<ARENA>(unsorted4, slab_start, sorted4, slab_xmin, slab_xmax, m, cap, big_list, arena, arena_cap, ytab, slab_cnt,
                          ARENA ? (int)blockIdx.x : first_slab + (int)blockIdx.x);
}
template <bool ARENA>
__global__ void __launch_bounds__(SLICE_KD_T) k_slice_kd(const float4 *__restrict__ sorted4, const int *__restrict__ slab_start,
                                                  DevMeta *m, const float *__restrict__ px, const float *__restrict__ lo,
                                                  const float *__restrict__ hi, int capb_lds, float *node_x, float *node_y,
                                                  float *node_z, int node_cap, int *node_start, int *node_cnt, int *band_cnt, int *big_list,
                                                  char *arena, unsigned long long arena_cap, double trim, double res, int W_cap, int *slice_wpcnt)
{
    slice_kd_body<ARENA>(sorted4, slab_start, m, px, lo, hi, capb_lds, node_x, node_y, node_z, node_cap, node_start, node_cnt, band_cnt,
                         big_list, arena, arena_cap, trim, res, W_cap, slice_wpcnt, blockIdx.x);
}
/* TMAX: the most threads a launch uses (256, 512 or POSE_T): the register budget follows from it -- the 1024-thread form is
   held to 128 VGPRs and spills a few values, the smaller forms are not */
template <bool ALIGNED, int TMAX>
__global__ void __launch_bounds__(TMAX) k_pose(DevMeta *m, DevParams P, const float4 *__restrict__ sorted4,
                                              const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
                                              const float *__restrict__ slab_xmax, const float *__restrict__ px,
                                              const float *__restrict__ node_x, const float *__restrict__ node_y,
                                              const float *__restrict__ node_z,
                                              const int *__restrict__ node_start, const int *__restrict__ node_cnt,
                                              int *wp_cnt, int *wp_off, int *tail, int W_cap, int arena_ran, int knot_cap, int stage_cap,
                                              int tab_slabs, float pad, float4 *wp_xyz, int *wp_nn, float4 *wp_normal, float *wp_pre, PoseBack back, const int *ytab,
                                              const int *slice_wpcnt)
{
    pose_body<ALIGNED, (TMAX <= 768 ? POSE_PRE : 0)>(m, P, sorted4, slab_start, slab_xmin, slab_xmax, px, node_x, node_y, node_z, node_start, node_cnt, wp_cnt, wp_off,
                       tail, W_cap, arena_ran, knot_cap, stage_cap, tab_slabs, pad, wp_xyz, wp_nn, wp_normal, wp_pre, back, ytab, slice_wpcnt, blockIdx.x);
}
PPP_KERNEL void __launch_bounds__(SMF_T) k_smooth_solve(DevMeta *m, DevParams P, int W_cap, const float *__restrict__ wp_pre,
                                                        float *wp_smooth, float *wp_out, const int *__restrict__ tail,
                                                        float *dst2, int cap2)
{
    smooth_solve_body(m, P, W_cap, wp_pre, wp_smooth, wp_out, tail, dst2, cap2, blockIdx.x);
}

/* ---- batched forms ---- */
PPP_KERNEL void __launch_bounds__(MM_T) k_minmax_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x >= M.g_minmax) return;
    minmax_body<true>(M.X, M.Y, M.Z, M.n, M.mm_part, M.slab_x0, M.slab_invw, M.B, M.slab_cnt, M.incl_lo, M.incl_hi, M.slab_cursor, blockIdx.x, M.g_minmax);
}
PPP_KERNEL void __launch_bounds__(SETUP_T) k_setup_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    setup_body(M.m, M.P, M.mm_part, M.g_minmax, M.px, M.lo, M.hi, M.S_cap, M.B, M.slab_cnt, M.slab_x0, M.slab_invw, M.slab_start,
               M.slab_cursor, M.coarse_cursor);
}
template <int PPT>
__global__ void __launch_bounds__(SCAT_T) k_slab_scatter_b(const BatchMember *__restrict__ mem)
{   /* the fused form (k_scatter_setup): workgroup g_scatter of every member is its set-up workgroup */
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x > M.g_scatter) return;
    ScatGrid G;
    G.x0 = M.slab_x0; G.invw = M.slab_invw; G.xlo = M.incl_lo; G.xhi = M.incl_hi; G.B = M.B;
    if ((int)blockIdx.x == M.g_scatter) {
        setup_body(M.m, M.P, M.mm_part, M.g_minmax, M.px, M.lo, M.hi, M.S_cap, M.B, M.slab_cnt, M.slab_x0, M.slab_invw, M.slab_start,
                   nullptr, nullptr, false);
        return;
    }
    slab_scatter_fused_body<PPT>(M.X, M.Y, M.Z, M.n, G, M.slab_cnt, M.slab_cursor, M.unsorted4, nullptr, blockIdx.x);
}
PPP_KERNEL void __launch_bounds__(SORT_T) k_slab_sort_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x >= M.g_sort) return;
    slab_sort_body<false>(M.unsorted4, M.slab_start, M.sorted4, M.slab_xmin, M.slab_xmax, M.m, M.slab_cap, M.big_slabs, nullptr, 0ull, M.ytab, M.slab_cnt, blockIdx.x);
}
PPP_KERNEL void __launch_bounds__(SLICE_KD_T) k_slice_kd_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x >= M.g_slice) return;
    slice_kd_body<false>(M.sorted4, M.slab_start, M.m, M.px, M.lo, M.hi, M.capb, M.node_x, M.node_y, M.node_z, M.node_cap, M.node_start,
                         M.node_cnt, M.band_cnt, M.big_slices, nullptr, 0ull, M.P.trim, M.P.path_resolution, M.W_cap, M.slice_wpcnt, blockIdx.x);
}
template <int TMAX>
__global__ void __launch_bounds__(TMAX) k_pose_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x >= M.g_pose) return;
    PoseBack none;
    none.sorted4 = nullptr; none.slab_start = nullptr; none.slab_xmin = nullptr; none.slab_xmax = nullptr; none.m = nullptr; none.ytab = nullptr;
    pose_body<false, (TMAX <= 768 ? POSE_PRE : 0)>(M.m, M.P, M.sorted4, M.slab_start, M.slab_xmin, M.slab_xmax, M.px, M.node_x, M.node_y, M.node_z, M.node_start, M.node_cnt,
                     M.wp_cnt, M.wp_off, M.tail, M.W_cap, 0, M.knot_cap, M.stage_cap, M.tab_slabs, M.pose_pad, M.wp_xyz, M.wp_nn, M.wp_normal, M.wp_pre, none, M.ytab, M.slice_wpcnt, blockIdx.x);
}
PPP_KERNEL void __launch_bounds__(SMF_T) k_smooth_solve_b(const BatchMember *__restrict__ mem)
{
    const BatchMember &M = mem[blockIdx.y];
    if ((int)blockIdx.x >= M.g_smooth) return;
    smooth_solve_body(M.m, M.P, M.W_cap, M.wp_pre, M.wp_smooth, M.wp_out, M.tail, M.out2, M.out2_cap, blockIdx.x);
}
/* the members' meta blocks side by side, so that ONE copy publishes the batch to the host */
PPP_KERNEL void __launch_bounds__(64) k_collect_meta(const BatchMember *__restrict__ mem, int count, DevMeta *out)
{
    const int i = blockIdx.x;
    if (i >= count) return;
    const int *src = (const int *)mem[i].m;
    int *dst = (int *)(out + i);
    for (int q = threadIdx.x; q < (int)(sizeof(DevMeta) / sizeof(int)); q += blockDim.x) dst[q] = src[q];
}
