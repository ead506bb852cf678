/*
 * ppp_window.h -- the WINDOW path of the hot path: three launches instead of six.
 *
 *   k_win_scatter   every point is read ONCE (12 B): bounds (a2) + binning straight into the window of the slice it can
 *                   matter to -- [Px - pad, Px + pad], pad = band half-width / nearest-neighbour ball / normal radius;
 *                   the ~1/3 of the cloud between two windows is never written.  Plan-sized window capacity, a count per
 *                   window (overflow -> the run falls back to the slab-index path), no separate histogram pass.
 *   k_win_slice     one workgroup per slice: the window staged once into LDS and bucket-sorted there on (class, y, index)
 *                   (classes = the x intervals left of the band | Er | on the plane | El | right of the band); on the same
 *                   staged points: rangedX_index + insert_point (path_slicing_alg.cpp:152-237), OnePath's map flattening
 *                   (:240-267), getPath's sampling (path_translation_alg.cpp:156-169), nearest point + PCL normal + pose +
 *                   HandEyeTransform (:178-211).  The bucket table of the sort doubles as the y index of every search.
 *                   Waypoints go to per-slice slots; one more workgroup re-derives the slice walk (a3) from the bounds
 *                   and checks the plan it was launched with.
 *   k_win_finish    scans the per-slice waypoint counts, compacts the list and runs postion_smooth / reduceRPY /
 *                   TransFlangeposition (:212-214) as k_smooth_solve does.
 *
 * The slab index (k_minmax .. k_slab_sort, ppp_kernels.h) stays: it serves the API mirrors, the dynamic adjustment, brute
 * pairing, alignment, overlapping windows (small tools) and every run the window path hands back (DevMeta.win_flag).
 *
 * Speculation, and how it is checked: the launches are sized by the PLAN (slice positions from the bounds cached when the
 * cloud was set -- the same values the slab path's grid comes from).  The bounds and the walk are recomputed in every
 * pass (a2, a3 stay inside the timed region) and compared with the plan; a difference raises WIN_FLAG_STALE and the run is
 * repeated on the slab path.  Every nearest-neighbour ball and normal neighbourhood is checked to lie inside its window.
 */
#pragma once
#include "ppp_kernels.h"
#include "ppp_window_decl.h"

/* ------------------------------------------------------------------ */
/* launch 1: bounds + window binning                                    */
/* ------------------------------------------------------------------ */
template <int PPT, bool STAGED>
__device__ __forceinline__ void win_scatter_body(const WinArgs &A, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    float *s_px = (float *)s_dyn;       /* S plane positions; STAGED: later the windows' places in the stage */
    int *s_cnt = s_dyn + A.S;           /* S counts, then bases */
    __shared__ int s_scr[17];
    __shared__ float s_mn[3][WSC_T / 64], s_mx[3][WSC_T / 64];
    __shared__ int s_n[WSC_T / 64];
    const int S = A.S, n = A.n;
    const int i0 = bx * (PPT * (int)blockDim.x);
    STAMP_BEGIN();
    float4 p[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = i0 + threadIdx.x + k * (int)blockDim.x;
        if (i < n) p[k] = make_float4(A.X[i], A.Y[i], A.Z[i], __int_as_float(A.idmap ? A.idmap[i] : i));
        else p[k] = make_float4(NAN, 0.f, 0.f, 0.f);
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) { s_px[s] = A.plan_px[s]; s_cnt[s] = 0; }
    if (bx == 0 && threadIdx.x == 0) { /* the run state of this pass: nothing else in this launch touches the meta block */
        DevMeta *m = A.m;
        m->W = 0; m->err = 0; m->err_slice = 0x7fffffff; m->sweeps = 0; m->any_short = 0; m->rpy_oob = 0;
        m->node_cursor = 0; m->smooth_done = -1; m->emit_ticket = 0; m->big_slabs = 0; m->big_slices = 0; m->arena_cursor = 0; m->win_flag = 0;
        /* (the finish launch's arrival counters: its last workgroup clears them; a pass that was cut short must not leave them counting) */
        if (A.fin_ticket) for (int q = 0; q <= WIN_FIN_GROUPS; ++q) A.fin_ticket[q] = 0;
    }
    __syncthreads();
    STAMP(7, 0); /* loads issued, plane table staged */
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    int pw[PPT], pr[PPT]; /* window (or -1) and rank inside this workgroup's run */
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const float x = p[k].x;
        pw[k] = -1; pr[k] = 0;
        if (x == x) {
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], p[k].y); mx[1] = fmaxf(mx[1], p[k].y);
            mn[2] = fminf(mn[2], p[k].z); mx[2] = fmaxf(mx[2], p[k].z);
            cnt++;
            /* the slice whose plane is nearest: within one of the lattice guess (the planes are `step` apart up to the
               rounding of the float walks and the off-lattice centre plane of the centre-out integer walk) */
            const float fj = fminf(fmaxf(floorf((x - s_px[0]) * A.inv_step + 0.5f), 0.f), (float)(S - 1)); /* (the table's own first plane: a pass may run on a plan made before this cloud's bounds were known) */
            const int j = (int)fj;
            const int ja = j > 0 ? j - 1 : j, jb = j + 1 < S ? j + 1 : j;
            const float da = fabsf(x - s_px[ja]), dj = fabsf(x - s_px[j]), db = fabsf(x - s_px[jb]);
            int w = -1;
            if (dj <= A.pad) w = j; else if (da <= A.pad) w = ja; else if (db <= A.pad) w = jb; /* windows are disjoint (plan) */
            if (w >= A.sb && w < A.se) { pw[k] = w; pr[k] = atomicAdd(&s_cnt[w], 1); }
        }
    }
    __syncthreads();
    STAMP(7, 1); /* points arrived, windows found, ranks from the LDS counters */
    if (!STAGED) {
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            const int c = s_cnt[s];
            if (c) s_cnt[s] = atomicAdd(&A.win_cnt[(size_t)s * WIN_CNT_STRIDE], c); /* this workgroup's run inside the window */
        }
        __syncthreads();
        STAMP(7, 2); /* runs reserved (global atomics) */
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (pw[k] >= 0) {
                const int pos = s_cnt[pw[k]] + pr[k];
                if (pos < A.capw) A.win_pts[(size_t)pw[k] * A.capw + pos] = p[k]; /* beyond: the slice sees count > capw and hands the run back */
            }
        }
    } else {
        int *s_loc = (int *)s_px;                 /* the planes are used up: the windows' first places in the stage */
        int *s_gb = s_dyn + 2 * S;                /* ... and in their global windows */
        float4 *stage = (float4 *)(s_dyn + ((3 * S + 3) & ~3));
        u16 *widx = (u16 *)(stage + PPT * (int)blockDim.x);
        int K;
        {
            const int per = (S + (int)blockDim.x - 1) / (int)blockDim.x;
            const int b0 = threadIdx.x * per;
            int sum = 0;
            for (int q = 0; q < per; ++q) if (b0 + q < S) sum += s_cnt[b0 + q];
            int pre = block_exscan_w(sum, s_scr, &K);
            for (int q = 0; q < per; ++q) {
                if (b0 + q < S) {
                    const int c = s_cnt[b0 + q];
                    s_loc[b0 + q] = pre; pre += c;
                    s_gb[b0 + q] = c ? atomicAdd(&A.win_cnt[(size_t)(b0 + q) * WIN_CNT_STRIDE], c) : 0;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (pw[k] >= 0) {
                const int q = s_loc[pw[k]] + pr[k];
                stage[q] = p[k]; widx[q] = (u16)pw[k];
            }
        }
        __syncthreads();
        for (int q = threadIdx.x; q < K; q += blockDim.x) {
            const int w = widx[q];
            const int pos = s_gb[w] + (q - s_loc[w]);
            if (pos < A.capw) A.win_pts[(size_t)w * A.capw + pos] = stage[q];
        }
    }
    STAMP(7, 3); /* stores issued */
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_n[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPart r;
        r.cnt = 0; r.pad = 0;
        const int nw = (int)(blockDim.x >> 6);
        for (int w = 0; w < nw; ++w) r.cnt += s_n[w];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < nw; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            r.mn[d] = a; r.mx[d] = b;
        }
        A.win_part[bx] = r;
    }
    STAMP(7, 4); /* bounds partial */
}

/* A workgroup barrier for phases that meet through LDS only: waits for this wave's LDS operations, not for its loads from
   memory (vmcnt counts loads and stores alike on this target, so __syncthreads() would sit out the prefetch below). */
__device__ __forceinline__ void win_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* The staged binning of a large cloud as a LOOP: a workgroup takes chunks bx, bx + nwg, ... of PPT x blockDim points.  Its LDS
   stage admits one workgroup per CU, whose phases -- points from memory, ranks, reservations, stage, stores -- used to follow one
   another with the memory pipes idle in between; here the next chunk's points are requested as soon as the reservations of the
   current one have come back and travel while it is staged and stored.  Same runs, same counters: the windows only see another
   arrival order (the slice kernel orders a window by (y, index) itself). */
template <int PPT>
__device__ __forceinline__ void win_scatter_staged_loop(const WinArgs &A, const int bx, const int nwg)
{
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    float *s_px = (float *)s_dyn;
    int *s_cnt = s_dyn + A.S;
    __shared__ int s_scr[17];
    __shared__ float s_mn[3][WSC_T / 64], s_mx[3][WSC_T / 64];
    __shared__ int s_n[WSC_T / 64];
    const int S = A.S, n = A.n, T = (int)blockDim.x;
    const int chunk_pts = PPT * T;
    const int nchunks = (n + chunk_pts - 1) / chunk_pts;
    int *s_loc = (int *)s_px;
    int *s_gb = s_dyn + 2 * S;
    float4 *stage = (float4 *)(s_dyn + ((3 * S + 3) & ~3));
    u16 *widx = (u16 *)(stage + PPT * T);
    if (bx == 0 && threadIdx.x == 0) { /* the run state of this pass: nothing else in this launch touches the meta block */
        DevMeta *m = A.m;
        m->W = 0; m->err = 0; m->err_slice = 0x7fffffff; m->sweeps = 0; m->any_short = 0; m->rpy_oob = 0;
        m->node_cursor = 0; m->smooth_done = -1; m->emit_ticket = 0; m->big_slabs = 0; m->big_slices = 0; m->arena_cursor = 0; m->win_flag = 0;
        /* (the finish launch's arrival counters: its last workgroup clears them; a pass that was cut short must not leave them counting) */
        if (A.fin_ticket) for (int q = 0; q <= WIN_FIN_GROUPS; ++q) A.fin_ticket[q] = 0;
    }
    float4 p[PPT], pn[PPT];
    auto request = [&](int chunk, float4 *dst) {
        const int i0 = chunk * chunk_pts;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = i0 + (int)threadIdx.x + k * T;
            if (i < n) dst[k] = make_float4(A.X[i], A.Y[i], A.Z[i], __int_as_float(A.idmap ? A.idmap[i] : i));
            else dst[k] = make_float4(NAN, 0.f, 0.f, 0.f);
        }
    };
    if (bx < nchunks) request(bx, p);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int chunk = bx; chunk < nchunks; chunk += nwg) {
        for (int s = threadIdx.x; s < S; s += T) { s_px[s] = A.plan_px[s]; s_cnt[s] = 0; }
        __syncthreads();
        int pw[PPT], pr[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float x = p[k].x;
            pw[k] = -1; pr[k] = 0;
            if (x == x) {
                mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
                mn[1] = fminf(mn[1], p[k].y); mx[1] = fmaxf(mx[1], p[k].y);
                mn[2] = fminf(mn[2], p[k].z); mx[2] = fmaxf(mx[2], p[k].z);
                cnt++;
                const float fj = fminf(fmaxf(floorf((x - s_px[0]) * A.inv_step + 0.5f), 0.f), (float)(S - 1));
                const int j = (int)fj;
                const int ja = j > 0 ? j - 1 : j, jb = j + 1 < S ? j + 1 : j;
                const float da = fabsf(x - s_px[ja]), dj = fabsf(x - s_px[j]), db = fabsf(x - s_px[jb]);
                int w = -1;
                if (dj <= A.pad) w = j; else if (da <= A.pad) w = ja; else if (db <= A.pad) w = jb; /* windows are disjoint (plan) */
                if (w >= A.sb && w < A.se) { pw[k] = w; pr[k] = atomicAdd(&s_cnt[w], 1); }
            }
        }
        __syncthreads();
        int K;
        {
            const int per = (S + T - 1) / T;
            const int b0 = threadIdx.x * per;
            int sum = 0;
            for (int q = 0; q < per; ++q) if (b0 + q < S) sum += s_cnt[b0 + q];
            int pre = block_exscan_w(sum, s_scr, &K);
            for (int q = 0; q < per; ++q) {
                if (b0 + q < S) {
                    const int c = s_cnt[b0 + q];
                    s_loc[b0 + q] = pre; pre += c;
                    s_gb[b0 + q] = c ? atomicAdd(&A.win_cnt[(size_t)(b0 + q) * WIN_CNT_STRIDE], c) : 0;
                }
            }
        }
        __syncthreads(); /* (the reservations have come back: nothing below waits on memory) */
        const int next = chunk + nwg;
        if (next < nchunks) request(next, pn);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (pw[k] >= 0) {
                const int q = s_loc[pw[k]] + pr[k];
                stage[q] = p[k]; widx[q] = (u16)pw[k];
            }
        }
        win_lds_barrier();
        for (int q = threadIdx.x; q < K; q += T) {
            const int w = widx[q];
            const int pos = s_gb[w] + (q - s_loc[w]);
            if (pos < A.capw) A.win_pts[(size_t)w * A.capw + pos] = stage[q];
        }
        win_lds_barrier(); /* the tables and the stage are free for the next chunk */
#pragma unroll
        for (int k = 0; k < PPT; ++k) p[k] = pn[k];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_n[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPart r;
        r.cnt = 0; r.pad = 0;
        const int nw = (int)(blockDim.x >> 6);
        for (int w = 0; w < nw; ++w) r.cnt += s_n[w];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < nw; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            r.mn[d] = a; r.mx[d] = b;
        }
        A.win_part[bx] = r;
    }
}

/* ------------------------------------------------------------------ */
/* launch 2: the per-slice kernel                                       */
/* ------------------------------------------------------------------ */
struct WinView {
    const float4 *pts;
    const int *tab; /* tab[b] = first place of bucket b (tab[NB] = the window's population) */
    int NBc;
    float y0, yscale;
    const int *cs; /* LDS: class c = pts[cs[c], cs[c + 1]) (a private array indexed by a run-time class would live in scratch memory) */
    __device__ inline int ybucket(float y) const
    {
        int q = (int)((y - y0) * yscale);
        q = q < 0 ? 0 : q;
        return q >= NBc ? NBc - 1 : q;
    }
    /* first position of class c whose y >= qy (cs[c + 1] when there is none): one table look-up, then the bucket's few points */
    __device__ inline int lower_bound(int c, float qy) const
    {
        const int b = c * NBc + ybucket(qy);
        int lo = tab[b];
        int hi = tab[b + 1];
        while (hi - lo > 4) { const int mid = (lo + hi) >> 1; if (pts[mid].y < qy) lo = mid + 1; else hi = mid; } /* (dense windows: buckets of a dozen points) */
        while (lo < hi && pts[lo].y < qy) ++lo;
        return lo;
    }
};

/* one direction of one class: candidates in the reference's order of decreasing promise, four reads in flight */
template <typename Visit>
__device__ __forceinline__ void win_walk(const WinView &V, int c, int p, bool up, Visit visit)
{
    const int s0 = V.cs[c], s1 = V.cs[c + 1];
    if (up) {
        for (int i = p; i < s1; i += 4) {
            const int e = s1 - 1;
            const float4 c0 = V.pts[i], c1 = V.pts[min(i + 1, e)], c2 = V.pts[min(i + 2, e)], c3 = V.pts[min(i + 3, e)];
            if (!visit(c0, i)) break;
            if (i + 1 > e || !visit(c1, i + 1)) break;
            if (i + 2 > e || !visit(c2, i + 2)) break;
            if (i + 3 > e || !visit(c3, i + 3)) break;
        }
    } else {
        for (int i = p - 1; i >= s0; i -= 4) {
            const float4 c0 = V.pts[i], c1 = V.pts[max(i - 1, s0)], c2 = V.pts[max(i - 2, s0)], c3 = V.pts[max(i - 3, s0)];
            if (!visit(c0, i)) break;
            if (i - 1 < s0 || !visit(c1, i - 1)) break;
            if (i - 2 < s0 || !visit(c2, i - 2)) break;
            if (i - 3 < s0 || !visit(c3, i - 3)) break;
        }
    }
}

/* kd-tree 1-NN of q inside ONE class (insert_point's per-side trees): ties -> lowest cloud index.  Returns the position. */
__device__ inline int win_nn_class(const WinView &V, int c, const float4 q)
{
    float best = INFINITY;
    int bidx = 0x7fffffff, bj = V.cs[c];
    auto visit = [&](const float4 &k, int i) {
        const float dy = q.y - k.y;
        if (dy * dy > best) return false;
        const float d = dist2_flann(q.x, q.y, q.z, k.x, k.y, k.z);
        const int id = idx_of(k);
        if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bj = i; }
        return true;
    };
    const int p = V.lower_bound(c, q.y);
    win_walk(V, c, p, true, visit);
    win_walk(V, c, p, false, visit);
    return bj;
}

/* insert_point's brute-force argmin (Path_Generation.cpp:141-150 / :163-170) of q inside ONE class: the key is Vector3f::norm()
   -- sqrt in float, Eigen's x*x + (y*y + z*z) -- and `compare[norm] = j` lets the LAST j of equal keys win: the highest cloud
   index (the sides are walked in ascending cloud index).  Windowed on the class's y order: norm >= |dy| holds in float arithmetic
   (every partial sum is >= y*y rounded, sqrt is monotone and sqrtf(fl(y*y)) >= |y|), so a candidate beyond |dy| > best cannot tie. */
__device__ inline int win_nn_class_brute(const WinView &V, int c, const float4 q)
{
    float best = INFINITY;
    int bidx = -1, bj = V.cs[c];
    auto visit = [&](const float4 &k, int i) {
        const float dy = q.y - k.y;
        if (fabsf(dy) > best) return false;
        const float d = norm_eigen3(q.x - k.x, dy, q.z - k.z);
        const int id = idx_of(k);
        if (d < best || (d == best && id > bidx)) { best = d; bidx = id; bj = i; }
        return true;
    };
    const int p = V.lower_bound(c, q.y);
    win_walk(V, c, p, true, visit);
    win_walk(V, c, p, false, visit);
    return bj;
}

/* Units of a waypoint's two searches: (class, direction) -- the two band sides (up, down), the two outer classes (up, down),
   the points on the plane.  A waypoint has G = 4 lanes (a launch whose workgroups have a CU to themselves: the band sides first,
   the outer classes second -- the nearest-neighbour search has closed those by their x gap by then, almost always) or G = 2 (one
   class per round: where it saves a whole pass over the slice's waypoints).  Partial sums are combined by one fixed tree, ((Er_up + Er_down) + (El_up + El_down)) + ((L_up + L_down) + (R_up + R_down)),
   then + (plane_up + plane_down): the same bits for either G. */
__device__ __forceinline__ int win_unit_class(int G, int round, int g)
{
    if (G == 2) return round == 0 ? 1 : (round == 1 ? 3 : (round == 2 ? 0 : (round == 3 ? 4 : 2)));
    return round == 0 ? (g < 2 ? 1 : 3) : (round == 1 ? (g < 2 ? 0 : 4) : 2);
}

/* diagnostic builds (tools/phase_costs.sh): the slice kernel leaves after phase WIN_STOP_AFTER, so that the difference of two
   builds' durations and instruction counters is that phase's bill.  `sink` keeps the phase's register results alive.  Never
   defined in the product. */
#ifdef WIN_STOP_AFTER
#define WIN_STOP(slot, sink) do { if (WIN_STOP_AFTER == (slot)) { A.wps_nn[tid] = (int)(sink); \
        if ((slot) < 5 && tid == 0) { A.node_start[s] = 0; A.node_cnt[s] = 0; if (kept) A.wp_cnt[k] = 0; } return; } } while (0)
#else
#define WIN_STOP(slot, sink) do { } while (0)
#endif

template <int TMAX>
__device__ __forceinline__ void win_slice_body(const WinArgs &A, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_cs[WIN_CLASSES + 1];
    __shared__ int s_m, s_base, s_nk, s_cnt;
    DevMeta *m = A.m;
    const DevParams &P = A.P;
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int capw = A.capw, cap_el = A.cap_el, NB = A.NB, NBc = A.NBc;
    float4 *pts = (float4 *)s_raw;
    int *tab = (int *)(pts + capw);
    char *scr = (char *)tab + (((size_t)NB + 1) * 4 + 15) / 16 * 16;
    float *cz = (float *)scr;                 /* z of the node candidate of the i-th El point */
    u64 *ckeys = (u64 *)(cz + cap_el);        /* candidates sorted by y; later the knots as (y, z) pairs */
    int *hc = (int *)(ckeys + cap_el);        /* histogram of that sort, then the kept candidates */
    float2 *knot = (float2 *)ckeys;

    const int s = A.sb + bx;
    if (s >= A.se) return;
    const int k = s - A.first_kept; /* index among the kept slices (getPath drops the first and the last, :149-150) */
    const bool kept = k >= 0 && k < A.nkept;
    const float Px = A.plan_px[s];
    const int position = (int)Px;   /* rangedX_index(int position), path_slicing_alg.cpp:152,247 */
    const float blo = (float)(-2 + position), bhi = (float)(2 + position);
    auto slice_fails = [&](int code) { /* code < 0: not the slice's fault -- hand the pass back (-code = WIN_FLAG_*) */
        if (tid == 0) {
            if (code < 0) win_flag(m, -code); else set_err(m, code, s);
            A.node_start[s] = 0; A.node_cnt[s] = 0; if (kept) A.wp_cnt[k] = 0;
        }
    };
    /* ---- stage the window: points to registers, class + bucket, bucket sort into LDS ---- */
    STAMP_BEGIN();
    /* the window's points are requested together with its count (slots beyond the count hold leftovers of earlier passes and
       are ignored below): one round trip to memory instead of two */
    float4 pr[WIN_EMAX];
    int pb[WIN_EMAX];
    const float4 *src = A.win_pts + (size_t)s * capw;
    const int n = A.win_cnt[(size_t)s * WIN_CNT_STRIDE];
#pragma unroll
    for (int e = 0; e < WIN_EMAX; ++e) {
        const int i = tid + e * T;
        pb[e] = -1;
        if (i < capw) pr[e] = src[i];
    }
    __syncthreads(); /* every thread has read the count ... */
    if (tid == 0) A.win_cnt[(size_t)s * WIN_CNT_STRIDE] = 0; /* ... this workgroup is its only reader: cleared for the next pass */
    if (n > capw) { slice_fails(-WIN_FLAG_OVERFLOW); return; }
    for (int b = tid; b <= NB; b += T) tab[b] = 0;
    __syncthreads();
    STAMP(6, 0); /* window load issued, table cleared */
    WIN_STOP(0, __float_as_int(pr[0].x) ^ __float_as_int(pr[WIN_EMAX - 1].y));
#pragma unroll
    for (int e = 0; e < WIN_EMAX; ++e) {
        const int i = tid + e * T;
        if (i < n) {
            const float4 p = pr[e];
            int c;
            if (p.x < blo) c = 0;
            else if (p.x > bhi) c = 4;
            else {
                const float distance2plane = (p.x - Px) * 1.f + (p.y - 0.f) * 0.f + (p.z - 0.f) * 0.f; /* path_slicing_alg.cpp:179 */
                c = distance2plane > 0 ? 3 : (distance2plane < 0 ? 1 : 2);
            }
            int q = (int)((p.y - A.y0) * A.yscale);
            q = q < 0 ? 0 : (q >= NBc ? NBc - 1 : q);
            const int b = c * NBc + q;
            pb[e] = b | (atomicAdd(&tab[b], 1) << 15); /* bucket (< 5 x 4096) and the point's arrival number in it: ONE LDS atomic per point */
        }
    }
    __syncthreads();
    {   /* exclusive scan of the histogram */
        const int per = (NB + T - 1) / T;
        const int b0 = tid * per;
        int sum = 0;
        for (int q = 0; q < per; ++q) if (b0 + q < NB) sum += tab[b0 + q];
        int total;
        int pre = block_exscan_w(sum, s_scr, &total);
        for (int q = 0; q < per; ++q) if (b0 + q < NB) { const int c = tab[b0 + q]; tab[b0 + q] = pre; pre += c; }
        if (tid == 0) tab[NB] = total;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < WIN_EMAX; ++e) if (pb[e] >= 0) { const int b = pb[e] & 0x7fff; pts[tab[b] + (pb[e] >> 15)] = pr[e]; pb[e] = b; }
    __syncthreads();
    STAMP(6, 1); /* histogram, scan, placement */
    WIN_STOP(1, pb[0] ^ pb[WIN_EMAX - 1]);
    /* tab[b] is the first place of bucket b, tab[b + 1] its end.  Inside a bucket (a point or two, a dozen in the densest windows) the points stand in
       arrival order: every point counts the members of its bucket that precede it on (y, cloud index) -- independent reads, no
       chain of dependent moves as in an insertion sort -- and takes that place. */
#pragma unroll
    for (int e = 0; e < WIN_EMAX; ++e) {
        const int b = pb[e];
        pb[e] = -1;
        if (b >= 0) {
            const int s0 = tab[b], s1 = tab[b + 1];
            if (s1 - s0 > 1) {
                const float my = pr[e].y;
                const int mi = idx_of(pr[e]);
                int rank = 0;
                for (int q = s0; q < s1; ++q) {
                    const float4 o = pts[q];
                    rank += (o.y < my || (o.y == my && idx_of(o) < mi)) ? 1 : 0;
                }
                pb[e] = s0 + rank;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < WIN_EMAX; ++e) if (pb[e] >= 0) pts[pb[e]] = pr[e];
    if (tid <= WIN_CLASSES) s_cs[tid] = 0;
    __syncthreads();
    STAMP(6, 2); /* buckets finished */
    WIN_STOP(2, pb[0] ^ __float_as_int(pts[tid % 64].y));
    if (tid < WIN_CLASSES) s_cs[tid + 1] = tab[(tid + 1) * NBc];
    __syncthreads();
    WinView V;
    V.pts = pts; V.tab = tab; V.NBc = NBc; V.y0 = A.y0; V.yscale = A.yscale; V.cs = s_cs;
    const int el0 = s_cs[3];
    const int nEr = s_cs[2] - s_cs[1], nEl = s_cs[4] - el0;
    if (tid == 0) A.band_cnt[s] = s_cs[4] - s_cs[1]; /* |rangedX_index| */
    if (nEl == 0 || nEr == 0) { slice_fails(DERR_SLICE); return; }  /* empty map -> < 3 knots; empty FLANN tree */
#ifdef WIN_NO_BRUTE /* (experiment builds: the kd kernel without the brute flavour's code) */
    const bool brute = false;
#else
    const bool brute = P.pairing != 0; /* ppp_params.pairing: 0 = PPP_PAIR_KD, 1 = PPP_PAIR_BRUTE (include/ppp_hip.h) */
#endif
    if (nEl > cap_el || (brute && nEr > cap_el)) { slice_fails(-WIN_FLAG_OVERFLOW); return; }
    constexpr int CE = WIN_CE; /* candidates per thread at most (cap_el <= CE * blockDim is checked by the plan) */
    u64 kr[CE];
    int kb[CE];
    int ncand = nEl;      /* candidates of the std::map: one per left point (kd), one per pair of left_pair (brute) */
    int NBcand = 64;
    float yr_scale = 0.f;
    auto cand_buckets = [&]() { /* about a candidate per bucket, as many as the histogram's cap_el + 1 slots allow; over the windows' y range */
        NBcand = next_pow2(max(ncand, 64));
        while (NBcand > cap_el) NBcand >>= 1;
        yr_scale = A.yscale * (float)NBcand / (float)NBc;
    };
    auto make_cand = [&](int e, int ci, const float4 &R, const float4 &Lp) { /* the pair interpolated onto the plane (:220-232 / Path_Generation.cpp:189-201) */
        const float t = (Px - R.x) / (Lp.x - R.x);
        float y = R.y + t * (Lp.y - R.y);
        const float z = R.z + t * (Lp.z - R.z);
        if (y == 0.f) y = 0.f; /* -0.0 and +0.0 are one std::map key */
        kr[e] = YK_MAKE(y, ci);
        int q2 = (int)((y - A.y0) * yr_scale);
        kb[e] = q2 < 0 ? 0 : (q2 >= NBcand ? NBcand - 1 : q2);
        return z;
    };
#pragma unroll
    for (int e = 0; e < CE; ++e) { kb[e] = -1; kr[e] = 0; }
    if (!brute) {
        /* ---- insert_point, kd flavour (path_slicing_alg.cpp:184-233): for every left point the nearest right point, for
           that one the nearest left point, the pair interpolated onto the plane ---- */
        cand_buckets();
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int i = tid + e * T;
            if (i < nEl) {
                const float4 q = pts[el0 + i];
                const float4 R = pts[win_nn_class(V, 1, q)];
                const float4 Lp = pts[win_nn_class(V, 3, R)];
                cz[i] = make_cand(e, i, R, Lp);
            }
        }
    } else {
        /* ---- insert_point, brute flavour (Path_Generation.cpp:129-201).  The two argmins do not depend on the used-flags: every
           left point's nearest right point and every right point's nearest left point are searched in parallel, on the staged
           window's y order (Vector3f::norm keys, the last index of equal keys).  The greedy walk over the left points in ascending
           cloud index with its two flag arrays stays one thread's loop, but over records made beforehand -- (i, j*(i), k*(j*)) in
           walk order -- so that an iteration is three flag reads, not a chain of look-ups.  Pairs are (right_pair[p], left_pair[p])
           for p < |left_pair|, the reference's misaligned indexing (App. B.3) included. ---- */
        const int er0 = s_cs[1];
        int jr[CE], kq[CE], ord[CE];
        int mymax = 0;
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int i = tid + e * T;
            jr[e] = kq[e] = ord[e] = 0;
            if (i < nEl) { const float4 q = pts[el0 + i]; jr[e] = win_nn_class_brute(V, 1, q) - er0; mymax = max(mymax, idx_of(q)); }
            if (i < nEr) kq[e] = win_nn_class_brute(V, 3, pts[er0 + i]) - el0;
        }
        for (int o = 32; o > 0; o >>= 1) mymax = max(mymax, __shfl_xor(mymax, o, 64));
        if ((tid & 63) == 0) s_scr[tid >> 6] = mymax;
        __syncthreads();
        int maxidx = 0;
        for (int w = 0; w < (T + 63) / 64; ++w) maxidx = max(maxidx, s_scr[w]);
        __syncthreads();
        {   /* the left points in ascending cloud index (the order El inherits from `indices`): bucket sort on the index */
            int NBs = next_pow2(max(nEl, 64));
            while (NBs > cap_el) NBs >>= 1;
            const u64 span = (u64)(u32)maxidx + 1ull;
            auto gen = [&](int i) { return ((u64)(u32)idx_of(pts[el0 + i]) << 32) | (u64)(u32)i; };
            auto bucket = [&](u64 k) { return (int)(((k >> 32) * (u64)NBs) / span); };
            auto less = [&](u64 a, u64 b) { return a < b; };
            block_bucket_sort_cached<CE>(ckeys, nEl, hc, NBs, s_scr, gen, bucket, less);
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) { const int t = tid + e * T; if (t < nEl) ord[e] = (int)(u32)(ckeys[t] & 0xffffffffull); }
        __syncthreads();
        u16 *jstar = (u16 *)cz, *kstar = jstar + cap_el; /* (cz: 4 bytes per left point) */
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int i = tid + e * T;
            if (i < nEl) jstar[i] = (u16)jr[e];
            if (i < nEr) kstar[i] = (u16)kq[e];
        }
        __syncthreads();
        u64 *trip = ckeys; /* walk step t: i | j*(i) << 16 | k*(j*) << 32 */
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int t = tid + e * T;
            if (t < nEl) { const int i = ord[e], j = jstar[i], k2 = kstar[j]; trip[t] = (u64)(u32)i | ((u64)(u32)j << 16) | ((u64)(u32)k2 << 32); }
        }
        __syncthreads();
        unsigned char *lf = (unsigned char *)cz, *rf = lf + 2 * cap_el; /* El_flag / Er_flag (the look-up tables are used up) */
        for (int w = tid; w < cap_el; w += T) ((u32 *)cz)[w] = 0u;
        u16 *rpair = (u16 *)hc, *lpair = rpair + cap_el;
        __syncthreads();
        if (tid == 0) {
            int nl = 0, nr = 0;
            u64 nx = trip[0];
            for (int t = 0; t < nEl; ++t) {
                const u64 tr = nx;
                if (t + 1 < nEl) nx = trip[t + 1]; /* (the next step's record travels while this one decides) */
                const int i = (int)(tr & 0xffffu), j = (int)((tr >> 16) & 0xffffu), k2 = (int)((tr >> 32) & 0xffffu);
                const int fi = lf[i], fj = rf[j], fk = lf[k2];
                if (fi) continue;                   /* if (El_flag[i] == 0) */
                if (fj) continue;                   /* Er_flag[compare.begin()->second] != 0: continue */
                rpair[nr++] = (u16)j; rf[j] = 1;    /* right_pair.push_back */
                if (!fk) { lpair[nl++] = (u16)k2; lf[k2] = 1; } /* left_pair.push_back only when that left point is unused */
            }
            s_m = nl; /* the reference loops i < left_pair.size() (Path_Generation.cpp:189) */
        }
        __syncthreads();
        ncand = s_m;
        cand_buckets();
        float zr[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int pq = tid + e * T;
            zr[e] = 0.f;
            if (pq < ncand) zr[e] = make_cand(e, pq, pts[er0 + rpair[pq]], pts[el0 + lpair[pq]]);
        }
        __syncthreads(); /* the pairs are in registers: their scratch becomes the candidates' */
#pragma unroll
        for (int e = 0; e < CE; ++e) { const int pq = tid + e * T; if (pq < ncand) cz[pq] = zr[e]; }
    }
    /* ---- std::map by y: bucket sort of the candidates (keys made once, kept in registers) ---- */
    for (int b = tid; b <= NBcand; b += T) hc[b] = 0;
    __syncthreads();
    STAMP(6, 3); /* pairing: two nearest-neighbour queries per left point + lerp */
    WIN_STOP(3, (int)kr[0] ^ kb[0] ^ kb[CE - 1] ^ (int)(kr[CE - 1] >> 32));
#pragma unroll
    for (int e = 0; e < CE; ++e) if (kb[e] >= 0) kb[e] |= atomicAdd(&hc[kb[e]], 1) << 13; /* bucket (< 4096) and arrival number in it: one LDS atomic per candidate */
    __syncthreads();
    {
        const int per = (NBcand + T - 1) / T;
        const int b0 = tid * per;
        int sum = 0;
        for (int q = 0; q < per; ++q) if (b0 + q < NBcand) sum += hc[b0 + q];
        int total;
        int pre = block_exscan_w(sum, s_scr, &total);
        for (int q = 0; q < per; ++q) if (b0 + q < NBcand) { const int c = hc[b0 + q]; hc[b0 + q] = pre; pre += c; }
        if (tid == 0) hc[NBcand] = total;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CE; ++e) if (kb[e] >= 0) { const int b = kb[e] & 0x1fff; ckeys[hc[b] + (kb[e] >> 13)] = kr[e]; kb[e] = b; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CE; ++e) { /* buckets finished by rank, as above (the keys are all different: they carry the candidate's number) */
        const int b = kb[e];
        kb[e] = -1;
        if (b >= 0) {
            const int s0 = hc[b], s1 = hc[b + 1];
            if (s1 - s0 > 1) {
                int rank = 0;
                for (int q = s0; q < s1; ++q) rank += ckeys[q] < kr[e] ? 1 : 0;
                kb[e] = s0 + rank;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CE; ++e) if (kb[e] >= 0) ckeys[kb[e]] = kr[e];
    if (tid == 0) s_m = 0;
    __syncthreads();
    STAMP(6, 4); /* candidate sort */
    WIN_STOP(4, (int)ckeys[tid % 64]);
    /* one knot per distinct y: Node[y] = ... is overwritten by every later writer, so the z kept is the last writer's inside the run
       of equal keys.  The knots take
       the place of the sorted keys: everything a chunk needs of them is read before the scan's barriers, written after. */
    for (int base = 0; base < ncand; base += T) {
        const int j = base + tid;
        const int o0 = s_m; /* slots below o0 hold knots of earlier chunks by now; a run of equal keys that reaches into this chunk
                               starts at or above o0 (every finished run left one knot and has at least one member) */
        int keep = 0;
        float ky = 0.f, kz = 0.f;
        if (j < ncand) {
            const u64 kj = ckeys[j];
            keep = (j == ncand - 1) || (YK_Y(ckeys[j + 1]) != YK_Y(kj));
            if (keep) {
                /* the last writer of a key: the left point with the highest cloud index (kd: candidate = left point, walked in ascending
                   index), the highest pair number (brute: candidate = pair, written in that order) */
                int best_i = YK_POS(kj);
                int best_idx = brute ? best_i : idx_of(pts[el0 + best_i]);
                for (int q = j - 1; q >= o0 && YK_Y(ckeys[q]) == YK_Y(kj); --q) {
                    const int ci = YK_POS(ckeys[q]);
                    const int id = brute ? ci : idx_of(pts[el0 + ci]);
                    if (id > best_idx) { best_idx = id; best_i = ci; }
                }
                ky = ord2f(YK_Y(kj)); kz = cz[best_i];
            }
        }
        int tot;
        const int pre = block_exscan_w(keep, s_scr, &tot);
        const int o = o0;
        __syncthreads();
        if (keep) knot[o + pre] = make_float2(ky, kz);
        if (tid == 0) s_m = o + tot;
        __syncthreads();
    }
    const int mm = s_m;
    if (tid == 0) {
        int tot = mm;
        /* the slice's knots go to its own segment of the knot arrays (cap_el each: a knot per left point at most) -- reserving
           a segment with an atomic on one shared cursor put a round trip to memory on every workgroup's critical path */
        int base = s * cap_el;
        if ((long long)base + tot > (long long)A.node_cap) { set_err(m, DERR_CAPACITY, s); base = 0; tot = 0; }
        s_base = base; s_nk = tot;
        A.node_start[s] = base; A.node_cnt[s] = tot;
        if (tot < 3) set_err(m, DERR_SLICE, s); /* gsl_spline_alloc needs >= 3 knots */
        int cnt = 0;
        if (kept && tot >= 1) cnt = sample_count((double)knot[0].x, (double)knot[tot - 1].x, P.trim, P.path_resolution, A.stride);
        if (cnt > A.stride) { win_flag(m, WIN_FLAG_OVERFLOW); cnt = 0; } /* (the slots per slice are the plan's: one made ahead of this cloud's bounds may have fewer than its y extent needs) */
        if (tot < 3) cnt = 0;
        s_cnt = cnt;
        if (kept) A.wp_cnt[k] = cnt;
    }
    __syncthreads();
    const int nknots = s_nk, cnt = s_cnt;
    STAMP(6, 5); /* map flattening, knot count, waypoint count */
    WIN_STOP(5, nknots ^ cnt ^ __float_as_int(knot[0].y));
    for (int i = tid; i < nknots; i += T) {
        const float2 kn = knot[i];
        A.node_x[s_base + i] = Px; /* insert_cloud.points[i].x = PlanePoint[0] */
        A.node_y[s_base + i] = kn.x;
        A.node_z[s_base + i] = kn.y;
    }
    if (cnt == 0) return;
    /* ---- getPath for this slice: sampling (:156-169), nearest point, PCL normal, frame, Euler, hand-eye (:178-208) ---- */
    auto Yf = [&](int i) { return (double)knot[i].x; };
    auto Zf = [&](int i) { return (double)knot[i].y; };
    const double ystart = (double)knot[0].x + P.trim;
    const double yfirst = (double)knot[0].x, ylast = (double)knot[mm - 1].x;
    const double inv_span = ylast > yfirst ? (double)(mm - 1) / (ylast - yfirst) : 0.0;
    const bool dy_closed_form = sums_exact(ystart, P.path_resolution, (double)cnt);
    float HE[3][3];
    handeye_rotation(P.handeye, HE);
    /* (8 lanes per waypoint -- both pairs of classes at once -- give the same bits but were slower where tried: 1 M points /
       256 slices 39.6 us against 35.0 with 4; twice the waves run the double-precision spline and the pose arithmetic) */
    /* 4 lanes per waypoint, or 2 where that saves a pass over the slice's waypoints (10 M points / 1024 slices: 258 waypoints x 4 lanes are
       eight more than a 1024-thread workgroup has -- two passes; with 2 lanes one: 228 -> 195 us).  Same bits either way (the tree above). */
    const int G = ((4 * cnt + T - 1) / T > (2 * cnt + T - 1) / T) ? 2 : 4, gshift = G == 4 ? 2 : 1;
    const int g = tid & (G - 1), per = T >> gshift;
    const int plane_round = G == 4 ? 2 : 4; /* the unit round of the on-plane class */
    const int rounds = (cnt + per - 1) / per;
    const int per_round = (cnt + rounds - 1) / rounds; /* the waypoints spread evenly over the rounds */
    const bool has_plane_class = s_cs[3] > s_cs[2];
    const float r2 = P.normal_radius * P.normal_radius;
    /* what is left of a waypoint once its searches are done: pcl::NormalEstimation's plane fit, the frame, the Euler angles, the
       hand-eye transform, the stores into the slice's slots */
    auto pose_and_store = [&](float *acc, const int count, const bool found, const int bidx, const float4 c0, const float4 q, const int t) {
        const bool finite = q.y == q.y && q.z == q.z;
        float n4[4] = {NAN, NAN, NAN, NAN};
        if (found && count >= 3) {
            const float cntf = (float)count;
            for (int i = 0; i < 9; ++i) acc[i] /= cntf;
            float cov[9];
            cov[0] = acc[0] - acc[6] * acc[6];
            cov[1] = acc[1] - acc[6] * acc[7];
            cov[2] = acc[2] - acc[6] * acc[8];
            cov[4] = acc[3] - acc[7] * acc[7];
            cov[5] = acc[4] - acc[7] * acc[8];
            cov[8] = acc[5] - acc[8] * acc[8];
            cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
            float ev, nn[3];
            pcl_eigen33_smallest<false>(cov, &ev, nn); /* continuous outputs only: the device float trig is enough */
            const float eig_sum = cov[0] + cov[4] + cov[8];
            const float curv = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
            const float vx = P.viewpoint[0] - c0.x, vy = P.viewpoint[1] - c0.y, vz = P.viewpoint[2] - c0.z;
            if (vx * nn[0] + vy * nn[1] + vz * nn[2] < 0) { nn[0] *= -1; nn[1] *= -1; nn[2] *= -1; }
            n4[0] = nn[0]; n4[1] = nn[1]; n4[2] = nn[2]; n4[3] = curv;
        }
        if (!finite) set_err(m, DERR_QUERY, -1);
        float wp[6], rpy[3];
        pose_from_normal(n4, rpy);
        if (P.change_range) { wp[0] = q.x / 1000; wp[1] = q.y / 1000; wp[2] = q.z / 1000; }
        else { wp[0] = q.x; wp[1] = q.y; wp[2] = q.z; }
        wp[3] = rpy[0]; wp[4] = rpy[1]; wp[5] = rpy[2];
        handeye_apply(HE, P.handeye, wp);
        /* Vector4f(point), std::reverse on every second slice (:166-168) */
        const size_t slot = (size_t)k * A.stride + (size_t)((k & 1) ? (cnt - 1 - t) : t);
        A.wps_xyz[slot] = q;
        A.wps_nn[slot] = found ? bidx : -1;
        A.wps_normal[slot] = make_float4(n4[0], n4[1], n4[2], n4[3]);
#pragma unroll
        for (int d = 0; d < 6; ++d) A.wps_pre[6 * slot + d] = wp[d];
    };
    /* Where a waypoint's record waits between its searches and its pose: in the pairing scratch that the knots do not use (cz and
       the candidate histogram, 4 bytes per left point each: word w of waypoint t at [w][t], seven words in the one, six in the other),
       when the plan found room there (A.rec_lds = waypoints per word row); else in the waypoint's global slot. */
    float *rec_a = cz, *rec_b = (float *)hc;
    const int rcap = A.rec_lds;
    float *rec_g = (float *)(A.wps_rec + 4 * (size_t)k * A.stride);
    auto rec_put = [&](int t, int w, float v) {
        if (rcap) { if (w < 7) rec_a[w * rcap + t] = v; else rec_b[(w - 7) * rcap + t] = v; }
        else rec_g[16 * (size_t)t + w] = v;
    };
    auto rec_get = [&](int t, int w) -> float {
        if (rcap) return w < 7 ? rec_a[w * rcap + t] : rec_b[(w - 7) * rcap + t];
        return rec_g[16 * (size_t)t + w];
    };
    for (int rd = 0; rd < rounds; ++rd) {
        const int lt = tid >> gshift;
        const int t = rd * per_round + lt;
        const bool act = lt < per_round && t < cnt;
        double dy = ystart;
        if (dy_closed_form) dy = ystart + (double)(act ? t : 0) * P.path_resolution; /* every partial sum is exact: same bits */
        else for (int r = 0; r < (act ? t : 0); ++r) dy += P.path_resolution;       /* the reference accumulates */
        /* gsl_interp_bsearch's interval -- the largest i <= mm - 2 with y_i <= dy, 0 below the first knot -- from a guess and a short walk */
        int iv;
        {
            int gi = (int)((dy - yfirst) * inv_span);
            gi = gi < 0 ? 0 : (gi > mm - 2 ? mm - 2 : gi);
            int steps = 0;
            while (gi > 0 && Yf(gi) > dy && steps < 12) { --gi; ++steps; }
            while (gi < mm - 2 && Yf(gi + 1) <= dy && steps < 12) { ++gi; ++steps; }
            iv = gi;
            if (steps >= 12) iv = gsl_bsearch(mm, dy, Yf);
        }
        /* x: every knot lies on its plane, so the y -> x spline is that constant, exactly (all slopes zero) */
        const double zd = steffen_eval_at(iv, mm, dy, Yf, Zf);
        const float4 q = make_float4((float)(double)Px, (float)dy, (float)zd, 1.f);
        const bool finite = q.y == q.y && q.z == q.z;
        const bool on = act && finite;
        STAMP(6, 6); /* knots out, dy, interval, Steffen */
        WIN_STOP(6, __float_as_int(q.z) ^ iv);
        /* -- kdtree.nearestKSearch(q, 1) (:189): inner classes first, the outer ones are closed by their x gap almost always -- */
        float best = P.nn_hint2;
        int bidx = 0x7fffffff, bpos = 0; /* cloud index (the tie rule) and place in the staged window of the best candidate so far */
        for (int pass = 0; pass < 2; ++pass) {
            for (int round = 0; round <= plane_round; ++round) {
                if (round == plane_round && !has_plane_class) break;
                const int c = win_unit_class(G, round, g);
                const bool up = (g & 1) == 0;
                const bool mine = on && !(round == plane_round && g >= 2);
                if (mine) {
                    /* every point of class 0 lies left of the band, of class 4 right of it */
                    const float gap = c == 0 ? q.x - blo : (c == 4 ? bhi - q.x : 0.f);
                    if (!(gap > 0.f && gap * gap > best)) {
                        const int p0 = V.lower_bound(c, q.y);
                        win_walk(V, c, p0, up, [&](const float4 &kk, int at) {
                            const float dyy = q.y - kk.y;
                            if (dyy * dyy > best) return false;
                            const float d = dist2_flann(q.x, q.y, q.z, kk.x, kk.y, kk.z);
                            const int id = idx_of(kk);
                            if (d < best || (d == best && id < bidx)) { best = d; bidx = id; bpos = at; }
                            return true;
                        });
                    }
                }
                for (int o = G >> 1; o > 0; o >>= 1) { /* minimum of the waypoint's lanes on (distance, cloud index) */
                    const float od = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bidx, o, 64), op = __shfl_xor(bpos, o, 64);
                    if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; bpos = op; }
                }
            }
            if (__all(bidx != 0x7fffffff || !on)) break;
            best = INFINITY; /* nothing within the hint: the same search without a bound */
        }
        const bool found = on && bidx != 0x7fffffff;
        const float4 bp = found ? pts[bpos] : make_float4(NAN, NAN, NAN, 0.f); /* (the lanes passed its place around, not the point) */
        STAMP(6, 7); /* nearest point */
        WIN_STOP(7, bidx ^ __float_as_int(bp.z));
        /* the window holds every point within `pad` of the plane: the answers are the whole cloud's as long as the ball that
           proves the nearest neighbour and the normal's radius search stay inside it */
        if (on && g == 0) {
            bool ok = found;
            if (found) {
                const float dq = sqrtf(best) * 1.0001f + 1.0e-4f;
                ok = dq <= A.pad && fabsf(bp.x - Px) + P.normal_radius * 1.0001f + 1.0e-4f <= A.pad;
            }
            if (!ok) win_flag(m, WIN_FLAG_REACH);
        }
        /* -- pcl::NormalEstimation at that point (path_slicing_alg.cpp:141-150): radius search, covariance about the point -- */
        float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pairsum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int count = 0, paircnt = 0;
        const float4 c0 = bp;
        for (int round = 0; round <= plane_round; ++round) {
            if (round == plane_round && !has_plane_class) break;
            float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            int cn = 0;
            const int c = win_unit_class(G, round, g);
            const bool up = (g & 1) == 0;
            if (found && !(round == plane_round && g >= 2)) {
                const float gap = c == 0 ? c0.x - blo : (c == 4 ? bhi - c0.x : 0.f);
                if (!(gap > 0.f && gap * gap > r2)) {
                    const int p0 = V.lower_bound(c, c0.y);
                    win_walk(V, c, p0, up, [&](const float4 &kk, int) {
                        const float dyy = c0.y - kk.y;
                        if (dyy * dyy > r2) return false;
                        if (dist2_flann(c0.x, c0.y, c0.z, kk.x, kk.y, kk.z) <= r2) {
                            const float x = kk.x - c0.x, y = kk.y - c0.y, z = kk.z - c0.z;
                            a[0] += x * x; a[1] += x * y; a[2] += x * z;
                            a[3] += y * y; a[4] += y * z; a[5] += z * z;
                            a[6] += x; a[7] += y; a[8] += z;
                            cn++;
                        }
                        return true;
                    });
                }
            }
            for (int o = 1; o < G; o <<= 1) { /* (up + down) of a class, then (4 lanes) the two classes of a pair: a fixed tree */
#pragma unroll
                for (int i = 0; i < 9; ++i) a[i] += __shfl_xor(a[i], o, 64);
                cn += __shfl_xor(cn, o, 64);
            }
            if (G == 2 && round < 4) { /* a pair of classes is two rounds here: (first + second), then into the total -- the tree above */
                if ((round & 1) == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) pairsum[i] = a[i];
                    paircnt = cn;
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 9; ++i) a[i] = pairsum[i] + a[i];
                cn = paircnt + cn;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] += a[i];
            count += cn;
        }
        STAMP(6, 8); /* normal: radius search + covariance */
        WIN_STOP(8, count ^ __float_as_int(acc[0] + acc[4] + acc[8]));
        /* What is left of a waypoint -- eigen33, frame, Euler angles, hand-eye: some 850 instructions, a third of this kernel's
           VALU work when the one lane in G that holds the sums runs them with the others idle -- is done below by one thread per
           waypoint: the searches park their result in the waypoint's slot (the same workgroup reads it back: L2 / L1 of this CU).
           (-DWIN_SPARSE_POSE: the earlier form, that lane finishes its waypoint here; kept for the A/B of DESIGN.md A.2.) */
#ifdef WIN_SPARSE_POSE
        if (act && g == 0) pose_and_store(acc, count, found, bidx, c0, q, t);
#else
        if (act && g == 0) { /* 13 words: the nine sums, the count (-1: no nearest point), the nearest point's place in the window, the sample */
#pragma unroll
            for (int w = 0; w < 9; ++w) rec_put(t, w, acc[w]);
            rec_put(t, 9, __int_as_float(found ? count : -1));
            rec_put(t, 10, __int_as_float(bpos));
            rec_put(t, 11, q.y);
            rec_put(t, 12, q.z);
        }
#endif
    }
#ifndef WIN_SPARSE_POSE
    __syncthreads();
    STAMP(6, 9); /* records parked */
    for (int t = tid; t < cnt; t += T) {
        float acc[9];
#pragma unroll
        for (int w = 0; w < 9; ++w) acc[w] = rec_get(t, w);
        const int count = __float_as_int(rec_get(t, 9));
        const bool found = count >= 0;
        const float4 c0 = found ? pts[__float_as_int(rec_get(t, 10))] : make_float4(NAN, NAN, NAN, 0.f);
        pose_and_store(acc, count, found, idx_of(c0), c0, make_float4((float)(double)Px, rec_get(t, 11), rec_get(t, 12), 1.f), t);
    }
#endif
    STAMP(6, 10); /* eigen33, frame, Euler, hand-eye, stores: one thread per waypoint */
}

/* the extra workgroup of launch 2: a2 finished (the scatter's partials reduced), a3 (the slice walk) recomputed from those
   bounds, both compared with the plan the launches were sized by; the meta block and the slice tables for the host */
__device__ __forceinline__ void win_verify_body(const WinArgs &A)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    float *s_px = (float *)s_raw;              /* S floats, then 2048 of scratch for the walk (the launch's dynamic LDS covers both) */
    float *s_front = s_px + A.S;
    __shared__ float s_mn[3][WSL_T / 64], s_mx[3][WSL_T / 64];
    __shared__ int s_cnt[WSL_T / 64];
    __shared__ int s_S, s_bad;
    DevMeta *m = A.m;
    const DevParams &P = A.P;
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int i = tid; i < A.g_scatter; i += T) {
        const MinMaxPart r = A.win_part[i];
        cnt += r.cnt;
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], r.mn[d]); mx[d] = fmaxf(mx[d], r.mx[d]); }
    }
    const int lane = tid & 63, wid = tid >> 6;
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    cnt = wave_sum(cnt);
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_mn[d][wid] = mn[d]; s_mx[d][wid] = mx[d]; } s_cnt[wid] = cnt; }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        const int nw = T >> 6;
        for (int w = 0; w < nw; ++w) c += s_cnt[w];
        float rmn[3], rmx[3];
        for (int d = 0; d < 3; ++d) {
            float a = INFINITY, b = -INFINITY;
            for (int w = 0; w < nw; ++w) { a = fminf(a, s_mn[d][w]); b = fmaxf(b, s_mx[d][w]); }
            rmn[d] = c ? a : 3.402823466e+38f; /* getMinMax3D starts from +-FLT_MAX */
            rmx[d] = c ? b : -3.402823466e+38f;
        }
        int bad = 0;
        if (P.bounds_given) { /* this handle saw its part of the cloud only: the whole cloud's bounds came with it; the part must lie inside */
            for (int d = 0; d < 3; ++d) { if (c && (rmn[d] < P.g_mn[d] || rmx[d] > P.g_mx[d])) bad = 1; rmn[d] = P.g_mn[d]; rmx[d] = P.g_mx[d]; }
            c = P.g_nvalid;
        }
        if (A.plan_rec) { /* the record the conversion pass of THIS cloud left on the device: the plan may be older than the cloud */
            const PlanAuto R = *A.plan_rec;
            for (int d = 0; d < 3; ++d) if (rmn[d] != R.fin.mn[d] || rmx[d] != R.fin.mx[d]) bad = 1;
            if (c != R.fin.cnt || R.S != A.S || __float_as_int(R.pad) != __float_as_int(A.pad)) bad = 1;
        } else {
            for (int d = 0; d < 3; ++d) if (rmn[d] != A.plan_mn[d] || rmx[d] != A.plan_mx[d]) bad = 1;
            if (c != A.plan_nvalid) bad = 1;
        }
        int S = c ? slice_walk_device(P.walk, rmn[0], rmx[0], P.tool_radius, s_px, A.S, s_front, 2048) : 0;
        if (S != A.S) bad = 1;
        for (int d = 0; d < 3; ++d) { m->mn[d] = rmn[d]; m->mx[d] = rmx[d]; m->mn_ord[d] = f2ord(rmn[d]); m->mx_ord[d] = f2ord(rmx[d]); }
        m->n_valid = c;
        m->S = S > A.S ? A.S : S;
        m->first_kept = P.drop_ends ? 1 : 0;
        { const int Sc = S > A.S ? A.S : S, nk = P.drop_ends ? Sc - 2 : Sc; m->nkept = nk < 0 ? 0 : nk; } /* (never beyond the plan's tables: whoever reads the block walks them by these counts) */
        m->sb = A.sb; m->se = A.se;
        m->incl_lo = P.incl_lo; m->incl_hi = P.incl_hi;
        s_S = S > A.S ? A.S : S;
        if (bad) s_bad = 1;
    }
    __syncthreads();
    const int S = s_S;
    int bad = 0;
    for (int s = tid; s < S; s += T) {
        const float v = s_px[s];
        if (v != A.plan_px[s]) bad = 1;
        const int position = (int)v;
        A.px[s] = v; A.lo[s] = (float)(-2 + position); A.hi[s] = (float)(2 + position);
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid == 0 && s_bad) win_flag(m, WIN_FLAG_STALE);
}

/* ------------------------------------------------------------------ */
/* launch 3: offsets + compaction + postion_smooth / reduceRPY / flange */
/* ------------------------------------------------------------------ */
/* words of the meta block written by the finish launch: agent-scope (write-through) stores, so that the workgroup that publishes
   the block to the host (win_publish_meta) sees them without any workgroup paying a release fence -- on this device a fence at
   agent scope writes back the whole L2, once per workgroup that executes it (64 x 250 k points: the finish launch 25 -> 126 us) */
#define META_PUT(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ void win_finish_body(const WinArgs &A, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) int s_off[]; /* nkept + 1 offsets */
    __shared__ float s_x[3][SMF_M];
    __shared__ int s_scan[17];
    __shared__ int s_run, s_short, s_last, s_err;
    __shared__ double s_rp[SMF_END];
    __shared__ double s_e[2][3];
    DevMeta *m = A.m;
    const DevParams &P = A.P;
    const int nk = A.nkept, tid = (int)threadIdx.x;
    if (tid == 0) { s_err = m->err | m->win_flag; s_run = 0; s_short = 0; }
    __syncthreads();
    if (s_err) return; /* a failed slice -- or a pass that is handed back (its slots and counts are another plan's: nothing list-wide may walk them) */
    /* a9 bookkeeping: every workgroup scans the waypoint counts of all kept slices (left by the slice workgroups) */
    const int res_i = (int)P.rpy_resolution;
    for (int base = 0; base < nk; base += blockDim.x) {
        const int k2 = base + tid;
        int c2 = 0;
        if (k2 < nk) {
            const int s2 = k2 + A.first_kept;
            if (s2 >= A.sb && s2 < A.se) c2 = A.wp_cnt[k2];
        }
        int tot;
        const int pre = block_exscan_w(c2, s_scan, &tot);
        const int run = s_run;
        if (k2 < nk) {
            s_off[k2] = run + pre;
            if (P.rpy_resolution > 2 && c2 <= res_i) s_short = 1; /* App. B.6: reduceRPY reads past a slice this short */
        }
        __syncthreads();
        if (tid == 0) s_run = run + tot;
        __syncthreads();
    }
    const int W = s_run;
    if (tid == 0) s_off[nk] = W;
    __syncthreads();
    if (W > A.W_cap) { if (bx == 0 && tid == 0) { set_err(m, DERR_CAPACITY, -1); META_PUT(&m->W, 0); } return; }
    if (bx == 0) { /* TailIndex.push_back(WayPointsList.size() - 1), the offsets, W: for the host and for the in-order finish */
        for (int k2 = tid; k2 < nk; k2 += blockDim.x) {
            A.wp_off[k2] = s_off[k2]; A.tail[k2] = s_off[k2 + 1] - 1;
            const int s2 = k2 + A.first_kept;
            if (s2 < A.sb || s2 >= A.se) A.wp_cnt[k2] = 0; /* another handle's slice */
        }
        if (tid == 0) { A.wp_off[nk] = W; META_PUT(&m->W, W); META_PUT(&m->any_short, s_short); META_PUT(&m->sweeps, 0); META_PUT(&m->smooth_done, 0); }
    }
    const int ntiles = smooth_tiles(W);
    const int tile = bx;
    if (W == 0 || tile >= ntiles) return;
    /* list index -> kept slice: the counts are nearly equal, so a proportional guess lands within a slice or two; a short
       walk settles it (bounded: a binary search takes over on very uneven lists) */
    auto slice_of = [&](int g) {
        int kk = (int)(((long long)g * (long long)nk) / (long long)W);
        kk = kk < 0 ? 0 : (kk > nk - 1 ? nk - 1 : kk);
        int steps = 0;
        while (kk > 0 && s_off[kk] > g && steps < 6) { --kk; ++steps; }
        while (kk < nk - 1 && s_off[kk + 1] <= g && steps < 6) { ++kk; ++steps; }
        if (steps >= 6) { int lo = 0, hi = nk - 1; while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= g) lo = mid; else hi = mid - 1; } kk = lo; }
        return kk;
    };
    auto row_of = [&](int g) { const int kk = slice_of(g); return (size_t)kk * A.stride + (size_t)(g - s_off[kk]); };
    const double wd = 0.65, ws = 1 - wd, dg = wd + 2 * ws;
    const double r = (dg - sqrt(dg * dg - 4 * ws * ws)) / (2 * ws), c = wd / (dg - 2 * ws * r);
    const int t0 = tile * SMF_T;
    const int base = t0 - SMF_K;
    const bool solve = A.finish && P.smooth && W > 2;
    /* this thread's waypoint: its slice and its six values, requested once; the position part also fills the filter's tile */
    const int g = t0 + tid;
    int kk = 0;
    size_t rowbase = 0;
    float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g < W) {
        kk = slice_of(g);
        rowbase = (size_t)kk * A.stride - (size_t)s_off[kk]; /* row of list index w of this slice = rowbase + w */
#pragma unroll
        for (int d = 0; d < 6; ++d) p[d] = A.wps_pre[6 * (rowbase + g) + d];
    }
    if (A.finish) {
        const bool src = g >= 1 && g <= W - 2; /* the sources are the INTERIOR waypoints: the two fixed ends enter through A and B */
        for (int j = 0; j < 3; ++j) s_x[j][tid + SMF_K] = src ? p[j] : 0.f;
        if (tid < 2 * SMF_K) { /* the halo: SMF_K waypoints either side of the tile */
            const int l = tid < SMF_K ? tid : SMF_T + tid;
            const int g2 = base + l;
            const bool src2 = g2 >= 1 && g2 <= W - 2;
            size_t row = 0;
            if (src2) row = row_of(g2);
            for (int j = 0; j < 3; ++j) s_x[j][l] = src2 ? A.wps_pre[6 * row + j] : 0.f;
        }
    }
    const bool in_order = A.finish && P.rpy_resolution > 2 && s_short;
    const bool copy2 = A.finish && A.out2 != nullptr && W <= A.out2_cap;
    if (A.finish && A.out2 != nullptr && W > A.out2_cap && tile == 0 && tid == 0) set_err(m, DERR_CAPACITY, -1);
    const bool near_end = solve && (t0 < SMF_END || t0 + SMF_T - 1 > W - 1 - SMF_END);
    if (near_end) {
        if (tid < SMF_END) s_rp[tid] = smooth_rpow(r, tid);
        const int j = tid >> 6, lane = tid & 63;
        if (j < 3) {
            const int kq = (lane & 31) + 1;
            const int gq = lane < 32 ? kq : W - 1 - kq;
            double v = (gq >= 1 && gq <= W - 2) ? (double)A.wps_pre[6 * row_of(gq) + j] * smooth_rpow(r, kq) : 0.0;
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((lane & 31) == 0) s_e[lane >> 5][j] = (double)A.wps_pre[6 * row_of(lane < 32 ? 0 : W - 1) + j] - c * v;
        }
    }
    __syncthreads();
    if (g < W) {
#pragma clang fp contract(on) /* the 65-tap filter and the flange offset: continuous output only (same text as smooth_solve_body) */
        for (int d = 0; d < 6; ++d) A.wp_pre[6 * (size_t)g + d] = p[d]; /* the compact list after HandEyeTransform */
        if (A.finish) {
            if (solve) {
                const int l = g - base;
                double y[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double acc = (double)s_x[j][l - SMF_K] + (double)s_x[j][l + SMF_K];
#pragma unroll 8
                    for (int q = SMF_K - 1; q >= 1; --q) acc = ((double)s_x[j][l - q] + (double)s_x[j][l + q]) + r * acc;
                    y[j] = c * ((double)s_x[j][l] + r * acc);
                }
                if (near_end) {
                    const double D = W - 1 < SMF_END ? s_rp[W - 1] : 0.0;
                    const double r0 = g < SMF_END ? s_rp[g] : 0.0, r1 = W - 1 - g < SMF_END ? s_rp[W - 1 - g] : 0.0;
                    const double det = 1.0 - D * D;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double e0 = s_e[0][j], e1 = s_e[1][j];
                        const double Aa = (e0 - D * e1) / det, Bb = (e1 - D * e0) / det;
                        y[j] = y[j] + (Aa * r0 + Bb * r1);
                    }
                }
                if (g >= 1 && g <= W - 2) { p[0] = (float)y[0]; p[1] = (float)y[1]; p[2] = (float)y[2]; }
            }
            for (int d = 0; d < 6; ++d) A.wp_smooth[6 * (size_t)g + d] = p[d];
            if (!in_order) {
                /* reduceRPY for one waypoint (see finish_one_waypoint): the slice of g is known, no search over TailIndex */
                if (P.rpy_resolution > 2) {
                    const int res = res_i;
                    const int p0 = s_off[kk], tl = s_off[kk + 1] - 1;
                    const int nfull = (tl - p0) / res;
                    const int lastkey = p0 + nfull * res;
                    const int rr = g - p0;
                    const float *src = A.wps_pre + 6 * rowbase;
                    if (g > lastkey) {
                        for (int D = 3; D < 6; ++D) p[D] = src[6 * (size_t)lastkey + D];
                    } else if (rr % res != 0) {
                        const int pre = p0 + (rr / res) * res, last = pre + res, wi = g - pre;
                        for (int D = 3; D < 6; D++) {
                            const float a = src[6 * (size_t)last + D], bb = src[6 * (size_t)pre + D];
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
            for (int d = 0; d < 6; ++d) A.wp_out[6 * (size_t)g + d] = p[d];
            if (copy2 && !in_order) for (int d = 0; d < 6; ++d) A.out2[6 * (size_t)g + d] = p[d];
        }
    }
    if (in_order) { /* App. B.6: a slice shorter than RPYres + 1 -- the last tile to arrive finishes the whole list in order */
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&m->emit_ticket, 1) == ntiles - 1;
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (tid == 0) { finish_list_in_order(m, P, A.tail, A.wp_out); m->emit_ticket = 0; }
            __threadfence();
            __syncthreads();
            if (copy2) for (size_t i = tid; i < 6 * (size_t)W; i += blockDim.x) A.out2[i] = A.wp_out[i];
        }
    }
}

/* the per-waypoint stage lists (sampled point, nearest cloud index, normal) from the per-slice slots into list order: only
   when a caller asks for them (ppp_get_stage) */
__global__ void __launch_bounds__(256) k_win_gather_stage(WinArgs A, float4 *wp_xyz, int *wp_nn, float4 *wp_normal)
{
    const int k = blockIdx.x;
    if (k >= A.nkept) return;
    const int s = k + A.first_kept;
    if (s < A.sb || s >= A.se) return;
    const int off = A.wp_off[k], cnt = A.wp_cnt[k];
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        const size_t slot = (size_t)k * A.stride + t;
        wp_xyz[off + t] = A.wps_xyz[slot];
        wp_nn[off + t] = A.wps_nn[slot];
        wp_normal[off + t] = A.wps_normal[slot];
    }
}


/* ------------------------------------------------------------------ */
/* plan time (once per cloud + parameters, not part of a pass): the exact population of every window and of its left band   */
/* side, so that the LDS capacities of the slice workgroups are the cloud's own maxima instead of a density guess (a jittered */
/* grid puts 5 or 6 columns of points into an 8 mm window: +-12 % around the mean).                                            */
/* ------------------------------------------------------------------ */

/* census of one point (both census kernels): its window and, inside the band, its side */
__device__ __forceinline__ void win_census_point(float x, const float *__restrict__ px, int S, float px0, float inv_step, float pad,
                                                 int *cw, int *ce, int *cr)
{
    if (!(x == x)) return;
    const float fj = fminf(fmaxf(floorf((x - px0) * inv_step + 0.5f), 0.f), (float)(S - 1));
    const int j = (int)fj;
    const int ja = j > 0 ? j - 1 : j, jb = j + 1 < S ? j + 1 : j;
    int w = -1;
    if (fabsf(x - px[j]) <= pad) w = j; else if (fabsf(x - px[ja]) <= pad) w = ja; else if (fabsf(x - px[jb]) <= pad) w = jb;
    if (w < 0) return;
    atomicAdd(&cw[w], 1);
    const float Px = px[w];
    const int position = (int)Px;
    if (!(x < (float)(-2 + position) || x > (float)(2 + position))) {
        const float d = (x - Px) * 1.f;
        if (d > 0) atomicAdd(&ce[w], 1); else if (d < 0) atomicAdd(&cr[w], 1);
    }
}
#define WIN_CENSUS_UNROLL 8 /* x values a thread requests before it looks at the first (one read per dependent loop trip was the census: 22 us for 4 MB) */
template <bool IN_LDS>
__global__ void __launch_bounds__(256) k_win_census(const float *__restrict__ X, int n, const float *__restrict__ px, int S, float px0,
                                                    float inv_step, float pad, int *cnt_win, int *cnt_el, int *cnt_er)
{
    /* IN_LDS: the 3 S counters privatised per workgroup (a million points on a few hundred global counters serialise:
       360 us at 1 M points / 256 windows; 12 us this way) */
    extern __shared__ __attribute__((aligned(16))) int s_c[];
    if (IN_LDS) {
        for (int i = threadIdx.x; i < 3 * S; i += blockDim.x) s_c[i] = 0;
        __syncthreads();
    }
    int *cw = IN_LDS ? s_c : cnt_win, *ce = IN_LDS ? s_c + S : cnt_el, *cr = IN_LDS ? s_c + 2 * S : cnt_er;
    const int step = (int)(gridDim.x * blockDim.x);
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += WIN_CENSUS_UNROLL * step) {
        float xv[WIN_CENSUS_UNROLL];
#pragma unroll
        for (int u = 0; u < WIN_CENSUS_UNROLL; ++u) { const int i = i0 + u * step; xv[u] = i < n ? X[i] : __int_as_float(0x7fc00000); }
#pragma unroll
        for (int u = 0; u < WIN_CENSUS_UNROLL; ++u) win_census_point(xv[u], px, S, px0, inv_step, pad, cw, ce, cr);
    }
    if (IN_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < S; i += blockDim.x) {
            if (s_c[i]) atomicAdd(&cnt_win[i], s_c[i]);
            if (s_c[S + i]) atomicAdd(&cnt_el[i], s_c[S + i]);
            if (s_c[2 * S + i]) atomicAdd(&cnt_er[i], s_c[2 * S + i]);
        }
    }
}

/* The census of a cloud that has just arrived, straight behind k_ingest_minmax in the stream: walk, slice count and pad come
   from the record that kernel's last workgroup left (PlanAuto), so the host is not needed in between; the last workgroup here
   hands the counters, the plane table and the record to pinned host memory and clears the counters again -- no copy or fill
   command on the way (each costs the host 10-20 us on this runtime; a new cloud's plan was 150 us of which 30 us were kernels). */
__global__ void __launch_bounds__(256) k_win_census_auto(const float *__restrict__ X, int n, const float *px, const PlanAuto *plan, float inv_step,
                                                         int *cnt, int *ticket, PlanAuto *plan_host, float *px_host, int *census_host)
{
    extern __shared__ __attribute__((aligned(16))) int s_c[];
    __shared__ int s_last;
    const int S = plan->S;
    const float pad = plan->pad;
    const bool valid = S >= 1 && S <= WIN_AUTO_SCAP;
    if (valid) {
        for (int i = threadIdx.x; i < 3 * S; i += blockDim.x) s_c[i] = 0;
        __syncthreads();
        const float px0 = px[0];
        int *cw = s_c, *ce = s_c + S, *cr = s_c + 2 * S;
        const int step = (int)(gridDim.x * blockDim.x);
        for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += WIN_CENSUS_UNROLL * step) {
            float xv[WIN_CENSUS_UNROLL];
#pragma unroll
            for (int u = 0; u < WIN_CENSUS_UNROLL; ++u) { const int i = i0 + u * step; xv[u] = i < n ? X[i] : __int_as_float(0x7fc00000); }
#pragma unroll
            for (int u = 0; u < WIN_CENSUS_UNROLL; ++u) win_census_point(xv[u], px, S, px0, inv_step, pad, cw, ce, cr);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * S; i += blockDim.x)
            if (s_c[i]) atomicAdd(&cnt[i], s_c[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (valid) {
        for (int i = threadIdx.x; i < 3 * S; i += blockDim.x) { census_host[i] = __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cnt[i] = 0; }
        for (int i = threadIdx.x; i < S; i += blockDim.x) px_host[i] = px[i];
    }
    if (threadIdx.x == 0) {
        PlanAuto r = *plan;
        r.census = valid ? 1 : 0;
        *plan_host = r;
        *ticket = 0;
    }
}

/* ---- launch forms: single (arguments by value) and batched (blockIdx.y = member of the batch) ---- */
template <int PPT, bool STAGED>
__global__ void __launch_bounds__(WSC_T) k_win_scatter(WinArgs A)
{
    if (STAGED) win_scatter_staged_loop<PPT>(A, blockIdx.x, A.g_scatter); /* g_scatter workgroups share the chunks of the cloud */
    else win_scatter_body<PPT, false>(A, blockIdx.x);
}
template <int PPT, bool STAGED>
__global__ void __launch_bounds__(WSC_T) k_win_scatter_b(const WinArgs *__restrict__ mem)
{
    const WinArgs &A = mem[blockIdx.y];
    if ((int)blockIdx.x >= A.g_scatter) return;
    if (STAGED) win_scatter_staged_loop<PPT>(A, blockIdx.x, A.g_scatter);
    else win_scatter_body<PPT, false>(A, blockIdx.x);
}
template <int TMAX>
__global__ void __launch_bounds__(TMAX) k_win_slice(WinArgs A)
{
    if ((int)blockIdx.x == A.g_slice) { win_verify_body(A); return; }
    win_slice_body<TMAX>(A, blockIdx.x);
}
template <int TMAX>
__global__ void __launch_bounds__(TMAX) k_win_slice_b(const WinArgs *__restrict__ mem)
{
    const WinArgs &A = mem[blockIdx.y];
    if ((int)blockIdx.x > A.g_slice) return;
    if ((int)blockIdx.x == A.g_slice) { win_verify_body(A); return; }
    win_slice_body<TMAX>(A, blockIdx.x);
}
/* The meta block of a pass for the host, without a copy command behind the pass (each costs the host 10-20 us on this runtime
   and the device a 4 us blit kernel): the last of the finish launch's workgroups to get here writes it to pinned host memory.
   Every workgroup of the member arrives, whatever its own part of the finish was. */
__device__ __forceinline__ void win_publish_meta(const WinArgs &A)
{
    if (!A.meta_host) return;
    __shared__ int s_lastw;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0); /* vmcnt(0): what this workgroup put into the meta block (META_PUT, atomics) has been acknowledged */
        /* two levels of arrival counters: the workgroups of a launch finish together, and a thousand atomics on ONE address queue
           for ~10 ns each (10 M points: the launch 23 -> 30 us with a single counter) */
        const int grp = (int)blockIdx.x % WIN_FIN_GROUPS;
        const int members = (A.g_finish - grp + WIN_FIN_GROUPS - 1) / WIN_FIN_GROUPS;
        int last = 0;
        if (__hip_atomic_fetch_add(A.fin_ticket + 1 + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(A.fin_ticket + 1 + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int groups = A.g_finish < WIN_FIN_GROUPS ? A.g_finish : WIN_FIN_GROUPS;
            last = __hip_atomic_fetch_add(A.fin_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
        }
        s_lastw = last;
    }
    __syncthreads();
    if (!s_lastw) return;
    const int *src = (const int *)A.m;
    int *dst = (int *)A.meta_host;
    for (int q = threadIdx.x; q < (int)(sizeof(DevMeta) / sizeof(int)); q += blockDim.x) dst[q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) __hip_atomic_store(A.fin_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(SMF_T) k_win_finish(WinArgs A) { win_finish_body(A, blockIdx.x); win_publish_meta(A); }
__global__ void __launch_bounds__(SMF_T) k_win_finish_b(const WinArgs *__restrict__ mem)
{
    const WinArgs &A = mem[blockIdx.y];
    if ((int)blockIdx.x >= A.g_finish) return;
    win_finish_body(A, blockIdx.x);
    win_publish_meta(A);
}
/* the members' meta blocks side by side, so that ONE copy publishes the batch to the host */
__global__ void __launch_bounds__(64) k_collect_meta_win(const WinArgs *__restrict__ mem, int count, DevMeta *out)
{
    const int i = blockIdx.x;
    if (i >= count) return;
    const int *src = (const int *)mem[i].m;
    int *dst = (int *)(out + i);
    for (int q = threadIdx.x; q < (int)(sizeof(DevMeta) / sizeof(int)); q += blockDim.x) dst[q] = src[q];
}
