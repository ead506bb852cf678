/*
 * ppp_dynamic.h -- dynamic adjustment of the slice paths:
 * compute_transform / Area2Cloud / compute_boundary / bisection / dynamic_adjust_path of
 * src/Path_Alg/path_dynamic_alg.cpp:77-306 (= dynamic_alg_sdir.cpp:77-306).
 *
 * Work decomposition: one WAVE per Area2Cloud evaluation (k-NN gather and selection, projected
 * normal covariance, 721-point ellipse extremum are all lane-parallel); the slice-to-slice
 * dependency (slice s is adjusted against the boundary of the already adjusted slice s-1) is a
 * chain of small launches on the handle's stream, left and right of the centre path side by side
 * (gridDim.y = chain).
 */
#pragma once
#include "ppp_kernels.h"

#define DYN_KNN_CAP 448   /* candidates a wave keeps while growing the search radius */
#ifndef DYN_WAVES
#define DYN_WAVES 4       /* waves (= Area2Cloud evaluations) per workgroup; 1, 2, 4 measure the same, 8 slower */
#endif
#define DYN_ELL 721       /* for (float angle = 0; angle <= 360; angle += 0.5)       */

struct DynParams {
    double tool_radius, depth, toolthickness, adjust_threshold;
    int k;          /* 50: path_dynamic_alg.cpp:87 */
    float r0;       /* first search radius of the k-NN gather */
    float r1;       /* first search radius of the 1-NN snap   */
};

struct __attribute__((aligned(16))) DynWaveLds {
    u64 key[DYN_KNN_CAP];  /* (bits of the squared distance) << 32 | cloud index: one compare ranks a candidate */
    int pos[DYN_KNN_CAP];
    int sel[64];
    int off[65];  /* exclusive prefix of the y-window sizes of 64 neighbouring slabs */
    int w0[64];   /* first position of each window                                  */
};

/* exact k nearest neighbours of q (ascending (distance, cloud index)): returns kk <= k, positions
   in L.sel[0..kk).  All 64 lanes of the wave call this together.
   Every slab the search ball touches is binary-searched for its y-window by its own lane (the
   searches are chains of dependent loads: side by side they cost one chain, not one per slab); the
   windows are then walked as one flat list, 64 candidates per step. */
__device__ inline int wave_knn(const SlabView &V, DynWaveLds &L, float qx, float qy, float qz, int k, float r0, StampCtx &sc)
{
    const int lane = threadIdx.x & 63;
    const int B = V.m->B;
    const int total = V.m->n_sorted;
    float r = r0;
    int count = 0;
    for (int attempt = 0; attempt < 48; ++attempt) {
        const float r2 = r * r;
        count = 0;
        bool overflow = false;
        const float pady = 1e-5f * (fabsf(qy) + r) + 1e-6f, padx = 1e-5f * (fabsf(qx) + r) + 1e-6f;
        const float ylo = qy - r - pady, yhi = qy + r + pady;
        int blo = slab_of(V.m, qx - r - padx) - 1, bhi = slab_of(V.m, qx + r + padx) + 1;
        blo = blo < 0 ? 0 : blo;
        bhi = bhi >= B ? B - 1 : bhi;
        for (int cb = blo; cb <= bhi && !overflow; cb += 64) {
            const int bb = cb + lane;
            int a = 0, e = 0;
            if (bb <= bhi) {
                const int s0 = V.slab_start[bb], s1 = V.slab_start[bb + 1];
                int l0 = s0, l1 = s1, u0 = s0, u1 = s1; /* first y >= ylo, first y > yhi */
                while (l0 < l1 || u0 < u1) {
                    if (l0 < l1) { const int mid = (l0 + l1) >> 1; if (V.at(mid).y < ylo) l0 = mid + 1; else l1 = mid; }
                    if (u0 < u1) { const int mid = (u0 + u1) >> 1; if (V.at(mid).y <= yhi) u0 = mid + 1; else u1 = mid; }
                }
                a = l0; e = u0 < l0 ? l0 : u0;
            }
            const int cnt = e - a;
            int inc = cnt;
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
            const int T = __shfl(inc, 63, 64);
            __builtin_amdgcn_wave_barrier();
            L.off[lane] = inc - cnt; L.w0[lane] = a;
            if (lane == 63) L.off[64] = T;
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                bool in = false;
                float d = 0.f;
                int i = 0, id = 0;
                if (t < T) {
                    int lo = 0, hi = 63; /* the window holding flat position t: last lane with off <= t */
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L.off[mid] <= t) lo = mid; else hi = mid - 1; }
                    i = L.w0[lo] + (t - L.off[lo]);
                    const float4 c = V.at(i);
                    d = dist2_flann(qx, qy, qz, c.x, c.y, c.z);
                    id = idx_of(c);
                    in = d <= r2;
                }
                const u64 mask = __ballot(in);
                if (in) {
                    const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
                    if (slot < DYN_KNN_CAP) { L.key[slot] = ((u64)__float_as_uint(d) << 32) | (u32)id; L.pos[slot] = i; }
                }
                count += __popcll(mask);
                if (count > DYN_KNN_CAP) { overflow = true; break; }
            }
        }
        if (overflow) { r *= 0.8f; continue; }
        if (count >= k || count >= total) break;
        r *= 1.5f;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    sc.mark(2);
    const int kk = count < k ? count : k;
    /* rank by counting: (distance, cloud index) is a total order; d >= 0, so its bit pattern orders like the value.
       The keys are read two per LDS access (every lane the same address: a broadcast), padded to a multiple of 8 */
    const int cpad = (count + 7) & ~7;
    for (int c = count + lane; c < cpad; c += 64) L.key[c] = ~0ull;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int c = lane; c < count; c += 64) {
        const u64 kc = L.key[c];
        int rank = 0;
        const ulonglong2 *kv = (const ulonglong2 *)L.key;
        for (int o = 0; o < cpad / 2; o += 4) {
            const ulonglong2 a0 = kv[o], a1 = kv[o + 1], a2 = kv[o + 2], a3 = kv[o + 3];
            rank += (a0.x < kc) + (a0.y < kc) + (a1.x < kc) + (a1.y < kc) + (a2.x < kc) + (a2.y < kc) + (a3.x < kc) + (a3.y < kc);
        }
        if (rank < kk) L.sel[rank] = L.pos[c];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    sc.mark(3);
    return kk;
}

/* Area2Cloud(point, flag, key): key 0 = left (min x), 1 = right (max x).  Wave-cooperative. */
__device__ inline void wave_area2cloud(const SlabView &V, DynWaveLds &L, const float4 *__restrict__ normals4,
                                       const float *__restrict__ ell_cs, const DynParams &D, const double point[3], int key,
                                       float bound[3], StampCtx &sc)
{
    const int lane = threadIdx.x & 63;
    const float sp[3] = {(float)point[0], (float)point[1], (float)point[2]};
    bound[0] = bound[1] = bound[2] = NAN;
    if (!(sp[0] == sp[0] && sp[1] == sp[1] && sp[2] == sp[2])) return;
    const int kk = wave_knn(V, L, sp[0], sp[1], sp[2], D.k, D.r0, sc);
    if (kk <= 0) return;
    /* computePointPrincipalCurvatures: lane r holds the neighbour of rank r */
    float nn[3] = {0.f, 0.f, 0.f};
    if (lane < kk) {
        const float4 nv = normals4[idx_of(V.at(L.sel[lane]))];
        nn[0] = nv.x; nn[1] = nv.y; nn[2] = nv.z;
    }
    float n0[3];
    for (int i = 0; i < 3; ++i) n0[i] = __shfl(nn[i], 0, 64);
    float proj[3] = {0.f, 0.f, 0.f};
    if (lane < kk)
        for (int i = 0; i < 3; ++i) {
            const float m0 = (i == 0 ? 1.f : 0.f) - n0[i] * n0[0], m1 = (i == 1 ? 1.f : 0.f) - n0[i] * n0[1],
                        m2 = (i == 2 ? 1.f : 0.f) - n0[i] * n0[2];
            proj[i] = m0 * nn[0] + m1 * nn[1] + m2 * nn[2];
        }
    /* centroid and covariance of the projected normals: summed neighbour by neighbour in rank order, as the reference's
       loops do (a tree reduction gives other last bits, and the ellipse extremum below is a discontinuous function of
       them) -- every lane runs the same sequential sums on values broadcast from lane r */
    auto lane_value = [](float v, int r) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r)); }; /* r is wave-uniform */
    /* lanes >= kk hold zeros: adding them is exact, so the loops run all 64 ranks without a branch */
    float cen[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 64; ++r)
        for (int i = 0; i < 3; ++i) cen[i] += lane_value(proj[i], r);
    for (int i = 0; i < 3; ++i) cen[i] /= (float)kk;
    float d[3] = {0.f, 0.f, 0.f};
    if (lane < kk) for (int i = 0; i < 3; ++i) d[i] = proj[i] - cen[i];
    float cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const float dx = lane_value(d[0], r), dy = lane_value(d[1], r), dz = lane_value(d[2], r);
        cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz;
        cov[4] += dy * dy; cov[5] += dy * dz; cov[8] += dz * dz;
    }
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    /* pcl::eigen33(mat, evals) + computeCorrespondingEigenVector(mat, evals[2]) */
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(cov[i]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float mm[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mm[i][j] = cov[3 * i + j] / scale;
    float ev[3];
    pcl_roots(mm, ev);
    for (int i = 0; i < 3; ++i) ev[i] *= scale;
    const float shift = ev[2] / scale;
    mm[0][0] -= shift; mm[1][1] -= shift; mm[2][2] -= shift;
    float cp[3][3];
    cross3f(mm[0], mm[1], cp[0]); cross3f(mm[0], mm[2], cp[1]); cross3f(mm[1], mm[2], cp[2]);
    float len[3];
    for (int i = 0; i < 3; ++i) len[i] = sqrtf(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
    int bi = 0;
    if (len[1] > len[bi]) bi = 1;
    if (len[2] > len[bi]) bi = 2;
    float cv[3];
    for (int q = 0; q < 3; ++q) cv[q] = cp[bi][q] / len[bi];
    const float inv = 1.0f / (float)kk;
    const float pc0 = ev[2] * inv, pc1 = ev[1] * inv;
    sc.mark(4);
    /* compute_transform: [n x c | c | n | p] */
    float cr[3];
    cross3f(n0, cv, cr);
    /* ellipse axes (path_dynamic_alg.cpp:123-141), double arithmetic as std::pow / std::sqrt give */
    const double toolRadius = D.tool_radius, depth = D.depth, toolthickness = D.toolthickness;
    double longAxis, shortAxis;
    if ((pc0 >= 0) && (pc1 >= 0)) {
        const double i1 = (double)(1 / pc1), i0 = (double)(1 / pc0);
        const double a1 = (double)fabsf(1 / pc1) - depth, a0 = (double)fabsf(1 / pc0) - depth;
        longAxis = sqrt(i1 * i1 - a1 * a1);
        if (longAxis > toolRadius) longAxis = toolRadius;
        shortAxis = sqrt(i0 * i0 - a0 * a0);
        if (shortAxis > toolRadius) shortAxis = toolRadius;
    } else {
        const double i1 = (double)(1 / pc1), i0 = (double)(1 / pc0);
        longAxis = (double)fabsf(1 / pc1) - sqrt(i1 * i1 - toolRadius * toolRadius);
        if (longAxis > toolthickness) longAxis = toolthickness;
        shortAxis = (double)fabsf(1 / pc0) - sqrt(i0 * i0 - toolRadius * toolRadius);
        if (shortAxis > toolthickness) shortAxis = toolthickness;
    }
    /* 721-point ellipse, transformed (SSE order c0*x + (c1*y + (c2*0 + c3))), first extremum in x */
    float bx = 0.f, by = 0.f, bz = 0.f;
    int ba = 0x7fffffff;
    bool have = false, first_nan = false;
    for (int a = lane; a < DYN_ELL; a += 64) {
        const float ex = (float)(longAxis * (double)ell_cs[2 * a]);
        const float ey = (float)(shortAxis * (double)ell_cs[2 * a + 1]);
        float t[3];
        for (int i = 0; i < 3; ++i) t[i] = cr[i] * ex + (cv[i] * ey + (n0[i] * 0.f + sp[i]));
        if (a == 0 && !(t[0] == t[0])) first_nan = true;
        if (t[0] == t[0]) {
            if (!have || (key == 1 ? (bx < t[0]) : (t[0] < bx))) { bx = t[0]; by = t[1]; bz = t[2]; ba = a; have = true; }
        }
    }
    /* the reference folds in ascending angle: a NaN at angle 0 poisons the whole fold */
    if (__ballot(first_nan)) return;
    for (int o = 32; o > 0; o >>= 1) {
        const float ox = __shfl_xor(bx, o, 64), oy = __shfl_xor(by, o, 64), oz = __shfl_xor(bz, o, 64);
        const int oa = __shfl_xor(ba, o, 64);
        const bool ohave = __shfl_xor(have ? 1 : 0, o, 64) != 0;
        bool take = false;
        if (ohave) {
            if (!have) take = true;
            else if (key == 1 ? (ox > bx) : (ox < bx)) take = true;
            else if (ox == bx && oa < ba) take = true;
        }
        if (take) { bx = ox; by = oy; bz = oz; ba = oa; have = true; }
    }
    if (have) { bound[0] = bx; bound[1] = by; bound[2] = bz; }
    sc.mark(5);
}

/* API: Area2Cloud for k query points (one wave each) */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_area2cloud_api(DevMeta *m, DynParams D, const float4 *__restrict__ sorted4,
                                                                   const int *__restrict__ slab_start,
                                                                   const float *__restrict__ slab_xmin,
                                                                   const float *__restrict__ slab_xmax,
                                                                   const float4 *__restrict__ normals4,
                                                                   const float *__restrict__ ell_cs, const double *__restrict__ pts,
                                                                   int k, int key, float *out)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    const int wv = threadIdx.x >> 6;
    const int q = blockIdx.x * DYN_WAVES + wv;
    if (q >= k) return;
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    double p[3] = {pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]};
    float b[3];
    StampCtx sc; sc.begin(15, false);
    wave_area2cloud(V, s_w[wv], normals4, ell_cs, D, p, key, b, sc);
    if ((threadIdx.x & 63) == 0) { out[3 * q] = b[0]; out[3 * q + 1] = b[1]; out[3 * q + 2] = b[2]; }
}

/* ------------------------------------------------------------------ */
/* The slice-to-slice chains (thread_worker, path_dynamic_alg.cpp:308-334; GenPath of            */
/* dynamic_alg_sdir.cpp:349-374).  Step t of chain c adjusts one slice against the boundary of   */
/* the slice adjusted in step t-1.                                                                */
/* ------------------------------------------------------------------ */
struct DynChain { int s, prev, key, active; };
__device__ inline DynChain dyn_chain(int walk, int chain, int t, int centre, int S)
{
    DynChain c;
    if (walk == 1) { /* centre-out: chain 0 = left of the centre path, chain 1 = right */
        if (chain == 0) { c.s = centre - 1 - t; c.prev = c.s + 1; c.key = 0; c.active = c.s >= 0; }
        else { c.s = centre + 1 + t; c.prev = c.s - 1; c.key = 1; c.active = c.s < S; }
    } else { /* single direction */
        c.s = 1 + t; c.prev = c.s - 1; c.key = 1; c.active = (chain == 0) && c.s < S;
    }
    return c;
}

struct DynBuffers {
    float4 *bnd_pts;  /* [chain][maxNB]  bx by bz in_loop                      */
    double *bnd_knots; /* [chain][3][maxNB + 2]  y, x, z of the boundary spline */
    int *bnd_n;        /* [chain] knots of the current boundary (0 = none yet)  */
    float4 *adj_pts;   /* [chain][maxNA]  y x z valid                           */
    int maxNB, maxNA;
};

/* Spline::point on float knots (y, x, z): same operations as GSL's steffen.c */
__device__ inline void spline_point_f(const float *ny, const float *nx, const float *nz, int mm, double dy, double out[3])
{
    auto Yf = [&](int i) { return (double)ny[i]; };
    auto Xf = [&](int i) { return (double)nx[i]; };
    auto Zf = [&](int i) { return (double)nz[i]; };
    const int iv = gsl_bsearch(mm, dy, Yf);
    out[0] = steffen_eval_at(iv, mm, dy, Yf, Xf);
    out[1] = dy;
    out[2] = steffen_eval_at(iv, mm, dy, Yf, Zf);
}

/* compute_boundary, first half (path_dynamic_alg.cpp:191-203): one wave per sample of the previous path */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_dyn_boundary_pts(DevMeta *m, DynParams D, int walk, int t, int centre,
        const float4 *__restrict__ sorted4, const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
        const float *__restrict__ slab_xmax, const float4 *__restrict__ normals4, const float *__restrict__ ell_cs,
        const float *__restrict__ node_x, const float *__restrict__ node_y, const float *__restrict__ node_z,
        const int *__restrict__ node_start, const int *__restrict__ node_cnt, DynBuffers Bf)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    StampCtx sc; sc.begin(3, blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    if (m->err) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * DYN_WAVES + wv, chain = blockIdx.y;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active || j >= Bf.maxNB) return;
    const int st = node_start[c.prev], mm = node_cnt[c.prev];
    float4 *dst = Bf.bnd_pts + (size_t)chain * Bf.maxNB + j;
    if (mm < 3) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    const double miny = (double)node_y[st], maxy = (double)node_y[st + mm - 1];
    /* dy = miny + 2; dy += toolRadius/4 per sample.  When start and step are multiples of 2^-20 (float knots and the
       usual radii are) every partial sum is exact in double and the closed form gives the same bits; else accumulate */
    const double dstep = D.tool_radius / 4, dy0 = miny + 2;
    double dy;
    if (floor(dstep * 1048576.0) == dstep * 1048576.0 && floor(dy0 * 1048576.0) == dy0 * 1048576.0 &&
        fabs(dy0) + (double)j * fabs(dstep) < 4294967296.0)
        dy = dy0 + (double)j * dstep;
    else {
        dy = dy0;
        for (int q = 0; q < j; ++q) dy += dstep;
    }
    if (!(dy < maxy - 2)) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    sc.mark(0);
    double point[3];
    spline_point_f(node_y + st, node_x + st, node_z + st, mm, dy, point);
    sc.mark(1);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    float b[3];
    wave_area2cloud(V, s_w[wv], normals4, ell_cs, D, point, c.key, b, sc);
    if (lane == 0) *dst = make_float4(b[0], b[1], b[2], 1.f);
    sc.mark(6);
}

/* compute_boundary, second half (:211-234): std::map by y (last writer wins), two extra end knots */
__global__ void __launch_bounds__(256) k_dyn_boundary_fit(DevMeta *m, int walk, int t, int centre, DynBuffers Bf)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_n, s_m;
    if (m->err) return;
    const int chain = blockIdx.x;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active) return;
    const int cap = 4096;
    u64 *keys = (u64 *)s_raw;            /* cap: (ord(by) << 31) | j */
    int *hist = (int *)(keys + cap);     /* cap + 1 */
    u64 *stage = (u64 *)(hist + cap + 2); /* cap */
    const float4 *pts = Bf.bnd_pts + (size_t)chain * Bf.maxNB;
    if (threadIdx.x == 0) { s_n = 0; s_m = 0; }
    __syncthreads();
    /* gather the valid samples in loop order (the order only matters inside equal keys, and there
       the key carries j) */
    float ymin = INFINITY, ymax = -INFINITY;
    for (int j = threadIdx.x; j < Bf.maxNB; j += blockDim.x) {
        float4 p = pts[j];
        if (p.w != 0.f && p.x == p.x && p.y == p.y) { /* isnan(boundpoint[0]) -> continue; a NaN key is skipped too */
            float y = p.y == 0.f ? 0.f : p.y;
            int slot = atomicAdd(&s_n, 1);
            if (slot < cap) stage[slot] = YK_MAKE(y, j);
            ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > cap) { if (threadIdx.x == 0) set_err(m, DERR_CAPACITY, c.s); return; }
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    __shared__ float s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = ymin; s_hi[threadIdx.x >> 6] = ymax; }
    __syncthreads();
    const float y0 = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    const float y1 = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    const int NB = next_pow2(max(n, 64));
    const float scale = (y1 > y0) ? (float)NB / (y1 - y0) : 0.f;
    auto gen = [&](int i) { return stage[i]; };
    auto bucket = [&](u64 k) { int q = (int)((ord2f(YK_Y(k)) - y0) * scale); return q < 0 ? 0 : (q >= NB ? NB - 1 : q); };
    auto less = [&](u64 a, u64 b) { return a < b; };
    block_bucket_sort(keys, n, hist, min(NB, cap), s_scr, gen, bucket, less);
    /* one knot per distinct y: the last writer (highest j) of an equal-y run */
    int mcount = 0;
    for (int q = threadIdx.x; q < n; q += blockDim.x) mcount += (q == n - 1) || (YK_Y(keys[q + 1]) != YK_Y(keys[q]));
    int node_number;
    block_exscan(mcount, s_scr, &node_number);
    if (node_number <= 2) { /* compute_boundary returns 0: the previous boundary stays; v1 has none then (Path_Generation.cpp:589-592) */
        if (walk == 3 && threadIdx.x == 0) Bf.bnd_n[chain] = 0;
        return;
    }
    double *ky = Bf.bnd_knots + (size_t)chain * 3 * (Bf.maxNB + 2), *kx = ky + (Bf.maxNB + 2), *kz = kx + (Bf.maxNB + 2);
    for (int base = 0; base < n; base += blockDim.x) {
        const int q = base + threadIdx.x;
        int keep = 0;
        if (q < n) keep = (q == n - 1) || (YK_Y(keys[q + 1]) != YK_Y(keys[q]));
        int tot;
        const int pre = block_exscan(keep, s_scr, &tot);
        const int o = s_m;
        if (keep) {
            const float4 p = pts[YK_POS(keys[q])];
            ky[1 + o + pre] = (double)(p.y == 0.f ? 0.f : p.y); kx[1 + o + pre] = (double)p.x; kz[1 + o + pre] = (double)p.z;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_m = o + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { /* get longer boundary, add new node at start & end (:223-229) */
        const int idx = node_number;
        kx[0] = kx[1]; ky[0] = ky[1] - 20; kz[0] = kz[1];
        kx[idx + 1] = kx[idx]; ky[idx + 1] = ky[idx] + 20; kz[idx + 1] = kz[idx];
        Bf.bnd_n[chain] = node_number + 2;
    }
}

/* dynamic_adjust_path, first half (path_dynamic_alg.cpp:278-295): one wave per node of the path */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_dyn_adjust_pts(DevMeta *m, DynParams D, int walk, int t, int centre,
        const float4 *__restrict__ sorted4, const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
        const float *__restrict__ slab_xmax, const float4 *__restrict__ normals4, const float *__restrict__ ell_cs,
        const float *__restrict__ node_x, const float *__restrict__ node_y, const float *__restrict__ node_z,
        const int *__restrict__ node_start, const int *__restrict__ node_cnt, DynBuffers Bf)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    StampCtx sc; sc.begin(4, blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    if (m->err) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * DYN_WAVES + wv, chain = blockIdx.y;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active || i >= Bf.maxNA) return;
    float4 *dst = Bf.adj_pts + (size_t)chain * Bf.maxNA + i;
    const int st = node_start[c.s], mm = node_cnt[c.s];
    if (mm < 3) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    const double miny = (double)node_y[st], maxy = (double)node_y[st + mm - 1];
    const int NumOfNode = (int)((maxy - miny) / 5);
    const bool v1 = walk == 3; /* Path_Generation.cpp:607: for (i = 1; i < NumOfNode; i++) */
    const int ii = v1 ? i + 1 : i;
    if (v1 ? ii >= NumOfNode : ii > NumOfNode) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    const int nb = Bf.bnd_n[chain];
    if (nb < 3) { /* no boundary: v1 leaves the path alone; v2 would read an unconstructed Spline */
        if (lane == 0) { if (v1) *dst = make_float4(0, 0, 0, 0); else set_err(m, DERR_SLICE, c.s); }
        return;
    }
    double dy = ((maxy - miny) / NumOfNode * ii) + miny;
    if (dy > maxy) dy = maxy; /* B.13: the reference aborts in GSL when the last sample lands an ulp past the last knot */
    if (!(dy >= miny && dy <= maxy)) { if (lane == 0) set_err(m, DERR_DOMAIN, c.s); return; } /* gsl_spline_eval: GSL_EDOM */
    sc.mark(0);
    double node[3];
    spline_point_f(node_y + st, node_x + st, node_z + st, mm, dy, node);
    sc.mark(1);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0};
    const double *ky = Bf.bnd_knots + (size_t)chain * 3 * (Bf.maxNB + 2), *kx = ky + (Bf.maxNB + 2);
    const double bminy = ky[0], bbigy = ky[nb - 1];
    auto BY = [&](int q) { return ky[q]; };
    auto BX = [&](int q) { return kx[q]; };
    /* bisection (:237-265), at most 6 Area2Cloud evaluations */
    for (int itr = 0; itr <= 5; ++itr) {
        float ab[3];
        wave_area2cloud(V, s_w[wv], normals4, ell_cs, D, node, c.key == 0 ? 1 : 0, ab, sc);
        if ((double)ab[1] < bminy || (double)ab[1] > bbigy) break; /* a NaN bound passes, as in the reference: the node turns NaN below */
        const int iv = gsl_bsearch(nb, (double)ab[1], BY);
        const double bpx = steffen_eval_at(iv, nb, (double)ab[1], BY, BX);
        const double norm0 = (double)ab[0] - bpx;
        if (fabs(norm0) < D.adjust_threshold) break;
        node[0] = node[0] - norm0;
        if (!(node[0] == node[0])) break;
        sc.mark(6);
    }
    sc.mark(7);
    /* kdtree.nearestKSearch(point, 3): only pointIdx[0] is used (:291-294).  B.14: a node that Area2Cloud turned NaN
       ("adjust path node NAN", :258-261) adds no knot (the reference hands the NaN to FLANN and reads whatever
       pointIdx holds afterwards) */
    const float qx = (float)node[0], qy = (float)node[1], qz = (float)node[2];
    const bool finite = fabsf(qx) <= 3.402823466e+38f && fabsf(qy) <= 3.402823466e+38f && fabsf(qz) <= 3.402823466e+38f;
    const int got = finite ? wave_knn(V, s_w[wv], qx, qy, qz, 1, D.r1, sc) : 0;
    if (lane == 0) {
        if (!finite) *dst = make_float4(0, 0, 0, 0);
        else if (got < 1) { set_err(m, DERR_QUERY, c.s); *dst = make_float4(0, 0, 0, 0); }
        else { const float4 p = V.at(s_w[wv].sel[0]); *dst = make_float4(p.y, p.x, p.z, 1.f); }
    }
    sc.mark(8);
}

/* dynamic_adjust_path, second half (:297-305): map by y, Spline::restart */
__global__ void __launch_bounds__(256) k_dyn_adjust_fit(DevMeta *m, int walk, int t, int centre, DynBuffers Bf, float *node_x,
                                                        float *node_y, float *node_z, int node_cap, int *node_start, int *node_cnt)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_n, s_m, s_base;
    __shared__ float s_lo[4], s_hi[4];
    if (m->err) return;
    const int chain = blockIdx.x;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active) return;
    if (walk == 3 && Bf.bnd_n[chain] < 3) return; /* "generate boundary fail": the path stays as fitted */
    const int cap = 4096;
    u64 *keys = (u64 *)s_raw;
    int *hist = (int *)(keys + cap);
    u64 *stage = (u64 *)(hist + cap + 2);
    const float4 *pts = Bf.adj_pts + (size_t)chain * Bf.maxNA;
    if (threadIdx.x == 0) { s_n = 0; s_m = 0; }
    __syncthreads();
    float ymin = INFINITY, ymax = -INFINITY;
    for (int i = threadIdx.x; i < Bf.maxNA; i += blockDim.x) {
        float4 p = pts[i];
        if (p.w != 0.f) {
            float y = p.x == 0.f ? 0.f : p.x; /* adj_pts = (y, x, z, valid) */
            int slot = atomicAdd(&s_n, 1);
            if (slot < cap) stage[slot] = YK_MAKE(y, i);
            ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > cap) { if (threadIdx.x == 0) set_err(m, DERR_CAPACITY, c.s); return; }
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = ymin; s_hi[threadIdx.x >> 6] = ymax; }
    __syncthreads();
    const float y0 = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    const float y1 = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    const int NB = next_pow2(max(n, 64));
    const float scale = (y1 > y0) ? (float)NB / (y1 - y0) : 0.f;
    auto gen = [&](int i) { return stage[i]; };
    auto bucket = [&](u64 k) { int q = (int)((ord2f(YK_Y(k)) - y0) * scale); return q < 0 ? 0 : (q >= NB ? NB - 1 : q); };
    auto less = [&](u64 a, u64 b) { return a < b; };
    block_bucket_sort(keys, n, hist, min(NB, cap), s_scr, gen, bucket, less);
    int mcount = 0;
    for (int q = threadIdx.x; q < n; q += blockDim.x) mcount += (q == n - 1) || (YK_Y(keys[q + 1]) != YK_Y(keys[q]));
    int tot;
    block_exscan(mcount, s_scr, &tot);
    if (threadIdx.x == 0) {
        int base = 0;
        if (tot < 3) set_err(m, DERR_SLICE, c.s); /* gsl_spline_alloc */
        else {
            base = atomicAdd(&m->node_cursor, tot);
            if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, c.s); base = 0; tot = 0; }
        }
        s_base = base;
    }
    __syncthreads();
    if (m->err) return;
    float *ox = node_x + s_base, *oy = node_y + s_base, *oz = node_z + s_base;
    for (int base = 0; base < n; base += blockDim.x) {
        const int q = base + threadIdx.x;
        int keep = 0;
        if (q < n) keep = (q == n - 1) || (YK_Y(keys[q + 1]) != YK_Y(keys[q]));
        int t2;
        const int pre = block_exscan(keep, s_scr, &t2);
        const int o = s_m;
        if (keep) {
            const float4 p = pts[YK_POS(keys[q])];
            oy[o + pre] = p.x == 0.f ? 0.f : p.x; ox[o + pre] = p.y; oz[o + pre] = p.z;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_m = o + t2;
        __syncthreads();
    }
    if (threadIdx.x == 0) { node_start[c.s] = s_base; node_cnt[c.s] = s_m; }
}
