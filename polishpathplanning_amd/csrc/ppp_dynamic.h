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
#define DYN_ELL_PER ((DYN_ELL + 63) / 64) /* ellipse angles per lane */

struct DynParams {
    double tool_radius, depth, toolthickness, adjust_threshold;
    int k;          /* 50: path_dynamic_alg.cpp:87 */
    float r0;       /* first search radius of the k-NN gather */
    float r1;       /* first search radius of the 1-NN snap   */
};

struct __attribute__((aligned(16))) DynWaveLds {
    u64 key[DYN_KNN_CAP];  /* (bits of the squared distance) << 32 | cloud index: one compare ranks a candidate */
    u32 dk[DYN_KNN_CAP + 16]; /* the distance bits alone (d >= 0: they order like the value): what the ranking reads, four per access */
    int pos[DYN_KNN_CAP];
    int cnt[64];           /* candidates per rank: more than one = equal distances, ranked again on the whole key */
    int sel[64];
    int sel_id[64]; /* cloud index of neighbour r (the low half of its key) */
    int sel_slot[64]; /* the candidate slot neighbour r was kept in */
    int off[65];  /* exclusive prefix of the y-window sizes of 64 neighbouring slabs */
    int w0[64];   /* first position of each window                                  */
};

/* The slab grid and the y-bucket scale of the pass: read ONCE when a kernel starts, together with its other first reads, so
   that no search begins with a trip to the meta block (slab_of / ytab_bucket on the values, same expressions). */
struct DynGrid { int B, total; float x0, invw, y0, ysc; };
__device__ inline DynGrid dyn_grid(const DevMeta *m)
{
    DynGrid G;
    G.B = m->B; G.total = m->n_sorted; G.x0 = m->slab_x0; G.invw = m->slab_invw; G.y0 = m->mn[1]; G.ysc = m->ytab_scale;
    return G;
}
__device__ inline int dyn_slab_of(const DynGrid &G, float x)
{
    int b = (int)((x - G.x0) * G.invw);
    b = b < 0 ? 0 : b;
    return b >= G.B ? G.B - 1 : b;
}
__device__ inline int dyn_ybucket(const DynGrid &G, float y)
{
    int q = (int)((y - G.y0) * G.ysc);
    q = q < 0 ? 0 : q;
    return q >= YTB ? YTB - 1 : q;
}
/* exact k nearest neighbours of q (ascending (distance, cloud index)): returns kk <= k, positions
   in L.sel[0..kk).  All 64 lanes of the wave call this together.
   Every slab the search ball touches gives its y-window from its y-bucket row (own lane each); the windows are then walked
   as one flat list, 64 candidates per step.  With `normals4` the normals of the candidates inside the ball are requested as
   soon as the ball is settled and travel while the ranking runs; lane r returns neighbour r's in nn[] (zeros beyond kk).
   (Measured and dropped in round 3: loading the windows one after the other, lane = place inside the window, eight windows in
   flight -- no flat index to search, but two and a half times the instructions: 3.65 -> 3.88 ms at cfg 2; an LDS copy of the
   index rows of the step's slabs, requested with the kernel's first reads so that a search goes straight to its candidates:
   3.54 -> 3.69 ms, the rows come from the L2 faster than nine more loads per thread cost; ranking in registers -- candidate j's
   distance broadcast from its lane, its rank the population count of two compare masks, scalar arithmetic only --: the ranking
   1.9 -> 5.4 us; the rank-order sums with all their LDS reads issued ahead of the chain of additions: no faster, and 30 more
   registers cost k_dyn_first_eval a wave per SIMD, 159 -> 200 us; the index rows requested from the x of the knot in front of a
   sample while its spline is still being evaluated: 3.42 -> 3.45 ms.) */
__device__ inline int wave_knn(const SlabView &V, const DynGrid &G, DynWaveLds &L, float qx, float qy, float qz, int k,
                               float r0, const float4 *__restrict__ normals4, float nn[3], StampCtx &sc)
{
    const int lane = threadIdx.x & 63;
    const int B = G.B;
    const int total = G.total;
    float r = r0;
    int count = 0;
    for (int attempt = 0; attempt < 48; ++attempt) {
        const float r2 = r * r;
        count = 0;
        bool overflow = false;
        const float pady = 1e-5f * (fabsf(qy) + r) + 1e-6f, padx = 1e-5f * (fabsf(qx) + r) + 1e-6f;
        const float ylo = qy - r - pady, yhi = qy + r + pady;
        /* slab_of is monotone and is what binned the points: every point with |x - qx| <= r lies in [blo, bhi] (the band gathers
           of the hot path rest on the same argument); a slab of margin either side was a third more candidates for nothing */
        int blo = dyn_slab_of(G, qx - r - padx), bhi = dyn_slab_of(G, qx + r + padx);
        blo = blo < 0 ? 0 : blo;
        bhi = bhi >= B ? B - 1 : bhi;
        const int q0 = dyn_ybucket(G, ylo), q1 = dyn_ybucket(G, yhi) + 1;
        auto keep = [&](bool in, float d, int id, int pos) { /* candidates inside the ball take the next slots, in lane order */
            const u64 mask = __ballot(in);
            if (in) {
                const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
                if (slot < DYN_KNN_CAP) { L.key[slot] = ((u64)__float_as_uint(d) << 32) | (u32)id; L.dk[slot] = __float_as_uint(d); L.pos[slot] = pos; }
            }
            count += __popcll(mask);
            if (count > DYN_KNN_CAP) overflow = true;
        };
        for (int cb = blo; cb <= bhi && !overflow; cb += 64) {
            const int bb = cb + lane;
            int a = 0, e = 0;
            if (bb <= bhi) {
                if (V.ytab) {
                    /* the slab's y-bucket row bounds the window from outside (bucket() is monotone in y): no search at all,
                       a few candidates more -- each is tested against r2 below anyway */
                    const int s0 = V.slab_start[bb];
                    const int *T = V.ytab + (size_t)bb * (YTB + 1);
                    a = s0 + T[q0]; e = s0 + T[q1];
                } else {
                    const int s0 = V.slab_start[bb];
                    const int s1 = V.slab_start[bb + 1];
                    int l0 = s0, l1 = s1, u0 = s0, u1 = s1; /* first y >= ylo, first y > yhi */
                    while (l0 < l1 || u0 < u1) {
                        if (l0 < l1) { const int mid = (l0 + l1) >> 1; if (V.at(mid).y < ylo) l0 = mid + 1; else l1 = mid; }
                        if (u0 < u1) { const int mid = (u0 + u1) >> 1; if (V.at(mid).y <= yhi) u0 = mid + 1; else u1 = mid; }
                    }
                    a = l0; e = u0 < l0 ? l0 : u0;
                }
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
            const int wtop = bhi - cb < 63 ? bhi - cb : 63; /* windows past it are empty */
            for (int base = 0; base < T && !overflow; base += 256) {
                /* four rounds of 64 candidates: their reads are in flight together */
                float4 c4[4];
                int i4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = base + 64 * u + lane;
                    i4[u] = -1;
                    if (t < T) {
                        int lo = 0, hi = wtop; /* the window holding flat position t: last lane with off <= t */
                        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L.off[mid] <= t) lo = mid; else hi = mid - 1; }
                        i4[u] = L.w0[lo] + (t - L.off[lo]);
                        c4[u] = V.at(i4[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (base + 64 * u >= T || overflow) break;
                    bool in = false;
                    float d = 0.f;
                    int id = 0;
                    if (i4[u] >= 0) {
                        d = dist2_flann(qx, qy, qz, c4[u].x, c4[u].y, c4[u].z);
                        id = idx_of(c4[u]);
                        in = d <= r2;
                    }
                    keep(in, d, id, i4[u]);
                }
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
    /* the normals of the candidates inside the ball (two per lane: slots lane and lane + 64) are requested now and travel
       while the ranking runs; a fuller ball gathers the k it needs afterwards, as before */
    const bool pre = normals4 != nullptr && count <= 128;
    float4 nq0 = make_float4(0.f, 0.f, 0.f, 0.f), nq1 = nq0;
    if (pre) {
        if (lane < count) nq0 = normals4[(u32)L.key[lane]];
        if (lane + 64 < count) nq1 = normals4[(u32)L.key[lane + 64]];
    }
    /* rank by counting: (distance, cloud index) is a total order.  Distances are almost always all different, so a candidate's
       rank is the number of smaller DISTANCES -- 32-bit compares on values read four per LDS access (every lane the same
       address: a broadcast), two candidates per lane in one sweep; candidates that land on the same rank (equal distances)
       are ranked again on the whole key.  Only ranks below k matter. */
    const int cpad = (count + 7) & ~7;
    for (int c = count + lane; c < cpad; c += 64) L.dk[c] = 0xffffffffu;
    L.cnt[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int rr[(DYN_KNN_CAP + 127) / 128][2];
#pragma unroll
    for (int sw = 0; sw < (DYN_KNN_CAP + 127) / 128; ++sw) {
        rr[sw][0] = rr[sw][1] = 0x7fffffff;
        if (sw * 128 >= count) continue; /* (wave-uniform) */
        const int c0 = sw * 128 + lane, c1 = c0 + 64;
        const u32 d0 = c0 < count ? L.dk[c0] : 0u, d1 = c1 < count ? L.dk[c1] : 0u;
        int r0 = 0, r1 = 0;
        const uint4 *dv = (const uint4 *)L.dk;
        for (int o = 0; o < cpad / 4; o += 2) {
            const uint4 a = dv[o], b = dv[o + 1];
            r0 += (a.x < d0) + (a.y < d0) + (a.z < d0) + (a.w < d0) + (b.x < d0) + (b.y < d0) + (b.z < d0) + (b.w < d0);
            r1 += (a.x < d1) + (a.y < d1) + (a.z < d1) + (a.w < d1) + (b.x < d1) + (b.y < d1) + (b.z < d1) + (b.w < d1);
        }
        if (c0 < count && r0 < kk) { rr[sw][0] = r0; atomicAdd(&L.cnt[r0], 1); }
        if (c1 < count && r1 < kk) { rr[sw][1] = r1; atomicAdd(&L.cnt[r1], 1); }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
#pragma unroll
    for (int sw = 0; sw < (DYN_KNN_CAP + 127) / 128; ++sw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rank = rr[sw][h];
            if (rank == 0x7fffffff) continue;
            const int c = sw * 128 + 64 * h + lane;
            const u64 kc = L.key[c];
            if (L.cnt[rank] > 1) { /* equal distances: the whole key decides */
                rank = 0;
                for (int j = 0; j < count; ++j) rank += L.key[j] < kc;
            }
            if (rank < kk) { L.sel[rank] = L.pos[c]; L.sel_id[rank] = (int)(u32)kc; L.sel_slot[rank] = c; }
        }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    sc.mark(3);
    if (normals4 != nullptr) { /* lane r: the normal of neighbour r */
        nn[0] = nn[1] = nn[2] = 0.f;
        if (pre) {
            const int sl = lane < kk ? L.sel_slot[lane] : 0;
            const int from = sl & 63;
            const float x0 = __shfl(nq0.x, from, 64), y0 = __shfl(nq0.y, from, 64), z0 = __shfl(nq0.z, from, 64);
            const float x1 = __shfl(nq1.x, from, 64), y1 = __shfl(nq1.y, from, 64), z1 = __shfl(nq1.z, from, 64);
            if (lane < kk) { nn[0] = sl >= 64 ? x1 : x0; nn[1] = sl >= 64 ? y1 : y0; nn[2] = sl >= 64 ? z1 : z0; }
        } else if (lane < kk) {
            const float4 nv = normals4[L.sel_id[lane]];
            nn[0] = nv.x; nn[1] = nv.y; nn[2] = nv.z;
        }
    }
    sc.mark(9);
    return kk;
}

/* the 721 ellipse angles (cos, sin) as a table in LDS: the whole workgroup stages it when the kernel starts (a barrier follows
   before the first use) */
__device__ inline void dyn_stage_ellipse(const float *__restrict__ ell_cs, float2 *s_ell)
{
    for (int a = threadIdx.x; a < DYN_ELL; a += blockDim.x) s_ell[a] = ((const float2 *)ell_cs)[a];
}

/* Area2Cloud(point, flag, key): key 0 = left (min x), 1 = right (max x).  Wave-cooperative.  Returns the number of neighbours
   its search found (0: none, or the point is not a number); L.sel[0] / L.sel_id[0] then still hold the nearest of them, which is
   what the 1-NN snap of the same point asks for. */
__device__ inline int wave_area2cloud(const SlabView &V, const DynGrid &G, DynWaveLds &L, const float4 *__restrict__ normals4,
                                       const float2 *ell, const DynParams &D, const double point[3], int key,
                                       float bound[3], StampCtx &sc)
{
    const int lane = threadIdx.x & 63;
    const float sp[3] = {(float)point[0], (float)point[1], (float)point[2]};
    bound[0] = bound[1] = bound[2] = NAN;
    if (!(sp[0] == sp[0] && sp[1] == sp[1] && sp[2] == sp[2])) return 0;
    /* computePointPrincipalCurvatures: lane r holds the neighbour of rank r */
    float nn[3] = {0.f, 0.f, 0.f};
    const int kk = wave_knn(V, G, L, sp[0], sp[1], sp[2], D.k, D.r0, normals4, nn, sc);
    if (kk <= 0) return 0;
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
       them).  The nine sums are independent of each other, so each gets a lane of its own: the values go through LDS
       (the key area is free after the ranking), lane a < 3 walks component a, then lane a < 6 walks one product --
       64 dependent additions per phase instead of 64 x 3 and 64 x 6.  Lanes >= kk hold +0: adding it is exact. */
    auto lane_value = [](float v, int r) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r)); }; /* r is wave-uniform */
    float *sv = (float *)L.key; /* [3][64] */
    const int kk4 = (kk + 3) >> 2;
    __builtin_amdgcn_wave_barrier();
    sv[lane] = proj[0]; sv[64 + lane] = proj[1]; sv[128 + lane] = proj[2];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float acc = 0.f;
    {
        const float4 *row = (const float4 *)(sv + 64 * (lane < 3 ? lane : 0));
#pragma unroll 4
        for (int r = 0; r < kk4; ++r) { const float4 v = row[r]; acc += v.x; acc += v.y; acc += v.z; acc += v.w; }
    }
    float cen[3];
    for (int i = 0; i < 3; ++i) cen[i] = lane_value(acc, i) / (float)kk;
    float d[3] = {0.f, 0.f, 0.f};
    if (lane < kk) for (int i = 0; i < 3; ++i) d[i] = proj[i] - cen[i];
    __builtin_amdgcn_wave_barrier();
    sv[lane] = d[0]; sv[64 + lane] = d[1]; sv[128 + lane] = d[2];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    acc = 0.f;
    {
        /* lane 0..5: xx xy xz yy yz zz */
        const int ia = lane < 3 ? 0 : (lane < 5 ? 1 : 2), ja = lane < 3 ? lane : (lane < 5 ? lane - 2 : 2);
        const float4 *ra = (const float4 *)(sv + 64 * (lane < 6 ? ia : 0)), *rb = (const float4 *)(sv + 64 * (lane < 6 ? ja : 0));
#pragma unroll 4
        for (int r = 0; r < kk4; ++r) {
            const float4 a = ra[r], b = rb[r];
            acc += a.x * b.x; acc += a.y * b.y; acc += a.z * b.z; acc += a.w * b.w;
        }
    }
    float cov[9];
    cov[0] = lane_value(acc, 0); cov[1] = lane_value(acc, 1); cov[2] = lane_value(acc, 2);
    cov[4] = lane_value(acc, 3); cov[5] = lane_value(acc, 4); cov[8] = lane_value(acc, 5);
    sc.mark(10);
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
    sc.mark(11);
    /* 721-point ellipse, transformed (SSE order c0*x + (c1*y + (c2*0 + c3))), first extremum in x.  Returns 1 with the point,
       0 when no sample is a number, 2 when the fold is poisoned (below) */
    auto ellipse_extremum = [&](const bool allow_window, float res[3]) -> int {
    float bx = 0.f, by = 0.f, bz = 0.f;
    int ba = 0x7fffffff;
    bool have = false, first_nan = false;
    auto sample = [&](int a) {
        const float2 csn = ell[a];
        const float ex = (float)(longAxis * (double)csn.x);
        const float ey = (float)(shortAxis * (double)csn.y);
        float t[3];
        for (int i = 0; i < 3; ++i) t[i] = cr[i] * ex + (cv[i] * ey + (n0[i] * 0.f + sp[i]));
        if (a == 0 && !(t[0] == t[0])) first_nan = true;
        if (t[0] == t[0]) {
            if (!have || (key == 1 ? (bx < t[0]) : (t[0] < bx))) { bx = t[0]; by = t[1]; bz = t[2]; ba = a; have = true; }
        }
    };
    /* Where the extremum can be is known beforehand: x(theta) = A cos theta + B sin theta + C peaks at atan2(B, A) (the minimum
       half a turn further), and a sample k half-degree steps away from the peak lies 0.49 R (k delta)^2 below it (R = |(A, B)|)
       while the float evaluation of a sample is off by at most E = 4 ulp(|C| + R) + 8 R 2^-23.  Samples further away than
       K = sqrt((2 E / (R delta^2) + 1/8) / 0.49) steps cannot hold the extremum, nor tie with it; when K < 27 the 63 samples
       around the peak (and the 360-degree sample when the 0-degree one is among them) are all that is evaluated, one per
       lane instead of twelve -- the same values compared in the same order, so the same sample wins. */
    bool windowed = false;
    int a_star = 0;
#ifndef DYN_ELL_FULL
    if (allow_window) { /* (float arithmetic is plenty for locating the window; its width carries the slack) */
        const float A = cr[0] * (float)longAxis, Bq = cv[0] * (float)shortAxis;
        const float R = sqrtf(A * A + Bq * Bq), Cm = fabsf(sp[0]) + R;
        if (R > 0.f && R < 1e30f && Cm < 1e30f && cr[1] == cr[1] && cr[2] == cr[2] && cv[1] == cv[1] && cv[2] == cv[2]) {
            const float ulp = __uint_as_float(__float_as_uint(Cm) & 0x7f800000u) * 1.1920929e-7f; /* of a float of Cm's size */
            const float E = 4.f * ulp + 8.f * R * 1.1920929e-7f;
            const float dl = 0.008726646f; /* half a degree */
            const float K2 = (2.f * E / (R * dl * dl) + 0.125f) / 0.49f;
            if (K2 < 27.f * 27.f) {
                float th = atan2f(Bq, A);
                if (key != 1) th += 3.14159265f;
                if (th < 0.f) th += 6.2831853f;
                a_star = (int)floorf(th / dl + 0.5f) % 720;
                windowed = true;
            }
        }
    }
#endif
    if (windowed) {
        int a = (a_star + lane - 31 + 720) % 720;
        const bool zero_in = __ballot(lane < 63 && a == 0) != 0;
        if (lane == 63) a = 720;
        if (lane < 63 || zero_in) sample(a);
    } else {
#pragma unroll
        for (int q = 0; q < DYN_ELL_PER; ++q) {
            const int a = lane + 64 * q;
            if (a >= DYN_ELL) break;
            sample(a);
        }
    }
    /* the reference folds in ascending angle: a NaN at angle 0 poisons the whole fold */
    sc.mark(12);
    if (__ballot(first_nan)) return 2;
    /* the wave's extremum: one 64-bit key per lane -- x in an order-preserving integer form (complemented for the minimum; a
       zero of either sign is one value), then the earlier angle first -- maximised over the lanes; the lane that held it
       hands over the point */
    u64 ek = 0;
    if (have) {
        const u32 ox = f2ord(bx == 0.f ? 0.f : bx);
        ek = ((u64)(key == 1 ? ox : ~ox) << 32) | (u32)(0x7fffffff - ba);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const u32 hi = __shfl_xor((u32)(ek >> 32), o, 64), lo = __shfl_xor((u32)ek, o, 64);
        const u64 other = ((u64)hi << 32) | lo;
        ek = other > ek ? other : ek;
    }
    have = ek != 0;
    if (have) {
        const int wa = 0x7fffffff - (int)(u32)ek; /* the winning angle: in lane a mod 64, or at its place in the window */
        const int wl = __builtin_amdgcn_readfirstlane(windowed ? (wa == 720 ? 63 : (wa - a_star + 31 + 720) % 720) : (wa & 63));
        bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), wl));
        by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), wl));
        bz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), wl));
    }
    if (have) { res[0] = bx; res[1] = by; res[2] = bz; }
    return have ? 1 : 0;
    };
    float res[3];
    const int st = ellipse_extremum(true, res);
#ifdef DYN_ELL_CHECK /* test build: the windowed evaluation against all 721 samples, every time */
    {
        float ref[3];
        const int st2 = ellipse_extremum(false, ref);
        if (st != st2 || (st == 1 && (__float_as_uint(res[0]) != __float_as_uint(ref[0]) || __float_as_uint(res[1]) != __float_as_uint(ref[1]) ||
                                      __float_as_uint(res[2]) != __float_as_uint(ref[2]))))
            set_err(const_cast<DevMeta *>(V.m), DERR_DOMAIN, -1);
    }
#endif
    if (st == 1) { bound[0] = res[0]; bound[1] = res[1]; bound[2] = res[2]; }
    sc.mark(5);
    return kk;
}

/* API: Area2Cloud for k query points (one wave each) */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_area2cloud_api(DevMeta *m, DynParams D, const float4 *__restrict__ sorted4,
                                                                   const int *__restrict__ slab_start,
                                                                   const float *__restrict__ slab_xmin,
                                                                   const float *__restrict__ slab_xmax,
                                                                   const float4 *__restrict__ normals4,
                                                                   const float *__restrict__ ell_cs, const int *__restrict__ ytab,
                                                                   const double *__restrict__ pts, int k, int key, float *out)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    __shared__ float2 s_ell[DYN_ELL];
    dyn_stage_ellipse(ell_cs, s_ell);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const int q = blockIdx.x * DYN_WAVES + wv;
    if (q >= k) return;
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0, ytab};
    const DynGrid G = dyn_grid(m);
    double p[3] = {pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]};
    float b[3];
    StampCtx sc; sc.begin(15, false);
    wave_area2cloud(V, G, s_w[wv], normals4, s_ell, D, p, key, b, sc);
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
    double *bnd_knots; /* [slot][3][maxNB + 2]  y, x, z of a boundary spline; slot = the slice adjusted against it (every boundary of the
                          pass is kept: drawpath(*boundary, ...) of the viewer, ppp_get_boundary) or, keep_all == 0, the chain */
    int *bnd_n;        /* [chain] knots of the current boundary (0 = none yet)  */
    int *bnd_slot;     /* [chain] the slot that boundary is in                  */
    int *bnd_all_n;    /* [slice] knots of the boundary in the slice's slot (0 = compute_boundary gave none at that step) */
    int keep_all;
    float4 *adj_pts;   /* [chain][maxNA]  y x z valid                           */
    int maxNB, maxNA;
    /* what k_dyn_first_eval leaves for every node i of every slice s, at [s][i] (it does not depend on the chain) */
    float4 *first_ab;   /* Area2Cloud of the node as sampled: x y z, w = 0 no such node, 1 evaluated, 2 sample outside the spline */
    double *first_node; /* [3]: the node as sampled                                                                           */
    float4 *first_snap; /* its nearest cloud point: y x z, w = 1 found, 0 node not finite, 2 no answer                        */
};

/* LDS of the chain kernels: the scratch of a fit (keys, their staging, the samples themselves while there are at most
   2048 of them, a histogram) shares its bytes with the per-wave search areas that are used after it; the knots the fit
   produces follow */
__host__ __device__ inline int dyn_fit_cap(int maxN) { const int c = maxN < 64 ? 64 : maxN; return c > 4096 ? 4096 : c; }
__host__ __device__ inline bool dyn_fit_keeps_samples(int cap) { return cap <= 2048; }
__host__ __device__ inline size_t dyn_scratch_bytes(int maxN)
{
    const int cap = dyn_fit_cap(maxN);
    const size_t s = (dyn_fit_keeps_samples(cap) ? 36 : 20) * (size_t)cap + 16, w = DYN_WAVES * sizeof(DynWaveLds);
    return ((s > w ? s : w) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t dyn_boundary_pts_lds(int maxNA) { return dyn_scratch_bytes(maxNA) + 12 * (size_t)dyn_fit_cap(maxNA); }
__host__ __device__ inline size_t dyn_adjust_pts_lds(int maxNB) { return dyn_scratch_bytes(maxNB) + 16 * ((size_t)dyn_fit_cap(maxNB) + 2); }
struct DynFitLds { u64 *keys, *stage; float4 *pay; int *hist; };
__device__ inline DynFitLds dyn_fit_lds(char *raw, int cap)
{
    DynFitLds F;
    F.keys = (u64 *)raw; F.stage = F.keys + cap;
    F.pay = dyn_fit_keeps_samples(cap) ? (float4 *)(F.stage + cap) : nullptr;
    F.hist = F.pay ? (int *)(F.pay + cap) : (int *)(F.stage + cap);
    return F;
}

/* A fit, first part: the samples of one chain step are requested and staged (with their sort keys; ~0 marks an invalid
   one).  No barrier in here: a kernel starts with it, so that these reads travel together with its other first reads.
   Returns the thread's number of valid samples. */
template <typename Valid, typename YOf>
__device__ inline int dyn_stage_samples(const float4 *pts, int maxN, const DynFitLds &F, Valid valid, YOf yof)
{
    int nv = 0;
#pragma unroll 4
    for (int j = threadIdx.x; j < maxN; j += blockDim.x) {
        const float4 p = pts[j];
        if (F.pay) F.pay[j] = p;
        u64 key = ~0ull;
        if (valid(p)) {
            float y = yof(p);
            y = y == 0.f ? 0.f : y;
            key = YK_MAKE(y, j);
            ++nv;
        }
        F.stage[j] = key;
    }
    return nv;
}

/* Second part: the staged samples sorted by (y, sample number) -- the order of the reference's std::map insertions, whose
   last writer wins: one knot per distinct y, the highest sample number of an equal-y run --, invalid ones behind the valid
   ones.  The samples are taken at ascending y and what is kept of each lies close to it, so as a rule they arrive in that
   order already, every y new (`identity`: knot o is sample o) -- found out in one pass; otherwise they are sorted (buckets
   over the cloud's y range: they only have to be monotone in y) or, if only equal y occur, just counted.  maxN <= cap.
   Whole workgroup. */
struct DynFit { int n, tot, identity, pre, q0, q1; };
__device__ inline DynFit dyn_fit_prepare(int nv, int maxN, const DynFitLds &F, int cap, int *scr17, const DevMeta *m)
{
    __shared__ int s_n, s_flags;
    if (threadIdx.x == 0) { s_n = 0; s_flags = 0; }
    nv = wave_sum(nv);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && nv) atomicAdd(&s_n, nv);
    int flags = 0; /* 1: not the identity, 2: not even sorted */
    for (int j = threadIdx.x; j < maxN; j += blockDim.x) {
        const u64 k = F.stage[j];
        F.keys[j] = k;
        if (j + 1 < maxN) {
            const u64 k1 = F.stage[j + 1];
            if (k1 != ~0ull && (k == ~0ull || YK_Y(k) >= YK_Y(k1))) flags |= 1;
            if (k > k1) flags |= 2;
        }
    }
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    flags = s_flags;
#ifdef DYN_FIT_FORCE /* test builds: take the counting (1) or the sorting (3) path whatever the samples look like */
    flags |= DYN_FIT_FORCE;
#endif
    DynFit ft;
    ft.n = s_n; ft.identity = !(flags & 1); ft.pre = ft.q0 = ft.q1 = 0;
    if (ft.identity) { ft.tot = ft.n; return ft; }
    if (flags & 2) {
        const float y0 = m->mn[1], y1 = m->mx[1];
        const int NB = min(next_pow2(max(maxN, 64)), cap);
        const float scale = (y1 > y0) ? (float)NB / (y1 - y0) : 0.f;
        auto gen = [&](int i) { return F.stage[i]; };
        auto bucket = [&](u64 k) {
            if (k == ~0ull) return NB - 1;
            const int q = (int)((ord2f(YK_Y(k)) - y0) * scale);
            return q < 0 ? 0 : (q >= NB ? NB - 1 : q);
        };
        auto less = [&](u64 a, u64 b) { return a < b; };
        block_bucket_sort(F.keys, maxN, F.hist, NB, scr17, gen, bucket, less);
    }
    /* every thread takes a run of consecutive positions [q0, q1); pre = the place of its first knot */
    const int n = ft.n;
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    ft.q0 = min((int)threadIdx.x * per, n); ft.q1 = min(ft.q0 + per, n);
    int c = 0;
    for (int q = ft.q0; q < ft.q1; ++q) c += (q == n - 1) || (YK_Y(F.keys[q + 1]) != YK_Y(F.keys[q]));
    ft.pre = block_exscan(c, scr17, &ft.tot);
    return ft;
}
/* emit(o, sample) for knot o = 0 .. tot-1 (no barrier in here) */
template <typename Emit>
__device__ inline void dyn_emit_knots(const float4 *pts, const DynFitLds &F, const DynFit &ft, Emit emit)
{
    if (ft.identity) {
        for (int j = threadIdx.x; j < ft.n; j += blockDim.x) emit(j, F.pay ? F.pay[j] : pts[j]);
        return;
    }
    int o = ft.pre;
    for (int q = ft.q0; q < ft.q1; ++q)
        if ((q == ft.n - 1) || (YK_Y(F.keys[q + 1]) != YK_Y(F.keys[q]))) {
            const int j = YK_POS(F.keys[q]);
            emit(o++, F.pay ? F.pay[j] : pts[j]);
        }
}

/* gsl_interp_bsearch over ascending float knots for a finite dy, by the whole wave: the largest i in [0, mm-2] with
   y[i] <= dy (0 if there is none) from two rounds of 64 reads instead of log2(mm) dependent ones */
__device__ inline int wave_gsl_bsearch_f(const float *ny, int mm, double dy)
{
    const int lane = threadIdx.x & 63;
    const int top = mm - 1;
    const int stride = (top + 63) >> 6;
    if (stride > 64) { auto Yf = [&](int i) { return (double)ny[i]; }; return gsl_bsearch(mm, dy, Yf); }
    const int i1 = lane * stride;
    const u64 m1 = __ballot(i1 < top && (double)ny[i1] <= dy);
    if (m1 == 0) return 0;
    const int seg = (63 - __clzll(m1)) * stride;
    const int i2 = seg + lane;
    const u64 m2 = __ballot(lane < stride && i2 < top && (double)ny[i2] <= dy);
    return seg + (63 - __clzll(m2));
}

/* the same over double knots, for any dy (a NaN takes gsl_interp_bsearch's own way through the comparisons) */
__device__ inline int wave_gsl_bsearch_d(const double *ny, int mm, double dy)
{
    const int lane = threadIdx.x & 63;
    const int top = mm - 1;
    const int stride = (top + 63) >> 6;
    if (stride > 64 || !(dy == dy)) { auto Yd = [&](int i) { return ny[i]; }; return gsl_bsearch(mm, dy, Yd); }
    const int i1 = lane * stride;
    const u64 m1 = __ballot(i1 < top && ny[i1] <= dy);
    if (m1 == 0) return 0;
    const int seg = (63 - __clzll(m1)) * stride;
    const int i2 = seg + lane;
    const u64 m2 = __ballot(lane < stride && i2 < top && ny[i2] <= dy);
    return seg + (63 - __clzll(m2));
}

/* Spline::point on float knots (y, x, z): same operations as GSL's steffen.c.  Called by all lanes of a wave together. */
__device__ inline void spline_point_f(const float *ny, const float *nx, const float *nz, int mm, double dy, double out[3])
{
    auto Yf = [&](int i) { return (double)ny[i]; };
    const int iv = wave_gsl_bsearch_f(ny, mm, dy);
    /* the two splines side by side instead of one after the other: even lanes evaluate y -> x, odd lanes y -> z (twenty
       double-precision divisions between them), lanes 0 and 1 hand the results to the wave */
    const float *nv = (threadIdx.x & 1) ? nz : nx;
    auto Vf = [&](int i) { return (double)nv[i]; };
    const double v = steffen_eval_at(iv, mm, dy, Yf, Vf);
    auto lane_double = [](double d, int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(d), l), hi = __builtin_amdgcn_readlane(__double2hiint(d), l);
        return __hiloint2double(hi, lo);
    };
    out[0] = lane_double(v, 0);
    out[1] = dy;
    out[2] = lane_double(v, 1);
}

/* The first Area2Cloud of dynamic_adjust_path's bisection (path_dynamic_alg.cpp:237-243) is taken at the node as sampled
   from the slice's own spline (:278-284), and so is the final snap (:291-294) of a node the bisection leaves where it is:
   neither depends on the neighbour's boundary.  They are evaluated here for all slices at once, ahead of the chain, which
   then starts every node at its first comparison.  One wave per node; blockIdx.y = slice. */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_dyn_first_eval(DevMeta *m, DynParams D, int walk, int centre,
        const float4 *__restrict__ sorted4, const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
        const float *__restrict__ slab_xmax, const float4 *__restrict__ normals4, const float *__restrict__ ell_cs, const int *__restrict__ ytab,
        const float *__restrict__ node_x, const float *__restrict__ node_y, const float *__restrict__ node_z,
        const int *__restrict__ node_start, const int *__restrict__ node_cnt, DynBuffers Bf)
{
    __shared__ DynWaveLds s_w[DYN_WAVES];
    __shared__ float2 s_ell[DYN_ELL];
    StampCtx sc; sc.begin(6, blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0);
    dyn_stage_ellipse(ell_cs, s_ell);
    const DynGrid G = dyn_grid(m);
    __syncthreads();
    if (m->err) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.y, i = blockIdx.x * DYN_WAVES + wv;
    if (s >= m->S || i >= Bf.maxNA) return;
    int key; /* of the chain that adjusts slice s */
    if (walk == 1) { if (s == centre) return; key = s < centre ? 0 : 1; }
    else { if (s == 0) return; key = 1; }
    const size_t at = (size_t)s * Bf.maxNA + i;
    auto none = [&](float status) { if (lane == 0) Bf.first_ab[at] = make_float4(0.f, 0.f, 0.f, status); };
    const int st = node_start[s], mm = node_cnt[s];
    if (mm < 3) { none(0.f); return; }
    const double miny = (double)node_y[st], maxy = (double)node_y[st + mm - 1];
    const int NumOfNode = (int)((maxy - miny) / 5);
    const bool v1 = walk == 3; /* Path_Generation.cpp:607: for (i = 1; i < NumOfNode; i++) */
    const int ii = v1 ? i + 1 : i;
    if (v1 ? ii >= NumOfNode : ii > NumOfNode) { none(0.f); return; }
    double dy = ((maxy - miny) / NumOfNode * ii) + miny;
    if (dy > maxy) dy = maxy; /* B.13: the reference aborts in GSL when the last sample lands an ulp past the last knot */
    if (!(dy >= miny && dy <= maxy)) { none(2.f); return; } /* gsl_spline_eval: GSL_EDOM (raised by the chain when it gets here) */
    double node[3];
    spline_point_f(node_y + st, node_x + st, node_z + st, mm, dy, node);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0, ytab};
    float ab[3];
    const int kk = wave_area2cloud(V, G, s_w[wv], normals4, s_ell, D, node, key == 0 ? 1 : 0, ab, sc);
    const float qx = (float)node[0], qy = (float)node[1], qz = (float)node[2];
    const bool finite = fabsf(qx) <= 3.402823466e+38f && fabsf(qy) <= 3.402823466e+38f && fabsf(qz) <= 3.402823466e+38f;
    /* the snap asks for the nearest point of the very query Area2Cloud has just ranked 50 neighbours of: rank 0, no second search */
    float no_nn[3];
    const int got = finite ? (kk > 0 ? 1 : wave_knn(V, G, s_w[wv], qx, qy, qz, 1, D.r1, nullptr, no_nn, sc)) : 0;
    if (lane == 0) {
        Bf.first_ab[at] = make_float4(ab[0], ab[1], ab[2], 1.f);
        Bf.first_node[3 * at] = node[0]; Bf.first_node[3 * at + 1] = node[1]; Bf.first_node[3 * at + 2] = node[2];
        if (!finite) Bf.first_snap[at] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (got < 1) Bf.first_snap[at] = make_float4(0.f, 0.f, 0.f, 2.f);
        else { const float4 p = V.at(s_w[wv].sel[0]); Bf.first_snap[at] = make_float4(p.y, p.x, p.z, 1.f); }
    }
}

/* Step t, first launch.  Every workgroup begins with the second half of dynamic_adjust_path for the slice of step t-1
   (path_dynamic_alg.cpp:297-305: map by y, Spline::restart) -- the same few hundred samples sorted by every workgroup for
   itself, which costs less than a launch of its own between two dependent kernels; one extra workgroup per chain, which
   has no samples, commits the knots to the node arrays -- and goes on with compute_boundary's first half (:191-203) on those knots: one wave per
   sample of the previous path. */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_dyn_boundary_pts(DevMeta *m, DynParams D, int walk, int t, int centre,
        const float4 *__restrict__ sorted4, const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
        const float *__restrict__ slab_xmax, const float4 *__restrict__ normals4, const float *__restrict__ ell_cs, const int *__restrict__ ytab,
        float *node_x, float *node_y, float *node_z, int node_cap, int *node_start, int *node_cnt, DynBuffers Bf)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_base;
    __shared__ float2 s_ell[DYN_ELL];
    StampCtx sc; sc.begin(3, blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chain = blockIdx.y;
    const int capA = dyn_fit_cap(Bf.maxNA);
    float *ly = (float *)(s_raw + dyn_scratch_bytes(Bf.maxNA)), *lx = ly + capA, *lz = lx + capA;
    const DynFitLds F = dyn_fit_lds(s_raw, capA);
    const float4 *pts = Bf.adj_pts + (size_t)chain * Bf.maxNA; /* (y, x, z, valid) */
    /* first reads, all at once: the samples of the step before, the ellipse table, the slab grid, the state of the pass */
    const int nv = t > 0 ? dyn_stage_samples(pts, Bf.maxNA, F, [](const float4 &p) { return p.w != 0.f; }, [](const float4 &p) { return p.x; }) : 0;
    const int bn = (t > 0 && walk == 3) ? Bf.bnd_n[chain] : 3;
    dyn_stage_ellipse(ell_cs, s_ell);
    const DynGrid G = dyn_grid(m);
    const int err = m->err;
    __syncthreads();
    if (err) return;
    const int S = m->S;
    const float *ky = nullptr, *kx = nullptr, *kz = nullptr; /* knots of the previous path */
    int mm = 0;
    if (t > 0) {
        const DynChain cp = dyn_chain(walk, chain, t - 1, centre, S);
        /* v1, "generate boundary fail": the path stays as fitted */
        if (cp.active && !(walk == 3 && bn < 3)) {
            const DynFit ft = dyn_fit_prepare(nv, Bf.maxNA, F, capA, s_scr, m);
            const int tot = ft.tot;
            if (tot < 3) { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) set_err(m, DERR_SLICE, cp.s); return; } /* gsl_spline_alloc */
            dyn_emit_knots(pts, F, ft, [&](int o, const float4 &p) { ly[o] = p.x == 0.f ? 0.f : p.x; lx[o] = p.y; lz[o] = p.z; });
            __syncthreads();
            if (blockIdx.x == gridDim.x - 1) { /* the launch's extra workgroup: it commits the knots and has no samples */
                if (threadIdx.x == 0) {
                    int base = atomicAdd(&m->node_cursor, tot);
                    if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, cp.s); base = -1; }
                    s_base = base;
                }
                __syncthreads();
                const int base = s_base;
                if (base < 0) return;
                for (int q = threadIdx.x; q < tot; q += blockDim.x) { node_y[base + q] = ly[q]; node_x[base + q] = lx[q]; node_z[base + q] = lz[q]; }
                if (threadIdx.x == 0) { node_start[cp.s] = base; node_cnt[cp.s] = tot; }
                return;
            }
            ky = ly; kx = lx; kz = lz; mm = tot;
        }
    }
    const DynChain c = dyn_chain(walk, chain, t, centre, S);
    const int j = blockIdx.x * DYN_WAVES + wv;
    if (!c.active || j >= Bf.maxNB || blockIdx.x == gridDim.x - 1) return;
    if (!ky) { const int st = node_start[c.prev]; mm = node_cnt[c.prev]; ky = node_y + st; kx = node_x + st; kz = node_z + st; }
    float4 *dst = Bf.bnd_pts + (size_t)chain * Bf.maxNB + j;
    if (mm < 3) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    const double miny = (double)ky[0], maxy = (double)ky[mm - 1];
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
    spline_point_f(ky, kx, kz, mm, dy, point);
    sc.mark(1);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0, ytab};
    float b[3];
    wave_area2cloud(V, G, ((DynWaveLds *)s_raw)[wv], normals4, s_ell, D, point, c.key, b, sc);
    if (lane == 0) *dst = make_float4(b[0], b[1], b[2], 1.f);
    sc.mark(6);
}

/* Step t, second launch.  Every workgroup begins with compute_boundary's second half (path_dynamic_alg.cpp:211-234:
   std::map by y, last writer wins, two extra end knots) on the samples the first launch left -- again sorted by each
   workgroup for itself, the knots staying in its LDS; workgroup 0 of the chain keeps the copy in HBM that a later step falls
   back to when its own boundary has fewer than 3 knots -- and goes on with dynamic_adjust_path's first half (:278-295),
   one wave per node, from the node's first comparison on (its first Area2Cloud is k_dyn_first_eval's). */
__global__ void __launch_bounds__(64 * DYN_WAVES) k_dyn_adjust_pts(DevMeta *m, DynParams D, int walk, int t, int centre,
        const float4 *__restrict__ sorted4, const int *__restrict__ slab_start, const float *__restrict__ slab_xmin,
        const float *__restrict__ slab_xmax, const float4 *__restrict__ normals4, const float *__restrict__ ell_cs, const int *__restrict__ ytab,
        int S_cap, DynBuffers Bf)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ float2 s_ell[DYN_ELL];
    StampCtx sc; sc.begin(4, blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chain = blockIdx.y;
    const int capB = dyn_fit_cap(Bf.maxNB);
    const DynFitLds F = dyn_fit_lds(s_raw, capB);
    const float4 *pts = Bf.bnd_pts + (size_t)chain * Bf.maxNB;
    /* first reads, all at once: the boundary samples of this step (isnan(boundpoint[0]) -> continue; a NaN key is skipped
       too), this wave's node as k_dyn_first_eval left it, the state of the pass */
    const int nv = dyn_stage_samples(pts, Bf.maxNB, F, [](const float4 &p) { return p.w != 0.f && p.x == p.x && p.y == p.y; },
                                     [](const float4 &p) { return p.y; });
    const int i = blockIdx.x * DYN_WAVES + wv;
    const DynChain cg = dyn_chain(walk, chain, t, centre, S_cap); /* the step's slice, whether or not it exists */
    const size_t at = (cg.s >= 0 && cg.s < S_cap && i < Bf.maxNA) ? (size_t)cg.s * Bf.maxNA + i : 0;
    const float4 first = Bf.first_ab[at];
    double node[3] = {Bf.first_node[3 * at], Bf.first_node[3 * at + 1], Bf.first_node[3 * at + 2]};
    dyn_stage_ellipse(ell_cs, s_ell); /* (the barriers of the fit come before any use) */
    const DynGrid G = dyn_grid(m);
    if (m->err) return;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active) return;
    double *ly = (double *)(s_raw + dyn_scratch_bytes(Bf.maxNB)), *lx = ly + capB + 2;
    const size_t slot_doubles = 3 * ((size_t)Bf.maxNB + 2);
    double *gy = Bf.bnd_knots + (size_t)(Bf.keep_all ? c.s : chain) * slot_doubles, *gx = gy + (Bf.maxNB + 2), *gz = gx + (Bf.maxNB + 2);
    const double *ky = gy, *kx = gx;
    int nb;
    {
        sc.mark(13);
        const DynFit ft = dyn_fit_prepare(nv, Bf.maxNB, F, capB, s_scr, m);
        const int node_number = ft.tot;
        sc.mark(14);
        if (node_number <= 2) { /* compute_boundary returns 0: the previous boundary stays; v1 has none then (Path_Generation.cpp:589-592) */
            if (walk == 3) { nb = 0; if (blockIdx.x == 0 && threadIdx.x == 0) Bf.bnd_n[chain] = 0; }
            else { nb = Bf.bnd_n[chain]; ky = Bf.bnd_knots + (size_t)Bf.bnd_slot[chain] * slot_doubles; kx = ky + (Bf.maxNB + 2); }
            __syncthreads(); /* the scratch becomes the waves' search areas */
        } else {
            const bool keep_copy = blockIdx.x == 0;
            dyn_emit_knots(pts, F, ft, [&](int o, const float4 &p) {
                const double y = (double)(p.y == 0.f ? 0.f : p.y);
                ly[1 + o] = y; lx[1 + o] = (double)p.x;
                if (keep_copy) { gy[1 + o] = y; gx[1 + o] = (double)p.x; gz[1 + o] = (double)p.z; }
            });
            __syncthreads();
            if (threadIdx.x == 0) { /* get longer boundary, add new node at start & end (:223-229) */
                const int idx = node_number;
                lx[0] = lx[1]; ly[0] = ly[1] - 20;
                lx[idx + 1] = lx[idx]; ly[idx + 1] = ly[idx] + 20;
                if (keep_copy) {
                    gx[0] = lx[0]; gy[0] = ly[0]; gz[0] = gz[1];
                    gx[idx + 1] = lx[idx + 1]; gy[idx + 1] = ly[idx + 1]; gz[idx + 1] = gz[idx];
                    Bf.bnd_n[chain] = node_number + 2;
                    Bf.bnd_slot[chain] = Bf.keep_all ? c.s : chain;
                    if (Bf.keep_all) Bf.bnd_all_n[c.s] = node_number + 2;
                }
            }
            __syncthreads();
            nb = node_number + 2; ky = ly; kx = lx;
        }
    }
    if (i >= Bf.maxNA) return;
    float4 *dst = Bf.adj_pts + (size_t)chain * Bf.maxNA + i;
    if (first.w == 0.f) { if (lane == 0) *dst = make_float4(0, 0, 0, 0); return; }
    const bool v1 = walk == 3;
    if (nb < 3) { /* no boundary: v1 leaves the path alone; v2 would read an unconstructed Spline */
        if (lane == 0) { if (v1) *dst = make_float4(0, 0, 0, 0); else set_err(m, DERR_SLICE, c.s); }
        return;
    }
    if (first.w == 2.f) { if (lane == 0) set_err(m, DERR_DOMAIN, c.s); return; } /* gsl_spline_eval: GSL_EDOM */
    sc.mark(0);
    SlabView V{sorted4, slab_start, slab_xmin, slab_xmax, m, nullptr, 0, 0, ytab};
    DynWaveLds &L = ((DynWaveLds *)s_raw)[wv];
    const double bminy = ky[0], bbigy = ky[nb - 1];
    auto BY = [&](int q) { return ky[q]; };
    auto BX = [&](int q) { return kx[q]; };
    /* bisection (:237-265), at most 6 Area2Cloud evaluations, the first of them taken ahead of the chain */
    float ab[3] = {first.x, first.y, first.z};
    bool moved = false;
    int kk_here = 0; /* neighbours Area2Cloud found at the node as it stands (0: not evaluated there, or none) */
    for (int itr = 0; itr <= 5; ++itr) {
        if (itr > 0) kk_here = wave_area2cloud(V, G, L, normals4, s_ell, D, node, c.key == 0 ? 1 : 0, ab, sc);
        if ((double)ab[1] < bminy || (double)ab[1] > bbigy) break; /* a NaN bound passes, as in the reference: the node turns NaN below */
        const int iv = wave_gsl_bsearch_d(ky, nb, (double)ab[1]);
        const double bpx = steffen_eval_at(iv, nb, (double)ab[1], BY, BX);
        const double norm0 = (double)ab[0] - bpx;
        if (fabs(norm0) < D.adjust_threshold) break;
        node[0] = node[0] - norm0;
        moved = true;
        kk_here = 0;
        if (!(node[0] == node[0])) break;
        sc.mark(6);
    }
    sc.mark(7);
    /* kdtree.nearestKSearch(point, 3): only pointIdx[0] is used (:291-294).  B.14: a node that Area2Cloud turned NaN
       ("adjust path node NAN", :258-261) adds no knot (the reference hands the NaN to FLANN and reads whatever
       pointIdx holds afterwards) */
    if (!moved) { /* the node as sampled: its snap is k_dyn_first_eval's */
        if (lane == 0) {
            const float4 sn = Bf.first_snap[at];
            if (sn.w == 1.f) *dst = sn;
            else { if (sn.w == 2.f) set_err(m, DERR_QUERY, c.s); *dst = make_float4(0, 0, 0, 0); }
        }
        return;
    }
    const float qx = (float)node[0], qy = (float)node[1], qz = (float)node[2];
    const bool finite = fabsf(qx) <= 3.402823466e+38f && fabsf(qy) <= 3.402823466e+38f && fabsf(qz) <= 3.402823466e+38f;
    /* (a bisection that ended on an evaluation at the node's final place has its nearest point already: rank 0 of that search) */
    float no_nn[3];
    const int got = finite ? (kk_here > 0 ? 1 : wave_knn(V, G, L, qx, qy, qz, 1, D.r1, nullptr, no_nn, sc)) : 0;
    if (lane == 0) {
        if (!finite) *dst = make_float4(0, 0, 0, 0);
        else if (got < 1) { set_err(m, DERR_QUERY, c.s); *dst = make_float4(0, 0, 0, 0); }
        else { const float4 p = V.at(L.sel[0]); *dst = make_float4(p.y, p.x, p.z, 1.f); }
    }
    sc.mark(8);
}

/* dynamic_adjust_path, second half (:297-305) as a launch of its own: after the last step of the chains (every earlier step
   is finished by the next step's first launch) */
__global__ void __launch_bounds__(256) k_dyn_adjust_fit(DevMeta *m, int walk, int t, int centre, DynBuffers Bf, float *node_x,
                                                        float *node_y, float *node_z, int node_cap, int *node_start, int *node_cnt)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    __shared__ int s_scr[17];
    __shared__ int s_base;
    if (m->err) return;
    const int chain = blockIdx.x;
    const DynChain c = dyn_chain(walk, chain, t, centre, m->S);
    if (!c.active) return;
    if (walk == 3 && Bf.bnd_n[chain] < 3) return; /* "generate boundary fail": the path stays as fitted */
    const int capA = dyn_fit_cap(Bf.maxNA);
    const DynFitLds F = dyn_fit_lds(s_raw, capA);
    const float4 *pts = Bf.adj_pts + (size_t)chain * Bf.maxNA;
    const int nv = dyn_stage_samples(pts, Bf.maxNA, F, [](const float4 &p) { return p.w != 0.f; }, [](const float4 &p) { return p.x; });
    const DynFit ft = dyn_fit_prepare(nv, Bf.maxNA, F, capA, s_scr, m);
    const int tot = ft.tot;
    if (threadIdx.x == 0) {
        int base = -1;
        if (tot < 3) set_err(m, DERR_SLICE, c.s); /* gsl_spline_alloc */
        else {
            base = atomicAdd(&m->node_cursor, tot);
            if (base + tot > node_cap) { set_err(m, DERR_CAPACITY, c.s); base = -1; }
        }
        s_base = base;
    }
    __syncthreads();
    const int base = s_base;
    if (base < 0) return;
    dyn_emit_knots(pts, F, ft, [&](int o, const float4 &p) { node_y[base + o] = p.x == 0.f ? 0.f : p.x; node_x[base + o] = p.y; node_z[base + o] = p.z; });
    if (threadIdx.x == 0) { node_start[c.s] = base; node_cnt[c.s] = tot; }
}
