/*
 * ppp_device.h -- device-side helpers shared by the gfx950 kernels.
 *
 * All float/double expressions that feed a DISCRETE decision of the reference
 * (band membership, side split, nearest neighbour, node keys, waypoint counts)
 * are evaluated with one IEEE rounding per operation, in the reference's
 * operation order: the translation unit is compiled with -ffp-contract=off and
 * without fast-math, so what is written here is what executes.
 */
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;

#define PPP_WAVE 64

/* order-preserving float <-> uint map (for atomics and radix-style keys) */
__host__ __device__ inline u32 f2ord(float f)
{
    u32 b = __builtin_bit_cast(u32, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float ord2f(u32 o)
{
    u32 b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}

/* flann::L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz  (pcl::KdTreeFLANN metric) */
__device__ inline float dist2_flann(float ax, float ay, float az, float bx, float by, float bz)
{
    float r = 0.f, d;
    d = ax - bx; r += d * d;
    d = ay - by; r += d * d;
    d = az - bz; r += d * d;
    return r;
}
/* Eigen Vector3f::norm(): sqrt(x*x + (y*y + z*z))  (Path_Generation.cpp:145-149) */
__device__ inline float norm_eigen3(float vx, float vy, float vz)
{
    return sqrtf(vx * vx + (vy * vy + vz * vz));
}

/* ---- slice walk: shared by the host planner and the device ---- */
#define PPP_WALK_HARD_MAX (1 << 22)
__host__ __device__ inline int ppp_slice_walk(int walk, float min_x, float max_x, double toolRadius, float *px, int cap)
{
    int step_size = (int)(toolRadius * 2);
    if (step_size <= 0 || !(min_x <= max_x)) return 0;
    int k = 0;
    switch (walk) {
    case 0: { /* SectPath::GenPath, path_slicing_alg.cpp:308-330 */
        int nfront = 0;
        float loc = (min_x + max_x) / 2 - step_size;
        while (loc > min_x && nfront < PPP_WALK_HARD_MAX) { nfront++; loc -= step_size; }
        loc = (min_x + max_x) / 2 - step_size;
        int i = nfront - 1;
        while (loc > min_x && i >= 0) { if (i < cap) px[i] = loc; i--; loc -= step_size; }
        k = nfront;
        loc = (min_x + max_x) / 2;
        while (loc < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = loc; k++; loc += step_size; }
        return k;
    }
    case 1: { /* path_generater::GenPath + thread_worker, path_dynamic_alg.cpp:308-372 */
        int imin = (int)min_x, imax = (int)max_x; /* thread_wrap_data{int min_pt, max_pt} */
        int c = (imax + imin) / 2;
        int nfront = 0;
        int loc = c - step_size;
        while (imax > loc && loc > imin && nfront < PPP_WALK_HARD_MAX) { nfront++; loc -= step_size; }
        loc = c - step_size;
        int i = nfront - 1;
        while (imax > loc && loc > imin && i >= 0) { if (i < cap) px[i] = (float)loc; i--; loc -= step_size; }
        k = nfront;
        if (k < cap) px[k] = (min_x + max_x) / 2; /* Center_path, :353 */
        k++;
        loc = c + step_size;
        while (imax > loc && loc > imin && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = (float)loc; k++; loc += step_size; }
        return k;
    }
    case 2: { /* dynamic_alg_sdir.cpp:349-374 */
        int loc = (int)(min_x + toolRadius);
        if (k < cap) px[k] = (float)loc;
        k++;
        loc += step_size;
        while (loc < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = (float)loc; k++; loc += step_size; }
        return k;
    }
    case 3: { /* Contact_Path_Generation, Path_Generation.cpp:711-725 */
        float locateX = (float)(min_x + toolRadius);
        while (locateX < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = locateX; k++; locateX += step_size; }
        return k;
    }
    case 4: { /* slicing_method, Path_Generation.cpp:295-304 */
        float x = min_x;
        x += step_size / 2;
        while (x < max_x && k < PPP_WALK_HARD_MAX) { if (k < cap) px[k] = x; k++; x += step_size; }
        return k;
    }
    }
    return 0;
}

/* Workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding
   global store (s_waitcnt vmcnt(0)), which costs a memory round trip per sweep in kernels that
   stream results out while they keep iterating in LDS. */
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* In-kernel phase stamps, DIAGNOSTIC BUILD ONLY (make stamps -> libppp_hip_stamps.so): thread 0
   of the middle workgroup accumulates s_memtime deltas per (kernel, slot).  In the product build
   the macros expand to nothing, so no stamp executes and no output depends on one. */
#ifdef PPP_STAMPS
__device__ unsigned long long g_stamps[16][16];
__device__ inline unsigned long long ppp_stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP_BEGIN() const bool _st_on = (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0); unsigned long long _st_t = ppp_stamp()
#define STAMP(kid, slot) do { unsigned long long _n = ppp_stamp(); if (_st_on) g_stamps[kid][slot] += _n - _st_t; _st_t = _n; } while (0)
/* the same for helpers called from a kernel: the kernel owns the context and passes it down */
struct StampCtx {
    bool on; unsigned long long t; int kid;
    __device__ inline void begin(int k, bool enable) { kid = k; on = enable; t = ppp_stamp(); }
    __device__ inline void mark(int slot) { unsigned long long n = ppp_stamp(); if (on) g_stamps[kid][slot] += n - t; t = n; }
};
#else
#define STAMP_BEGIN() do { } while (0)
#define STAMP(kid, slot) do { } while (0)
struct StampCtx {
    __device__ inline void begin(int, bool) {}
    __device__ inline void mark(int) {}
};
#endif

/* a, a + step, a + 2 step, ... accumulated in double (the reference's `dy += PathResolution` loops): when a and step
   are multiples of 2^-39 and the sums stay below 2^13 every partial sum has at most 52 significant bits, so no
   addition rounds and a + k * step computed directly has the same bits as k accumulated additions. */
__host__ __device__ inline bool sums_exact(double a, double step, double kmax)
{
    const double S = 549755813888.0; /* 2^39 */
    if (!(fabs(a) < 8192.0) || !(fabs(step) < 8192.0)) return false;
    const double as = a * S, ss = step * S;
    return floor(as) == as && floor(ss) == ss && fabs(a) + kmax * fabs(step) < 8192.0;
}

/* ---- block-wide helpers (blockDim.x multiple of 64, <= 1024) ---- */
template <typename T>
__device__ inline T wave_sum(T v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

/* exclusive scan of one int per thread across the block; returns this thread's prefix,
   *total receives the block sum.  scratch: >= 17 ints of LDS. */
__device__ inline int block_exscan(int v, int *scratch, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { int t = scratch[w]; scratch[w] = run; run += t; }
        scratch[16] = run;
    }
    __syncthreads();
    int pre = scratch[wid] + inc - v;
    *total = scratch[16];
    return pre;
}

/* The same scan with two barriers and no serial part: every wave adds up the waves' totals for itself (a shuffle scan over
   at most 16 values) instead of waiting for thread 0 to walk them.  scratch: >= 16 ints of LDS. */
__device__ inline int block_exscan_w(int v, int *scratch, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    const int part = lane < nw ? scratch[lane] : 0;
    int incw = part;
    for (int o = 1; o < 16; o <<= 1) {
        int t = __shfl_up(incw, o, 64);
        if (lane >= o) incw += t;
    }
    const int wave_off = __shfl(incw - part, wid, 64);
    *total = __shfl(incw, nw - 1, 64);
    return wave_off + inc - v;
}

/* ---------------------------------------------------------------------- */
/* Block sort of 256*E 64-bit keys held in registers (blocked layout: thread */
/* t owns elements t*E .. t*E+E-1), ascending, for a 256-thread workgroup.   */
/* Bitonic network; a compare-exchange runs in registers when the partner    */
/* is in the same thread, through wave shuffles when it is in the same wave, */
/* and through LDS only for the three cross-wave strides -- so the whole     */
/* sort has 3 barriers-pairs instead of one per step.                        */
/* ---------------------------------------------------------------------- */
template <int E, int S>
__device__ inline void sort_reg_stride(u64 (&k)[E], int size, int base_i)
{
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if ((r & S) == 0) {
            const bool up = ((base_i + r) & size) == 0;
            u64 a = k[r], b = k[r | S];
            bool sw = (a > b) == up;
            k[r] = sw ? b : a;
            k[r | S] = sw ? a : b;
        }
    }
}
/* all in-register strides below FROM (a power of two <= E/2), down to 1 */
template <int E, int FROM>
__device__ inline void sort_reg_tail(u64 (&k)[E], int size, int base_i)
{
    if constexpr (FROM >= 1) {
        sort_reg_stride<E, FROM>(k, size, base_i);
        sort_reg_tail<E, FROM / 2>(k, size, base_i);
    }
}
template <int E, int SIZE>
__device__ inline void sort_reg_phase(u64 (&k)[E], int base_i)
{   /* merge sizes 2 .. E that live entirely in one thread */
    if constexpr (SIZE <= E) {
        sort_reg_tail<E, SIZE / 2>(k, SIZE, base_i);
        sort_reg_phase<E, SIZE * 2>(k, base_i);
    }
}
template <int E>
__device__ inline void block_sort_regs(u64 (&k)[E], u64 *lds /* 256*E keys of scratch */)
{
    const int t = threadIdx.x;
    const int base_i = t * E;
    constexpr int P = 256 * E;
    const bool live = t < 256; /* a wider workgroup (k_slice's 512 threads): its other waves hold padding, keep the barriers and stay out of the scratch */
    sort_reg_phase<E, 2>(k, base_i);
    for (int size = 2 * E; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride >= E; stride >>= 1) {
            if (stride >= 64 * E) {
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int r = 0; r < E; ++r) lds[base_i + r] = k[r];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int i = base_i + r;
                    u64 o = live ? lds[i ^ stride] : ~0ull;
                    const bool keep_min = (((i & stride) == 0) == ((i & size) == 0));
                    u64 a = k[r];
                    k[r] = keep_min ? (a < o ? a : o) : (a > o ? a : o);
                }
            } else {
                const int lx = stride / E;
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int i = base_i + r;
                    u64 a = k[r];
                    u64 o = __shfl_xor(a, lx, 64);
                    const bool keep_min = (((i & stride) == 0) == ((i & size) == 0));
                    k[r] = keep_min ? (a < o ? a : o) : (a > o ? a : o);
                }
            }
        }
        sort_reg_tail<E, E / 2>(k, size, base_i);
    }
}

/* Sorts keys[0..n) (LDS, n <= 4096) ascending with the first 256 threads of a workgroup (wider ones keep the barriers); on return the
   keys are back in LDS.  Picks the smallest register tile that covers n. */
template <int E>
__device__ inline void block_sort_lds_e(u64 *keys, int n)
{
    u64 k[E];
    const int base_i = threadIdx.x * E;
#pragma unroll
    for (int r = 0; r < E; ++r) k[r] = (base_i + r) < n ? keys[base_i + r] : ~0ull;
    block_sort_regs<E>(k, keys);
    __syncthreads();
    if (threadIdx.x < 256) {
#pragma unroll
        for (int r = 0; r < E; ++r) keys[base_i + r] = k[r]; /* slots >= n hold the ~0 padding */
    }
    __syncthreads();
}
/* keys must have room for 256 * E entries, E = max(1, next_pow2(n) / 256) */
__device__ inline void block_sort_lds(u64 *keys, int n)
{
    if (n <= 256) block_sort_lds_e<1>(keys, n);
    else if (n <= 512) block_sort_lds_e<2>(keys, n);
    else if (n <= 1024) block_sort_lds_e<4>(keys, n);
    else if (n <= 2048) block_sort_lds_e<8>(keys, n);
    else block_sort_lds_e<16>(keys, n);
}

/* ---------------------------------------------------------------------- */
/* Exact LDS bucket sort for one workgroup (<= 1024 threads): n <= 4096 keys. */
/*   gen(i)      -> the i-th key (recomputed in both passes, so no copy)     */
/*   bucket(key) -> [0, NB), monotone non-decreasing in the sort order       */
/*   less(a, b)  -> strict weak order (refines the bucket order)             */
/* Histogram (LDS atomics) + scan + scatter, then every bucket (mean size    */
/* <= 1) is finished by an insertion sort.  ~15x fewer instructions than a   */
/* 64-bit bitonic network at these sizes (measured: 2100 VALU/wave).         */
/* hist: NB + 1 ints of LDS; out: n keys of LDS.                             */
/* ---------------------------------------------------------------------- */
template <typename Gen, typename Bucket, typename Less>
__device__ inline void block_bucket_sort(u64 *out, int n, int *hist, int NB, int *scratch17, Gen gen, Bucket bucket, Less less)
{
    for (int b = threadIdx.x; b <= NB; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[bucket(gen(i))], 1);
    __syncthreads();
    /* exclusive scan of hist[0..NB): every thread owns NB/blockDim consecutive buckets */
    {
        const int per = (NB + blockDim.x - 1) / blockDim.x;
        const int b0 = threadIdx.x * per;
        int sum = 0;
        for (int k = 0; k < per; ++k) if (b0 + k < NB) sum += hist[b0 + k];
        int total;
        int pre = block_exscan(sum, scratch17, &total);
        for (int k = 0; k < per; ++k) {
            if (b0 + k < NB) { int c = hist[b0 + k]; hist[b0 + k] = pre; pre += c; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        u64 k = gen(i);
        out[atomicAdd(&hist[bucket(k)], 1)] = k;
    }
    __syncthreads();
    /* hist[b] is now the END of bucket b */
    for (int b = threadIdx.x; b < NB; b += blockDim.x) {
        const int s = b ? hist[b - 1] : 0, e = hist[b];
        for (int p = s + 1; p < e; ++p) {
            u64 kp = out[p];
            int q = p - 1;
            while (q >= s && less(kp, out[q])) { out[q + 1] = out[q]; --q; }
            out[q + 1] = kp;
        }
    }
    __syncthreads();
}

/* The same sort with every key made ONCE and kept in registers between the counting and the placing pass (gen may be a
   global load): for n <= E * blockDim. */
template <int E, typename Gen, typename Bucket, typename Less>
__device__ inline void block_bucket_sort_cached(u64 *out, int n, int *hist, int NB, int *scratch17, Gen gen, Bucket bucket, Less less)
{
    u64 kr[E];
    int br[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)threadIdx.x + e * (int)blockDim.x;
        br[e] = -1;
        kr[e] = 0;
        if (i < n) { kr[e] = gen(i); br[e] = bucket(kr[e]); }
    }
    for (int b = threadIdx.x; b <= NB; b += blockDim.x) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) if (br[e] >= 0) atomicAdd(&hist[br[e]], 1);
    __syncthreads();
    {
        const int per = (NB + blockDim.x - 1) / blockDim.x;
        const int b0 = threadIdx.x * per;
        int sum = 0;
        for (int k = 0; k < per; ++k) if (b0 + k < NB) sum += hist[b0 + k];
        int total;
        int pre = block_exscan(sum, scratch17, &total);
        for (int k = 0; k < per; ++k) {
            if (b0 + k < NB) { int c = hist[b0 + k]; hist[b0 + k] = pre; pre += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) if (br[e] >= 0) out[atomicAdd(&hist[br[e]], 1)] = kr[e];
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += blockDim.x) { /* hist[b] is now the END of bucket b */
        const int s = b ? hist[b - 1] : 0, e = hist[b];
        for (int p = s + 1; p < e; ++p) {
            u64 kp = out[p];
            int q = p - 1;
            while (q >= s && less(kp, out[q])) { out[q + 1] = out[q]; --q; }
            out[q + 1] = kp;
        }
    }
    __syncthreads();
}

__host__ __device__ inline int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

/* ---------------------------------------------------------------------- */
/* float restatements of the third-party numerics (same text as the        */
/* published algorithms; see DESIGN.md "Third-party semantics")            */
/* ---------------------------------------------------------------------- */

/* pcl::computeRoots2 / computeRoots / eigen33 (smallest eigenvalue) */
__device__ inline void pcl_roots2(float b, float c, float roots[3])
{
    roots[0] = 0.f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
/* EXACT_TRIG: atan2 / cos / sin through double, rounded (the correctly rounded float, as the oracle computes it) -- for
   the normal FIELD and the principal curvatures, whose last bits feed the dynamic adjustment's discontinuous
   decisions.  Otherwise the device's float functions: k_pose's per-waypoint normals feed continuous outputs only, and
   the double versions cost its hot loop 10-20 %. */
template <bool EXACT_TRIG = true>
__device__ inline void pcl_roots(const float m[3][3], float roots[3])
{
    float c0 = m[0][0] * m[1][1] * m[2][2] + 2.f * m[0][1] * m[0][2] * m[1][2] -
               m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] +
               m[1][1] * m[2][2] - m[1][2] * m[1][2];
    float c2 = m[0][0] + m[1][1] + m[2][2];
    if (fabsf(c0) < 1.1920929e-07f) { pcl_roots2(c2, c1, roots); return; }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    float rho = sqrtf(-a_over_3);
    /* correctly rounded float results through double (see the oracle's computeRoots): device and host libm float
       functions differ by an ulp now and then, their double functions rounded to float do not */
    float theta, cos_theta, sin_theta;
    if (EXACT_TRIG) {
        theta = (float)atan2((double)sqrtf(-q), (double)half_b) * s_inv3;
        cos_theta = (float)cos((double)theta);
        sin_theta = (float)sin((double)theta);
    } else {
        theta = atan2f(sqrtf(-q), half_b) * s_inv3;
        cos_theta = cosf(theta);
        sin_theta = sinf(theta);
    }
    roots[0] = c2_over_3 + 2.f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
        t = roots[1]; roots[1] = roots[2]; roots[2] = t;
        if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0) pcl_roots2(c2, c1, roots);
}
__device__ inline void cross3f(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <bool EXACT_TRIG = true>
__device__ inline void pcl_eigen33_smallest(const float cov[9], float *eigenvalue, float ev[3])
{
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(cov[i]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = cov[3 * i + j] / scale;
    float roots[3];
    pcl_roots<EXACT_TRIG>(m, roots);
    *eigenvalue = roots[0] * scale;
    m[0][0] -= roots[0]; m[1][1] -= roots[0]; m[2][2] -= roots[0];
    float cp[3][3];
    cross3f(m[0], m[1], cp[0]);
    cross3f(m[0], m[2], cp[1]);
    cross3f(m[1], m[2], cp[2]);
    float len[3];
    for (int i = 0; i < 3; ++i) len[i] = sqrtf(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
    int idx = 0;
    if (len[1] > len[idx]) idx = 1;
    if (len[2] > len[idx]) idx = 2;
    for (int d = 0; d < 3; ++d) ev[d] = cp[idx][d] / len[idx];
}

/* Eigen: Quaternionf from AngleAxisf about a unit axis, product, toRotationMatrix.
   The translation unit is compiled -ffp-contract=off because the reference's DECISIONS (band membership, nearest neighbours, map
   keys, waypoint counts) must see one rounding per written operation.  The frame / Euler / hand-eye arithmetic below and the
   smoothing filter feed continuous outputs only (compared at 1e-6 m / 1e-4 rad): they re-enable contraction locally -- contract(on): only a product and a sum written in ONE expression fuse, a decision of the front end, so the batched and the single launch forms of a kernel body carry the same bits (contract(fast) lets the optimiser fuse across statements, and two instantiations of one body came out one bit apart). */
struct Quatf { float w, x, y, z; };
__device__ __forceinline__ Quatf quat_axis(float angle, int axis)
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    float ha = 0.5f * angle;
    float s, c;
    sincosf(ha, &s, &c); /* (one argument reduction for both) */
    Quatf q;
    q.w = c; q.x = 0.f; q.y = 0.f; q.z = 0.f;
    if (axis == 0) q.x = s; else if (axis == 1) q.y = s; else q.z = s;
    return q;
}
__device__ __forceinline__ Quatf quat_mul(const Quatf &a, const Quatf &b)
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    Quatf r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
__device__ __forceinline__ void quat_to_mat(const Quatf &q, float R[3][3])
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1.f - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1.f - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1.f - (txx + tyy);
}
/* AngleAxisf(rz,Z) * AngleAxisf(ry,Y) * AngleAxisf(rx,X) -> Matrix3f */
__device__ __forceinline__ void rot_zyx(float rx, float ry, float rz, float R[3][3])
{
    Quatf q = quat_mul(quat_mul(quat_axis(rz, 2), quat_axis(ry, 1)), quat_axis(rx, 0));
    quat_to_mat(q, R);
}
/* Matrix3f::eulerAngles(2,1,0) -> (yaw, pitch, roll) */
__device__ __forceinline__ void euler_zyx(const float m[3][3], float e[3])
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    const float kPi = 3.14159265358979323846f;
    e[0] = atan2f(m[1][0], m[0][0]);
    float c2 = sqrtf(m[2][2] * m[2][2] + m[2][1] * m[2][1]);
    if (e[0] < 0.f) {
        e[0] += kPi;
        e[1] = atan2f(-m[2][0], -c2);
    } else {
        e[1] = atan2f(-m[2][0], c2);
    }
    float s1, c1;
    sincosf(e[0], &s1, &c1);
    e[2] = atan2f(s1 * m[0][2] - c1 * m[1][2], c1 * m[1][1] - s1 * m[0][1]);
}
/* SectPath::HandEyeTransform, path_translation_alg.cpp:3-35 */
/* (the rotation of the calibration itself is the same for every waypoint: handeye_rotation once, handeye_apply per waypoint) */
__device__ __forceinline__ void handeye_rotation(const float he[6], float HE[3][3]) { rot_zyx(he[3], he[4], he[5], HE); }
__device__ __forceinline__ void handeye_apply(const float HE[3][3], const float he[6], float wp[6])
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    float P[3][3];
    rot_zyx(wp[3], wp[4], wp[5], P);
    float R[3][3], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            R[i][j] = HE[i][0] * P[0][j] + HE[i][1] * P[1][j] + HE[i][2] * P[2][j] + he[i] * 0.f;
        t[i] = HE[i][0] * wp[0] + HE[i][1] * wp[1] + HE[i][2] * wp[2] + he[i] * 1.f;
    }
    float e[3];
    euler_zyx(R, e);
    wp[0] = t[0]; wp[1] = t[1]; wp[2] = t[2];
    wp[3] = e[2]; wp[4] = e[1]; wp[5] = e[0];
}
__device__ __forceinline__ void handeye_transform(const float he[6], float wp[6])
{
    float HE[3][3];
    handeye_rotation(he, HE);
    handeye_apply(HE, he, wp);
}
/* Approach / Orientation / Normal frame, path_translation_alg.cpp:192-202 */
__device__ __forceinline__ void pose_from_normal(const float n[3], float rpy[3])
{
#pragma clang fp contract(on) /* continuous output only (no decision hangs on it): fused multiply-adds allowed */
    float A[3] = {-n[0], -n[1], -n[2]};
    const float X[3] = {1.f, 0.f, 0.f};
    float O[3], Nn[3];
    cross3f(A, X, O);
    cross3f(O, A, Nn);
    float M[3][3];
    for (int i = 0; i < 3; ++i) { M[i][0] = Nn[i]; M[i][1] = O[i]; M[i][2] = A[i]; }
    float e[3];
    euler_zyx(M, e);
    rpy[0] = e[2]; rpy[1] = e[1]; rpy[2] = e[0];
}

/* GSL steffen.c evaluated locally: knots ys[0..m) (float, widened), values vs[0..m).
   Same operation order as steffen_init/steffen_eval, so the double result is identical. */
__device__ inline double steffen_sgn(double x, double y)
{
    if ((x < 0 && y > 0) || (x > 0 && y < 0)) return -x;
    return x;
}
template <typename LoadY, typename LoadV>
__device__ inline double steffen_yprime(int i, int m, LoadY Y, LoadV V)
{
    if (i == 0) {
        double h0 = Y(1) - Y(0);
        return (V(1) - V(0)) / h0;
    }
    if (i == m - 1) return (V(m - 1) - V(m - 2)) / (Y(m - 1) - Y(m - 2));
    double hi = Y(i + 1) - Y(i);
    double him1 = Y(i) - Y(i - 1);
    double si = (V(i + 1) - V(i)) / hi;
    double sim1 = (V(i) - V(i - 1)) / him1;
    double pi = (sim1 * hi + si * him1) / (him1 + hi);
    return (steffen_sgn(1.0, sim1) + steffen_sgn(1.0, si)) * fmin(fabs(sim1), fmin(fabs(si), 0.5 * fabs(pi)));
}
template <typename LoadY, typename LoadV>
__device__ inline double steffen_eval_at(int i, int m, double xq, LoadY Y, LoadV V)
{
    double yp0 = steffen_yprime(i, m, Y, V);
    double yp1 = steffen_yprime(i + 1, m, Y, V);
    double hi = Y(i + 1) - Y(i);
    double si = (V(i + 1) - V(i)) / hi;
    double a = (yp0 + yp1 - 2 * si) / hi / hi;
    double b = (3 * si - 2 * yp0 - yp1) / hi;
    double c = yp0;
    double d = V(i);
    double delta = xq - Y(i);
    return d + delta * (c + delta * (b + delta * a));
}
/* gsl_interp_bsearch(x_array, x, 0, m-1) */
template <typename LoadY>
__device__ inline int gsl_bsearch(int m, double xq, LoadY Y)
{
    int ilo = 0, ihi = m - 1;
    while (ihi > ilo + 1) {
        int i = (ihi + ilo) / 2;
        if (Y(i) > xq) ihi = i; else ilo = i;
    }
    return ilo;
}
