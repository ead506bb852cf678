/*
 * ppp_oracle.cpp -- CPU oracle for the polishing-path hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ppp_oracle.h).  Single-threaded, dependency
 * free restatement of the reference semantics.  File:line citations are
 * relative to the reference tree (tsai0507/PolishPathPlanning @ v1).
 *
 * PARITY UNPINNED: no reference test / golden vector exists for this path and
 * the reference cannot be compiled here (needs PCL >= 1.11, GSL >= 2.0,
 * Eigen3 -- none installed, no network).  The third-party arithmetic is
 * restated from the published algorithms of the pinned-by-usage versions:
 *   - PCL 1.12 (code uses pcl::Indices + pcl::make_shared => >= 1.11;
 *     Ubuntu 22.04 ships 1.12.1): PassThrough, KdTreeFLANN (exact NN,
 *     flann::L2_Simple<float>), NormalEstimation (radius search,
 *     computeMeanAndCovarianceMatrix shifted by the first neighbour,
 *     pcl::eigen33, flipNormalTowardsViewpoint), getMinMax3D.
 *     Dynamic adjustment: PrincipalCurvaturesEstimation, nearestKSearch order.
 *     Preprocessing: StatisticalOutlierRemoval, VoxelGrid, MovingLeastSquares
 *     (MLSResult::computeMLSSurface + SIMPLE projection), compute3DCentroid /
 *     computeCovarianceMatrix (float running sums), transformPointCloud (SSE2 order).
 *   - GSL >= 2.0 interpolation/steffen.c (gsl_interp_steffen).
 *   - Eigen 3.3/3.4: Matrix3f::eulerAngles(2,1,0), AngleAxisf products
 *     (quaternion route), Matrix4f products; for trans2center EigenSolver<Matrix3f>
 *     (RealSchur, eigenvectors unsorted) and the generic 4 x 4 inverse; LLT, unitOrthogonal for MLS.
 * Build with -ffp-contract=off: every float expression below is evaluated
 * with one rounding per operation, in the order written.
 *
 * Known reference defect handled here (SURVEY.md App. B + DESIGN.md B.12):
 * postion_smooth() (path_translation_alg.cpp:114-141) stores float but sums
 * the un-rounded double deltas, so its `change < 1e-5` test can never be met
 * once 3*W*ulp/4 > 1e-5 (W >~ 250): the reference spins forever.  The oracle
 * (and the product) stop at the first sweep k >= 2 whose change is >= 0.9 x
 * the previous sweep's change (the geometric decay has hit the float floor),
 * or at `smooth_max_sweeps`, whichever comes first.
 */
#include "ppp_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <vector>
#include <math.h>

namespace {

/* pcl::PointXYZRGB is 32 bytes: xyz + pad, rgba + pad. */
struct Pt {
    float x, y, z, pad0;
    uint32_t rgba;
    float pad1[3];
};
static_assert(sizeof(Pt) == 32, "PointXYZRGB layout");

inline float dist2(const float *a, const float *b)
{
    /* flann::L2_Simple<float>: result += diff*diff over x, y, z in order */
    float r = 0.f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

/* ------------------------------------------------------------------ */
/* Exact kd-tree (stand-in for pcl::KdTreeFLANN).  */
/* Results: ascending (distance, index); ties resolve to lowest index. */
/* ------------------------------------------------------------------ */
class KdTree {
public:
    void build(const Pt *pts, const int *ids, int n)
    {
        pts_ = pts;
        /* local id = position in ids (like the per-slice sub-clouds cloudEl / cloudEr) */
        if (ids) sub_.assign(ids, ids + n);
        else sub_.clear();
        /* pcl::KdTreeFLANN::convertCloudToArray leaves non-finite points out of the index (a cloud read with NaNs is
           not dense); they would also break the ordering the median split relies on */
        order_.clear();
        order_.reserve(n);
        for (int i = 0; i < n; ++i) {
            const float *c = p(i);
            if (std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2])) order_.push_back(i);
        }
        n = (int)order_.size();
        nodes_.clear();
        if (n > 0) { nodes_.reserve(2 * n / kLeaf + 8); build_rec(0, n); }
    }
    bool empty() const { return order_.empty(); }
    /* returns local id (position in ids, or cloud index if ids == nullptr) */
    int nearest(const float *q, float *d2out = nullptr) const
    {
        float best = std::numeric_limits<float>::infinity();
        int bi = -1;
        if (!nodes_.empty()) nn_rec(0, q, best, bi);
        if (d2out) *d2out = best;
        return bi;
    }
    /* k nearest, ascending (distance, id) */
    void knn(const float *q, int k, std::vector<std::pair<float, int>> &out) const
    {
        out.clear();
        if (nodes_.empty() || k <= 0) return;
        knn_rec(0, q, k, out);
        std::sort_heap(out.begin(), out.end());
    }
    void radius(const float *q, float r, std::vector<std::pair<float, int>> &out) const
    {
        out.clear();
        if (nodes_.empty()) return;
        rad_rec(0, q, r * r, out);
        std::sort(out.begin(), out.end());
    }
private:
    static constexpr int kLeaf = 12;
    struct Node { int lo, hi, dim; float split; int left, right; };
    const Pt *pts_ = nullptr;
    std::vector<int> sub_, order_;
    std::vector<Node> nodes_;

    const float *p(int local) const { return &pts_[sub_.empty() ? local : sub_[local]].x; }

    int build_rec(int lo, int hi)
    {
        int me = (int)nodes_.size();
        nodes_.push_back({lo, hi, -1, 0.f, -1, -1});
        if (hi - lo <= kLeaf) return me;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; ++i) {
            const float *c = p(order_[i]);
            for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
        }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (!(mx[dim] > mn[dim])) return me; /* all coincident: keep as a (big) leaf */
        int mid = (lo + hi) / 2;
        std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi,
                         [&](int a, int b) {
                             float ca = p(a)[dim], cb = p(b)[dim];
                             return ca < cb || (ca == cb && a < b);
                         });
        float split = p(order_[mid])[dim];
        int l = build_rec(lo, mid);
        int r = build_rec(mid, hi);
        nodes_[me].dim = dim; nodes_[me].split = split; nodes_[me].left = l; nodes_[me].right = r;
        return me;
    }
    void nn_rec(int ni, const float *q, float &best, int &bi) const
    {
        const Node &n = nodes_[ni];
        if (n.dim < 0) {
            for (int i = n.lo; i < n.hi; ++i) {
                int id = order_[i];
                float d = dist2(q, p(id));
                if (d < best || (d == best && id < bi)) { best = d; bi = id; }
            }
            return;
        }
        float diff = q[n.dim] - n.split;
        int near = diff < 0 ? n.left : n.right, far = diff < 0 ? n.right : n.left;
        nn_rec(near, q, best, bi);
        if (diff * diff <= best) nn_rec(far, q, best, bi);
    }
    void knn_rec(int ni, const float *q, int k, std::vector<std::pair<float, int>> &heap) const
    {
        const Node &n = nodes_[ni];
        if (n.dim < 0) {
            for (int i = n.lo; i < n.hi; ++i) {
                int id = order_[i];
                std::pair<float, int> e(dist2(q, p(id)), id);
                if ((int)heap.size() < k) { heap.push_back(e); std::push_heap(heap.begin(), heap.end()); }
                else if (e < heap.front()) { std::pop_heap(heap.begin(), heap.end()); heap.back() = e; std::push_heap(heap.begin(), heap.end()); }
            }
            return;
        }
        float diff = q[n.dim] - n.split;
        int near = diff < 0 ? n.left : n.right, far = diff < 0 ? n.right : n.left;
        knn_rec(near, q, k, heap);
        if ((int)heap.size() < k || diff * diff <= heap.front().first) knn_rec(far, q, k, heap);
    }
    void rad_rec(int ni, const float *q, float r2, std::vector<std::pair<float, int>> &out) const
    {
        const Node &n = nodes_[ni];
        if (n.dim < 0) {
            for (int i = n.lo; i < n.hi; ++i) {
                int id = order_[i];
                float d = dist2(q, p(id));
                if (d <= r2) out.emplace_back(d, id); /* flann radiusSearch: dist <= radius^2 */
            }
            return;
        }
        float diff = q[n.dim] - n.split;
        int near = diff < 0 ? n.left : n.right, far = diff < 0 ? n.right : n.left;
        rad_rec(near, q, r2, out);
        if (diff * diff <= r2) rad_rec(far, q, r2, out);
    }
};

/* ------------------------------------------------------------------ */
/* GSL steffen.c                                  */
/* ------------------------------------------------------------------ */
inline double steffen_copysign(double x, double y)
{
    if ((x < 0 && y > 0) || (x > 0 && y < 0)) return -x;
    return x;
}

struct Steffen {
    std::vector<double> x, a, b, c, d;
    void init(const double *xs, const double *ys, int n)
    {
        x.assign(xs, xs + n);
        a.assign(n, 0); b.assign(n, 0); c.assign(n, 0); d.assign(n, 0);
        std::vector<double> yp(n);
        double h0 = xs[1] - xs[0];
        double s0 = (ys[1] - ys[0]) / h0;
        yp[0] = s0;
        for (int i = 1; i < n - 1; ++i) {
            double hi = xs[i + 1] - xs[i];
            double him1 = xs[i] - xs[i - 1];
            double si = (ys[i + 1] - ys[i]) / hi;
            double sim1 = (ys[i] - ys[i - 1]) / him1;
            double pi = (sim1 * hi + si * him1) / (him1 + hi);
            yp[i] = (steffen_copysign(1.0, sim1) + steffen_copysign(1.0, si)) *
                    std::min(fabs(sim1), std::min(fabs(si), 0.5 * fabs(pi)));
        }
        yp[n - 1] = (ys[n - 1] - ys[n - 2]) / (xs[n - 1] - xs[n - 2]);
        for (int i = 0; i < n - 1; ++i) {
            double hi = xs[i + 1] - xs[i];
            double si = (ys[i + 1] - ys[i]) / hi;
            a[i] = (yp[i] + yp[i + 1] - 2 * si) / hi / hi;
            b[i] = (3 * si - 2 * yp[i] - yp[i + 1]) / hi;
            c[i] = yp[i];
            d[i] = ys[i];
        }
    }
    /* gsl_interp_bsearch(x_array, x, 0, size-1) */
    int bsearch(double xq) const
    {
        size_t ilo = 0, ihi = x.size() - 1;
        while (ihi > ilo + 1) {
            size_t i = (ihi + ilo) / 2;
            if (x[i] > xq) ihi = i; else ilo = i;
        }
        return (int)ilo;
    }
    bool in_domain(double xq) const { return !(xq < x.front() || xq > x.back()); }
    double eval(double xq) const
    {
        int i = bsearch(xq);
        double delta = xq - x[i];
        return d[i] + delta * (c[i] + delta * (b[i] + delta * a[i]));
    }
};

/* include/Spline.h:7-51 */
struct Spline {
    std::vector<double> y, x, z;
    Steffen yx, yz;
    double small_y = 0, big_y = 0;
    void fit()
    {
        int n = (int)y.size();
        yx.init(y.data(), x.data(), n);
        yz.init(y.data(), z.data(), n);
        small_y = y.front(); big_y = y.back();
    }
    void point(double yy, double out[3]) const { out[0] = yx.eval(yy); out[1] = yy; out[2] = yz.eval(yy); }
};

/* ------------------------------------------------------------------ */
/* pcl::eigen33 (smallest eigenvalue form), SURVEY.md App. A.4         */
/* ------------------------------------------------------------------ */
void compute_roots2(float b, float c, float roots[3])
{
    roots[0] = 0.f;
    float d = float(b * b - 4.0 * c); /* Scalar (b * b - 4.0 * c): double intermediate */
    if (d < 0.0) d = 0.0;
    float sd = std::sqrt(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}

void compute_roots(const float m[3][3], float roots[3])
{
    float c0 = m[0][0] * m[1][1] * m[2][2] + 2.f * m[0][1] * m[0][2] * m[1][2] -
               m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] +
               m[1][1] * m[2][2] - m[1][2] * m[1][2];
    float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::abs(c0) < std::numeric_limits<float>::epsilon()) {
        compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = float(1.0 / 3.0);
    const float s_sqrt3 = std::sqrt(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    float rho = std::sqrt(-a_over_3);
    /* std::atan2 / std::cos / std::sin on floats (pcl/common/impl/eigen.hpp computeRoots).  Evaluated in double and
       rounded: the correctly rounded float results, which any libm's float functions return or miss by one ulp; the
       product does the same, so the eigen-decompositions (normals, principal curvatures) agree bit for bit and the
       discontinuous decisions of the dynamic adjustment built on them do not flip between oracle and GPU. */
    float theta = (float)std::atan2((double)std::sqrt(-q), (double)half_b) * s_inv3;
    float cos_theta = (float)std::cos((double)theta);
    float sin_theta = (float)std::sin((double)theta);
    roots[0] = c2_over_3 + 2.f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        std::swap(roots[1], roots[2]);
        if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    }
    if (roots[0] <= 0) compute_roots2(c2, c1, roots);
}

inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

void eigen33_smallest(const float cov[9], float *eigenvalue, float ev[3])
{
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(cov[i]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = cov[3 * i + j] / scale;
    float roots[3];
    compute_roots(m, roots);
    *eigenvalue = roots[0] * scale;
    m[0][0] -= roots[0]; m[1][1] -= roots[0]; m[2][2] -= roots[0];
    /* detail::getLargest3x3Eigenvector: largest of the three row cross products */
    float cp[3][3];
    cross3(m[0], m[1], cp[0]);
    cross3(m[0], m[2], cp[1]);
    cross3(m[1], m[2], cp[2]);
    float len[3];
    for (int i = 0; i < 3; ++i) len[i] = std::sqrt(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
    int idx = 0; /* maxCoeff: first maximum */
    if (len[1] > len[idx]) idx = 1;
    if (len[2] > len[idx]) idx = 2;
    for (int d = 0; d < 3; ++d) ev[d] = cp[idx][d] / len[idx];
}

/* ------------------------------------------------------------------ */
/* Eigen::EigenSolver<Matrix3f> (trans2center, path_slicing_alg.cpp:92-94)                    */
/* Restated from Eigen 3.3 / 3.4: RealSchur::compute (scaling by the largest |entry|, HessenbergDecomposition,     */
/* computeFromHessenberg with Francis double-shift steps), EigenSolver::compute + doComputeEigenvectors (back       */
/* substitution on T, back transformation by U) and eigenvectors() (columns normalised).  The order and the signs   */
/* of the eigenvectors are whatever this iteration leaves on T's diagonal -- the reference takes them as they come. */
/* Real eigenvalues only (a symmetric input); a complex pair (a numerically degenerate covariance) is reported.     */
/* Matrices are row-major m[r][c] here.                                                                              */
/* ------------------------------------------------------------------ */
struct EigenSolver3f {
    float T[3][3], U[3][3];
    float ev[3];
    bool complex_pair = false, converged = true;

    /* MatrixBase::makeHouseholder: v[0..n-1] -> essential part (n-1), tau, beta */
    static void make_householder(const float *v, int n, float *ess, float &tau, float &beta)
    {
        float tail_sq = 0.f;
        for (int i = 1; i < n; ++i) tail_sq = (i == 1) ? v[i] * v[i] : tail_sq + v[i] * v[i];
        const float c0 = v[0];
        if (tail_sq <= std::numeric_limits<float>::min()) {
            tau = 0.f; beta = c0;
            for (int i = 0; i < n - 1; ++i) ess[i] = 0.f;
        } else {
            beta = std::sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0.f) beta = -beta;
            for (int i = 0; i < n - 1; ++i) ess[i] = v[i + 1] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
    }
    /* MatrixBase::applyHouseholderOnTheLeft on the block rows r0..r0+nr-1, cols c0..c0+nc-1 */
    static void apply_left(float M[3][3], int r0, int nr, int c0, int nc, const float *ess, float tau)
    {
        if (nr == 1) { for (int j = 0; j < nc; ++j) M[r0][c0 + j] *= 1.f - tau; return; }
        if (tau == 0.f) return;
        for (int j = 0; j < nc; ++j) {
            float tmp = ess[0] * M[r0 + 1][c0 + j];
            for (int i = 2; i < nr; ++i) tmp += ess[i - 1] * M[r0 + i][c0 + j];
            tmp += M[r0][c0 + j];
            M[r0][c0 + j] -= tau * tmp;
            for (int i = 1; i < nr; ++i) M[r0 + i][c0 + j] -= tmp * (tau * ess[i - 1]);
        }
    }
    static void apply_right(float M[3][3], int r0, int nr, int c0, int nc, const float *ess, float tau)
    {
        if (nc == 1) { for (int i = 0; i < nr; ++i) M[r0 + i][c0] *= 1.f - tau; return; }
        if (tau == 0.f) return;
        for (int i = 0; i < nr; ++i) {
            float tmp = M[r0 + i][c0 + 1] * ess[0];
            for (int j = 2; j < nc; ++j) tmp += M[r0 + i][c0 + j] * ess[j - 1];
            tmp += M[r0 + i][c0];
            M[r0 + i][c0] -= tau * tmp;
            for (int j = 1; j < nc; ++j) M[r0 + i][c0 + j] -= ess[j - 1] * (tau * tmp);
        }
    }
    /* JacobiRotation::makeGivens (real) */
    static void make_givens(float p, float q, float &c, float &s)
    {
        if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
        else if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
        else if (std::abs(p) > std::abs(q)) {
            float t = q / p, u = std::sqrt(1.f + t * t);
            if (p < 0.f) u = -u;
            c = 1.f / u; s = -t * c;
        } else {
            float t = p / q, u = std::sqrt(1.f + t * t);
            if (q < 0.f) u = -u;
            s = -1.f / u; c = -t * s;
        }
    }

    void compute(const float A[3][3])
    {
        const float eps = std::numeric_limits<float>::epsilon();
        /* RealSchur::compute */
        float scale = 0.f;
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) scale = std::max(scale, std::abs(A[i][j]));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T[i][j] = 0.f; U[i][j] = i == j ? 1.f : 0.f; }
        if (scale < std::numeric_limits<float>::min()) { finish(); return; }
        float H[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[i][j] = A[i][j] / scale;
        /* HessenbergDecomposition::_compute, n = 3: one reflector on (H10, H20); the second (length 1) has tau = 0 */
        float v2[2] = {H[1][0], H[2][0]}, ess0, h0, beta;
        make_householder(v2, 2, &ess0, h0, beta);
        H[2][0] = ess0; H[1][0] = beta;
        apply_left(H, 1, 2, 1, 2, &ess0, h0);
        apply_right(H, 0, 3, 1, 2, &ess0, h0);
        /* i = 1: tau = 0, beta = H21; the 1 x 1 / 1-column applications multiply by (1 - 0) */
        /* matrixQ().evalTo(U): identity, then the reflector on its bottom-right 2 x 2 corner */
        apply_left(U, 1, 2, 1, 2, &ess0, h0);
        /* matrixH() */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] = H[i][j];
        T[2][0] = 0.f;
        /* computeFromHessenberg */
        const int size = 3, max_iters = 40 * 3;
        int iu = size - 1, iter = 0, total_iter = 0;
        float exshift = 0.f, norm = 0.f;
        for (int j = 0; j < size; ++j) { /* computeNormOfT */
            float cs = std::abs(T[0][j]);
            for (int i = 1; i < std::min(size, j + 2); ++i) cs += std::abs(T[i][j]);
            norm += cs;
        }
        const float consider_as_zero = std::max(norm * (eps * eps), std::numeric_limits<float>::min());
        if (norm != 0.f) {
            while (iu >= 0) {
                int il = iu; /* findSmallSubdiagEntry */
                while (il > 0) {
                    float s = std::abs(T[il - 1][il - 1]) + std::abs(T[il][il]);
                    s = std::max(s * eps, consider_as_zero);
                    if (std::abs(T[il][il - 1]) <= s) break;
                    il--;
                }
                if (il == iu) { /* one root */
                    T[iu][iu] = T[iu][iu] + exshift;
                    if (iu > 0) T[iu][iu - 1] = 0.f;
                    iu--; iter = 0;
                } else if (il == iu - 1) { /* two roots: splitOffTwoRows */
                    float p = 0.5f * (T[iu - 1][iu - 1] - T[iu][iu]);
                    float q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
                    T[iu][iu] += exshift;
                    T[iu - 1][iu - 1] += exshift;
                    if (q >= 0.f) {
                        float z = std::sqrt(std::abs(q)), c, s;
                        if (p >= 0.f) make_givens(p + z, T[iu][iu - 1], c, s);
                        else make_givens(p - z, T[iu][iu - 1], c, s);
                        for (int j = iu - 1; j < size; ++j) { /* rightCols(size-iu+1).applyOnTheLeft(iu-1, iu, rot.adjoint()) */
                            float x = T[iu - 1][j], y = T[iu][j];
                            T[iu - 1][j] = c * x - s * y;
                            T[iu][j] = s * x + c * y;
                        }
                        for (int i = 0; i <= iu; ++i) { /* topRows(iu+1).applyOnTheRight(iu-1, iu, rot) */
                            float x = T[i][iu - 1], y = T[i][iu];
                            T[i][iu - 1] = c * x - s * y;
                            T[i][iu] = s * x + c * y;
                        }
                        T[iu][iu - 1] = 0.f;
                        for (int i = 0; i < size; ++i) {
                            float x = U[i][iu - 1], y = U[i][iu];
                            U[i][iu - 1] = c * x - s * y;
                            U[i][iu] = s * x + c * y;
                        }
                    } else complex_pair = true;
                    if (iu > 1) T[iu - 1][iu - 2] = 0.f;
                    iu -= 2; iter = 0;
                } else { /* Francis QR step on rows il..iu (here il = 0, iu = 2) */
                    float shift[3] = {T[iu][iu], T[iu - 1][iu - 1], T[iu][iu - 1] * T[iu - 1][iu]}; /* computeShift */
                    if (iter == 10) {
                        exshift += shift[0];
                        for (int i = 0; i <= iu; ++i) T[i][i] -= shift[0];
                        float s = std::abs(T[iu][iu - 1]) + std::abs(T[iu - 1][iu - 2]);
                        shift[0] = 0.75f * s; shift[1] = 0.75f * s; shift[2] = -0.4375f * s * s;
                    }
                    if (iter == 30) {
                        float s = (shift[1] - shift[0]) / 2.0f;
                        s = s * s + shift[2];
                        if (s > 0.f) {
                            s = std::sqrt(s);
                            if (shift[1] < shift[0]) s = -s;
                            s = s + (shift[1] - shift[0]) / 2.0f;
                            s = shift[0] - shift[2] / s;
                            exshift += s;
                            for (int i = 0; i <= iu; ++i) T[i][i] -= s;
                            shift[0] = shift[1] = shift[2] = 0.964f;
                        }
                    }
                    iter++; total_iter++;
                    if (total_iter > max_iters) { converged = false; break; }
                    int im; /* initFrancisQRStep */
                    float fv[3] = {0.f, 0.f, 0.f};
                    for (im = iu - 2; im >= il; --im) {
                        const float Tmm = T[im][im], r = shift[0] - Tmm, s = shift[1] - Tmm;
                        fv[0] = (r * s - shift[2]) / T[im + 1][im] + T[im][im + 1];
                        fv[1] = T[im + 1][im + 1] - Tmm - r - s;
                        fv[2] = T[im + 2][im + 1];
                        if (im == il) break;
                        const float lhs = T[im][im - 1] * (std::abs(fv[1]) + std::abs(fv[2]));
                        const float rhs = fv[0] * (std::abs(T[im - 1][im - 1]) + std::abs(Tmm) + std::abs(T[im + 1][im + 1]));
                        if (std::abs(lhs) < eps * rhs) break;
                    }
                    for (int k = im; k <= iu - 2; ++k) { /* performFrancisQRStep */
                        const bool first = (k == im);
                        float v[3];
                        if (first) { v[0] = fv[0]; v[1] = fv[1]; v[2] = fv[2]; }
                        else { v[0] = T[k][k - 1]; v[1] = T[k + 1][k - 1]; v[2] = T[k + 2][k - 1]; }
                        float ess[2], tau, bt;
                        make_householder(v, 3, ess, tau, bt);
                        if (bt != 0.f) {
                            if (first && k > il) T[k][k - 1] = -T[k][k - 1];
                            else if (!first) T[k][k - 1] = bt;
                            apply_left(T, k, 3, k, size - k, ess, tau);
                            apply_right(T, 0, std::min(iu, k + 3) + 1, k, 3, ess, tau);
                            apply_right(U, 0, size, k, 3, ess, tau);
                        }
                    }
                    {
                        float v[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]}, ess, tau, bt;
                        make_householder(v, 2, &ess, tau, bt);
                        if (bt != 0.f) {
                            T[iu - 1][iu - 2] = bt;
                            apply_left(T, iu - 1, 2, iu - 1, size - iu + 1, &ess, tau);
                            apply_right(T, 0, iu + 1, iu - 1, 2, &ess, tau);
                            apply_right(U, 0, size, iu - 1, 2, &ess, tau);
                        }
                    }
                    for (int i = im + 2; i <= iu; ++i) { /* clean up pollution due to round-off errors */
                        T[i][i - 2] = 0.f;
                        if (i > im + 2) T[i][i - 3] = 0.f;
                    }
                }
            }
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] *= scale; /* m_matT *= scale */
        finish();
    }

    /* EigenSolver::compute after the Schur form: eigenvalues off T, doComputeEigenvectors, eigenvectors() */
    void finish()
    {
        const float eps = std::numeric_limits<float>::epsilon();
        const int size = 3;
        for (int i = 0; i < size; ++i) ev[i] = T[i][i];
        for (int i = 0; i + 1 < size; ++i) if (T[i + 1][i] != 0.f) complex_pair = true;
        if (complex_pair || !converged) return;
        float norm = 0.f;
        for (int j = 0; j < size; ++j) {
            const int a = std::max(j - 1, 0);
            float rs = std::abs(T[j][a]);
            for (int k = a + 1; k < size; ++k) rs += std::abs(T[j][k]);
            norm += rs;
        }
        if (norm != 0.f) {
            for (int n = size - 1; n >= 0; n--) {
                const float p = ev[n];
                int l = n;
                T[n][n] = 1.f;
                for (int i = n - 1; i >= 0; i--) {
                    const float w = T[i][i] - p;
                    float r = T[i][l] * T[l][n];
                    for (int k = l + 1; k <= n; ++k) r += T[i][k] * T[k][n];
                    l = i;
                    if (w != 0.f) T[i][n] = -r / w;
                    else T[i][n] = -r / (eps * norm);
                    const float t = std::abs(T[i][n]);
                    if ((eps * t) * t > 1.f) for (int k = i; k < size; ++k) T[k][n] /= t;
                }
            }
            for (int j = size - 1; j >= 0; j--) { /* back transformation */
                float tmp[3];
                for (int i = 0; i < size; ++i) {
                    float a = U[i][0] * T[0][j];
                    for (int k = 1; k <= j; ++k) a += U[i][k] * T[k][j];
                    tmp[i] = a;
                }
                for (int i = 0; i < size; ++i) U[i][j] = tmp[i];
            }
        }
        for (int j = 0; j < size; ++j) { /* eigenvectors(): matV.col(j).normalize() */
            const float z = (U[0][j] * U[0][j] + U[1][j] * U[1][j]) + U[2][j] * U[2][j];
            if (z > 0.f) { const float s = std::sqrt(z); for (int i = 0; i < size; ++i) U[i][j] /= s; }
        }
    }
};

/* Matrix4f::inverse(): Eigen's generic 4 x 4 path (cofactors, InverseImpl.h compute_inverse_size4).  The SSE build
   of Eigen runs Intel's 4 x 4 routine instead, whose entries differ from these in the last bits. */
inline float det3_helper(const float m[4][4], int i1, int i2, int i3, int j1, int j2, int j3)
{
    return m[i1][j1] * (m[i2][j2] * m[i3][j3] - m[i2][j3] * m[i3][j2]);
}
inline float cofactor_4x4(const float m[4][4], int i, int j)
{
    const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
    return det3_helper(m, i1, i2, i3, j1, j2, j3) + det3_helper(m, i2, i3, i1, j1, j2, j3) + det3_helper(m, i3, i1, i2, j1, j2, j3);
}
inline void inverse_4x4(const float m[4][4], float r[4][4])
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r[j][i] = (((i + j) & 1) ? -1.f : 1.f) * cofactor_4x4(m, i, j);
    const float det = ((m[0][0] * r[0][0] + m[1][0] * r[0][1]) + m[2][0] * r[0][2]) + m[3][0] * r[0][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[i][j] /= det;
}
/* pcl::transformPointCloud, float transform, SSE2 build (common/impl/transforms.hpp detail::Transformer<float>::se3):
   p0 + (p1 + (p2 + c3)) per row; non-finite points pass unchanged */
inline void transform_se3(const float m[4][4], const float *src, float *dst)
{
    const float x = src[0], y = src[1], z = src[2];
    for (int r = 0; r < 3; ++r) dst[r] = m[r][0] * x + (m[r][1] * y + (m[r][2] * z + m[r][3]));
}

/* the same pcl::eigen33 path with Scalar = double (pcl::MLSResult::computeMLSSurface calls it on a Matrix3d) */
void compute_roots2_d(double b, double c, double roots[3])
{
    roots[0] = 0.0;
    double d = b * b - 4.0 * c;
    if (d < 0.0) d = 0.0;
    double sd = std::sqrt(d);
    roots[2] = 0.5 * (b + sd);
    roots[1] = 0.5 * (b - sd);
}
void compute_roots_d(const double m[3][3], double roots[3])
{
    double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[0][1] * m[0][2] * m[1][2] -
                m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    double c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] +
                m[1][1] * m[2][2] - m[1][2] * m[1][2];
    double c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::abs(c0) < std::numeric_limits<double>::epsilon()) { compute_roots2_d(c2, c1, roots); return; }
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = std::sqrt(3.0);
    double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0) a_over_3 = 0.0;
    double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0) q = 0.0;
    double rho = std::sqrt(-a_over_3);
    double theta = std::atan2(std::sqrt(-q), half_b) * s_inv3;
    double cos_theta = std::cos(theta), sin_theta = std::sin(theta);
    roots[0] = c2_over_3 + 2.0 * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        std::swap(roots[1], roots[2]);
        if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    }
    if (roots[0] <= 0) compute_roots2_d(c2, c1, roots);
}
void eigen33_smallest_d(const double cov[9], double *eigenvalue, double ev[3])
{
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(cov[i]));
    if (scale <= std::numeric_limits<double>::min()) scale = 1.0;
    double m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = cov[3 * i + j] / scale;
    double roots[3];
    compute_roots_d(m, roots);
    *eigenvalue = roots[0] * scale;
    m[0][0] -= roots[0]; m[1][1] -= roots[0]; m[2][2] -= roots[0];
    double cp[3][3];
    for (int k = 0; k < 3; ++k) {
        const double *a = m[k == 2 ? 1 : 0], *b = m[k == 0 ? 1 : 2];
        cp[k][0] = a[1] * b[2] - a[2] * b[1];
        cp[k][1] = a[2] * b[0] - a[0] * b[2];
        cp[k][2] = a[0] * b[1] - a[1] * b[0];
    }
    double len[3];
    for (int i = 0; i < 3; ++i) len[i] = std::sqrt(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
    int idx = 0;
    if (len[1] > len[idx]) idx = 1;
    if (len[2] > len[idx]) idx = 2;
    for (int d = 0; d < 3; ++d) ev[d] = cp[idx][d] / len[idx];
}

/* ------------------------------------------------------------------ */
/* Eigen restatements                             */
/* ------------------------------------------------------------------ */
struct Quat { float w, x, y, z; };
inline Quat quat_axis(float angle, int axis)
{   /* Quaternion = AngleAxis: ha = 0.5*angle; w = cos(ha); vec = sin(ha)*axis */
    float ha = 0.5f * angle;
    float s = std::sin(ha);
    Quat q{std::cos(ha), 0.f, 0.f, 0.f};
    (&q.x)[axis] = s * 1.0f;
    return q;
}
inline Quat quat_mul(const Quat &a, const Quat &b)
{
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline void quat_to_mat(const Quat &q, float R[3][3])
{
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1.f - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1.f - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1.f - (txx + tyy);
}
/* AngleAxisf(rz,Z)*AngleAxisf(ry,Y)*AngleAxisf(rx,X) assigned to a Matrix3f */
inline void rot_zyx(float rx, float ry, float rz, float R[3][3])
{
    Quat q = quat_mul(quat_mul(quat_axis(rz, 2), quat_axis(ry, 1)), quat_axis(rx, 0));
    quat_to_mat(q, R);
}
/* MatrixBase::eulerAngles(2,1,0): returns (e0,e1,e2) = (yaw,pitch,roll) */
inline void euler_zyx(const float m[3][3], float e[3])
{
    const float kPi = 3.14159265358979323846f; /* Scalar(EIGEN_PI) */
    e[0] = std::atan2(m[1][0], m[0][0]);
    float c2 = std::sqrt(m[2][2] * m[2][2] + m[2][1] * m[2][1]);
    if (e[0] < 0.f) {
        e[0] += kPi;
        e[1] = std::atan2(-m[2][0], -c2);
    } else {
        e[1] = std::atan2(-m[2][0], c2);
    }
    float s1 = std::sin(e[0]);
    float c1 = std::cos(e[0]);
    e[2] = std::atan2(s1 * m[0][2] - c1 * m[1][2], c1 * m[1][1] - s1 * m[0][1]);
}

/* path_translation_alg.cpp:3-35 */
void handeye_transform(const float he[6], float wp[6])
{
    float HE[3][3], P[3][3];
    rot_zyx(he[3], he[4], he[5], HE);
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

/* path_translation_alg.cpp:192-202 */
void pose_from_normal(const float n[3], float rpy[3])
{
    float A[3] = {-n[0], -n[1], -n[2]};
    const float X[3] = {1.f, 0.f, 0.f};
    float O[3], Nn[3];
    cross3(A, X, O);
    cross3(O, A, Nn);
    float M[3][3];
    for (int i = 0; i < 3; ++i) { M[i][0] = Nn[i]; M[i][1] = O[i]; M[i][2] = A[i]; }
    float e[3];
    euler_zyx(M, e);
    rpy[0] = e[2]; rpy[1] = e[1]; rpy[2] = e[0];
}

/* path_translation_alg.cpp:114-141 (+ the stop rule documented at the top) */
int position_smooth(std::vector<std::array<float, 6>> &list, int max_sweeps)
{
    std::vector<std::array<float, 6>> new_path(list);
    double tolerance = 0.00001, change = tolerance, weight_data = 0.65, weight_smooth = 1 - weight_data;
    double x_i, y_i, y_prev, y_next, y_i_saved;
    int dim = 3, path_len = (int)list.size();
    int sweeps = 0;
    double prev_change = INFINITY;
    while (change >= tolerance) {
        change = 0;
        for (int i = 1; i < path_len - 1; i++) {
            for (int j = 0; j < dim; j++) {
                x_i = list[i][j];
                y_i = new_path[i][j];
                y_prev = new_path[i - 1][j];
                y_next = new_path[i + 1][j];
                y_i_saved = y_i;
                y_i += (weight_data * (x_i - y_i) + weight_smooth * (y_next + y_prev - 2 * y_i));
                new_path[i][j] = (float)y_i;
                change += fabs(y_i - y_i_saved);
            }
        }
        ++sweeps;
        if (sweeps >= 2 && change >= 0.9 * prev_change) break; /* float floor reached */
        if (sweeps >= max_sweeps) break;
        prev_change = change;
    }
    list = std::move(new_path);
    return sweeps;
}

/* path_translation_alg.cpp:37-86.  `oob` is set when the B.6 read past the list is hit;
   the offending do-while pass is then skipped (the reference has UB there). */
void reduce_rpy(std::vector<std::array<float, 6>> &W, const std::vector<int> &Index, double RPYres, int *oob)
{
    if (RPYres <= 2) return;
    int preId = 0, lastId = 0, res = (int)RPYres;
    const int n = (int)W.size();
    for (int id = 0; id < (int)Index.size(); id++) {
        do {
            double dr[3];
            lastId = preId + res;
            if (lastId >= n) { if (oob) *oob = 1; break; }
            for (size_t D = 3; D < 6; D++) {
                if (W[lastId][D] * W[preId][D] >= 0) {
                    dr[D - 3] = (W[lastId][D] - W[preId][D]) / res;
                } else {
                    double no1, no2;
                    if (W[lastId][D] < 0) {
                        no2 = W[preId][D];
                        no1 = 2 * M_PI + W[lastId][D];
                    } else {
                        no2 = 2 * M_PI + W[preId][D];
                        no1 = W[lastId][D];
                    }
                    dr[D - 3] = std::abs(W[lastId][D] - W[preId][D]) < std::abs(no1 - no2)
                                    ? (W[lastId][D] - W[preId][D]) : (no1 - no2);
                    dr[D - 3] /= res;
                }
            }
            for (int wi = 1; wi < res; wi++) {
                W[preId + wi][3] = float(dr[0] + W[preId + wi - 1][3]);
                W[preId + wi][4] = float(dr[1] + W[preId + wi - 1][4]);
                W[preId + wi][5] = float(dr[2] + W[preId + wi - 1][5]);
            }
            preId = lastId;
        } while ((preId + res) <= Index[id]);
        if (preId != Index[id]) {
            for (int i = preId + 1; i <= Index[id]; i++) {
                W[i][3] = W[preId][3]; W[i][4] = W[preId][4]; W[i][5] = W[preId][5];
            }
        }
        preId = Index[id] + 1;
    }
    for (auto &p : W)
        for (int D = 3; D < 6; ++D) p[D] = p[D] > M_PI ? float(p[D] - 2 * M_PI) : p[D];
}

/* path_translation_alg.cpp:89-112 */
void trans_flange(std::vector<std::array<float, 6>> &W, float EElen)
{
    const float ee[3] = {0.f, 0.f, -EElen};
    for (auto &p : W) {
        float R[3][3];
        rot_zyx(p[3], p[4], p[5], R);
        for (int i = 0; i < 3; ++i)
            p[i] = R[i][0] * ee[0] + R[i][1] * ee[1] + R[i][2] * ee[2] + p[i] * 1.f;
    }
}

} // namespace

/* ====================================================================== */
struct ppo_handle {
    ppo_params P;
    std::vector<Pt> cloud;
    KdTree tree; bool tree_built = false;
    std::vector<float> normals; /* 4 per point */
    std::vector<uint8_t> normal_done;
    std::vector<float> px;
    std::vector<std::vector<int>> slice_idx;
    std::vector<Spline> path_set;
    std::vector<Spline> boundary_set; /* [s]: the boundary slice s was adjusted against (dynamic adjustment; empty = none) */
    std::vector<std::array<float, 6>> wp, wp_pre, wp_smooth;
    std::vector<std::array<float, 3>> wp_xyz;
    std::vector<std::array<float, 4>> wp_normal;
    std::vector<int> wp_nn, tail;
    int sweeps = 0, oob = 0;

    void ensure_tree() { if (!tree_built) { tree.build(cloud.data(), nullptr, (int)cloud.size()); tree_built = true; } }
    void rebuild_tree() { tree.build(cloud.data(), nullptr, (int)cloud.size()); tree_built = true; }

    /* pcl::NormalEstimation::computeFeature for one point */
    void point_normal(int idx, float out[4])
    {
        ensure_tree();
        std::vector<std::pair<float, int>> nb;
        const Pt &q = cloud[idx];
        tree.radius(&q.x, P.normal_radius, nb);
        const float nanv = std::numeric_limits<float>::quiet_NaN();
        if (nb.size() < 3) { out[0] = out[1] = out[2] = out[3] = nanv; return; }
        /* computeMeanAndCovarianceMatrix (PCL 1.12): shift by the first neighbour */
        const Pt &K = cloud[nb[0].second];
        float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (auto &e : nb) {
            const Pt &c = cloud[e.second];
            float x = c.x - K.x, y = c.y - K.y, z = c.z - K.z;
            accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
            accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
            accu[6] += x; accu[7] += y; accu[8] += z;
        }
        float cnt = (float)nb.size();
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
        eigen33_smallest(cov, &ev, n);
        float eig_sum = cov[0] + cov[4] + cov[8];
        float curv = eig_sum != 0 ? std::abs(ev / eig_sum) : 0.f;
        /* flipNormalTowardsViewpoint: vp = sensor origin, NOT scaled by the x1000 */
        float vx = P.viewpoint[0] - q.x, vy = P.viewpoint[1] - q.y, vz = P.viewpoint[2] - q.z;
        float cos_theta = vx * n[0] + vy * n[1] + vz * n[2];
        if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
        out[0] = n[0]; out[1] = n[1]; out[2] = n[2]; out[3] = curv;
    }
    void estimate_normal_all()
    {   /* path_slicing_alg.cpp:141-150: a fresh search tree + every point */
        size_t n = cloud.size();
        normals.assign(4 * n, 0.f);
        normal_done.assign(n, 1);
        rebuild_tree();
        const int nt = P.threads > 1 ? P.threads : 1; /* all-cores context figure of bench.py; 1 = the reference */
#pragma omp parallel for schedule(static, 4096) num_threads(nt) if (nt > 1)
        for (long long i = 0; i < (long long)n; ++i) point_normal((int)i, &normals[4 * (size_t)i]);
    }
    const float *normal_lazy(int idx)
    {
        if (normals.size() != 4 * cloud.size()) { normals.assign(4 * cloud.size(), 0.f); normal_done.assign(cloud.size(), 0); }
        if (!normal_done[idx]) { point_normal(idx, &normals[4 * idx]); normal_done[idx] = 1; }
        return &normals[4 * idx];
    }

    void minmax(float mn[3], float mx[3]) const
    {   /* pcl::getMinMax3D: finite points only */
        for (int d = 0; d < 3; ++d) { mn[d] = std::numeric_limits<float>::max(); mx[d] = -std::numeric_limits<float>::max(); }
        for (const Pt &p : cloud) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
            mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
        }
    }

    std::vector<float> slice_positions() const
    {
        float mn[3], mx[3];
        minmax(mn, mx);
        const float min_x = mn[0], max_x = mx[0];
        const double toolRadius = P.tool_radius;
        int step_size = int(toolRadius * 2);
        std::vector<float> out;
        if (step_size <= 0) return out;
        switch (P.walk) {
        case PPO_WALK_SECTPATH: { /* path_slicing_alg.cpp:308-330 */
            std::vector<float> front;
            float loc = (min_x + max_x) / 2 - step_size;
            while (loc > min_x) { front.insert(front.begin(), loc); loc -= step_size; }
            out = front;
            loc = (min_x + max_x) / 2;
            while (loc < max_x) { out.push_back(loc); loc += step_size; }
            break;
        }
        case PPO_WALK_CENTER_INT: { /* path_dynamic_alg.cpp:308-372, Path_Generate_Algorithm.h:112 */
            int imin = (int)min_x, imax = (int)max_x;
            std::vector<float> front, back;
            int loc = (imax + imin) / 2;
            loc -= step_size;
            while (imax > loc && loc > imin) { front.push_back((float)loc); loc -= step_size; }
            loc = (imax + imin) / 2 + step_size;
            while (imax > loc && loc > imin) { back.push_back((float)loc); loc += step_size; }
            out.assign(front.rbegin(), front.rend());
            out.push_back((min_x + max_x) / 2);
            out.insert(out.end(), back.begin(), back.end());
            break;
        }
        case PPO_WALK_SDIR_INT: { /* dynamic_alg_sdir.cpp:349-374 */
            int loc = int(min_x + toolRadius);
            out.push_back((float)loc);
            loc += step_size;
            while (loc < max_x) { out.push_back((float)loc); loc += step_size; }
            break;
        }
        case PPO_WALK_V1_CONTACT: { /* Path_Generation.cpp:711-725 */
            float locateX = float(min_x + toolRadius);
            while (locateX < max_x) { out.push_back(locateX); locateX += step_size; }
            break;
        }
        case PPO_WALK_V1_SLICING: { /* Path_Generation.cpp:295-304 */
            float x = min_x;
            x += step_size / 2;
            while (x < max_x) { out.push_back(x); x += step_size; }
            break;
        }
        }
        return out;
    }

    /* pcl::PassThrough on "x", limits [-2+position, 2+position] (App. A.2) */
    std::vector<int> ranged_x_index(int position) const
    {
        std::vector<int> indices;
        const float lo = float(-2 + position), hi = float(2 + position);
        const int n = (int)cloud.size();
        for (int i = 0; i < n; ++i) {
            const Pt &p = cloud[i];
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            if (p.x < lo || p.x > hi) continue;
            indices.push_back(i);
        }
        return indices;
    }

    /* one pass bucketing of all slices (fast mode; same lists as ranged_x_index) */
    void bucket_slices(const std::vector<float> &planes, std::vector<std::vector<int>> &lists) const
    {
        int S = (int)planes.size();
        lists.assign(S, {});
        std::vector<float> lo(S), hi(S);
        for (int s = 0; s < S; ++s) { int pos = (int)planes[s]; lo[s] = float(-2 + pos); hi[s] = float(2 + pos); }
        const int n = (int)cloud.size();
        for (int i = 0; i < n; ++i) {
            const Pt &p = cloud[i];
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            /* first slice whose hi >= x (hi ascending) */
            int s = int(std::lower_bound(hi.begin(), hi.end(), p.x) - hi.begin());
            for (; s < S && lo[s] <= p.x; ++s)
                if (!(p.x < lo[s] || p.x > hi[s])) lists[s].push_back(i);
        }
    }

    /* returns node map or error (<0) */
    int insert_point(const std::vector<int> &indices, float plane_x, std::map<double, std::array<double, 2>> &Node)
    {
        std::vector<int> El, Er;
        for (int i : indices) {
            float distance2plane = (cloud[i].x - plane_x) * 1.f + (cloud[i].y - 0.f) * 0.f + (cloud[i].z - 0.f) * 0.f;
            if (distance2plane > 0) El.push_back(i);
            else if (distance2plane < 0) Er.push_back(i);
        }
        std::vector<int> left_pair, right_pair;
        if (P.pairing == PPO_PAIR_KD) {
            /* path_slicing_alg.cpp:184-210 */
            if (El.empty()) { Node.clear(); return 0; }
            if (Er.empty()) return -1; /* empty FLANN tree: the reference crashes */
            KdTree treeEl, treeEr;
            treeEl.build(cloud.data(), El.data(), (int)El.size());
            treeEr.build(cloud.data(), Er.data(), (int)Er.size());
            for (int i = 0; i < (int)El.size(); i++) {
                const Pt &pl = cloud[El[i]];
                int r = treeEr.nearest(&pl.x);
                const Pt &pr = cloud[Er[r]];
                /* kdtree.nearestKSearch(pr,1) on the full cloud returns pr itself (or a
                   coordinate duplicate: same values either way) */
                right_pair.push_back(Er[r]);
                int l = treeEl.nearest(&pr.x);
                left_pair.push_back(El[l]);
            }
        } else {
            /* Path_Generation.cpp:129-179 */
            std::vector<int> El_flag(El.size(), 0), Er_flag(Er.size(), 0);
            int rp_index = 0;
            for (int i = 0; i < int(El.size()); i++) {
                if (El_flag[i] == 0) {
                    if (Er.empty()) return -1; /* compare.begin() on an empty map */
                    /* std::map<float,int> compare; compare[norm] = j  => smallest norm, last j */
                    float best = 0; int bj = -1;
                    for (int j = 0; j < int(Er.size()); j++) {
                        float vx = cloud[El[i]].x - cloud[Er[j]].x;
                        float vy = cloud[El[i]].y - cloud[Er[j]].y;
                        float vz = cloud[El[i]].z - cloud[Er[j]].z;
                        float nrm = std::sqrt(vx * vx + (vy * vy + vz * vz)); /* Vector3f::norm() */
                        if (bj < 0 || nrm <= best) { best = nrm; bj = j; }
                    }
                    if (Er_flag[bj] == 0) {
                        rp_index = Er[bj];
                        right_pair.push_back(rp_index);
                        Er_flag[bj] = 1;
                    } else
                        continue;
                    best = 0; bj = -1;
                    for (int j = 0; j < int(El.size()); j++) {
                        float vx = cloud[rp_index].x - cloud[El[j]].x;
                        float vy = cloud[rp_index].y - cloud[El[j]].y;
                        float vz = cloud[rp_index].z - cloud[El[j]].z;
                        float nrm = std::sqrt(vx * vx + (vy * vy + vz * vz));
                        if (bj < 0 || nrm <= best) { best = nrm; bj = j; }
                    }
                    if (El_flag[bj] == 0) {
                        left_pair.push_back(El[bj]);
                        El_flag[bj] = 1;
                    }
                }
            }
        }
        Node.clear();
        for (int i = 0; i < (int)left_pair.size(); i++) {
            int index_right = right_pair[i], index_left = left_pair[i];
            float t = (plane_x - cloud[index_right].x) / (cloud[index_left].x - cloud[index_right].x);
            float ix = plane_x;
            float iy = cloud[index_right].y + t * (cloud[index_left].y - cloud[index_right].y);
            float iz = cloud[index_right].z + t * (cloud[index_left].z - cloud[index_right].z);
            Node[(double)iy] = {(double)ix, (double)iz};
        }
        return (int)Node.size();
    }


    /* ---------------- dynamic adjustment (path_dynamic_alg.cpp:77-306) ---------------- */
    const float *normal_of(int idx) { return P.reference_complexity ? &normals[4 * (size_t)idx] : normal_lazy(idx); }

    /* pcl::PrincipalCurvaturesEstimation::computePointPrincipalCurvatures (PCL 1.12
       features/impl/principal_curvatures.hpp) */
    void principal_curvatures(int p_idx, const std::vector<std::pair<float, int>> &nb, float pc[5])
    {
        const float *np = normal_of(p_idx);
        const float n_idx[3] = {np[0], np[1], np[2]};
        float M[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i][j] = (i == j ? 1.f : 0.f) - n_idx[i] * n_idx[j];
        std::vector<std::array<float, 3>> proj(nb.size());
        float cen[3] = {0, 0, 0};
        for (size_t k = 0; k < nb.size(); ++k) {
            const float *nn = normal_of(nb[k].second);
            for (int i = 0; i < 3; ++i) proj[k][i] = M[i][0] * nn[0] + M[i][1] * nn[1] + M[i][2] * nn[2];
            for (int i = 0; i < 3; ++i) cen[i] += proj[k][i];
        }
        for (int i = 0; i < 3; ++i) cen[i] /= (float)nb.size();
        float cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = 0; k < nb.size(); ++k) {
            float d[3] = {proj[k][0] - cen[0], proj[k][1] - cen[1], proj[k][2] - cen[2]};
            double dxy = d[0] * d[1], dxz = d[0] * d[2], dyz = d[1] * d[2];
            cov[0] += d[0] * d[0]; cov[1] += (float)dxy; cov[2] += (float)dxz;
            cov[3] += (float)dxy; cov[4] += d[1] * d[1]; cov[5] += (float)dyz;
            cov[6] += (float)dxz; cov[7] += (float)dyz; cov[8] += d[2] * d[2];
        }
        /* pcl::eigen33(mat, evals): scaled roots */
        float scale = 0.f;
        for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(cov[i]));
        if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
        float m[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = cov[3 * i + j] / scale;
        float ev[3];
        compute_roots(m, ev);
        for (int i = 0; i < 3; ++i) ev[i] *= scale;
        /* pcl::computeCorrespondingEigenVector(mat, evals[2], vec) */
        float sm[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sm[i][j] = cov[3 * i + j] / scale;
        const float shift = ev[2] / scale;
        sm[0][0] -= shift; sm[1][1] -= shift; sm[2][2] -= shift;
        float cp[3][3];
        cross3(sm[0], sm[1], cp[0]); cross3(sm[0], sm[2], cp[1]); cross3(sm[1], sm[2], cp[2]);
        float len[3];
        for (int i = 0; i < 3; ++i) len[i] = std::sqrt(cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1] + cp[i][2] * cp[i][2]);
        int bi = 0;
        if (len[1] > len[bi]) bi = 1;
        if (len[2] > len[bi]) bi = 2;
        for (int d = 0; d < 3; ++d) pc[d] = cp[bi][d] / len[bi];
        float inv = 1.0f / (float)nb.size();
        pc[3] = ev[2] * inv;
        pc[4] = ev[1] * inv;
    }

    /* compute_transform (path_dynamic_alg.cpp:77-107): T = [n x c | c | n | p] */
    void compute_transform(const float sp[3], float T[3][4], float pcv[2])
    {
        ensure_tree();
        std::vector<std::pair<float, int>> nb;
        tree.knn(sp, P.curvature_k, nb);
        float pc[5];
        principal_curvatures(nb[0].second, nb, pc);
        pcv[0] = pc[3]; pcv[1] = pc[4];
        const float *nn = normal_of(nb[0].second);
        const float nv[3] = {nn[0], nn[1], nn[2]}, cv[3] = {pc[0], pc[1], pc[2]};
        float cr[3];
        cross3(nv, cv, cr);
        for (int i = 0; i < 3; ++i) { T[i][0] = cr[i]; T[i][1] = cv[i]; T[i][2] = nv[i]; T[i][3] = sp[i]; }
    }

    /* Area2Cloud (path_dynamic_alg.cpp:110-180); key: 0 = left (min x), 1 = right (max x) */
    void area2cloud(const double point[3], int key, float bound[3])
    {
        const float sp[3] = {(float)point[0], (float)point[1], (float)point[2]};
        float T[3][4], pc[2];
        compute_transform(sp, T, pc);
        const double toolRadius = P.tool_radius, depth = P.depth, toolthickness = P.toolthickness;
        double longAxis, shortAxis;
        if ((pc[0] >= 0) && (pc[1] >= 0)) {
            longAxis = std::sqrt(std::pow(1 / pc[1], 2) - std::pow(std::abs(1 / pc[1]) - depth, 2));
            if (longAxis > toolRadius) longAxis = toolRadius;
            shortAxis = std::sqrt(std::pow(1 / pc[0], 2) - std::pow(std::abs(1 / pc[0]) - depth, 2));
            if (shortAxis > toolRadius) shortAxis = toolRadius;
        } else {
            longAxis = std::abs(1 / pc[1]) - std::sqrt(std::pow(1 / pc[1], 2) - std::pow(toolRadius, 2));
            if (longAxis > toolthickness) longAxis = toolthickness;
            shortAxis = std::abs(1 / pc[0]) - std::sqrt(std::pow(1 / pc[0], 2) - std::pow(toolRadius, 2));
            if (shortAxis > toolthickness) shortAxis = toolthickness;
        }
        bool have = false;
        float best[3] = {NAN, NAN, NAN};
        for (float angle(0.0); angle <= 360.0; angle += 0.5) {
            const float rad = angle * 0.017453293f; /* pcl::deg2rad(float) */
            const float ex = (float)(longAxis * std::cos(rad));
            const float ey = (float)(shortAxis * std::sin(rad));
            /* pcl::transformPointCloud, SSE form: c0*x + (c1*y + (c2*z + c3)), z = 0 */
            float t[3];
            for (int i = 0; i < 3; ++i) t[i] = T[i][0] * ex + (T[i][1] * ey + (T[i][2] * 0.f + T[i][3]));
            /* std::max_element / std::min_element on x: first extremum, NaN never wins */
            if (!have) { best[0] = t[0]; best[1] = t[1]; best[2] = t[2]; have = true; }
            else if (key == 1 ? (best[0] < t[0]) : (t[0] < best[0])) { best[0] = t[0]; best[1] = t[1]; best[2] = t[2]; }
        }
        bound[0] = best[0]; bound[1] = best[1]; bound[2] = best[2];
    }

    /* compute_boundary (path_dynamic_alg.cpp:183-235); returns 0/1 */
    int compute_boundary(const Spline &path, Spline &boundary, int key)
    {
        double miny = path.small_y, maxy = path.big_y;
        std::map<double, std::array<double, 2>> boundary_node;
        double point[3] = {0, 0, 0};
        float bp[3];
        double dy = miny + 2;
        while (dy < maxy - 2) {
            path.point(dy, point);
            dy += P.tool_radius / 4;
            area2cloud(point, key, bp);
            if (std::isnan(bp[0])) continue;
            boundary_node[bp[1]] = {bp[0], bp[2]};
        }
        area2cloud(point, key, bp); /* "last point": the stale node again (App. B.9) */
        if (!std::isnan(bp[1])) boundary_node[bp[1]] = {bp[0], bp[2]}; /* a NaN key would corrupt the std::map */
        int node_number = (int)boundary_node.size();
        if (node_number <= 2) return 0;
        Spline b;
        b.y.resize(node_number + 2); b.x.resize(node_number + 2); b.z.resize(node_number + 2);
        int index = 0;
        for (auto &kv : boundary_node) { index++; b.x[index] = kv.second[0]; b.y[index] = kv.first; b.z[index] = kv.second[1]; }
        b.x[0] = b.x[1]; b.y[0] = b.y[1] - 20; b.z[0] = b.z[1];
        b.x[index + 1] = b.x[index]; b.y[index + 1] = b.y[index] + 20; b.z[index + 1] = b.z[index];
        b.fit();
        boundary = b;
        return 1;
    }

    /* bisection (path_dynamic_alg.cpp:237-265) */
    void bisection(double node[3], const Spline &boundary, int itr, int key)
    {
        if (itr > 5) return;
        float ab[3];
        area2cloud(node, key == 0 ? 1 : 0, ab);
        if (ab[1] < boundary.small_y || ab[1] > boundary.big_y) return;
        if (!boundary.yx.in_domain((double)ab[1])) return; /* (a NaN y passes this test and GSL's own: the spline then returns NaN) */
        double bpnt[3];
        boundary.point((double)ab[1], bpnt);
        double norm0 = (double)ab[0] - bpnt[0];
        if (std::fabs(norm0) < P.adjust_threshold) return;   /* norm.norm() of (n, 0, 0) */
        node[0] = node[0] - norm0;
        if (std::isnan(node[0])) return;
        bisection(node, boundary, itr + 1, key);
    }

    /* dynamic_adjust_path (path_dynamic_alg.cpp:267-306; v1: Path_Generation.cpp:585-634, whose loop
       leaves both end samples out); returns < 0 where the reference aborts */
    int dynamic_adjust_path(Spline &origin, const Spline &boundary, int key, bool v1 = false)
    {
        std::map<double, std::array<double, 2>> new_path;
        double miny = origin.small_y, maxy = origin.big_y, dy = 0;
        int NumOfNode = (int)((maxy - miny) / 5);
        ensure_tree();
        for (int i = v1 ? 1 : 0; v1 ? i < NumOfNode : i <= NumOfNode; i++) {
            dy = ((maxy - miny) / NumOfNode * i) + miny;
            /* B.13: for i == NumOfNode the rounded product can land an ulp beyond the last knot, where
               gsl_spline_eval raises GSL_EDOM and the reference aborts; evaluate at the last knot */
            if (dy > maxy) dy = maxy;
            if (!origin.yx.in_domain(dy)) return -1; /* NumOfNode == 0: 0/0 */
            double node[3];
            origin.point(dy, node);
            bisection(node, boundary, 0, key);
            const float q[3] = {(float)node[0], (float)node[1], (float)node[2]};
            /* B.14: a NaN Area2Cloud (negative curvature under a square root) makes the whole node NaN ("adjust path node
               NAN" is printed, :258-261) and nearestKSearch is then handed a NaN point; what FLANN leaves in pointIdx is
               the previous sample's result at best.  Defined behaviour here and in the product: such a sample adds no
               knot. */
            if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
            int id = tree.nearest(q); /* nearestKSearch(point, 3): only pointIdx[0] is used */
            if (id < 0) return -1;
            const Pt &c = cloud[id];
            new_path[(double)c.y] = {(double)c.x, (double)c.z};
        }
        if (new_path.size() < 3) return -1; /* gsl_spline_alloc */
        origin.y.clear(); origin.x.clear(); origin.z.clear();
        for (auto &kv : new_path) { origin.y.push_back(kv.first); origin.x.push_back(kv.second[0]); origin.z.push_back(kv.second[1]); }
        origin.fit();
        return 0;
    }

    /* the slice-to-slice chains of GenPath with Adjust = true */
    int adjust_all()
    {
        int S = (int)path_set.size();
        boundary_set.assign(S, Spline());
        auto chain = [&](int from, int to, int dir, int key) -> int {
            Spline boundary;
            bool have_boundary = false;
            for (int s = from; s != to; s += dir) {
                const Spline &pre = path_set[s - dir];
                if (compute_boundary(pre, boundary, key)) { have_boundary = true; boundary_set[s] = boundary; } /* drawpath(*boundary, 0,255,0), :321-322 */
                if (!have_boundary) return -(1 + s); /* the reference would use an unconstructed Spline */
                if (dynamic_adjust_path(path_set[s], boundary, key) < 0) return -(1 + s);
            }
            return 0;
        };
        if (P.walk == PPO_WALK_CENTER_INT) { /* thread_worker left / right of the centre path */
            int c = 0;
            { /* index of the centre slice = number of front slices */
                float mn[3], mx[3];
                minmax(mn, mx);
                int imin = (int)mn[0], imax = (int)mx[0], step = int(P.tool_radius * 2);
                int cc = (imax + imin) / 2, loc = cc - step;
                while (imax > loc && loc > imin) { c++; loc -= step; }
            }
            int rc = chain(c - 1, -1, -1, 0);
            if (rc) return rc;
            return chain(c + 1, S, +1, 1);
        }
        if (P.walk == PPO_WALK_SDIR_INT) return chain(1, S, +1, 1);
        if (P.walk == PPO_WALK_V1_CONTACT) {
            /* Contact_Path_Generation (Path_Generation.cpp:711-725): every path but the first is adjusted
               against the boundary (Area2Cloud key 0 = max x) of its already adjusted predecessor; a
               predecessor without a boundary ("generate boundary fail", :589-592) leaves the path as is */
            for (int s = 1; s < S; ++s) {
                Spline boundary;
                if (!compute_boundary(path_set[s - 1], boundary, 1)) continue;
                boundary_set[s] = boundary;
                if (dynamic_adjust_path(path_set[s], boundary, 1, true) < 0) return -(1 + s);
            }
            return 0;
        }
        return 0; /* SectPath::GenPath and slicing_method have no adjustment */
    }

    /* SectPath::remove_outlier (path_slicing_alg.cpp:101-108): pcl::StatisticalOutlierRemoval, setMeanK(50),
       setStddevMulThresh(1.0), filter(*cloud) -- PCL 1.12 filters/impl/statistical_outlier_removal.hpp, applyFilterIndices.
       Returns the new size, or -1 where PCL reads past its neighbour vectors (fewer than mean_k + 1 finite points). */
    int remove_outlier(int mean_k, double std_mul)
    {
        rebuild_tree();
        const size_t n = cloud.size();
        std::vector<float> distances(n, 0.f);
        int valid = 0;
        std::vector<std::pair<float, int>> nb;
        for (size_t i = 0; i < n; ++i) {
            const Pt &q = cloud[i];
            if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) { distances[i] = 0.f; continue; }
            tree.knn(&q.x, mean_k + 1, nb);
            if ((int)nb.size() < mean_k + 1) return -1;
            double dist_sum = 0.0;
            for (int k = 1; k < mean_k + 1; ++k) dist_sum += std::sqrt(nb[k].first); /* float sqrt, double sum */
            distances[i] = static_cast<float>(dist_sum / mean_k);
            valid++;
        }
        double sum = 0, sq_sum = 0;
        for (const float &distance : distances) { sum += distance; sq_sum += distance * distance; } /* float product */
        double mean = sum / static_cast<double>(valid);
        double variance = (sq_sum - sum * sum / static_cast<double>(valid)) / (static_cast<double>(valid) - 1);
        double stddev = std::sqrt(variance);
        double distance_threshold = mean + std_mul * stddev;
        std::vector<Pt> kept;
        kept.reserve(n);
        for (size_t i = 0; i < n; ++i)
            if (!(distances[i] > distance_threshold)) kept.push_back(cloud[i]); /* non-finite points carry 0 and stay */
        sor_threshold = distance_threshold;
        sor_distances.swap(distances);
        cloud.swap(kept);
        tree_built = false;
        normals.clear(); normal_done.clear();
        path_set.clear(); slice_idx.clear();
        return (int)cloud.size();
    }
    double sor_threshold = 0;
    std::vector<float> sor_distances;

    /* path_generater::voxel_down (Path_Generation.cpp:53-59): pcl::VoxelGrid<pcl::PointXYZRGB>, setLeafSize(x, y, z),
       filter(*cloud) -- PCL filters/impl/voxel_grid.hpp applyFilter with the class defaults (no filter field,
       downsample_all_data_ = true -> CentroidPoint, min_points_per_voxel_ = 0).  PCL orders the (voxel id, point index)
       pairs by id alone with an unstable sort (std::sort in 1.10, boost integer_sort from 1.11), so the order of the float
       additions inside a voxel is an artefact of that sort; this restatement adds in ascending point index.
       Returns the new size; *overflow = 1 and the cloud unchanged where PCL warns that the indices would overflow. */
    int voxel_down(float lx, float ly, float lz, int *overflow)
    {
        *overflow = 0;
        const float inv[3] = {1.0f / lx, 1.0f / ly, 1.0f / lz}; /* inverse_leaf_size_ = Array4f::Ones() / leaf_size_ */
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        size_t finite = 0;
        for (const Pt &p : cloud) { /* getMinMax3D, skipping non-finite points */
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            const float c[3] = {p.x, p.y, p.z};
            for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
            ++finite;
        }
        if (cloud.empty()) return 0;
        if (!finite) { cloud.clear(); invalidate(); return 0; }
        long long dxyz = 1, cells = 1;
        int min_b[3], div_b[3];
        for (int d = 0; d < 3; ++d) {
            dxyz *= static_cast<long long>((mx[d] - mn[d]) * inv[d]) + 1;
            min_b[d] = static_cast<int>(std::floor(mn[d] * inv[d]));
            const int max_b = static_cast<int>(std::floor(mx[d] * inv[d]));
            div_b[d] = max_b - min_b[d] + 1;
            cells *= div_b[d];
            /* PCL tests dx*dy*dz only; div_b can be one larger per axis, and its product is what the int id is built from */
            if (dxyz > INT_MAX || cells > INT_MAX || dxyz <= 0 || cells <= 0) { *overflow = 1; return (int)cloud.size(); }
        }
        const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
        std::vector<std::pair<unsigned, unsigned>> iv;
        iv.reserve(finite);
        for (size_t i = 0; i < cloud.size(); ++i) {
            const Pt &p = cloud[i];
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            const int i0 = static_cast<int>(std::floor(p.x * inv[0]) - static_cast<float>(min_b[0]));
            const int i1 = static_cast<int>(std::floor(p.y * inv[1]) - static_cast<float>(min_b[1]));
            const int i2 = static_cast<int>(std::floor(p.z * inv[2]) - static_cast<float>(min_b[2]));
            iv.emplace_back(static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i);
        }
        std::stable_sort(iv.begin(), iv.end(), [](const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) { return a.first < b.first; });
        std::vector<Pt> out;
        for (size_t a = 0; a < iv.size();) {
            size_t b = a;
            float sx = 0.f, sy = 0.f, sz = 0.f; /* AccumulatorXYZ: Eigen::Vector3f xyz += point; get: xyz / n */
            while (b < iv.size() && iv[b].first == iv[a].first) { const Pt &p = cloud[iv[b].second]; sx += p.x; sy += p.y; sz += p.z; ++b; }
            const float c = static_cast<float>(b - a);
            Pt q = cloud[iv[a].second];
            q.x = sx / c; q.y = sy / c; q.z = sz / c;
            out.push_back(q);
            a = b;
        }
        cloud.swap(out);
        invalidate();
        return (int)cloud.size();
    }
    /* SectPath::smooth (path_slicing_alg.cpp:111-139; v1 Path_Generation.cpp:340-360): pcl::MovingLeastSquares with
       setPolynomialOrder(3), setSearchRadius(15), the default SIMPLE projection and no upsampling; the result replaces
       the cloud (copyPointCloud).  PCL surface/include/pcl/surface/impl/mls.hpp (1.10 - 1.12): process ->
       performProcessing -> computeMLSPointNormal -> MLSResult::computeMLSSurface + projectQueryPoint(SIMPLE).
       Points with fewer than 3 neighbours in the radius (and non-finite points, whose search returns nothing) are not
       in the output.  All arithmetic in double on the float coordinates.  Returns the new size. */
    int smooth_mls(double search_radius, int order)
    {
        rebuild_tree();
        const int nr_coeff = (order + 1) * (order + 2) / 2;
        const double sqr_gauss = search_radius * search_radius;
        std::vector<Pt> out;
        out.reserve(cloud.size());
        std::vector<std::pair<float, int>> nb;
        std::vector<double> P, A, wv, fv;
        for (size_t i = 0; i < cloud.size(); ++i) {
            const Pt &qp = cloud[i];
            if (!std::isfinite(qp.x) || !std::isfinite(qp.y) || !std::isfinite(qp.z)) continue;
            tree.radius(&qp.x, (float)search_radius, nb); /* sorted by distance; the point itself first */
            if (nb.size() < 3) continue;
            const int nn = (int)nb.size();
            /* computeMeanAndCovarianceMatrix<PointT, double> (PCL 1.12: shifted by the first neighbour) */
            const Pt &K0 = cloud[nb[0].second];
            const double K[3] = {K0.x, K0.y, K0.z};
            double accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (auto &e : nb) {
                const Pt &c = cloud[e.second];
                const double x = c.x - K[0], y = c.y - K[1], z = c.z - K[2];
                accu[0] += x * x; accu[1] += x * y; accu[2] += x * z;
                accu[3] += y * y; accu[4] += y * z; accu[5] += z * z;
                accu[6] += x; accu[7] += y; accu[8] += z;
            }
            for (int k = 0; k < 9; ++k) accu[k] /= (double)nn;
            const double centroid[3] = {accu[6] + K[0], accu[7] + K[1], accu[8] + K[2]};
            double cov[9];
            cov[0] = accu[0] - accu[6] * accu[6];
            cov[1] = accu[1] - accu[6] * accu[7];
            cov[2] = accu[2] - accu[6] * accu[8];
            cov[4] = accu[3] - accu[7] * accu[7];
            cov[5] = accu[4] - accu[7] * accu[8];
            cov[8] = accu[5] - accu[8] * accu[8];
            cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
            double ev, n[3];
            eigen33_smallest_d(cov, &ev, n);
            const double q[3] = {qp.x, qp.y, qp.z};
            Pt o = qp;
            if (!std::isfinite(n[0]) || !std::isfinite(n[1]) || !std::isfinite(n[2])) { out.push_back(o); continue; } /* mean = query_point */
            const double d4 = -1 * (n[0] * centroid[0] + n[1] * centroid[1] + n[2] * centroid[2]);
            const double distance = (q[0] * n[0] + q[1] * n[1] + q[2] * n[2]) + d4;
            const double mean[3] = {q[0] - distance * n[0], q[1] - distance * n[1], q[2] - distance * n[2]};
            double res[3] = {mean[0], mean[1], mean[2]};
            if (order > 1 && nn >= nr_coeff) {
                /* v_axis = plane_normal.unitOrthogonal(); u_axis = plane_normal.cross(v_axis)  (Eigen OrthoMethods.h) */
                double va[3], ua[3];
                const double prec = 1e-12; /* NumTraits<double>::dummy_precision() */
                if (!(std::abs(n[0]) <= std::abs(n[2]) * prec) || !(std::abs(n[1]) <= std::abs(n[2]) * prec)) {
                    const double invnm = 1.0 / std::sqrt(n[0] * n[0] + n[1] * n[1]);
                    va[0] = -n[1] * invnm; va[1] = n[0] * invnm; va[2] = 0;
                } else {
                    const double invnm = 1.0 / std::sqrt(n[1] * n[1] + n[2] * n[2]);
                    va[0] = 0; va[1] = -n[2] * invnm; va[2] = n[1] * invnm;
                }
                ua[0] = n[1] * va[2] - n[2] * va[1];
                ua[1] = n[2] * va[0] - n[0] * va[2];
                ua[2] = n[0] * va[1] - n[1] * va[0];
                P.assign((size_t)nr_coeff * nn, 0.0); wv.assign(nn, 0.0); fv.assign(nn, 0.0);
                for (int ni = 0; ni < nn; ++ni) {
                    const Pt &c = cloud[nb[ni].second];
                    const double dm[3] = {c.x - mean[0], c.y - mean[1], c.z - mean[2]};
                    wv[ni] = std::exp(-(dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2]) / sqr_gauss); /* computeMLSWeight */
                    const double u_coord = dm[0] * ua[0] + dm[1] * ua[1] + dm[2] * ua[2];
                    const double v_coord = dm[0] * va[0] + dm[1] * va[1] + dm[2] * va[2];
                    fv[ni] = dm[0] * n[0] + dm[1] * n[1] + dm[2] * n[2];
                    int j = 0;
                    double u_pow = 1;
                    for (int ui = 0; ui <= order; ++ui) {
                        double v_pow = 1;
                        for (int vi = 0; vi <= order - ui; ++vi) { P[(size_t)(j++) * nn + ni] = u_pow * v_pow; v_pow *= v_coord; }
                        u_pow *= u_coord;
                    }
                }
                /* P_weight = P * w.asDiagonal(); P_weight_Pt = P_weight * P^T; c_vec = P_weight * f_vec
                   (Eigen's blocked products group the additions differently: last-bit differences in double) */
                A.assign((size_t)nr_coeff * nr_coeff, 0.0);
                std::vector<double> c(nr_coeff, 0.0);
                for (int a = 0; a < nr_coeff; ++a) {
                    for (int b = 0; b < nr_coeff; ++b) {
                        double sum = 0;
                        for (int ni = 0; ni < nn; ++ni) sum += (P[(size_t)a * nn + ni] * wv[ni]) * P[(size_t)b * nn + ni];
                        A[(size_t)a * nr_coeff + b] = sum;
                    }
                    double sum = 0;
                    for (int ni = 0; ni < nn; ++ni) sum += (P[(size_t)a * nn + ni] * wv[ni]) * fv[ni];
                    c[a] = sum;
                }
                /* P_weight_Pt.llt().solveInPlace(c_vec): Eigen's unblocked LLT (size < 32) on the lower triangle; on a
                   non-positive pivot it stops and the solve runs on what is there (Eigen does not check info()) */
                const int N = nr_coeff;
                for (int k = 0; k < N; ++k) {
                    double x = A[(size_t)k * N + k];
                    for (int j = 0; j < k; ++j) x -= A[(size_t)k * N + j] * A[(size_t)k * N + j];
                    if (x <= 0.0) break;
                    A[(size_t)k * N + k] = x = std::sqrt(x);
                    for (int r = k + 1; r < N; ++r) {
                        double t = A[(size_t)r * N + k];
                        for (int j = 0; j < k; ++j) t -= A[(size_t)r * N + j] * A[(size_t)k * N + j];
                        A[(size_t)r * N + k] = t / x;
                    }
                }
                for (int r = 0; r < N; ++r) { /* L y = c */
                    double t = c[r];
                    for (int j = 0; j < r; ++j) t -= A[(size_t)r * N + j] * c[j];
                    c[r] = t / A[(size_t)r * N + r];
                }
                for (int r = N - 1; r >= 0; --r) { /* L^T x = y */
                    double t = c[r];
                    for (int j = r + 1; j < N; ++j) t -= A[(size_t)j * N + r] * c[j];
                    c[r] = t / A[(size_t)r * N + r];
                }
                if (std::isfinite(c[0])) { /* projectQueryPoint, SIMPLE: mean + c_vec[0] * plane_normal */
                    res[0] = mean[0] + c[0] * n[0]; res[1] = mean[1] + c[0] * n[1]; res[2] = mean[2] + c[0] * n[2];
                }
            }
            o.x = static_cast<float>(res[0]); o.y = static_cast<float>(res[1]); o.z = static_cast<float>(res[2]);
            out.push_back(o);
        }
        cloud.swap(out);
        invalidate();
        return (int)cloud.size();
    }
    /* SectPath::trans2center (path_slicing_alg.cpp:82-99): pcl::compute3DCentroid and pcl::computeCovarianceMatrix
       (float accumulators in cloud order, non-finite points skipped), Eigen::EigenSolver<Matrix3f> eigenvectors,
       TransAlign = [V^T | -V^T c], pcl::transformPointCloud(*cloud, *cloud, TransAlign).
       Returns 0, -1 without finite points, -2 when the float Schur form keeps a complex pair / does not converge. */
    float TA[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    bool aligned = false;
    float centroid3[3] = {0, 0, 0}, cov33[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int trans2center()
    {
        float c[3] = {0.f, 0.f, 0.f};
        size_t cp = 0;
        for (const Pt &p : cloud) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            c[0] += p.x; c[1] += p.y; c[2] += p.z;
            ++cp;
        }
        if (!cp) return -1;
        for (int d = 0; d < 3; ++d) c[d] /= static_cast<float>(cp);
        float cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (const Pt &p : cloud) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            float pt[3] = {p.x - c[0], p.y - c[1], p.z - c[2]};
            cov[1][1] += pt[1] * pt[1];
            cov[1][2] += pt[1] * pt[2];
            cov[2][2] += pt[2] * pt[2];
            const float x = pt[0];
            pt[0] *= x; pt[1] *= x; pt[2] *= x; /* pt *= pt.x () */
            cov[0][0] += pt[0];
            cov[0][1] += pt[1];
            cov[0][2] += pt[2];
        }
        cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
        for (int d = 0; d < 3; ++d) { centroid3[d] = c[d]; for (int e = 0; e < 3; ++e) cov33[d][e] = cov[d][e]; }
        EigenSolver3f es;
        es.compute(cov);
        if (es.complex_pair || !es.converged) return -2;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) TA[i][j] = es.U[j][i];                                  /* eigen_vectors.transpose() */
            TA[i][3] = ((-es.U[0][i]) * c[0] + (-es.U[1][i]) * c[1]) + (-es.U[2][i]) * c[2];       /* -V^T * centroid */
        }
        TA[3][0] = TA[3][1] = TA[3][2] = 0.f; TA[3][3] = 1.f;
        for (Pt &p : cloud) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            float o[3];
            transform_se3(TA, &p.x, o);
            p.x = o[0]; p.y = o[1]; p.z = o[2];
        }
        aligned = true;
        invalidate();
        return 0;
    }
    void invalidate()
    {
        tree_built = false;
        normals.clear(); normal_done.clear();
        path_set.clear(); slice_idx.clear();
    }

    int gen_path()
    {
        path_set.clear();
        slice_idx.clear();
        bool derived = (P.walk == PPO_WALK_CENTER_INT || P.walk == PPO_WALK_SDIR_INT);
        if (P.reference_complexity) {
            /* path_dynamic_alg.cpp:343-344 / path_slicing_alg.cpp:295 */
            if (derived) estimate_normal_all();
            rebuild_tree();
        }
        px = slice_positions();
        int S = (int)px.size();
        if (!P.reference_complexity) bucket_slices(px, slice_idx);
        else slice_idx.resize(S);
        path_set.resize(S);
        int first_bad = S; /* the reference stops at the first slice that aborts */
        const int nt = P.threads > 1 ? P.threads : 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) if (nt > 1)
        for (int s = 0; s < S; ++s) {
            /* OnePath / path_track: rangedX_index(int(plane_point[0])) */
            if (P.reference_complexity) slice_idx[s] = ranged_x_index((int)px[s]);
            std::map<double, std::array<double, 2>> Node;
            int m = insert_point(slice_idx[s], px[s], Node);
            if (m < 3) { /* gsl_spline_alloc / FLANN would abort */
#pragma omp critical
                first_bad = std::min(first_bad, s);
                continue;
            }
            Spline &sp = path_set[s];
            for (auto &kv : Node) { sp.y.push_back(kv.first); sp.x.push_back(kv.second[0]); sp.z.push_back(kv.second[1]); }
            sp.fit();
        }
        if (first_bad < S) return -(1 + first_bad);
        if (P.dynamic_adjustment) {
            int rc = adjust_all();
            if (rc) return rc;
        }
        return S;
    }

    int get_path()
    {
        wp.clear(); wp_pre.clear(); wp_smooth.clear(); wp_xyz.clear(); wp_normal.clear(); wp_nn.clear(); tail.clear();
        sweeps = 0; oob = 0;
        /* path_translation_alg.cpp:149-150 */
        std::vector<const Spline *> kept;
        int S = (int)path_set.size();
        int first = P.drop_ends ? 1 : 0, last = P.drop_ends ? S - 1 : S;
        for (int s = first; s < last; ++s) kept.push_back(&path_set[s]);
        /* :146, :156-169: invTransAlign = TransAlign.inverse() (identity unless trans2center ran) */
        float inv[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        if (aligned) inverse_4x4(TA, inv);
        std::vector<std::vector<std::array<float, 3>>> lists;
        int flag = 1;
        for (const Spline *path : kept) {
            std::vector<std::array<float, 3>> one;
            double dy = path->small_y + P.trim;
            while (dy < path->big_y - P.trim) {
                double p[3];
                path->point(dy, p);
                std::array<float, 3> q = {(float)p[0], (float)p[1], (float)p[2]};
                if (aligned) { /* wayPointXYZ = invTransAlign * wayPointXYZ (Matrix4f * Vector4f, w = 1) */
                    std::array<float, 3> t;
                    for (int r = 0; r < 3; ++r) t[r] = ((inv[r][0] * q[0] + inv[r][1] * q[1]) + inv[r][2] * q[2]) + inv[r][3] * 1.f;
                    q = t;
                }
                one.push_back(q);
                dy += P.path_resolution;
            }
            if (flag == -1) std::reverse(one.begin(), one.end());
            lists.push_back(one);
            flag *= -1;
        }
        /* :171-174: the cloud goes back through invTransAlign, the search tree and the normals are rebuilt on it */
        std::vector<Pt> aligned_cloud;
        if (aligned) {
            aligned_cloud = cloud;
            for (Pt &p : cloud) {
                if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
                float o[3];
                transform_se3(inv, &p.x, o);
                p.x = o[0]; p.y = o[1]; p.z = o[2];
            }
            tree_built = false; normals.clear(); normal_done.clear();
        }
        if (P.reference_complexity) { rebuild_tree(); estimate_normal_all(); }
        else ensure_tree();
        for (auto &one : lists) {
            for (auto &q : one) {
                int id = tree.nearest(q.data());
                const float *N = P.reference_complexity ? &normals[4 * (size_t)id] : normal_lazy(id);
                float rpy[3];
                pose_from_normal(N, rpy);
                std::array<float, 6> w;
                if (P.change_range) w = {q[0] / 1000, q[1] / 1000, q[2] / 1000, rpy[0], rpy[1], rpy[2]};
                else w = {q[0], q[1], q[2], rpy[0], rpy[1], rpy[2]};
                handeye_transform(P.handeye, w.data());
                wp.push_back(w);
                wp_xyz.push_back(q);
                wp_nn.push_back(id);
                wp_normal.push_back({N[0], N[1], N[2], N[3]});
            }
            tail.push_back((int)wp.size() - 1);
        }
        if (aligned) { /* the reference leaves the cloud in the sensor frame; this handle stays reusable */
            cloud.swap(aligned_cloud);
            tree_built = false; normals.clear(); normal_done.clear();
        }
        wp_pre = wp;
        if (P.smooth && wp.size() > 0) sweeps = position_smooth(wp, P.smooth_max_sweeps);
        wp_smooth = wp;
        reduce_rpy(wp, tail, P.rpy_resolution, &oob);
        trans_flange(wp, P.ee_length);
        return (int)wp.size();
    }
};

/* ====================================================================== */
extern "C" {

void ppo_default_params(ppo_params *p)
{   /* config.txt:1-13 and Path_Generate_Algorithm.h:43-48 */
    memset(p, 0, sizeof(*p));
    p->tool_radius = 12; p->path_resolution = 7; p->rpy_resolution = 7; p->ee_length = 0.3f;
    p->change_range = 1; p->pairing = PPO_PAIR_KD; p->walk = PPO_WALK_CENTER_INT;
    p->trim = 10; p->drop_ends = 1; p->smooth = 1;
    const float he[6] = {-0.764091f, 0.025886f, 0.663790f, -3.1270175f, -0.040124f, -1.6063578f};
    memcpy(p->handeye, he, sizeof(he));
    p->normal_radius = 2.5f;
    p->reference_complexity = 0;
    p->smooth_max_sweeps = 32;
    p->threads = 1;
    p->dynamic_adjustment = 0; /* config.txt says true; the benchmarks of this round run without */
    p->depth = 0.01; p->adjust_threshold = 1; p->toolthickness = 10; p->curvature_k = 50;
}

ppo_handle *ppo_create(const float *xyz, size_t n, size_t stride, const ppo_params *p)
{
    ppo_handle *h = new ppo_handle();
    h->P = *p;
    h->cloud.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Pt &q = h->cloud[i];
        q.x = xyz[i * stride + 0]; q.y = xyz[i * stride + 1]; q.z = xyz[i * stride + 2];
        q.pad0 = 1.f; q.rgba = 0x00ffffffu; q.pad1[0] = q.pad1[1] = q.pad1[2] = 0.f;
        if (p->change_range) { q.x *= 1000; q.y *= 1000; q.z *= 1000; } /* path_slicing_alg.cpp:19-23 */
    }
    return h;
}
void ppo_destroy(ppo_handle *h) { delete h; }
size_t ppo_num_points(const ppo_handle *h) { return h->cloud.size(); }
void ppo_get_points(const ppo_handle *h, float *o)
{
    for (size_t i = 0; i < h->cloud.size(); ++i) { o[3 * i] = h->cloud[i].x; o[3 * i + 1] = h->cloud[i].y; o[3 * i + 2] = h->cloud[i].z; }
}
void ppo_minmax(const ppo_handle *h, float mn[3], float mx[3]) { h->minmax(mn, mx); }

int ppo_slice_positions(const ppo_handle *h, float *px, int cap)
{
    std::vector<float> v = h->slice_positions();
    for (int i = 0; i < (int)v.size() && i < cap; ++i) px[i] = v[i];
    return (int)v.size();
}
int ppo_ranged_x_index(const ppo_handle *h, int position, int *out, int cap)
{
    std::vector<int> v = h->ranged_x_index(position);
    for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
int ppo_insert_point(ppo_handle *h, const int *indices, int n, float plane_x, double *y, double *x, double *z, int cap)
{
    std::vector<int> idx(indices, indices + n);
    std::map<double, std::array<double, 2>> Node;
    int m = h->insert_point(idx, plane_x, Node);
    if (m < 0) return m;
    int k = 0;
    for (auto &kv : Node) {
        if (k < cap) { y[k] = kv.first; x[k] = kv.second[0]; z[k] = kv.second[1]; }
        ++k;
    }
    return m;
}
int ppo_gen_path(ppo_handle *h) { return h->gen_path(); }
int ppo_num_slices(const ppo_handle *h) { return (int)h->path_set.size(); }
int ppo_get_nodes(const ppo_handle *h, int s, double *y, double *x, double *z, int cap)
{
    const Spline &sp = h->path_set[s];
    int m = (int)sp.y.size();
    for (int i = 0; i < m && i < cap; ++i) { y[i] = sp.y[i]; x[i] = sp.x[i]; z[i] = sp.z[i]; }
    return m;
}
/* the boundary spline slice s was adjusted against (the curve thread_worker paints green before it adjusts the slice); 0 knots: none */
int ppo_get_boundary(const ppo_handle *h, int s, double *y, double *x, double *z, int cap)
{
    if (s < 0 || s >= (int)h->boundary_set.size()) return 0;
    const Spline &sp = h->boundary_set[s];
    int m = (int)sp.y.size();
    for (int i = 0; i < m && i < cap; ++i) { y[i] = sp.y[i]; x[i] = sp.x[i]; z[i] = sp.z[i]; }
    return m;
}
int ppo_get_slice_indices(const ppo_handle *h, int s, int *out, int cap)
{
    const std::vector<int> &v = h->slice_idx[s];
    for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
int ppo_eval_spline(const ppo_handle *h, int s, const double *y, int k, double *xyz)
{
    const Spline &sp = h->path_set[s];
    int rc = 0;
    for (int i = 0; i < k; ++i) {
        if (!sp.yx.in_domain(y[i])) { xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = NAN; rc = -1; continue; }
        sp.point(y[i], &xyz[3 * i]);
    }
    return rc;
}
int ppo_get_path(ppo_handle *h) { return h->get_path(); }
int ppo_num_waypoints(const ppo_handle *h) { return (int)h->wp.size(); }
static void copy6(const std::vector<std::array<float, 6>> &v, float *o) { for (size_t i = 0; i < v.size(); ++i) memcpy(o + 6 * i, v[i].data(), 24); }
void ppo_get_waypoints(const ppo_handle *h, float *o) { copy6(h->wp, o); }
void ppo_get_waypoints_presmooth(const ppo_handle *h, float *o) { copy6(h->wp_pre, o); }
void ppo_get_waypoints_smoothed(const ppo_handle *h, float *o) { copy6(h->wp_smooth, o); }
int ppo_get_tail_index(const ppo_handle *h, int *tail, int cap)
{
    for (int i = 0; i < (int)h->tail.size() && i < cap; ++i) tail[i] = h->tail[i];
    return (int)h->tail.size();
}
void ppo_get_waypoints_xyz(const ppo_handle *h, float *o) { for (size_t i = 0; i < h->wp_xyz.size(); ++i) memcpy(o + 3 * i, h->wp_xyz[i].data(), 12); }
void ppo_get_waypoint_nn(const ppo_handle *h, int *nn) { for (size_t i = 0; i < h->wp_nn.size(); ++i) nn[i] = h->wp_nn[i]; }
void ppo_get_waypoint_normals(const ppo_handle *h, float *o) { for (size_t i = 0; i < h->wp_normal.size(); ++i) memcpy(o + 4 * i, h->wp_normal[i].data(), 16); }
int ppo_smooth_sweeps(const ppo_handle *h) { return h->sweeps; }
int ppo_rpy_oob(const ppo_handle *h) { return h->oob; }

void ppo_estimate_normals(ppo_handle *h, float *n4)
{
    h->estimate_normal_all();
    memcpy(n4, h->normals.data(), h->normals.size() * sizeof(float));
}
void ppo_normal_at(ppo_handle *h, int idx, float n4[4]) { h->point_normal(idx, n4); }
int ppo_remove_outlier(ppo_handle *h, int mean_k, double std_mul, double *threshold, float *distances)
{
    const size_t n0 = h->cloud.size();
    int rc = h->remove_outlier(mean_k, std_mul);
    if (rc >= 0) {
        if (threshold) *threshold = h->sor_threshold;
        if (distances) memcpy(distances, h->sor_distances.data(), n0 * sizeof(float));
    }
    return rc;
}

int ppo_voxel_down(ppo_handle *h, float lx, float ly, float lz, int *overflow) { return h->voxel_down(lx, ly, lz, overflow); }

int ppo_smooth_mls(ppo_handle *h, double radius, int order) { return h->smooth_mls(radius, order); }
int ppo_trans2center(ppo_handle *h, float T16[16], float centroid[3], float cov9[9])
{
    int rc = h->trans2center();
    if (T16) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T16[4 * i + j] = h->TA[i][j];
    if (centroid) for (int d = 0; d < 3; ++d) centroid[d] = h->centroid3[d];
    if (cov9) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov9[3 * i + j] = h->cov33[i][j];
    return rc;
}
int ppo_eigensolver3f(const float A9[9], float evals[3], float evecs9[9])
{
    float A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = A9[3 * i + j];
    EigenSolver3f es;
    es.compute(A);
    for (int i = 0; i < 3; ++i) { evals[i] = es.ev[i]; for (int j = 0; j < 3; ++j) evecs9[3 * i + j] = es.U[i][j]; }
    return es.complex_pair ? -2 : (es.converged ? 0 : -3);
}
int ppo_knn(ppo_handle *h, const float q[3], int k, int *out)
{
    h->ensure_tree();
    std::vector<std::pair<float, int>> nb;
    h->tree.knn(q, k, nb);
    for (size_t i = 0; i < nb.size(); ++i) out[i] = nb[i].second;
    return (int)nb.size();
}
void ppo_principal_curvature(ppo_handle *h, const float q[3], float out[5])
{
    h->ensure_tree();
    std::vector<std::pair<float, int>> nb;
    h->tree.knn(q, h->P.curvature_k, nb);
    h->principal_curvatures(nb[0].second, nb, out);
}
void ppo_area2cloud(ppo_handle *h, const double p[3], int key, float out[3]) { h->area2cloud(p, key, out); }
int ppo_nearest(ppo_handle *h, const float q[3], float *d2) { h->ensure_tree(); return h->tree.nearest(q, d2); }
int ppo_radius_search(ppo_handle *h, const float q[3], float r, int *out, int cap)
{
    h->ensure_tree();
    std::vector<std::pair<float, int>> nb;
    h->tree.radius(q, r, nb);
    for (int i = 0; i < (int)nb.size() && i < cap; ++i) out[i] = nb[i].second;
    return (int)nb.size();
}

int ppo_steffen(int n, const double *xs, const double *ys, const double *xq, int k, double *out)
{
    if (n < 3) return -1; /* steffen min_size = 3 */
    for (int i = 1; i < n; ++i) if (!(xs[i] > xs[i - 1])) return -2;
    Steffen st; st.init(xs, ys, n);
    int rc = 0;
    for (int i = 0; i < k; ++i) {
        if (!st.in_domain(xq[i])) { out[i] = NAN; rc = -3; continue; }
        out[i] = st.eval(xq[i]);
    }
    return rc;
}
void ppo_eigen33(const float cov[9], float *ev, float vec[3]) { eigen33_smallest(cov, ev, vec); }
void ppo_euler_zyx(const float m[9], float e[3])
{
    float M[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = m[3 * i + j];
    euler_zyx(M, e);
}
void ppo_handeye(const float he[6], float wp[6]) { handeye_transform(he, wp); }
void ppo_pose_from_normal(const float n[3], float rpy[3]) { pose_from_normal(n, rpy); }
int ppo_position_smooth(float *wp6, int n, int max_sweeps)
{
    std::vector<std::array<float, 6>> v(n);
    for (int i = 0; i < n; ++i) memcpy(v[i].data(), wp6 + 6 * i, 24);
    int s = position_smooth(v, max_sweeps);
    for (int i = 0; i < n; ++i) memcpy(wp6 + 6 * i, v[i].data(), 24);
    return s;
}
int ppo_reduce_rpy(float *wp6, int n, const int *tail, int ntail, double rpy_res)
{
    std::vector<std::array<float, 6>> v(n);
    for (int i = 0; i < n; ++i) memcpy(v[i].data(), wp6 + 6 * i, 24);
    std::vector<int> t(tail, tail + ntail);
    int oob = 0;
    reduce_rpy(v, t, rpy_res, &oob);
    for (int i = 0; i < n; ++i) memcpy(wp6 + 6 * i, v[i].data(), 24);
    return oob;
}
void ppo_trans_flange(float *wp6, int n, float ee_len)
{
    std::vector<std::array<float, 6>> v(n);
    for (int i = 0; i < n; ++i) memcpy(v[i].data(), wp6 + 6 * i, 24);
    trans_flange(v, ee_len);
    for (int i = 0; i < n; ++i) memcpy(wp6 + 6 * i, v[i].data(), 24);
}

} /* extern "C" */
