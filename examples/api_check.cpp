// Class-surface check of the drop-in headers beyond what the CLIs call (tests/test_gpu_parity.py reads the output):
//   api_check cloud.pcd position  -> rangedX_index(position) through the v1 class (public there, Path_Generate.h:50),
//                                    estimate_normal() + the readable field, class Spline on caller-supplied knots
//                                    (Spline.h:10-42: constructor, point, miny/bigy, restart, copies by value).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "Path_Generate.h"

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: api_check cloud.pcd position\n"); return 2; }
    path_generater planner(argv[1], 6.0);
    const int position = std::atoi(argv[2]);
    std::vector<int> idx = planner.rangedX_index(position);
    long long sum = 0;
    bool ascending = true;
    for (size_t i = 0; i < idx.size(); ++i) { sum += idx[i]; if (i && idx[i] <= idx[i - 1]) ascending = false; }
    std::printf("ranged %d %zu %lld %d %d %d\n", position, idx.size(), sum, ascending ? 1 : 0, idx.empty() ? -1 : idx.front(), idx.empty() ? -1 : idx.back());

    planner.estimate_normal();
    const std::vector<float> &nrm = planner.cloud_normals();
    size_t nan_rows = 0;
    double acc = 0;
    for (size_t i = 0; i + 3 < nrm.size(); i += 4) { if (nrm[i] != nrm[i]) ++nan_rows; else acc += (double)nrm[i + 2] + 0.5 * (double)nrm[i + 3]; }
    std::printf("normals %zu %zu %.17g\n", nrm.size() / 4, nan_rows, acc);

    // class Spline on knots the caller supplies
    const int n = 9;
    double y[n], x[n], z[n];
    for (int i = 0; i < n; ++i) { y[i] = -3.0 + 1.25 * i + 0.01 * i * i; x[i] = 100.0 + 0.5 * i * (i % 3); z[i] = 1500.0 - 0.75 * i + (i % 2); }
    Spline sp(n, y, x, z);
    Spline copy = sp; // by value, like the reference's Path_set.push_back(Spline)
    std::printf("spline %d %.17g %.17g", copy.nodes(), copy.miny(), copy.bigy());
    for (int q = 0; q <= 16; ++q) {
        const double yq = q == 16 ? y[n - 1] : y[0] + (y[n - 1] - y[0]) * q / 16.0; /* (the last sum may round above bigy: GSL_EDOM) */
        Eigen::Vector3d p = copy.point(yq);
        std::printf(" %.17g %.17g %.17g", p[0], p[1], p[2]);
    }
    std::printf("\n");
    // restart on other knots (dynamic_adjust_path: path_dynamic_alg.cpp:297-303)
    double y2[4] = {0.0, 1.0, 2.5, 4.0}, x2[4] = {1.0, 3.0, 2.0, 5.0}, z2[4] = {0.0, -1.0, -1.5, 2.0};
    sp.restart(4, y2, x2, z2);
    std::printf("restart %d %.17g %.17g", sp.nodes(), sp.miny(), sp.bigy());
    for (int q = 0; q <= 8; ++q) { Eigen::Vector3d p = sp.point(0.5 * q); std::printf(" %.17g %.17g", p[0], p[2]); }
    std::printf("\n");
    Eigen::Vector3d still = copy.point(y[3]); // the copy keeps the first fit
    std::printf("copy %.17g %.17g\n", still[0], still[2]);
    Eigen::Vector3d out = sp.point(4.5);       // GSL_EDOM: a line on stderr, NaNs
    std::printf("edom %d\n", (out[0] != out[0]) ? 1 : 0);
    Spline bad(2, y2, x2, z2);                 // GSL_EINVAL
    std::printf("einval %d\n", bad.nodes());
    return 0;
}
