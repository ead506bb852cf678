/*
 * Spline.h -- drop-in for the reference's include/Spline.h (class Spline, lines 7-51).
 * The knots live in HBM; point(y) evaluates the two Steffen interpolants on the device through
 * ppp_eval_spline (same formulas and operation order as GSL's steffen.c).  A Spline is a light
 * view (engine handle + slice number), copied by value like the reference's.
 */
#ifndef SPLINE
#define SPLINE

#include <cmath>
#include "ppp_planner.hpp"

class Spline {
public:
    Spline() {}
    Spline(ppp_handle h, int slice) : h_(h), slice_(slice)
    {
        size_t m = 0;
        if (ppp_get_nodes(h_, slice_, nullptr, nullptr, nullptr, 0, &m) == PPP_OK && m > 0) {
            std::vector<double> y(m);
            ppp_get_nodes(h_, slice_, y.data(), nullptr, nullptr, m, &m);
            node_number = (int)m; small_y = y.front(); big_y = y.back();
        }
    }
    /* must use double type (Vector3d) -- Spline.h:21-25 */
    Eigen::Vector3d point(double y)
    {
        double xyz[3] = {NAN, NAN, NAN};
        if (h_) ppp_eval_spline(h_, slice_, &y, 1, xyz);
        return Eigen::Vector3d(xyz[0], xyz[1], xyz[2]);
    }
    double miny() { return small_y; }
    double bigy() { return big_y; }
    int nodes() const { return node_number; }

private:
    ppp_handle h_ = nullptr;
    int slice_ = 0;
    int node_number = 0;
    double big_y = 0, small_y = 0;
};

#endif
