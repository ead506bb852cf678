/*
 * Spline.h -- drop-in for the reference's include/Spline.h (class Spline, lines 7-51): same constructors, point(),
 * miny(), bigy(), restart().  The two GSL Steffen interpolants are evaluated on the device (same formulas and
 * operation order as GSL's steffen.c, in double):
 *   - Spline(number, point_y, point_x, point_z) / restart(...)  (Spline.h:10-20, 30-42) fit caller-supplied knots
 *     through ppp_spline_create / ppp_spline_restart -- what OnePath, path_track and dynamic_adjust_path do;
 *   - Spline(handle, slice) is a view of a slice the planner fitted itself (knots resident with the slice,
 *     ppp_eval_spline) -- how the drop-in planner classes fill Path_set.
 * Copied by value like the reference's (which shares its raw GSL pointers between copies and never frees them; here
 * the copies share one reference-counted object).  Where GSL would call its error handler and abort (fewer than 3
 * knots, y not strictly increasing, evaluation outside [miny, bigy]) a line goes to stderr and point() returns NaNs.
 */
#ifndef SPLINE
#define SPLINE

#include <cmath>
#include <cstdio>
#include <memory>
#include "ppp_planner.hpp"

class Spline {
public:
    Spline() {}
    /* include/Spline.h:10-20 */
    Spline(int number, const double *point_y, const double *point_x, const double *point_z) { fit(number, point_y, point_x, point_z); }
    /* a slice of the planner behind `h` (engine-side knots) */
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
        int rc = PPP_ERR_ARG;
        if (own_) rc = ppp_spline_eval(own_.get(), &y, 1, xyz);
        else if (h_) rc = ppp_eval_spline(h_, slice_, &y, 1, xyz);
        if (rc == PPP_ERR_DOMAIN) std::fprintf(stderr, "gsl: interpolation error (y = %g outside [%g, %g])\n", y, small_y, big_y);
        return Eigen::Vector3d(xyz[0], xyz[1], xyz[2]);
    }
    double miny() { return small_y; }
    double bigy() { return big_y; }
    int nodes() const { return node_number; }

    /* include/Spline.h:30-42 */
    void restart(int number, const double *point_y, const double *point_x, const double *point_z)
    {
        if (own_ && own_.use_count() == 1 && number >= 0) {
            int rc = ppp_spline_restart(own_.get(), (size_t)number, point_y, point_x, point_z);
            if (rc == PPP_OK) { node_number = number; small_y = point_y[0]; big_y = point_y[number - 1]; return; }
        }
        fit(number, point_y, point_x, point_z); /* a view, a shared copy, or a failed re-fit: a fresh object */
    }

private:
    void fit(int number, const double *point_y, const double *point_x, const double *point_z)
    {
        own_.reset(); h_ = nullptr; node_number = 0; small_y = big_y = 0;
        ppp_spline sp = nullptr;
        int rc = number < 0 ? PPP_ERR_ARG : ppp_spline_create(ppp::Planner::device_from_env(), (size_t)number, point_y, point_x, point_z, &sp);
        if (rc != PPP_OK) {
            std::fprintf(stderr, rc == PPP_ERR_ARG ? "gsl: Spline needs at least 3 knots with strictly increasing y (%d given)\n"
                                                   : "ppp: Spline: no MI355X device available (error %d)\n", rc == PPP_ERR_ARG ? number : rc);
            return;
        }
        own_ = std::shared_ptr<ppp_spline_s>(sp, [](ppp_spline p) { ppp_spline_destroy(p); });
        node_number = number; small_y = point_y[0]; big_y = point_y[number - 1];
    }
    std::shared_ptr<ppp_spline_s> own_; /* caller-supplied knots */
    ppp_handle h_ = nullptr;            /* or a slice of a planner */
    int slice_ = 0;
    int node_number = 0;
    double big_y = 0, small_y = 0;
};

#endif
