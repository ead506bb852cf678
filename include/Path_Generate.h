/*
 * Path_Generate.h -- drop-in for the reference header of the same name (class path_generater,
 * :33-75), the planner ./main links (src/main.cpp, src/Path_Generation.cpp): brute-force pairing
 * (insert_point, Path_Generation.cpp:107-206) and the float walks of slicing_method (:282-321)
 * and Contact_Path_Generation (:689-755) including its dynamic adjustment (compute_transform,
 * Area2Cloud, compute_boundary, bisection, dynamic_adjust_path, :362-634; k = 10 neighbours,
 * depth 0.005, Adjust_Threshold 1, toolthickness 10 as in the reference header :71).
 * compute_coverage / drawpath only colour the viewer's cloud and are not reproduced.
 */
#ifndef PATH_GENERATION
#define PATH_GENERATION

#include <chrono>
#include <fstream>
#include <string>
#include <vector>
#include "Spline.h"
/* The reference's drivers write `cout << ... << endl` unqualified (src/main.cpp:10,16, src/connect.cpp:13,19,
   src/connect1.cpp:13,19, src/contour.cpp:13,19).  Upstream those names reach them through THIS header ->
   pcl/visualization/cloud_viewer.h -> VTK's vtkIOStream.h (`using std::cout; using std::endl; using std::cerr;` at global
   scope).  The drop-in header takes PCL and VTK away, so it supplies the same three names itself -- here, in the header
   the drivers include, not in ppp_planner.hpp / Spline.h, whose other includers get nothing at global scope
   (tests/test_host_logic.py::test_reference_drivers_compile_unchanged).  -DPPP_NO_GLOBAL_IOSTREAM_NAMES leaves them out. */
#ifndef PPP_NO_GLOBAL_IOSTREAM_NAMES
#include <iostream>
using std::cerr;
using std::cout;
using std::endl;
#endif

class path_generater {
public:
    path_generater() {}
    path_generater(std::string cloud_name, double Radius) : toolRadius(Radius), file_name(cloud_name)
    {
        ppp_params &p = planner.config().params;
        p.tool_radius = Radius; p.pairing = PPP_PAIR_BRUTE; p.walk = PPP_WALK_V1_CONTACT; p.change_range = 1; /* always x1000, :28-30 */
        planner.open(cloud_name);
    }
    ~path_generater() {}

    /* Path_Generation.cpp:37-51: other_cloud (inserted nodes, blue :196-198) + the cloud with the paths in red (:724) */
    void show()
    {
        const unsigned char node_rgb[3] = {0, 0, 255}, path_rgb[3] = {255, 0, 0};
        planner.show_dump(node_rgb, path_rgb);
    }
    void voxel_down(const float x, const float y, const float z) { planner.voxel_down(x, y, z); } /* Path_Generation.cpp:53-59 */
    void trans2center() { planner.trans2center(); } /* Path_Generation.cpp:60-92 */
    void smooth() { planner.smooth_mls(15, 3, file_name, true); } /* Path_Generation.cpp:340-360 */
    void Set_kdtree() {}
    void estimate_normal() { planner.estimate_normal(); } /* Path_Generation.cpp:323-333 */
    const std::vector<float> &cloud_normals() const { return planner.cloud_normals(); } /* n x (nx ny nz curvature) */
    void get_coverage() { note("get_coverage"); }

    std::vector<int> rangedX_index(int position) { return planner.rangedX_index(position); }
    std::map<double, std::vector<double>> insert_point(std::vector<int> indices, Eigen::Vector3f PlanePoint)
    {
        return planner.insert_point(indices, PlanePoint[0]);
    }
    /* Path_Generation.cpp:282-321: insert_point on every slice of the min+step/2 walk */
    void slicing_method() { run(PPP_WALK_V1_SLICING, 0, "use time: "); }
    /* Path_Generation.cpp:689-755: contact paths, each adjusted against its predecessor's boundary */
    void Contact_Path_Generation()
    {
        printf("Start Path Planning!\n");
        run(PPP_WALK_V1_CONTACT, 1, "Toal Using Time: ");
    }
    std::vector<Spline> &paths() { return Path_set; }

private:
    void run(int walk, int adjust, const char *label)
    {
        auto t0 = std::chrono::high_resolution_clock::now();
        ppp_params &p = planner.config().params;
        p.walk = walk;
        p.dynamic_adjustment = adjust;
        p.curvature_k = 10; p.depth = depth; p.adjust_threshold = Adjust_Threshold; p.toolthickness = toolthickness;
        if (!planner.apply_params() || !planner.gen_path()) return;
        Path_set.clear();
        int S = planner.num_slices();
        for (int s = 0; s < S; ++s) Path_set.emplace_back(planner.handle(), s);
        auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        std::cout << label << us << std::endl;
        std::cout << "number of paths: " << S << std::endl;
        std::ofstream outputFile("output.csv", std::ios::app); /* Path_Generation.cpp:312-320 */
        if (outputFile.is_open()) outputFile << us << std::endl;
    }
    void note(const char *what) { fprintf(stderr, "ppp: %s() is outside the accelerated path (no-op)\n", what); }

    ppp::Planner planner;
    double toolRadius = 15, depth = 0.005, Adjust_Threshold = 1, toolthickness = 10; /* Path_Generate.h:71 */
    std::vector<Spline> Path_set;
    std::string file_name;
};

#endif
