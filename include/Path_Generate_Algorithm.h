/*
 * Path_Generate_Algorithm.h -- drop-in for the reference header of the same name
 * (class SectPath :57-98, class path_generater :102-135), used by src/connect.cpp and
 * src/connect1.cpp.  Same public methods and constructor arguments; the arithmetic runs on the
 * MI355X through include/ppp_hip.h.  Differences a caller can observe:
 *   - show() opens no PCL viewer: with PPP_SHOW_PCD=<file> it writes the cloud the viewer would show (inserted nodes +
 *     the cloud with the paths painted on it, path_slicing_alg.cpp:69-80, 269-288) as a PointXYZRGB PCD, else a notice;
 *   - Dynamic_adjustment = true runs the curvature-driven re-spacing (path_dynamic_alg.cpp:77-306)
 *     on the GPU, RemoveOutlier = true the statistical outlier removal (path_slicing_alg.cpp:101-108),
 *     Smooth = true the moving-least-squares smoothing (path_slicing_alg.cpp:111-139);
 *     Alignment = true the PCA alignment (path_slicing_alg.cpp:82-99; eigenvector order as Eigen's EigenSolver
 *     leaves it);
 *   - conditions on which the reference aborts (GSL / FLANN) are reported on stderr instead.
 * Build connect1's single-direction walk with -DPPP_SDIR (it links dynamic_alg_sdir.cpp in the
 * reference, CMakeLists.txt:38-47).
 */
#ifndef PATH_CONNECT
#define PATH_CONNECT

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

#ifndef HANDEYEx /* (include/contour_alg.h brings its own calibration and trim) */
#define HANDEYEx -0.764091
#define HANDEYEy 0.025886
#define HANDEYEz 0.663790
#define HANDEYErx -3.1270175
#define HANDEYEry -0.040124
#define HANDEYErz -1.6063578
#endif
#ifndef PPP_NODE_GB
#define PPP_NODE_GB 0 /* green / blue of the inserted nodes' colour: red (path_slicing_alg.cpp:228-230); contour_alg.h makes them white */
#endif
#ifndef PPP_GETPATH_TRIM
#define PPP_GETPATH_TRIM 10 /* path_translation_alg.cpp:158-159: dy = miny + 10 ... bigy - 10 */
#endif

/* Base class: equal spacing path + robot path (path_slicing_alg.cpp, path_translation_alg.cpp) */
class SectPath {
public:
    SectPath() {}
    SectPath(std::string configName, std::string CloudFileName) : cloud_name(CloudFileName)
    {
        read_config(configName);
        planner.config().params.walk = PPP_WALK_SECTPATH;
        init_common();
        planner.open(cloud_name);
        if (ifSmooth) smooth();         /* path_slicing_alg.cpp:27 */
        if (ifAlign) trans2center();    /* path_slicing_alg.cpp:28 */
        if (ifRemove) remove_outlier(); /* path_slicing_alg.cpp:29 */
    }
    virtual ~SectPath() {}

    /* path_slicing_alg.cpp:82-99: PCA alignment; getPath then goes back through the inverse (path_translation_alg.cpp:146-174) */
    void trans2center() { planner.trans2center(); }
    /* path_slicing_alg.cpp:111-139: pcl::MovingLeastSquares, order 3, radius 15, and the smooth_<name> side file */
    void smooth() { planner.smooth_mls(15, 3, cloud_name, ifChangeRange); }

    /* path_slicing_alg.cpp:101-108: pcl::StatisticalOutlierRemoval, 50 neighbours, 1 sigma, on the resident cloud */
    void remove_outlier() { planner.remove_outlier(50, 1.0); }

    /* path_slicing_alg.cpp:69-80: other_cloud (the inserted nodes, red -- white in contour_alg.cpp:228-230) + the cloud with
       the paths drawn on it (drawpath: red for SectPath::GenPath :315,326, blue for the derived planner, path_dynamic_alg.cpp:330,354) */
    void show()
    {
        const unsigned char node_rgb[3] = {255, PPP_NODE_GB, PPP_NODE_GB}, path_rgb[3] = {(unsigned char)(blue_paths ? 0 : 255), 0, (unsigned char)(blue_paths ? 255 : 0)};
        /* the derived planner also paints each boundary curve green before it adjusts a slice against it (path_dynamic_alg.cpp:320-322) */
        planner.show_dump(node_rgb, path_rgb, blue_paths && planner.config().params.dynamic_adjustment != 0);
    }
    /* path_slicing_alg.cpp:141-150: the whole-cloud normal field (the reference's GenPath and getPath call it themselves;
       here they evaluate normals only where getPath needs them, so this runs when the CALLER asks for the field) */
    void estimate_normal() { planner.estimate_normal(); }
    const std::vector<float> &cloud_normals() const { return planner.cloud_normals(); } /* n x (nx ny nz curvature) */
    virtual void GenPath()
    {
        printf("Start Path Planning!\n");
        auto startT = std::chrono::high_resolution_clock::now();
        if (!planner.gen_path()) return;
        build_path_set();
        auto duration = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - startT).count();
        printf("Toal Using Time: %ld \n", (long)duration);          /* path_slicing_alg.cpp:337-341 */
        printf("Number of paths: %d\n", (int)Path_set.size());
        printf("Number of Point Cloud: %ld\n", (long)planner.num_points());
    }
    void getPath()
    {
        /* Path_set.erase(begin) + pop_back (path_translation_alg.cpp:149-150) happen inside the engine */
        std::cout << "Path number is " << (Path_set.size() >= 2 ? Path_set.size() - 2 : 0) << std::endl;
        std::vector<float> wp;
        if (!planner.get_path(wp)) return;
        WayPointsList.assign(wp.size() / 6, std::vector<float>(6));
        for (size_t w = 0; w < WayPointsList.size(); ++w)
            for (int d = 0; d < 6; ++d) WayPointsList[w][d] = wp[6 * w + d];
    }
    const std::vector<std::vector<float>> &waypoints() const { return WayPointsList; }

protected:
    virtual void read_config(std::string filename)
    {
        ppp_read_config(filename.c_str(), &planner.config());
        ppp_config &c = planner.config();
        toolRadius = c.params.tool_radius; PathResolution = c.params.path_resolution; RPYres = c.params.rpy_resolution;
        EElen = c.params.ee_length; ifChangeRange = c.params.change_range; ifAlign = c.alignment; ifSmooth = c.smooth_cloud;
        ifRemove = c.remove_outlier; pathFile = c.path_file;
    }
    void init_common()
    {
        ppp_params &p = planner.config().params;
        p.pairing = PPP_PAIR_KD; p.trim = PPP_GETPATH_TRIM; p.drop_ends = 1; p.smooth = 1;
        const float he[6] = {(float)HANDEYEx, (float)HANDEYEy, (float)HANDEYEz, (float)HANDEYErx, (float)HANDEYEry, (float)HANDEYErz};
        for (int i = 0; i < 6; ++i) p.handeye[i] = he[i];
    }
    void build_path_set()
    {
        Path_set.clear();
        int S = planner.num_slices();
        for (int s = 0; s < S; ++s) Path_set.emplace_back(planner.handle(), s);
    }

    /* Path planning Alg. */
    std::vector<int> rangedX_index(int position) { return planner.rangedX_index(position); }
    MAP insert_point(std::vector<int> indices, Eigen::Vector3f PlanePoint) { return planner.insert_point(indices, PlanePoint[0]); }

    ppp::Planner planner;
    double toolRadius = 12, PathResolution = 7, RPYres = 7;
    float EElen = 0.3f;
    bool ifAlign = false, ifSmooth = false, ifChangeRange = true, ifRemove = false;
    std::vector<Spline> Path_set;
    std::string cloud_name, pathFile;
    std::vector<std::vector<float>> WayPointsList;
    bool blue_paths = false;
};

/* Derived class of the reference: dynamic adjustment (path_dynamic_alg.cpp / dynamic_alg_sdir.cpp) */
class path_generater : public SectPath {
public:
    path_generater() {}
    path_generater(std::string configName, std::string CloudFileName)
    {
        cloud_name = CloudFileName;
        blue_paths = true;
        read_config(configName);
#ifdef PPP_SDIR
        planner.config().params.walk = PPP_WALK_SDIR_INT;   /* dynamic_alg_sdir.cpp:349-374 */
#else
        planner.config().params.walk = PPP_WALK_CENTER_INT; /* path_dynamic_alg.cpp:308-372 */
#endif
        init_common();
        planner.open(cloud_name);
        if (ifSmooth) smooth();         /* path_dynamic_alg.cpp:29 */
        if (ifAlign) trans2center();    /* path_dynamic_alg.cpp:30 */
        if (ifRemove) remove_outlier(); /* path_dynamic_alg.cpp:32 */
    }
    void GenPath() override
    {
        printf("Start Path Planning!\n");
        auto startT = std::chrono::high_resolution_clock::now();
        if (!planner.gen_path()) return;
        build_path_set();
        auto duration = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - startT).count();
        printf("Toal Using Time: %ld \n", (long)duration);          /* path_dynamic_alg.cpp:374-378 */
        printf("Number of Point Cloud: %ld\n", (long)planner.num_points());
        std::ofstream outputFile("output.csv", std::ios::app);      /* path_dynamic_alg.cpp:380-388 */
        if (!outputFile.is_open()) std::cerr << "cannot open output.csv" << std::endl;
        else outputFile << duration << std::endl;
    }
    typedef enum { left, right } Dir;

private:
    void read_config(std::string filename) override
    {
        SectPath::read_config(filename);
        ppp_config &c = planner.config();
        depth = c.depth; Adjust_Threshold = c.adjust_threshold; toolthickness = c.toolthickness; Adjust = c.dynamic_adjustment;
        /* path_dynamic_alg.cpp:337-372: with Adjust the slices left/right of the centre are re-fitted
           against the boundary of their inner neighbour (compute_boundary / dynamic_adjust_path) */
        c.params.dynamic_adjustment = Adjust ? 1 : 0;
    }
    double depth = 0.01, Adjust_Threshold = 1, toolthickness = 10;
    bool Adjust = false;
};

#endif
