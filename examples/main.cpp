// Mirror of the reference's src/main.cpp (:6-36): ./main workpiece.pcd, tool radius 15.
// The calls main.cpp:25-29 keeps commented out can be switched on from the environment: PPP_MAIN_VOXEL=1 (voxel_down(0.1, 1, 1)),
// PPP_MAIN_ALIGN=1 (trans2center), PPP_MAIN_SLICING=1 (slicing_method), PPP_MAIN_SMOOTH=1 (smooth).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "Path_Generate.h"

int main(int argc, char **argv)
{
    std::string pcd;
    for (int i = 1; i < argc; ++i) {
        size_t n = strlen(argv[i]);
        if (n > 4 && strcmp(argv[i] + n - 4, ".pcd") == 0) pcd = argv[i];
    }
    if (pcd.empty()) {
        std::cout << "./slicing_method cad_name.pcd" << std::endl;
        return (-1);
    }
    double step_size = 15;
    path_generater path_planner(pcd, step_size);
    auto on = [](const char *name) { const char *v = std::getenv(name); return v && v[0] == '1'; };
    if (on("PPP_MAIN_VOXEL")) path_planner.voxel_down(0.1, 1, 1);
    if (on("PPP_MAIN_ALIGN")) path_planner.trans2center();
    if (on("PPP_MAIN_SLICING")) path_planner.slicing_method();
    if (on("PPP_MAIN_SMOOTH")) path_planner.smooth();
    path_planner.estimate_normal();
    path_planner.Contact_Path_Generation();
    path_planner.show();
    return 0;
}
