// Mirror of the reference's src/main.cpp (:6-36): ./main workpiece.pcd, tool radius 15.
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
    path_planner.estimate_normal();
    path_planner.Contact_Path_Generation();
    path_planner.show();
    return 0;
}
