/*
 * contour_alg.h -- drop-in for the reference header of the same name (class SectPath, contour_alg.h:49-91), the
 * self-contained copy of the equal-spacing planner that src/contour.cpp builds on (src/contour_alg.cpp; not in the
 * reference's CMakeLists.txt).  It is SectPath of Path_Generate_Algorithm.h with two differences, both parameters of the
 * same kernels: getPath samples from miny + 5 to bigy - 5 (contour_alg.cpp:496-497) and the hand-eye calibration is
 * the one of contour_alg.h:37-42.  (The inserted points are white instead of red in show()'s cloud, contour_alg.cpp:228-230; the millisecond
 * printout of GenPath is a console matter.)  Do not include it together with Path_Generate_Algorithm.h in one translation unit -- the
 * reference's two headers define the same class name as well.
 */
#ifndef PATH_CONTOUR
#define PATH_CONTOUR

#define HANDEYEx -0.858533
#define HANDEYEy 0.075348
#define HANDEYEz 0.672533
#define HANDEYErx -3.138775
#define HANDEYEry -0.0405313
#define HANDEYErz -1.5707969
#define PPP_GETPATH_TRIM 5
#define PPP_NODE_GB 255 /* inserted points are white here (contour_alg.cpp:228-230) */

#include "Path_Generate_Algorithm.h"

#endif
