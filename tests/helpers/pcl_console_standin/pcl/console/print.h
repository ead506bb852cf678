/* TEST INFRASTRUCTURE -- stand-in for PCL's <pcl/console/print.h> (see parse.h beside it): the reference's
   drivers include it and use nothing from it. */
#pragma once
