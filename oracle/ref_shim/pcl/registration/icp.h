// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in: relocator.cpp names pcl::KdTreeFLANN (relocator.cpp:112) through this header's transitive includes and uses nothing else of it.
#pragma once
#include <pcl/kdtree/kdtree_flann.h>
