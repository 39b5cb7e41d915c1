// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/sample_consensus/method_types.h>.
#pragma once
namespace pcl { const static int SAC_RANSAC = 0; }
