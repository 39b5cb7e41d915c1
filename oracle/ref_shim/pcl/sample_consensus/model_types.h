// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/sample_consensus/model_types.h>.
#pragma once
namespace pcl { enum SacModel { SACMODEL_PLANE = 0 }; }
