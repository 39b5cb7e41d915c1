// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <opencv2/core/eigen.hpp> (utility.h:9 includes it; nothing on the factor path uses it).
#pragma once
#include "../opencv.hpp"
