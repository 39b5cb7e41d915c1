// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pcl/common/io.h> (nothing needed).
#pragma once
#include "common_headers.h"
