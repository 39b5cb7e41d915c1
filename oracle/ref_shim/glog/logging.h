// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <glog/logging.h> (the functor headers log nothing).
#pragma once
