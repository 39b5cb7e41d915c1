// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <sophus/so3.hpp>: only the type name is needed (common.h:29 typedef).
#pragma once
namespace Sophus { class SO3d {}; }
