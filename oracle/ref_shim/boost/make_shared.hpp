// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for boost::make_shared (PCL <= 1.10 clouds are held by boost::shared_ptr; association.cpp:279,337
// copies the map cloud into one): the std smart pointer the PointCloud stand-in uses.
#pragma once
#include <memory>
#include <utility>
namespace boost {
template <typename T, typename... A> std::shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
}
