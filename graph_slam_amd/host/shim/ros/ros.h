// ROS-free stand-in for the handful of roscpp calls the reference drivers make (g2o/test_g2o_graph.cpp:28-29,
// 138-162): ros::init, NodeHandle::param and the ROS_* log macros.  Private parameters ("~name") are read
// from environment variables of the same name, so a launch file's <param> entries become `name=value`.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include "../qt_lite.h"   // boost::bind / _1 reach the reference's sources through roscpp's headers

#define ROS_INFO(...) do { std::fprintf(stderr, "[ INFO] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { std::fprintf(stderr, "[ WARN] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, "[ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)

namespace ros {
inline void init(int &, char **, const std::string &) {}
inline bool ok() { return true; }
class NodeHandle {
 public:
  NodeHandle() {}
  explicit NodeHandle(const std::string &) {}
  template <typename T, typename D>
  bool param(const std::string &name, T &var, const D &def) const {
    const char *e = std::getenv(name.c_str());
    if (!e) { var = def; return false; }
    std::istringstream is(e);
    T tmp;
    if (is >> tmp) { var = tmp; return true; }
    var = def;
    return false;
  }
  bool param(const std::string &name, std::string &var, const std::string &def) const {
    const char *e = std::getenv(name.c_str());
    var = e ? std::string(e) : def;
    return e != nullptr;
  }
};
}  // namespace ros
