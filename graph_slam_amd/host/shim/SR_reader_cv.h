// Stand-in for VRO's SR4000 reader + the OpenCV types the driver names (test_g2o_graph.cpp:40,58,67-70).
// The reference's sources rely on these headers for `using namespace std` and <iomanip>/<sstream>.
#pragma once
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include "vro_synth.h"
using namespace std;

#include "opencv2/opencv.hpp"

class CSReadCV {
 public:
  // "<dir>/<prefix>_<7-digit frame>.<suffix>": succeeds while the synthetic world has that frame (1-based)
  bool readOneFrameCV(const std::string &path, cv::Mat &intensity, cv::Mat &depth) {
    fgo_synth::World::instance().ensure();
    const size_t us = path.find_last_of('_'), dot = path.find_last_of('.');
    if (us == std::string::npos || dot == std::string::npos || dot < us) return false;
    const int frame = std::atoi(path.substr(us + 1, dot - us - 1).c_str());
    if (frame < 1 || frame > fgo_synth::World::instance().n_poses) return false;
    intensity.frame = depth.frame = frame - 1;
    return true;
  }
};
