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
#include "cam_model.h"

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
    // plane-aided runs: a 176 x 144 range image of the synthetic room (SR4000 geometry; depth in mm like the real reader,
    // CamModel::m_z_scale = 0.001) and a featureless intensity image
    fgo_synth::World &w = fgo_synth::World::instance();
    double R[9], t[3];
    if (w.has_room && w.camera_pose(frame, R, t)) {
      const CamModel *cam = CamModel::gCamModel();
      intensity = cv::Mat(144, 176, CV_8UC1); depth = cv::Mat(144, 176, CV_16UC1);
      intensity.frame = depth.frame = frame - 1;
      for (int v = 0; v < 144; ++v)
        for (int u = 0; u < 176; ++u) {
          double z; int wall;
          intensity.at<unsigned char>(v, u) = 128;
          unsigned short mm = 0;
          if (cam->fx > 0 && w.cast(R, t, (u - cam->cx) / cam->fx, (v - cam->cy) / cam->fy, z, wall) && z < 60.0) mm = (unsigned short)(z * 1000.0 + 0.5);
          depth.at<unsigned short>(v, u) = mm;
        }
    }
    return true;
  }
};
