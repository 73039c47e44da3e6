// Stand-in for the few OpenCV names the reference's wrappers / drivers mention: cv::Mat as a
// typed 2-D buffer (rows, cols, type, at<T>, zeros, clone, size), DMatch / KeyPoint, and no-op highgui calls.  Besides
// pixels a Mat carries the synthetic frame index of the stand-in front end (shim/vro_synth.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16UC1 2
#define CV_32FC1 5

namespace cv {
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0; };
struct DMatch { int queryIdx = 0, trainIdx = 0; float distance = 0; };
class Mat {
 public:
  int rows = 0, cols = 0;
  int frame = -1;                         // synthetic frame index (CSReadCV::readOneFrameCV)
  Mat() : type_(CV_8UC1) {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t), d_((size_t)r * c * elem(t), 0) {}
  Mat(Size s, int t) : rows(s.height), cols(s.width), type_(t), d_((size_t)s.height * s.width * elem(t), 0) {}
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  Mat clone() const { return *this; }
  int type() const { return type_; }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return d_.empty(); }
  template <class T> T &at(int r, int c) { return *reinterpret_cast<T *>(&d_[((size_t)r * cols + c) * elem(type_)]); }
  template <class T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(&d_[((size_t)r * cols + c) * elem(type_)]); }
  template <class T> T &at(int i) { return *reinterpret_cast<T *>(&d_[(size_t)i * sizeof(T)]); }
  template <class T> const T &at(int i) const { return *reinterpret_cast<const T *>(&d_[(size_t)i * sizeof(T)]); }
 private:
  static size_t elem(int t) { return t == CV_8UC3 ? 3 : (t == CV_16UC1 ? 2 : (t == CV_32FC1 ? 4 : 1)); }
  int type_;
  std::vector<unsigned char> d_;
};
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int = 0) { return 0; }
inline void namedWindow(const std::string &, int = 0) {}
inline void destroyWindow(const std::string &) {}
}  // namespace cv
typedef unsigned char uchar;
