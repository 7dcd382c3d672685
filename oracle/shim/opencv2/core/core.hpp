// ORACLE — TEST INFRASTRUCTURE ONLY: stand-in for the handful of OpenCV types the okvis camera / measurement
// headers mention (cv::Mat as an always-empty mask / image holder, cv::KeyPoint).  No image processing.
#pragma once
#include <cstdint>
#include <vector>
typedef unsigned char uchar;
#define CV_8UC1 0
#define CV_8U 0
namespace cv {
struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};
template <class T>
struct Point_ {
  T x, y;
  Point_(T a = 0, T b = 0) : x(a), y(b) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;
class Mat {
 public:
  int rows, cols;
  uchar* data;
  Mat() : rows(0), cols(0), data(0) {}
  Mat(int r, int c, int) : rows(r), cols(c), data(0), buf_((size_t)r * c) { data = buf_.data(); }
  Mat(const Mat& o) : rows(o.rows), cols(o.cols), data(0), buf_(o.buf_) { data = buf_.empty() ? 0 : buf_.data(); }
  Mat& operator=(const Mat& o) {
    rows = o.rows, cols = o.cols, buf_ = o.buf_;
    data = buf_.empty() ? 0 : buf_.data();
    return *this;
  }
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  static Mat ones(int r, int c, int t) {
    Mat m(r, c, t);
    for (auto& v : m.buf_) v = 1;
    return m;
  }
  bool empty() const { return rows == 0 || cols == 0; }
  int type() const { return CV_8UC1; }
  Mat clone() const { return *this; }
  void release() { *this = Mat(); }
  void resize(size_t r) {
    rows = (int)r;
    buf_.resize((size_t)rows * cols);
    data = buf_.empty() ? 0 : buf_.data();
  }
  template <class T>
  T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
  template <class T>
  const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }

 private:
  std::vector<uchar> buf_;
};
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
}  // namespace cv
