// ORACLE — TEST INFRASTRUCTURE ONLY: stand-in declarations so that okvis/Frame.hpp compiles; detectors are never run.
#pragma once
#include "../core/core.hpp"
namespace cv {
class FeatureDetector {
 public:
  virtual ~FeatureDetector() {}
  virtual void detect(const Mat&, std::vector<KeyPoint>&, const Mat& = Mat()) const {}
};
class DescriptorExtractor {
 public:
  virtual ~DescriptorExtractor() {}
  virtual void compute(const Mat&, std::vector<KeyPoint>&, Mat&) const {}
};
}  // namespace cv
