// stand-in (declarations only) for the BRISK detector / extractor classes okvis::Frontend constructs
#pragma once
#include <opencv2/features2d/features2d.hpp>
#include <cstddef>
#include <vector>
namespace brisk {
class HarrisScoreCalculator {};
template <class SCORE_CALCULATOR_T>
class ScaleSpaceFeatureDetector : public cv::FeatureDetector {
 public:
  ScaleSpaceFeatureDetector(size_t octaves, double uniformityRadius, double absoluteThreshold = 0, size_t maxNumKpt = 100000);
  void detect(const cv::Mat& image, std::vector<cv::KeyPoint>& keypoints, const cv::Mat& mask = cv::Mat()) const;
};
class BriskDescriptorExtractor : public cv::DescriptorExtractor {
 public:
  BriskDescriptorExtractor(bool rotationInvariant = true, bool scaleInvariant = true);
  void compute(const cv::Mat& image, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) const;
};
}
