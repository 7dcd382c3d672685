// stand-in (declarations only) for OpenGV's sample-consensus problem interface
#pragma once
#include <memory>
#include <vector>
namespace opengv { namespace sac {
template <typename M>
class SampleConsensusProblem {
 public:
  typedef M model_t;
  SampleConsensusProblem(bool randomSeed = true) : max_sample_checks_(10) { (void)randomSeed; }
  virtual ~SampleConsensusProblem() {}
  virtual void getSamples(int& iterations, std::vector<int>& samples);
  virtual bool isSampleGood(const std::vector<int>& sample) const;
  std::shared_ptr<std::vector<int> > getIndices() const;
  void drawIndexSample(std::vector<int>& sample);
  virtual int getSampleSize() const = 0;
  virtual bool computeModelCoefficients(const std::vector<int>& indices, model_t& outModel) const = 0;
  virtual void optimizeModelCoefficients(const std::vector<int>& inliers, const model_t& model_coefficients, model_t& optimized_coefficients) = 0;
  virtual void getSelectedDistancesToModel(const model_t& model, const std::vector<int>& indices, std::vector<double>& scores) const = 0;
  virtual void getDistancesToModel(const model_t& model_coefficients, std::vector<double>& distances);
  virtual void selectWithinDistance(const model_t& model_coefficients, const double threshold, std::vector<int>& inliers);
  virtual int countWithinDistance(const model_t& model_coefficients, const double threshold);
  void setIndices(const std::vector<int>& indices);
  void setUniformIndices(int N);
  int rnd();
  int max_sample_checks_;
  std::shared_ptr<std::vector<int> > indices_;
  std::vector<int> shuffled_indices_;
};
}}
