// stand-in (declarations only)
#pragma once
#include <memory>
#include <vector>
namespace opengv { namespace sac {
template <typename P>
class SampleConsensus {
 public:
  typedef P problem_t;
  typedef typename problem_t::model_t model_t;
  SampleConsensus(int maxIterations = 1000, double threshold = 1.0, double probability = 0.99)
      : max_iterations_(maxIterations), iterations_(0), threshold_(threshold), probability_(probability) {}
  virtual ~SampleConsensus() {}
  virtual bool computeModel(int debug_verbosity_level = 0) = 0;
  int max_iterations_;
  int iterations_;
  double threshold_;
  double probability_;
  model_t model_coefficients_;
  std::vector<int> model_;
  std::vector<int> inliers_;
  std::shared_ptr<P> sac_model_;
};
}}
