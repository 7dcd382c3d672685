// stand-in (declarations only) for OpenGV's RANSAC driver
#pragma once
#include <opengv/sac/SampleConsensus.hpp>
namespace opengv { namespace sac {
template <typename P>
class Ransac : public SampleConsensus<P> {
 public:
  typedef P problem_t;
  typedef typename problem_t::model_t model_t;
  using SampleConsensus<P>::max_iterations_;
  using SampleConsensus<P>::threshold_;
  using SampleConsensus<P>::iterations_;
  using SampleConsensus<P>::sac_model_;
  using SampleConsensus<P>::model_;
  using SampleConsensus<P>::model_coefficients_;
  using SampleConsensus<P>::inliers_;
  using SampleConsensus<P>::probability_;
  Ransac(int maxIterations = 1000, double threshold = 1.0, double probability = 0.99) : SampleConsensus<P>(maxIterations, threshold, probability) {}
  virtual ~Ransac() {}
  bool computeModel(int debug_verbosity_level = 0);
};
}}
