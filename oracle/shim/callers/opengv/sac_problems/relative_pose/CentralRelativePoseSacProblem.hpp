// stand-in (declarations only) for OpenGV's central relative-pose sample-consensus problem
#pragma once
#include <opengv/sac/SampleConsensusProblem.hpp>
#include <opengv/types.hpp>
#include <opengv/relative_pose/RelativeAdapterBase.hpp>
namespace opengv { namespace sac_problems { namespace relative_pose {
class CentralRelativePoseSacProblem : public sac::SampleConsensusProblem<transformation_t> {
 public:
  typedef transformation_t model_t;
  typedef opengv::relative_pose::RelativeAdapterBase adapter_t;
  typedef enum Algorithm { STEWENIUS = 0, NISTER = 1, SEVENPT = 2, EIGHTPT = 3 } algorithm_t;
  CentralRelativePoseSacProblem(adapter_t& adapter, algorithm_t algorithm, bool randomSeed = true)
      : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter), _algorithm(algorithm) {}
  CentralRelativePoseSacProblem(adapter_t& adapter, algorithm_t algorithm, const std::vector<int>& indices, bool randomSeed = true)
      : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter), _algorithm(algorithm) { (void)indices; }
  virtual ~CentralRelativePoseSacProblem() {}
  virtual bool computeModelCoefficients(const std::vector<int>& indices, model_t& outModel) const;
  virtual void getSelectedDistancesToModel(const model_t& model, const std::vector<int>& indices, std::vector<double>& scores) const;
  virtual void optimizeModelCoefficients(const std::vector<int>& inliers, const model_t& model, model_t& optimized_model);
  virtual int getSampleSize() const;
 protected:
  adapter_t& _adapter;
  algorithm_t _algorithm;
};
}}}
