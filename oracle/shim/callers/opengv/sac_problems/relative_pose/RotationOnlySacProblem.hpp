// stand-in (declarations only) for OpenGV's rotation-only sample-consensus problem
#pragma once
#include <opengv/sac/SampleConsensusProblem.hpp>
#include <opengv/types.hpp>
#include <opengv/relative_pose/RelativeAdapterBase.hpp>
namespace opengv { namespace sac_problems { namespace relative_pose {
class RotationOnlySacProblem : public sac::SampleConsensusProblem<rotation_t> {
 public:
  typedef rotation_t model_t;
  typedef opengv::relative_pose::RelativeAdapterBase adapter_t;
  RotationOnlySacProblem(adapter_t& adapter, bool randomSeed = true) : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter) {}
  RotationOnlySacProblem(adapter_t& adapter, const std::vector<int>& indices, bool randomSeed = true)
      : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter) { (void)indices; }
  virtual ~RotationOnlySacProblem() {}
  virtual bool computeModelCoefficients(const std::vector<int>& indices, model_t& outModel) const;
  virtual void getSelectedDistancesToModel(const model_t& model, const std::vector<int>& indices, std::vector<double>& scores) const;
  virtual void optimizeModelCoefficients(const std::vector<int>& inliers, const model_t& model, model_t& optimized_model);
  virtual int getSampleSize() const;
 protected:
  adapter_t& _adapter;
};
}}}
