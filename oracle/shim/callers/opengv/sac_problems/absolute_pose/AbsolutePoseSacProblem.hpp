// stand-in (declarations only) for OpenGV's absolute-pose sample-consensus problem
#pragma once
#include <opengv/sac/SampleConsensusProblem.hpp>
#include <opengv/types.hpp>
#include <opengv/absolute_pose/AbsoluteAdapterBase.hpp>
namespace opengv { namespace sac_problems { namespace absolute_pose {
class AbsolutePoseSacProblem : public sac::SampleConsensusProblem<transformation_t> {
 public:
  typedef transformation_t model_t;
  typedef opengv::absolute_pose::AbsoluteAdapterBase adapter_t;
  typedef enum Algorithm { TWOPT = 0, KNEIP = 1, GAO = 2, EPNP = 3, GP3P = 4 } algorithm_t;
  AbsolutePoseSacProblem(adapter_t& adapter, algorithm_t algorithm, bool randomSeed = true)
      : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter), _algorithm(algorithm) {}
  AbsolutePoseSacProblem(adapter_t& adapter, algorithm_t algorithm, const std::vector<int>& indices, bool randomSeed = true)
      : sac::SampleConsensusProblem<model_t>(randomSeed), _adapter(adapter), _algorithm(algorithm) { (void)indices; }
  virtual ~AbsolutePoseSacProblem() {}
  virtual bool computeModelCoefficients(const std::vector<int>& indices, model_t& outModel) const;
  virtual void getSelectedDistancesToModel(const model_t& model, const std::vector<int>& indices, std::vector<double>& scores) const;
  virtual void optimizeModelCoefficients(const std::vector<int>& inliers, const model_t& model, model_t& optimized_model);
  virtual int getSampleSize() const;
 protected:
  adapter_t& _adapter;
  algorithm_t _algorithm;
};
}}}
