// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C wrapper around the okvis reference's OWN classes (compiled unmodified from /root/reference by
// oracle/ref/Makefile against the stand-in headers in oracle/shim): okvis::ceres::ReprojectionError<G>,
// ImuError, PoseError, SpeedAndBiasError, RelativePoseError, PoseLocalParameterization,
// HomogeneousPointLocalParameterization, MarginalizationError, Map and okvis::cameras::PinholeCamera<D>.
// It exposes them through the same flat data format as include/okvis_amd_ba.h so that tests/ can check the
// restatement in oracle/ (and the committed golden fixtures) against the reference's own lines.
// Nothing here is a restatement of reference arithmetic: every number returned is produced by reference code;
// this file only builds the objects, moves data in and out, and reads protected members through derived
// "Access" classes.  Only tests/ and tests/golden/make_golden.py load the resulting library (oracle/_ref/).
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <vector>

#include <okvis/Measurements.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/Time.hpp>
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/NoDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/cameras/RadialTangentialDistortion8.hpp>
#include <okvis/ceres/HomogeneousPointLocalParameterization.hpp>
#include <okvis/ceres/HomogeneousPointParameterBlock.hpp>
#include <okvis/ceres/ImuError.hpp>
#include <okvis/ceres/Map.hpp>
#include <okvis/ceres/MarginalizationError.hpp>
#include <okvis/ceres/PoseError.hpp>
#include <okvis/ceres/PoseLocalParameterization.hpp>
#include <okvis/ceres/PoseParameterBlock.hpp>
#include <okvis/ceres/RelativePoseError.hpp>
#include <okvis/ceres/ReprojectionError.hpp>
#include <okvis/ceres/SpeedAndBiasError.hpp>
#include <okvis/ceres/SpeedAndBiasParameterBlock.hpp>
#include <okvis/kinematics/Transformation.hpp>

#include "okvis_amd_ba.h"

namespace google {
int eshim_log_warnings = 0;
}

namespace oc = okvis::ceres;
namespace cam = okvis::cameras;
using okvis::kinematics::Transformation;

namespace {

typedef cam::PinholeCamera<cam::NoDistortion> CamNone;
typedef cam::PinholeCamera<cam::RadialTangentialDistortion> CamRadtan;
typedef cam::PinholeCamera<cam::EquidistantDistortion> CamEqui;
typedef cam::PinholeCamera<cam::RadialTangentialDistortion8> CamRadtan8;

okvis::Time to_time(int64_t ns) { return okvis::Time((uint32_t)(ns / 1000000000LL), (uint32_t)(ns % 1000000000LL)); }

Transformation to_T(const double* p) {
  Transformation T;
  Eigen::Matrix<double, 7, 1> c;
  for (int i = 0; i < 7; ++i) c[i] = p[i];
  T.setCoeffs(c);  // raw coefficients, no renormalisation
  return T;
}

std::shared_ptr<cam::CameraBase> make_camera(const double* intr, int model) {
  const int W = 752, H = 480;
  switch (model) {
    case OKVIS_BA_DIST_NONE:
      return std::shared_ptr<cam::CameraBase>(new CamNone(W, H, intr[0], intr[1], intr[2], intr[3], cam::NoDistortion()));
    case OKVIS_BA_DIST_RADTAN:
      return std::shared_ptr<cam::CameraBase>(new CamRadtan(
          W, H, intr[0], intr[1], intr[2], intr[3], cam::RadialTangentialDistortion(intr[4], intr[5], intr[6], intr[7])));
    case OKVIS_BA_DIST_EQUIDISTANT:
      return std::shared_ptr<cam::CameraBase>(new CamEqui(W, H, intr[0], intr[1], intr[2], intr[3],
                                                          cam::EquidistantDistortion(intr[4], intr[5], intr[6], intr[7])));
    case OKVIS_BA_DIST_RADTAN8:
      return std::shared_ptr<cam::CameraBase>(
          new CamRadtan8(W, H, intr[0], intr[1], intr[2], intr[3],
                         cam::RadialTangentialDistortion8(intr[4], intr[5], intr[6], intr[7], intr[8], intr[9], intr[10],
                                                          intr[11])));
  }
  return std::shared_ptr<cam::CameraBase>();
}

// A ReprojectionError<G> whose squareRootInformation_ can be set directly (protected member).
template <class G>
struct ReprojAccess : oc::ReprojectionError<G> {
  using oc::ReprojectionError<G>::ReprojectionError;
  void setSqrt(const double s[4]) {
    this->squareRootInformation_(0, 0) = s[0], this->squareRootInformation_(0, 1) = s[1];
    this->squareRootInformation_(1, 0) = s[2], this->squareRootInformation_(1, 1) = s[3];
    this->information_ = this->squareRootInformation_.transpose() * this->squareRootInformation_;
  }
};
template <class G>
std::shared_ptr<oc::ErrorInterface> make_reproj_t(std::shared_ptr<cam::CameraBase> c, const double uv[2],
                                                  const double sqrtInfo[4], std::shared_ptr< ::ceres::CostFunction>* cf) {
  std::shared_ptr<const G> g = std::static_pointer_cast<const G>(c);
  Eigen::Vector2d m(uv[0], uv[1]);
  Eigen::Matrix2d info = Eigen::Matrix2d::Identity();
  std::shared_ptr<ReprojAccess<G> > e(new ReprojAccess<G>(g, 0, m, info));
  e->setSqrt(sqrtInfo);
  if (cf) *cf = e;
  return e;
}
std::shared_ptr<oc::ErrorInterface> make_reproj(std::shared_ptr<cam::CameraBase> c, int model, const double uv[2],
                                                const double sqrtInfo[4],
                                                std::shared_ptr< ::ceres::CostFunction>* cf = 0) {
  switch (model) {
    case OKVIS_BA_DIST_NONE: return make_reproj_t<CamNone>(c, uv, sqrtInfo, cf);
    case OKVIS_BA_DIST_RADTAN: return make_reproj_t<CamRadtan>(c, uv, sqrtInfo, cf);
    case OKVIS_BA_DIST_EQUIDISTANT: return make_reproj_t<CamEqui>(c, uv, sqrtInfo, cf);
    default: return make_reproj_t<CamRadtan8>(c, uv, sqrtInfo, cf);
  }
}

struct PoseErrorAccess : oc::PoseError {
  using oc::PoseError::PoseError;
  void setSqrt(const double* s) {
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) squareRootInformation_(i, j) = s[i * 6 + j];
  }
  const information_t& sqrtInfo() const { return squareRootInformation_; }
};
struct SbErrorAccess : oc::SpeedAndBiasError {
  using oc::SpeedAndBiasError::SpeedAndBiasError;
  void setSqrt(const double* s) {
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) squareRootInformation_(i, j) = s[i * 9 + j];
  }
  const information_t& sqrtInfo() const { return squareRootInformation_; }
};
struct RelPoseErrorAccess : oc::RelativePoseError {
  using oc::RelativePoseError::RelativePoseError;
  void setSqrt(const double* s) {
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) squareRootInformation_(i, j) = s[i * 6 + j];
  }
};
struct ImuErrorAccess : oc::ImuError {
  using oc::ImuError::ImuError;
  void primeAt(const double sb_ref[9]) {  // what a factor that was last re-preintegrated at sb_ref looks like
    okvis::SpeedAndBias s;
    for (int i = 0; i < 9; ++i) s[i] = sb_ref[i];
    redoPreintegration(Transformation(), s);
    redo_ = false;
  }
  const information_t& sqrtInfo() const { return squareRootInformation_; }
  void sbRef(double out[9]) const {
    for (int i = 0; i < 9; ++i) out[i] = speedAndBiases_ref_[i];
  }
  int redoCount() const { return redoCounter_; }
};
struct MargAccess : oc::MarginalizationError {
  using oc::MarginalizationError::MarginalizationError;
  // previous prior (H_, b0_ over blocks) as the reference object would hold it before the next addResidualBlock
  void injectPrior(const std::vector<std::shared_ptr<oc::ParameterBlock> >& blocks, const std::vector<int>& offs, int dim,
                   const double* H, const double* b0) {
    for (size_t i = 0; i < blocks.size(); ++i) {
      ParameterBlockInfo info(blocks[i]->id(), blocks[i], (size_t)offs[i], false);
      parameterBlockInfos_.push_back(info);
      parameterBlockId2parameterBlockInfoIdx_[blocks[i]->id()] = i;
      base_t::mutable_parameter_block_sizes()->push_back((int)info.dimension);
    }
    denseIndices_ = blocks.size();
    H_.resize(dim, dim);
    b0_.resize(dim);
    for (int i = 0; i < dim; ++i) {
      b0_[i] = b0[i];
      for (int j = 0; j < dim; ++j) H_(i, j) = H[(size_t)i * dim + j];
    }
    base_t::set_num_residuals(dim);
  }
  // a finished prior in error-term form (J_, e0_, linearisation points): what Evaluate reads
  void injectErrorTerm(const std::vector<std::shared_ptr<oc::ParameterBlock> >& blocks, const std::vector<int>& offs,
                       int dim, const double* J, const double* e0, const double* lin /*[n][9]*/) {
    std::vector<double> Hz((size_t)dim * dim, 0.0), bz((size_t)dim, 0.0);
    injectPrior(blocks, offs, dim, Hz.data(), bz.data());
    for (size_t i = 0; i < blocks.size(); ++i)
      std::memcpy(parameterBlockInfos_[i].linearizationPoint.get(), lin + 9 * i,
                  sizeof(double) * parameterBlockInfos_[i].dimension);
    J_.resize(dim, dim);
    e0_.resize(dim);
    for (int i = 0; i < dim; ++i) {
      e0_[i] = e0[i];
      for (int j = 0; j < dim; ++j) J_(i, j) = J[(size_t)i * dim + j];
    }
    errorComputationValid_ = true;
  }
  const Eigen::MatrixXd& H() const { return H_; }
  const Eigen::VectorXd& b0() const { return b0_; }
  const Eigen::MatrixXd& J() const { return J_; }
  const Eigen::VectorXd& e0() const { return e0_; }
  const Eigen::VectorXd& S() const { return S_; }
  size_t nInfo() const { return parameterBlockInfos_.size(); }
  uint64_t infoId(size_t i) const { return parameterBlockInfos_[i].parameterBlockId; }
  size_t infoOff(size_t i) const { return parameterBlockInfos_[i].orderingIdx; }
  size_t infoDim(size_t i) const { return parameterBlockInfos_[i].minimalDimension; }
};

okvis::ImuParameters to_imu_params(const okvis_ba_imu_params* p) {
  okvis::ImuParameters q;
  q.a_max = p->a_max, q.g_max = p->g_max, q.sigma_g_c = p->sigma_g_c, q.sigma_a_c = p->sigma_a_c;
  q.sigma_gw_c = p->sigma_gw_c, q.sigma_aw_c = p->sigma_aw_c, q.g = p->g;
  q.sigma_bg = 0.03, q.sigma_ba = 0.1, q.tau = 3600.0, q.rate = 200;
  q.a0 = Eigen::Vector3d(0, 0, 0);
  return q;
}
okvis::ImuMeasurementDeque to_deque(int n, const int64_t* t, const double* gyr, const double* acc) {
  okvis::ImuMeasurementDeque d;
  for (int i = 0; i < n; ++i)
    d.push_back(okvis::ImuMeasurement(
        to_time(t[i]), okvis::ImuSensorReadings(Eigen::Vector3d(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]),
                                                Eigen::Vector3d(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]))));
  return d;
}

// evaluate an ErrorInterface with minimal Jacobians into row-major buffers
void evaluate(const oc::ErrorInterface& e, const std::vector<const double*>& params, double* r,
              const std::vector<double*>& Jmin /* may hold NULLs */) {
  const size_t nb = e.parameterBlocks();
  std::vector<std::vector<double> > J(nb);
  std::vector<double*> Jp(nb), Jm(nb);
  for (size_t i = 0; i < nb; ++i) {
    J[i].assign(e.residualDim() * e.parameterBlockDim(i), 0.0);
    Jp[i] = J[i].data();
    Jm[i] = Jmin.size() > i ? Jmin[i] : 0;
  }
  bool any = false;
  for (size_t i = 0; i < nb; ++i) any = any || Jm[i];
  e.EvaluateWithMinimalJacobians(params.data(), r, any ? Jp.data() : 0, any ? Jm.data() : 0);
}

}  // namespace

// =====================================================================================================
extern "C" {

int ref_abi_version(void) { return 1; }

// ---- PoseLocalParameterization (PoseLocalParameterization.cpp:60-145) ----
void ref_pose_plus(const double x[7], const double d[6], double out[7]) { oc::PoseLocalParameterization::plus(x, d, out); }
void ref_pose_minus(const double x[7], const double xpd[7], double d[6]) {
  oc::PoseLocalParameterization::minus(x, xpd, d);
}
void ref_pose_lift_jacobian(const double x[7], double J[42]) { oc::PoseLocalParameterization::liftJacobian(x, J); }
void ref_pose_plus_jacobian(const double x[7], double J[42]) { oc::PoseLocalParameterization::plusJacobian(x, J); }
void ref_hp_plus(const double x[4], const double d[3], double out[4]) {
  oc::HomogeneousPointLocalParameterization::plus(x, d, out);
}
void ref_hp_minus(const double x[4], const double xpd[4], double d[3]) {
  oc::HomogeneousPointLocalParameterization::minus(x, xpd, d);
}

// ---- PinholeCamera<D>::project (implementation/PinholeCamera.hpp:148-226); returns ProjectionStatus ----
int ref_project(const double intr[12], int model, const double p[3], double kp[2], double* J_2x3) {
  std::shared_ptr<cam::CameraBase> c = make_camera(intr, model);
  Eigen::Vector2d ip(0, 0);
  Eigen::Matrix<double, 2, 3> J = Eigen::Matrix<double, 2, 3>::Zero();
  cam::CameraBase::ProjectionStatus st = c->project(Eigen::Vector3d(p[0], p[1], p[2]), &ip, J_2x3 ? &J : 0, 0);
  kp[0] = ip[0], kp[1] = ip[1];
  if (J_2x3)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) J_2x3[i * 3 + j] = J(i, j);
  return (int)st;
}

// ---- ReprojectionError<G>::EvaluateWithMinimalJacobians (implementation/ReprojectionError.hpp:87-242) ----
// returns 1 (kept for signature symmetry with the oracle; validity is visible as zeroed Jacobians)
int ref_reprojection(const double pose[7], const double point[4], const double extr[7], const double intr[12], int model,
                     const double uv[2], const double sqrtInfo[4], double r[2], double* Jp_2x6, double* Jl_2x3,
                     double* Je_2x6) {
  std::shared_ptr<cam::CameraBase> c = make_camera(intr, model);
  std::shared_ptr<oc::ErrorInterface> e = make_reproj(c, model, uv, sqrtInfo);
  std::vector<const double*> params = {pose, point, extr};
  std::vector<double*> Jm = {Jp_2x6, Jl_2x3, Je_2x6};
  if (Jp_2x6 || Jl_2x3 || Je_2x6) {  // the reference writes the minimal Jacobians only for requested blocks
    double dump[12];
    for (size_t i = 0; i < 3; ++i)
      if (!Jm[i]) Jm[i] = dump;
  }
  evaluate(*e, params, r, Jm);
  return 1;
}

// squareRootInformation_ as the reference computes it from an information matrix
// (n = 2: ReprojectionError::setInformation, 6: PoseError::setInformation (PoseError.cpp:70-76),
//  9: SpeedAndBiasError::setInformation)
int ref_sqrt_information(const double* info, int n, double* out) {
  if (n == 6) {
    Eigen::Matrix<double, 6, 6> I;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) I(i, j) = info[i * 6 + j];
    PoseErrorAccess e(Transformation(), I);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) out[i * 6 + j] = e.sqrtInfo()(i, j);
    return 0;
  }
  if (n == 9) {
    Eigen::Matrix<double, 9, 9> I;
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) I(i, j) = info[i * 9 + j];
    SbErrorAccess e(okvis::SpeedAndBias::Zero(), I);
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) out[i * 9 + j] = e.sqrtInfo()(i, j);
    return 0;
  }
  return -1;
}

// ---- small priors ----
void ref_pose_error(const double pose[7], const double meas[7], const double sqrtInfo[36], double r[6], double* J_6x6) {
  PoseErrorAccess e(to_T(meas), Eigen::Matrix<double, 6, 6>::Identity());
  e.setSqrt(sqrtInfo);
  evaluate(e, {pose}, r, {J_6x6});
}
void ref_speedbias_error(const double sb[9], const double meas[9], const double sqrtInfo[81], double r[9], double* J_9x9) {
  okvis::SpeedAndBias m;
  for (int i = 0; i < 9; ++i) m[i] = meas[i];
  SbErrorAccess e(m, Eigen::Matrix<double, 9, 9>::Identity());
  e.setSqrt(sqrtInfo);
  evaluate(e, {sb}, r, {J_9x9});
}
void ref_relative_pose_error(const double p0[7], const double p1[7], const double sqrtInfo[36], double r[6], double* J0,
                             double* J1) {
  RelPoseErrorAccess e(Eigen::Matrix<double, 6, 6>::Identity());
  e.setSqrt(sqrtInfo);
  double dump[36];
  std::vector<double*> Jm = {J0, J1};
  if (J0 || J1)
    for (size_t i = 0; i < 2; ++i)
      if (!Jm[i]) Jm[i] = dump;
  evaluate(e, {p0, p1}, r, Jm);
}

// ---- ImuError (ImuError.cpp:76-284 redoPreintegration, :514-685 EvaluateWithMinimalJacobians) ----
static void imu_eval(ImuErrorAccess& e, const double* pose0, const double* sb0, const double* pose1, const double* sb1,
                     double* r, double* J0, double* J1, double* J2, double* J3) {
  double d0[90], d1[135], d2[90], d3[135];
  const bool any = J0 || J1 || J2 || J3;
  std::vector<double*> Jm;
  if (any) Jm = {J0 ? J0 : d0, J1 ? J1 : d1, J2 ? J2 : d2, J3 ? J3 : d3};
  evaluate(e, {pose0, sb0, pose1, sb1}, r, Jm);
}
int ref_imu_evaluate_fresh(int n, const int64_t* t, const double* gyr, const double* acc, const okvis_ba_imu_params* p,
                           int64_t t0, int64_t t1, const double pose0[7], const double sb0[9], const double pose1[7],
                           const double sb1[9], double r[15], double* J0, double* J1, double* J2, double* J3,
                           double* sqrtInfo_15x15) {
  ImuErrorAccess e(to_deque(n, t, gyr, acc), to_imu_params(p), to_time(t0), to_time(t1));
  imu_eval(e, pose0, sb0, pose1, sb1, r, J0, J1, J2, J3);
  if (sqrtInfo_15x15)
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) sqrtInfo_15x15[i * 15 + j] = e.sqrtInfo()(i, j);
  return e.redoCount();
}
int ref_imu_evaluate_at_ref(int n, const int64_t* t, const double* gyr, const double* acc, const okvis_ba_imu_params* p,
                            int64_t t0, int64_t t1, const double sb_ref[9], const double pose0[7], const double sb0[9],
                            const double pose1[7], const double sb1[9], double r[15], double* J0, double* J1, double* J2,
                            double* J3) {
  ImuErrorAccess e(to_deque(n, t, gyr, acc), to_imu_params(p), to_time(t0), to_time(t1));
  e.primeAt(sb_ref);
  imu_eval(e, pose0, sb0, pose1, sb1, r, J0, J1, J2, J3);
  return e.redoCount();  // 0 = the first-order bias correction path was taken (no redo inside Evaluate)
}
// static ImuError::propagation (ImuError.cpp:287-504)
int ref_imu_propagation(int n, const int64_t* t, const double* gyr, const double* acc, const okvis_ba_imu_params* p,
                        double T_WS[7], double sb[9], int64_t t_start, int64_t t_end, double* cov_15x15,
                        double* jac_15x15) {
  Transformation T = to_T(T_WS);
  okvis::SpeedAndBias s;
  for (int i = 0; i < 9; ++i) s[i] = sb[i];
  Eigen::Matrix<double, 15, 15> cov, jac;
  const int k = oc::ImuError::propagation(to_deque(n, t, gyr, acc), to_imu_params(p), T, s, to_time(t_start),
                                          to_time(t_end), cov_15x15 ? &cov : 0, jac_15x15 ? &jac : 0);
  for (int i = 0; i < 7; ++i) T_WS[i] = T.coeffs()[i];
  for (int i = 0; i < 9; ++i) sb[i] = s[i];
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      if (cov_15x15) cov_15x15[i * 15 + j] = cov(i, j);
      if (jac_15x15) jac_15x15[i * 15 + j] = jac(i, j);
    }
  return k;
}

// =====================================================================================================
// window level: the reference's Map filled from the flat window description
// =====================================================================================================
struct ref_window {
  std::shared_ptr<oc::Map> map;
  std::vector<std::shared_ptr<oc::PoseParameterBlock> > pose;
  std::vector<std::shared_ptr<oc::SpeedAndBiasParameterBlock> > sb;
  std::vector<std::shared_ptr<oc::HomogeneousPointParameterBlock> > lm;
  std::vector<std::shared_ptr<cam::CameraBase> > cams;
  std::shared_ptr< ::ceres::LossFunction> loss;
  std::vector< ::ceres::ResidualBlockId> residuals;  // insertion order: priors, relative, imu, [marg], reprojection
  std::vector<std::shared_ptr<ImuErrorAccess> > imu;
  std::shared_ptr<MargAccess> margPrior;
  std::map<uint64_t, std::pair<int, int> > id2block;  // id -> (type 0 pose / 1 sb / 2 lm, index)
};
static const uint64_t POSE_ID0 = 1, SB_ID0 = 1000000, LM_ID0 = 2000000;

ref_window* ref_window_create(const okvis_ba_window* w) {
  ref_window* h = new ref_window();
  h->map.reset(new oc::Map());
  for (int i = 0; i < w->n_pose; ++i) {
    std::shared_ptr<oc::PoseParameterBlock> b(new oc::PoseParameterBlock(Transformation(), POSE_ID0 + i, okvis::Time(0)));
    b->setParameters(w->pose + 7 * i);  // raw copy, exactly the window's doubles
    h->map->addParameterBlock(b, oc::Map::Pose6d);
    if (w->pose_fixed && w->pose_fixed[i]) h->map->setParameterBlockConstant(b->id());
    h->pose.push_back(b);
    h->id2block[b->id()] = std::make_pair(0, i);
  }
  for (int i = 0; i < w->n_sb; ++i) {
    std::shared_ptr<oc::SpeedAndBiasParameterBlock> b(
        new oc::SpeedAndBiasParameterBlock(okvis::SpeedAndBias::Zero(), SB_ID0 + i, okvis::Time(0)));
    b->setParameters(w->sb + 9 * i);
    h->map->addParameterBlock(b);
    if (w->sb_fixed && w->sb_fixed[i]) h->map->setParameterBlockConstant(b->id());
    h->sb.push_back(b);
    h->id2block[b->id()] = std::make_pair(1, i);
  }
  for (int i = 0; i < w->n_lm; ++i) {
    std::shared_ptr<oc::HomogeneousPointParameterBlock> b(
        new oc::HomogeneousPointParameterBlock(Eigen::Vector4d(0, 0, 0, 1), LM_ID0 + i));
    b->setParameters(w->lm + 4 * i);
    h->map->addParameterBlock(b, oc::Map::HomogeneousPoint);
    h->lm.push_back(b);
    h->id2block[b->id()] = std::make_pair(2, i);
  }
  for (int i = 0; i < w->n_cam; ++i) h->cams.push_back(make_camera(w->cam_intr + 12 * i, w->cam_model[i]));
  if (w->cauchy_b > 0) h->loss.reset(new ::ceres::CauchyLoss(w->cauchy_b));

  for (int i = 0; i < w->n_pprior; ++i) {
    std::shared_ptr<PoseErrorAccess> e(
        new PoseErrorAccess(to_T(w->pprior_meas + 7 * i), Eigen::Matrix<double, 6, 6>::Identity()));
    e->setSqrt(w->pprior_sqrtinfo + 36 * i);
    h->residuals.push_back(h->map->addResidualBlock(e, NULL, h->pose[w->pprior_pose[i]]));
  }
  for (int i = 0; i < w->n_sbprior; ++i) {
    okvis::SpeedAndBias m;
    for (int k = 0; k < 9; ++k) m[k] = w->sbprior_meas[9 * i + k];
    std::shared_ptr<SbErrorAccess> e(new SbErrorAccess(m, Eigen::Matrix<double, 9, 9>::Identity()));
    e->setSqrt(w->sbprior_sqrtinfo + 81 * i);
    h->residuals.push_back(h->map->addResidualBlock(e, NULL, h->sb[w->sbprior_sb[i]]));
  }
  for (int i = 0; i < w->n_relpose; ++i) {
    std::shared_ptr<RelPoseErrorAccess> e(new RelPoseErrorAccess(Eigen::Matrix<double, 6, 6>::Identity()));
    e->setSqrt(w->rel_sqrtinfo + 36 * i);
    h->residuals.push_back(h->map->addResidualBlock(e, NULL, h->pose[w->rel_pose0[i]], h->pose[w->rel_pose1[i]]));
  }
  const okvis::ImuParameters ip = to_imu_params(&w->imu_params);
  for (int i = 0; i < w->n_imu; ++i) {
    const int b = w->imu_s_begin[i], n = w->imu_s_count[i];
    std::shared_ptr<ImuErrorAccess> e(new ImuErrorAccess(to_deque(n, w->imu_s_t + b, w->imu_s_gyr + 3 * b, w->imu_s_acc + 3 * b),
                                                         ip, to_time(w->imu_t0[i]), to_time(w->imu_t1[i])));
    if (w->imu_sb_ref && w->imu_sb_ref_valid && w->imu_sb_ref_valid[i]) e->primeAt(w->imu_sb_ref + 9 * i);
    h->imu.push_back(e);
    h->residuals.push_back(h->map->addResidualBlock(e, NULL, h->pose[w->imu_pose0[i]], h->sb[w->imu_sb0[i]],
                                                    h->pose[w->imu_pose1[i]], h->sb[w->imu_sb1[i]]));
  }
  if (w->marg_dim > 0) {
    std::vector<std::shared_ptr<oc::ParameterBlock> > blocks;
    std::vector<int> offs;
    for (int i = 0; i < w->marg_nblocks; ++i) {
      if (w->marg_block_type[i] == OKVIS_BA_BLOCK_POSE) blocks.push_back(h->pose[w->marg_block_idx[i]]);
      else blocks.push_back(h->sb[w->marg_block_idx[i]]);
      offs.push_back(w->marg_block_off[i]);
    }
    h->margPrior.reset(new MargAccess(*h->map));
    h->margPrior->injectErrorTerm(blocks, offs, w->marg_dim, w->marg_J, w->marg_e0, w->marg_lin);
    h->residuals.push_back(h->map->addResidualBlock(h->margPrior, NULL, blocks));
  }
  for (int i = 0; i < w->n_obs; ++i) {
    const int c = w->obs_cam[i];
    const double s = w->obs_sqrtw[i];
    const double si[4] = {s, 0, 0, s};
    std::shared_ptr< ::ceres::CostFunction> cf;
    make_reproj(h->cams[c], w->cam_model[c], w->obs_uv + 2 * i, si, &cf);
    h->residuals.push_back(h->map->addResidualBlock(cf, h->loss.get(), h->pose[w->obs_pose[i]], h->lm[w->obs_lm[i]],
                                                    h->pose[w->obs_ext[i]]));
  }
  return h;
}
void ref_window_destroy(ref_window* h) { delete h; }

void ref_window_get_state(ref_window* h, double* pose, double* sb, double* lm) {
  for (size_t i = 0; pose && i < h->pose.size(); ++i) std::memcpy(pose + 7 * i, h->pose[i]->parameters(), 56);
  for (size_t i = 0; sb && i < h->sb.size(); ++i) std::memcpy(sb + 9 * i, h->sb[i]->parameters(), 72);
  for (size_t i = 0; lm && i < h->lm.size(); ++i) std::memcpy(lm + 4 * i, h->lm[i]->parameters(), 32);
}
void ref_window_set_state(ref_window* h, const double* pose, const double* sb, const double* lm) {
  for (size_t i = 0; pose && i < h->pose.size(); ++i) h->pose[i]->setParameters(pose + 7 * i);
  for (size_t i = 0; sb && i < h->sb.size(); ++i) h->sb[i]->setParameters(sb + 9 * i);
  for (size_t i = 0; lm && i < h->lm.size(); ++i) h->lm[i]->setParameters(lm + 4 * i);
}

// 0.5 * sum rho(|r|^2) over every residual block of the Map, each evaluated by its own reference class
// (cost-only path, jacobians == NULL) with the LossFunction it was added with.
double ref_window_cost(ref_window* h) {
  double cost = 0;
  for (size_t k = 0; k < h->residuals.size(); ++k) {
    std::shared_ptr<oc::ErrorInterface> e = h->map->errorInterfacePtr(h->residuals[k]);
    oc::Map::ParameterBlockCollection pc = h->map->parameters(h->residuals[k]);
    std::vector<const double*> params;
    for (size_t i = 0; i < pc.size(); ++i) params.push_back(pc[i].second->parameters());
    std::vector<double> r(e->residualDim());
    e->EvaluateWithMinimalJacobians(params.data(), r.data(), 0, 0);
    double s = 0;
    for (size_t i = 0; i < r.size(); ++i) s += r[i] * r[i];
    const ::ceres::LossFunction* loss = h->map->residualBlockId2ResidualBlockSpecMap().at(h->residuals[k]).lossFunctionPtr;
    if (loss) {
      double rho[3];
      loss->Evaluate(s, rho);
      cost += 0.5 * rho[0];
    } else {
      cost += 0.5 * s;
    }
  }
  return cost;
}

// un-robustified weighted residual of the k-th residual block (insertion order) -> r; returns its dimension
int ref_window_residual(ref_window* h, int k, double* r) {
  std::shared_ptr<oc::ErrorInterface> e = h->map->errorInterfacePtr(h->residuals[(size_t)k]);
  oc::Map::ParameterBlockCollection pc = h->map->parameters(h->residuals[(size_t)k]);
  std::vector<const double*> params;
  for (size_t i = 0; i < pc.size(); ++i) params.push_back(pc[i].second->parameters());
  e->EvaluateWithMinimalJacobians(params.data(), r, 0, 0);
  return (int)e->residualDim();
}
int ref_window_num_residual_blocks(ref_window* h) { return (int)h->residuals.size(); }

static void export_blocks(ref_window* h, const MargAccess& m, int capacity, int32_t* type, int32_t* idx, int32_t* off,
                          int32_t* nblocks) {
  int n = 0;
  for (size_t i = 0; i < m.nInfo(); ++i) {
    if (m.infoDim(i) == 0) continue;  // fixed block: no columns
    const std::pair<int, int> b = h->id2block.at(m.infoId(i));
    if (n < capacity) type[n] = b.first, idx[n] = b.second, off[n] = (int32_t)m.infoOff(i);
    ++n;
  }
  *nblocks = n;
}

// The reference's normal equations of the whole window: MarginalizationError::addResidualBlock(id, keep = true)
// of every residual block (MarginalizationError.cpp:127-435: minimal Jacobians, Ceres' loss corrector as restated
// there, H_ += J^T J, b0_ -= J^T r).  H [dim][dim] row-major, b0 [dim]; the block list gives the column order
// (type 0 pose / 1 speed-bias / 2 landmark).
int ref_window_full_system(ref_window* h, int capacity_dim, int capacity_blocks, int32_t* dim, double* H, double* b0,
                           int32_t* nblocks, int32_t* type, int32_t* idx, int32_t* off) {
  MargAccess m(*h->map);
  for (size_t k = 0; k < h->residuals.size(); ++k) m.addResidualBlock(h->residuals[k], true);
  const int n = (int)m.H().rows();
  *dim = n;
  export_blocks(h, m, capacity_blocks, type, idx, off, nblocks);
  if (n > capacity_dim || *nblocks > capacity_blocks) return -1;
  for (int i = 0; i < n; ++i) {
    b0[i] = m.b0()[i];
    for (int j = 0; j < n; ++j) H[(size_t)i * n + j] = m.H()(i, j);
  }
  return 0;
}

// The numeric core of Estimator::applyMarginalizationStrategy done by the reference's own MarginalizationError:
// previous prior (H_, b0_) -> addResidualBlock of every residual of the window -> marginalizeOut(all landmarks +
// flagged blocks) -> updateErrorComputation.  CONSUMES the window (residuals and marginalised blocks leave the Map).
int ref_window_marginalize(ref_window* h, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* res) {
  MargAccess m(*h->map);
  if (spec->prior_dim > 0) {
    std::vector<std::shared_ptr<oc::ParameterBlock> > blocks;
    std::vector<int> offs;
    for (int i = 0; i < spec->prior_nblocks; ++i) {
      if (spec->prior_block_type[i] == OKVIS_BA_BLOCK_POSE) blocks.push_back(h->pose[spec->prior_block_idx[i]]);
      else blocks.push_back(h->sb[spec->prior_block_idx[i]]);
      offs.push_back(spec->prior_block_off[i]);
    }
    m.injectPrior(blocks, offs, spec->prior_dim, spec->prior_H, spec->prior_b0);
  }
  for (size_t k = 0; k < h->residuals.size(); ++k) m.addResidualBlock(h->residuals[k], false);
  h->residuals.clear();
  std::vector<uint64_t> ids;
  std::vector<bool> keep;
  for (size_t i = 0; i < h->lm.size(); ++i)
    if (m.isParameterBlockConnected(h->lm[i]->id())) ids.push_back(h->lm[i]->id()), keep.push_back(false);
  for (size_t i = 0; i < h->pose.size(); ++i)
    if (spec->pose_marg && spec->pose_marg[i] && !h->pose[i]->fixed()) ids.push_back(h->pose[i]->id()), keep.push_back(false);
  for (size_t i = 0; i < h->sb.size(); ++i)
    if (spec->sb_marg && spec->sb_marg[i] && !h->sb[i]->fixed()) ids.push_back(h->sb[i]->id()), keep.push_back(false);
  m.marginalizeOut(ids, keep);
  m.updateErrorComputation();
  const int n = (int)m.H().rows();
  res->dim = n;
  export_blocks(h, m, res->capacity_blocks, res->block_type, res->block_idx, res->block_off, &res->nblocks);
  if (n > res->capacity_dim || res->nblocks > res->capacity_blocks) return -1;
  int rank = 0;
  for (int i = 0; i < n; ++i) {
    if (m.S()[i] > 0) ++rank;
    res->b0[i] = m.b0()[i];
    res->e0[i] = m.e0()[i];
    for (int j = 0; j < n; ++j) {
      res->H[(size_t)i * n + j] = m.H()(i, j);
      res->J[(size_t)i * n + j] = m.J()(i, j);
    }
  }
  res->rank = rank;
  res->sweeps[0] = res->sweeps[1] = 0;
  return 0;
}

// landmark quality exactly as Estimator::optimize computes it after the solve (Estimator.cpp:880-896):
// Map::getLhs (Map.cpp:101-156) + SelfAdjointEigenSolver<Matrix3d>
void ref_window_lm_quality(ref_window* h, double* quality, double* H_3x3 /* [n_lm][9] or NULL */) {
  for (size_t i = 0; i < h->lm.size(); ++i) {
    Eigen::MatrixXd H(3, 3);
    h->map->getLhs(h->lm[i]->id(), H);
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> saes(H);
    Eigen::Vector3d ev = saes.eigenvalues();
    const double smallest = ev[0], largest = ev[2];
    quality[i] = smallest < 1.0e-12 ? 0.0 : std::sqrt(smallest) / std::sqrt(largest);
    if (H_3x3)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H_3x3[9 * i + 3 * a + b] = H(a, b);
  }
}

void ref_window_imu_sb_ref(ref_window* h, double* out /* [n_imu][9] */) {
  for (size_t i = 0; i < h->imu.size(); ++i) h->imu[i]->sbRef(out + 9 * i);
}

// ::ceres::Solve on the Map's Problem with the options Estimator::optimize sets (Estimator.cpp:854-873).  The solver
// behind it is oracle/ref/ceres_shim_solve.cpp (this repository's statement of Ceres' TRUST_REGION / DOGLEG policy,
// NOT Ceres), but every residual and Jacobian it consumes is evaluated by the reference's classes.
void ref_window_optimize(ref_window* h, const okvis_ba_options* opt, int num_iter, int use_dogleg, okvis_ba_summary* out) {
  ::ceres::Solver::Options& o = h->map->options;
  o.linear_solver_type = ::ceres::SPARSE_SCHUR;
  o.trust_region_strategy_type = use_dogleg ? ::ceres::DOGLEG : ::ceres::LEVENBERG_MARQUARDT;
  o.max_num_iterations = num_iter;
  o.minimizer_progress_to_stdout = false;
  if (opt) {
    o.initial_trust_region_radius = opt->initial_radius;
    o.max_trust_region_radius = opt->max_radius;
    o.min_trust_region_radius = opt->min_radius;
    o.min_lm_diagonal = opt->min_lm_diagonal;
    o.max_lm_diagonal = opt->max_lm_diagonal;
    o.min_relative_decrease = opt->min_relative_decrease;
    o.function_tolerance = opt->function_tolerance;
    o.gradient_tolerance = opt->gradient_tolerance;
    o.parameter_tolerance = opt->parameter_tolerance;
    o.jacobi_scaling = opt->jacobi_scaling != 0;
    o.max_num_consecutive_invalid_steps = opt->max_consecutive_invalid_steps > 0 ? opt->max_consecutive_invalid_steps : 5;
  }
  h->map->solve();
  const ::ceres::Solver::Summary& s = h->map->summary;
  if (out) {
    std::memset(out, 0, sizeof(*out));
    out->initial_cost = s.initial_cost;
    out->final_cost = s.final_cost;
    // same conventions as okvis_ba_summary: the iteration in which a tolerance fired counts (Ceres does not record it)
    int term = 0, extra = 0;
    if (s.message.find("Function tolerance") == 0) term = 1, extra = 1;
    else if (s.message.find("Gradient tolerance") == 0) term = 2;
    else if (s.message.find("Parameter tolerance") == 0) term = 3, extra = 1;
    else if (s.message.find("Termination. Minimum trust region") == 0) term = 4, extra = 1;
    else if (s.message.find("Number of successive invalid") == 0) term = 5, extra = 1;
    out->iterations = (int32_t)s.iterations.size() - 1 + extra;
    out->successful_steps = s.num_successful_steps;
    out->termination = term;
    if (!s.iterations.empty()) {
      out->final_radius = s.iterations.back().trust_region_radius;
      out->gradient_max_norm = s.iterations.back().gradient_max_norm;
    }
  }
}

}  // extern "C"
