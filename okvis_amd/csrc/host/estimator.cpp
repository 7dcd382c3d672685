// okvis_amd::Estimator implementation — window book-keeping on the host, optimisation on the GPU through
// the C-ABI.  Citations are to the reference okvis_ceres/src/Estimator.cpp unless noted.
//
// Derived work: the decision logic of addStates / applyMarginalizationStrategy follows okvis_ceres/src/Estimator.cpp
// (Copyright (c) 2015, Autonomous Systems Lab / ETH Zurich, BSD 3-clause) because a drop-in must behave identically;
// see LICENSE for the attribution and the original licence text.
#include "estimator.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>

#include "../ba_math.hpp"

namespace okvis_amd {

namespace {

// squareRootInformation_ = LLT(information).matrixL().transpose() with Eigen's unblocked-LLT early exit on a
// non-positive pivot (PoseError.cpp:70-76 applied to diag(1e8,1e8,1e8,0,0,1e8), Estimator.cpp:240-242).
template <size_t N2>
void sqrtInformation(const std::array<double, N2>& info, int n, std::array<double, N2>& out) {
  std::array<double, N2> A = info;
  for (int k = 0; k < n; ++k) {
    double x = A[k * n + k];
    for (int j = 0; j < k; ++j) x -= A[k * n + j] * A[k * n + j];
    if (x <= 0.0) break;
    x = std::sqrt(x);
    A[k * n + k] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = A[i * n + k];
      for (int j = 0; j < k; ++j) s -= A[i * n + j] * A[k * n + j];
      A[i * n + k] = s / x;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[i * n + j] = (j >= i) ? A[j * n + i] : 0.0;
}

std::array<double, 36> poseSqrtInfo(double translationVariance, double rotationVariance) {
  std::array<double, 36> info{}, out{};
  for (int i = 0; i < 3; ++i) {
    info[i * 6 + i] = 1.0 / translationVariance;
    info[(3 + i) * 6 + 3 + i] = 1.0 / rotationVariance;
  }
  sqrtInformation(info, 6, out);
  return out;
}

void check(int status, const char* what) {
  if (status != OKVIS_BA_OK)
    throw Estimator::Exception(std::string("okvis_amd backend: ") + what + ": " + okvis_ba_error_string(status));
}

}  // namespace

Estimator::Estimator(int device) : device_(device) {
  okvis_ba_default_options(&options_);
  // Every optimize() / applyMarginalizationStrategy() uploads a new window structure, so a captured hipGraph is
  // replayed for one call only and then destroyed (measured: ~0.27 ms per frame for capture + destruction against
  // ~10 x 3 plain kernel launches that overlap with the GPU work anyway): eager launches by default here.
  options_.use_graph = 0;
  std::memset(&summary_, 0, sizeof(summary_));
  if (const char* e = std::getenv("OKVIS_AMD_DEBUG")) {   // print / cross-check diagnostics only (estimator.hpp, Diagnostics)
    const std::string w = std::string(",") + e + ",";
    diag_.trace = w.find(",trace,") != std::string::npos;
    diag_.checkPatch = w.find(",check_patch,") != std::string::npos;
    diag_.syncAfterHandover = w.find(",sync_after_handover,") != std::string::npos;
  }
  if (device < 0) {   // book-keeping only (estimator.hpp): nothing below this class computes
    dry_ = true;
    return;
  }
  // no GPU => hard failure: there is no CPU optimisation path behind this class
  check(okvis_ba_create(&solver_, device), "okvis_ba_create");
  check(okvis_ba_set_patchable(solver_, 1), "okvis_ba_set_patchable");   // the solver keeps the window between optimize() calls
}

Estimator::~Estimator() {
  if (priorPending_ && margSolver_) (void)okvis_ba_marginalize_end(margSolver_, &margRes_);   // (nobody reads the numbers any more)
  if (solver_) okvis_ba_destroy(solver_);
  if (margSolver_) okvis_ba_destroy(margSolver_);
  if (dryStore_) okvis_ba_store_destroy(dryStore_);
}

int Estimator::addCamera(const ExtrinsicsEstimationParameters& p) {
  extrinsicsEstimationParametersVec_.push_back(p);
  return (int)extrinsicsEstimationParametersVec_.size() - 1;  // Estimator.cpp:83-89
}
int Estimator::addImu(const ImuParameters& p) {
  if (imuParametersVec_.size() > 1) return -1;  // "only one IMU currently supported", Estimator.cpp:93-96
  imuParametersVec_.push_back(p);
  return (int)imuParametersVec_.size() - 1;
}
void Estimator::clearCameras() { extrinsicsEstimationParametersVec_.clear(); }
void Estimator::clearImus() { imuParametersVec_.clear(); }

const Estimator::State* Estimator::findState(uint64_t id) const {
  for (const State& s : states_)
    if (s.id == id) return &s;
  return nullptr;
}
Estimator::State* Estimator::findState(uint64_t id) {
  for (State& s : states_)
    if (s.id == id) return &s;
  return nullptr;
}

// ---------------------------------------------------------------------------------------------------
// static ImuError::propagation (ImuError.cpp:287-504), state output only (covariance/jacobian == 0 as in
// the addStates call, Estimator.cpp:145-147)
// ---------------------------------------------------------------------------------------------------
int Estimator::propagation(const ImuMeasurementDeque& m, const ImuParameters& prm, Transformation& T_WS,
                           SpeedAndBias& sb, int64_t t_start, int64_t t_end) {
  using namespace ba;
  if (m.empty() || !(m.back().t_ns >= t_end)) return -1;  // :301-302
  double q0[4] = {T_WS.p[3], T_WS.p[4], T_WS.p[5], T_WS.p[6]};
  qnormalize(q0);
  double C_WS_0[9];
  qrot(q0, C_WS_0);
  double Dq[4] = {0, 0, 0, 1};
  double acc_integral[3] = {0, 0, 0}, acc_doubleintegral[3] = {0, 0, 0};
  double Delta_t = 0;
  bool hasStarted = false;
  int i = 0;
  int64_t time = t_start;
  const size_t n = m.size();
  for (size_t it = 0; it < n; ++it) {
    double w0[3], a0[3], w1[3], a1[3];
    const size_t nx = (it + 1 < n) ? it + 1 : it;
    for (int c = 0; c < 3; ++c) {
      w0[c] = m[it].gyr[c];
      a0[c] = m[it].acc[c];
      w1[c] = m[nx].gyr[c];
      a1[c] = m[nx].acc[c];
    }
    int64_t nexttime = (it + 1 == n) ? t_end : m[it + 1].t_ns;
    double dt = ns_to_sec(nexttime - time);
    if (t_end < nexttime) {
      const double interval = ns_to_sec(nexttime - m[it].t_ns);
      nexttime = t_end;
      dt = ns_to_sec(nexttime - time);
      const double r = dt / interval;
      for (int c = 0; c < 3; ++c) {
        w1[c] = (1.0 - r) * w0[c] + r * w1[c];
        a1[c] = (1.0 - r) * a0[c] + r * a1[c];
      }
    }
    if (dt <= 0.0) continue;
    Delta_t += dt;
    if (!hasStarted) {
      hasStarted = true;
      const double r = dt / ns_to_sec(nexttime - m[it].t_ns);
      for (int c = 0; c < 3; ++c) {
        w0[c] = r * w0[c] + (1.0 - r) * w1[c];
        a0[c] = r * a0[c] + (1.0 - r) * a1[c];
      }
    }
    double om[3], ab[3];
    for (int c = 0; c < 3; ++c) {
      om[c] = 0.5 * (w0[c] + w1[c]) - sb[3 + c];
      ab[c] = 0.5 * (a0[c] + a1[c]) - sb[6 + c];
    }
    const double th = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]) * 0.5 * dt;
    const double sc = sinc(th);
    const double dq[4] = {sc * om[0] * 0.5 * dt, sc * om[1] * 0.5 * dt, sc * om[2] * 0.5 * dt, std::cos(th)};
    double Dq1[4], C[9], C1[9];
    qmul(Dq, dq, Dq1);
    qrot(Dq, C);
    qrot(Dq1, C1);
    double h[9], t3[3];
    for (int c = 0; c < 9; ++c) h[c] = C[c] + C1[c];
    mat3_vec(h, ab, t3);
    for (int c = 0; c < 3; ++c) {
      acc_doubleintegral[c] += acc_integral[c] * dt + 0.25 * t3[c] * dt * dt;
      acc_integral[c] += 0.5 * t3[c] * dt;
    }
    for (int c = 0; c < 4; ++c) Dq[c] = Dq1[c];
    time = nexttime;
    ++i;
    if (nexttime == t_end) break;
  }
  // :470-477
  double r_new[3], tmp[3];
  mat3_vec(C_WS_0, acc_doubleintegral, tmp);
  for (int c = 0; c < 3; ++c) {
    const double gW = (c == 2) ? prm.g : 0.0;
    r_new[c] = T_WS.p[c] + sb[c] * Delta_t + tmp[c] - 0.5 * gW * Delta_t * Delta_t;
  }
  double qn[4];
  qmul(q0, Dq, qn);
  qnormalize(qn);
  mat3_vec(C_WS_0, acc_integral, tmp);
  for (int c = 0; c < 3; ++c) {
    const double gW = (c == 2) ? prm.g : 0.0;
    sb[c] += tmp[c] - gW * Delta_t;
    T_WS.p[c] = r_new[c];
  }
  for (int c = 0; c < 4; ++c) T_WS.p[3 + c] = qn[c];
  return i;
}

// Estimator.cpp:811-840
bool Estimator::initPoseFromImu(const ImuMeasurementDeque& imuMeasurements, Transformation& T_WS) {
  T_WS = Transformation();
  if (imuMeasurements.empty()) return false;
  double acc_B[3] = {0, 0, 0};
  for (const ImuMeasurement& m : imuMeasurements)
    for (int c = 0; c < 3; ++c) acc_B[c] += m.acc[c];
  for (int c = 0; c < 3; ++c) acc_B[c] /= double(imuMeasurements.size());
  const double n = std::sqrt(acc_B[0] * acc_B[0] + acc_B[1] * acc_B[1] + acc_B[2] * acc_B[2]);
  const double e[3] = {acc_B[0] / n, acc_B[1] / n, acc_B[2] / n};
  // poseIncrement.tail<3>() = ez_W.cross(e_acc).normalized() * acos(ez_W . e_acc); T_WS.oplus(-poseIncrement)
  double ax[3] = {-e[1], e[0], 0.0};  // (0,0,1) x e
  const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1]);
  const double angle = std::acos(e[2]);
  double d[6] = {0, 0, 0, 0, 0, 0};
  if (an > 0)
    for (int c = 0; c < 3; ++c) d[3 + c] = -ax[c] / an * angle;
  double out[7];
  ba::pose_oplus(T_WS.p.data(), d, out);
  for (int c = 0; c < 7; ++c) T_WS.p[c] = out[c];
  return true;
}

// ---------------------------------------------------------------------------------------------------
// addStates (Estimator.cpp:110-343)
// ---------------------------------------------------------------------------------------------------
bool Estimator::addStates(MultiFramePtr multiFrame, const ImuMeasurementDeque& imuMeasurements, bool asKeyframe) {
  lastRefusal_.clear();
  if (!multiFrame || imuParametersVec_.empty()) return refuse("addStates: no multiframe or no IMU parameters");
  Transformation T_WS;
  SpeedAndBias speedAndBias{};
  if (states_.empty()) {
    if (!initPoseFromImu(imuMeasurements, T_WS)) return refuse("addStates: initPoseFromImu failed (no IMU measurements)");  // :121-125
    for (int c = 0; c < 3; ++c) speedAndBias[6 + c] = imuParametersVec_.at(0).a0[c];  // :126-127
  } else {
    const State& last = states_.back();
    if (last.sbBlock < 0) return refuse("addStates: the previous state has no speed/bias block");
    T_WS.p = poseBlocks_[last.poseBlock].x;
    speedAndBias = sbBlocks_[last.sbBlock].x;
    const int used = propagation(imuMeasurements, imuParametersVec_.at(0), T_WS, speedAndBias, last.t_ns,
                                 multiFrame->t_ns);  // :145-147
    if (used < 1)
      return refuse("addStates: propagation used " + std::to_string(used) + " of " + std::to_string(imuMeasurements.size()) +
                    " IMU measurements between " + std::to_string(last.t_ns) + " and " + std::to_string(multiFrame->t_ns) +
                    (imuMeasurements.empty() ? std::string() : " (measurements " + std::to_string(imuMeasurements.front().t_ns) + " .. " +
                                                                    std::to_string(imuMeasurements.back().t_ns) + ")"));                    // :150-153
  }
  if (findState(multiFrame->id)) return refuse("addStates: pose id " + std::to_string(multiFrame->id) + " was used before");  // "pose ID was used before" (:161-163)
  // the reference orders its states by frame id (statesMap_, std::map) and takes rbegin() as the previous one; ids come from
  // IdProvider and increase.  This class keeps insertion order, which is the same thing only for increasing ids: enforced.
  if (!states_.empty() && multiFrame->id <= states_.back().id)
    return refuse("addStates: frame id " + std::to_string(multiFrame->id) + " does not follow " + std::to_string(states_.back().id));
  // nothing is changed before every input is known to be usable (extrinsics of the cameras that get a block of their own)
  for (size_t i = 0; i < extrinsicsEstimationParametersVec_.size(); ++i) {
    const ExtrinsicsEstimationParameters& ep = extrinsicsEstimationParametersVec_[i];
    const bool shared = (ep.sigma_c_relative_translation < 1e-12 || ep.sigma_c_relative_orientation < 1e-12) && !states_.empty();
    if (!shared && i >= multiFrame->T_SC.size()) return refuse("addStates: the multiframe has no extrinsics for camera " + std::to_string(i));
  }

  State st;
  st.id = multiFrame->id;
  st.t_ns = multiFrame->t_ns;
  st.isKeyframe = asKeyframe;
  st.poseBlock = (int)poseBlocks_.size();
  poseBlocks_.push_back(PoseBlock{T_WS.p, false, st.id});
  const bool first = states_.empty();
  // camera extrinsics (:191-218)
  for (size_t i = 0; i < extrinsicsEstimationParametersVec_.size(); ++i) {
    const ExtrinsicsEstimationParameters& ep = extrinsicsEstimationParametersVec_[i];
    if ((ep.sigma_c_relative_translation < 1e-12 || ep.sigma_c_relative_orientation < 1e-12) && !first) {
      st.extBlocks.push_back(states_.back().extBlocks.at(i));  // use the same block
    } else {
      if (i >= multiFrame->T_SC.size()) return refuse("addStates: the multiframe has no extrinsics for camera " + std::to_string(i));
      st.extBlocks.push_back((int)poseBlocks_.size());
      poseBlocks_.push_back(PoseBlock{multiFrame->T_SC[i].p, false, nextId_++});
    }
  }
  st.sbBlock = (int)sbBlocks_.size();
  sbBlocks_.push_back(SbBlock{speedAndBias, false, nextId_++});  // :222-235

  if (first) {
    // pose prior with information diag(1e8,1e8,1e8,0,0,1e8) (:238-243)
    std::array<double, 36> info{}, si{};
    info[0] = info[7] = info[14] = 1.0e8;
    info[35] = 1.0e8;
    sqrtInformation(info, 6, si);
    posePriors_.push_back(PosePrior{st.poseBlock, T_WS.p, si});
    familiesChanged_ |= OKVIS_BA_PATCH_POSE_PRIORS | OKVIS_BA_PATCH_SB_PRIORS;
    for (size_t i = 0; i < extrinsicsEstimationParametersVec_.size(); ++i) {  // :247-268
      const ExtrinsicsEstimationParameters& ep = extrinsicsEstimationParametersVec_[i];
      const double tv = ep.sigma_absolute_translation * ep.sigma_absolute_translation;
      const double rv = ep.sigma_absolute_orientation * ep.sigma_absolute_orientation;
      if (tv > 1.0e-16 && rv > 1.0e-16)
        posePriors_.push_back(PosePrior{st.extBlocks[i], poseBlocks_[st.extBlocks[i]].x, poseSqrtInfo(tv, rv)});
      else
        poseBlocks_[st.extBlocks[i]].fixed = true;  // setParameterBlockConstant
    }
    {  // speed and bias prior (:269-284)
      const ImuParameters& ip = imuParametersVec_.at(0);
      std::array<double, 81> info9{}, si9{};
      for (int c = 0; c < 3; ++c) {
        info9[c * 9 + c] = 1.0;
        info9[(3 + c) * 9 + 3 + c] = 1.0 / (ip.sigma_bg * ip.sigma_bg);
        info9[(6 + c) * 9 + 6 + c] = 1.0 / (ip.sigma_ba * ip.sigma_ba);
      }
      sqrtInformation(info9, 9, si9);
      sbPriors_.push_back(SbPrior{st.sbBlock, speedAndBias, si9});
    }
  } else {
    const State& last = states_.back();
    imuFactors_.push_back(ImuFactor{nextImuUid_++, last.poseBlock, last.sbBlock, st.poseBlock, st.sbBlock, last.t_ns, st.t_ns,
                                    imuMeasurements, {}, false});  // :288-307
    for (size_t i = 0; i < extrinsicsEstimationParametersVec_.size(); ++i) {           // :310-336
      if (last.extBlocks[i] != st.extBlocks[i]) {
        const ExtrinsicsEstimationParameters& ep = extrinsicsEstimationParametersVec_[i];
        const double dt = ba::ns_to_sec(st.t_ns - last.t_ns);
        familiesChanged_ |= OKVIS_BA_PATCH_RELPOSE;
        relPoses_.push_back(RelPose{last.extBlocks[i], st.extBlocks[i],
                                    poseSqrtInfo(ep.sigma_c_relative_translation * ep.sigma_c_relative_translation * dt,
                                                 ep.sigma_c_relative_orientation * ep.sigma_c_relative_orientation * dt)});
      }
    }
  }
  states_.push_back(st);
  multiFramePtrMap_[st.id] = multiFrame;
  return true;
}

// Estimator.cpp:346-365
bool Estimator::addLandmark(uint64_t landmarkId, const std::array<double, 4>& landmark) {
  if (landmarksMap_.count(landmarkId)) return false;  // Map::addParameterBlock fails on a duplicate id
  MapPoint mp;
  mp.id = landmarkId;
  mp.point = landmark;
  mp.quality = 0.0;
  mp.distance = 1.7976931348623157e308;
  if (std::fabs(landmark[3]) > 1.0e-8) {
    const double x = landmark[0] / landmark[3], y = landmark[1] / landmark[3], z = landmark[2] / landmark[3];
    mp.distance = std::sqrt(x * x + y * y + z * z);
  }
  landmarksMap_[landmarkId] = mp;
  landmarkInitialized_[landmarkId] = false;
  return true;
}

// implementation/Estimator.hpp:43-90
uint64_t Estimator::addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) {
  auto lit = landmarksMap_.find(landmarkId);
  if (lit == landmarksMap_.end()) throw Exception("landmark not added");
  const KeypointIdentifier kid{poseId, camIdx, keypointIdx};
  if (lit->second.observations.count(kid)) return 0;  // duplicate -> NULL (:52-56)
  auto mf = multiFramePtrMap_.find(poseId);
  if (mf == multiFramePtrMap_.end() || camIdx >= mf->second->keypoints.size() ||
      keypointIdx >= mf->second->keypoints[camIdx].size())
    throw Exception("addObservation: unknown frame / camera / keypoint");
  const Keypoint& kp = mf->second->keypoints[camIdx][keypointIdx];
  Observation o;
  o.handle = nextHandle_++;
  o.landmarkId = landmarkId;
  o.poseId = poseId;
  o.camIdx = camIdx;
  o.keypointIdx = keypointIdx;
  o.u = (double)kp.x;  // float -> double (:60-61)
  o.v = (double)kp.y;
  o.sqrtw = 8.0 / (double)kp.size;  // information = 64/size^2 * I (:62-65)
  {
    const State* st = findState(poseId);
    if (!st || camIdx >= st->extBlocks.size()) throw Exception("addObservation: unknown frame / camera");
    o.poseBlock = st->poseBlock;
    o.extBlock = st->extBlocks[camIdx];
  }
  observations_.insert(o);
  lit->second.observations[kid] = o.handle;
  touch(lit->second);
  if (synced_.valid) {
    obsAdded_.push_back(PendingObs{&lit->second, o.handle, o.poseBlock, o.extBlock, (int)camIdx, o.u, o.v, o.sqrtw});
    lit->second.pendingAdds++;
  }
  return o.handle;
}

// Estimator.cpp:368-413
bool Estimator::removeObservation(uint64_t handle) {
  const Observation* found = observations_.find(handle);
  if (!found) return false;
  MapPoint& mp = landmarksMap_.at(found->landmarkId);
  for (auto oit = mp.observations.begin(); oit != mp.observations.end();) {
    if (oit->second == handle)
      oit = mp.observations.erase(oit);
    else
      ++oit;
  }
  noteObservationRemoved(mp, handle);
  observations_.erase(handle);
  return true;
}
bool Estimator::removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) {
  auto lit = landmarksMap_.find(landmarkId);
  if (lit == landmarksMap_.end()) throw Exception("landmark not added");
  auto oit = lit->second.observations.find(KeypointIdentifier{poseId, camIdx, keypointIdx});
  if (oit == lit->second.observations.end()) return false;  // observation not present
  noteObservationRemoved(lit->second, oit->second);
  observations_.erase(oit->second);
  lit->second.observations.erase(oit);
  return true;
}

// Estimator.cpp:909-929
bool Estimator::setOptimizationTimeLimit(double timeLimit, int minIterations) {
  timeLimit_ = timeLimit;
  minIterations_ = minIterations;
  hasTimeLimit_ = true;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// flat window for the C-ABI
// ---------------------------------------------------------------------------------------------------
Estimator::WindowSel Estimator::selectAll() const {
  WindowSel sel;
  for (size_t i = 0; i < poseBlocks_.size(); ++i)
    if (poseBlocks_[i].alive) sel.pose.push_back((int)i);
  for (size_t i = 0; i < sbBlocks_.size(); ++i)
    if (sbBlocks_[i].alive) sel.sb.push_back((int)i);
  // a landmark nobody observes has no residual block: Ceres drops such parameter blocks from the program it solves (they
  // keep their value); optimize() gives them the quality 0 the reference derives from their zero H (Estimator.cpp:890-893)
  sel.landmarks.reserve(landmarksMap_.size());
  sel.lmPtr.reserve(landmarksMap_.size());
  for (const auto& kv : landmarksMap_)
    if (!kv.second.observations.empty()) {
      sel.landmarks.push_back(kv.first);
      sel.lmPtr.push_back(&kv.second);
    }
  sel.allObservations = true;   // (every observation belongs to a landmark of the map: nothing else to list)
  for (size_t i = 0; i < imuFactors_.size(); ++i) sel.imu.push_back((int)i);
  for (size_t i = 0; i < posePriors_.size(); ++i) sel.pprior.push_back((int)i);
  for (size_t i = 0; i < sbPriors_.size(); ++i) sel.sbprior.push_back((int)i);
  for (size_t i = 0; i < relPoses_.size(); ++i) sel.rel.push_back((int)i);
  sel.withPrior = true;
  return sel;
}

void Estimator::resolvePrior() const {
  if (!priorPending_) return;
  priorPending_ = false;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = okvis_ba_marginalize_end(margSolver_, &margRes_);
  if (debugFailPending_) {
    debugFailPending_ = false;
    rc = OKVIS_BA_ERR_NUMERIC;
  }
  if (rc != OKVIS_BA_OK) {
    // (the prior has its blocks but will never have its numbers: without it the estimator stays usable, if poorer)
    prior_ = MargPrior();   // (the family is still flagged as changed: applyMarginalizationStrategy did that when it set the blocks)
    throw Exception(std::string("okvis_amd::Estimator: the marginalisation enqueued by applyMarginalizationStrategy failed (") +
                    okvis_ba_error_string(rc) + "); its deletions cannot be taken back, the prior is dropped");
  }
  const size_t n = (size_t)margRes_.dim;
  if (prior_.dim > 0) {   // (a prior without residuals was dropped at once, Estimator.cpp:747-749)
    prior_.H.assign(margH_.begin(), margH_.begin() + n * n);
    prior_.b0.assign(margB_.begin(), margB_.begin() + n);
    prior_.J.assign(margJ_.begin(), margJ_.begin() + n * n);
    prior_.e0.assign(margE_.begin(), margE_.begin() + n);
  }
  margInfo_[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  margInfo_[3] = (double)margRes_.sweeps[0];
  margInfo_[4] = (double)margRes_.sweeps[1];
}

void Estimator::flatten(const WindowSel& sel, FlatWindow& fw) const {
  if (sel.withPrior) resolvePrior();
  okvis_ba_window& w = fw.w;
  std::memset(&w, 0, sizeof(w));
  auto& f64 = fw.f64;
  auto& i32 = fw.i32;
  auto& i64 = fw.i64;
  auto& u8 = fw.u8;
  f64.assign(24, {});
  i32.assign(24, {});
  i64.assign(4, {});
  u8.assign(3, {});
  enum { F_POSE, F_SB, F_LM, F_INTR, F_UV, F_SW, F_GYR, F_ACC, F_PPM, F_PPS, F_SBM, F_SBS, F_RELS, F_MJ, F_ME, F_ML, F_SBREF, F_ICACHE };
  enum { I_MODEL, I_OLM, I_OPOSE, I_OEXT, I_OCAM, I_IP0, I_IS0, I_IP1, I_IS1, I_SB, I_SC, I_PPP, I_SBP, I_R0, I_R1,
         I_MT, I_MI, I_MO };
  // parameter blocks (values: estimate, or the linearisation point of prior-connected blocks)
  fw.poseMap.assign(poseBlocks_.size(), -1);
  fw.sbMap.assign(sbBlocks_.size(), -1);
  std::vector<const double*> poseLin(poseBlocks_.size(), nullptr), sbLin(sbBlocks_.size(), nullptr);
  if (sel.atLinearizationPoint)
    for (size_t k = 0; k < prior_.block.size(); ++k)
      (prior_.type[k] == OKVIS_BA_BLOCK_POSE ? poseLin : sbLin)[prior_.block[k]] = prior_.lin[k].data();
  for (int b : sel.pose) {
    fw.poseMap[b] = (int)u8[0].size();
    const double* x = poseLin[b] ? poseLin[b] : poseBlocks_[b].x.data();
    f64[F_POSE].insert(f64[F_POSE].end(), x, x + 7);
    u8[0].push_back(poseBlocks_[b].fixed ? 1 : 0);
  }
  for (int b : sel.sb) {
    fw.sbMap[b] = (int)u8[1].size();
    const double* x = sbLin[b] ? sbLin[b] : sbBlocks_[b].x.data();
    f64[F_SB].insert(f64[F_SB].end(), x, x + 9);
    u8[1].push_back(sbBlocks_[b].fixed ? 1 : 0);
  }
  const bool walk = sel.allObservations && sel.lmPtr.size() == sel.landmarks.size();
  std::unordered_map<uint64_t, int> lmIndex;
  if (!walk) lmIndex.reserve(2 * sel.landmarks.size() + 1);
  f64[F_LM].reserve(4 * sel.landmarks.size());
  for (size_t n = 0; n < sel.landmarks.size(); ++n) {
    if (!walk) lmIndex[sel.landmarks[n]] = (int)n;
    const MapPoint& mp = walk ? *sel.lmPtr[n] : landmarksMap_.at(sel.landmarks[n]);
    f64[F_LM].insert(f64[F_LM].end(), mp.point.begin(), mp.point.end());
  }
  const size_t nLandmarks = sel.landmarks.size();
  // cameras: one intrinsics record per camera index of the newest multiframe
  size_t ncam = 0;
  if (!states_.empty()) {
    const MultiFramePtr& mf = multiFramePtrMap_.at(states_.back().id);
    ncam = mf->geometry.size();
    for (const CameraGeometry& g : mf->geometry) {
      f64[F_INTR].insert(f64[F_INTR].end(), g.intr.begin(), g.intr.end());
      i32[I_MODEL].push_back(g.model);
    }
  }
  // observations sorted by (landmark index, pose block, camera)
  struct Rec {
    int lm, pose, ext, cam;
    double u, v, sw;
    uint64_t handle;
  };
  std::vector<Rec> recs;
  auto byPoseCam = [](const Rec& a, const Rec& b) {
    if (a.pose != b.pose) return a.pose < b.pose;
    return a.cam < b.cam;
  };
  if (walk) {
    // landmark by landmark through its own observation map: the records of one landmark arrive together and in (frame, camera,
    // keypoint) order, which is (pose index, camera) order because pose blocks are created in frame order
    recs.reserve(observations_.size());
    for (size_t n = 0; n < nLandmarks; ++n) {
      const size_t first = recs.size();
      for (const auto& ob : sel.lmPtr[n]->observations) {
        const Observation& o = observations_.at(ob.second);
        if (o.camIdx >= ncam) continue;
        const int ip = fw.poseMap[o.poseBlock], ie = fw.poseMap[o.extBlock];
        if (ip < 0 || ie < 0) throw Exception("flatten: observation refers to a block outside the window");
        recs.push_back(Rec{(int)n, ip, ie, (int)o.camIdx, o.u, o.v, o.sqrtw, o.handle});
      }
      if (!std::is_sorted(recs.begin() + first, recs.end(), byPoseCam)) std::stable_sort(recs.begin() + first, recs.end(), byPoseCam);
    }
  } else {
    recs.reserve(sel.obs.size());
    for (uint64_t hnd : sel.obs) {
      const Observation& o = observations_.at(hnd);
      if (o.camIdx >= ncam) continue;
      auto li = lmIndex.find(o.landmarkId);
      if (li == lmIndex.end()) continue;
      const int ip = fw.poseMap[o.poseBlock], ie = fw.poseMap[o.extBlock];
      if (ip < 0 || ie < 0) throw Exception("flatten: observation refers to a block outside the window");
      recs.push_back(Rec{li->second, ip, ie, (int)o.camIdx, o.u, o.v, o.sqrtw, o.handle});
    }
  }
  if (!walk) {
    // counting sort by landmark (dense indices), then the handful of observations of each landmark by (pose, camera):
    // linear instead of n log n comparisons on the whole list
    const size_t nl = nLandmarks;
    std::vector<int> start(nl + 1, 0);
    for (const Rec& r : recs) ++start[r.lm + 1];
    for (size_t l = 0; l < nl; ++l) start[l + 1] += start[l];
    std::vector<Rec> sorted(recs.size());
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (const Rec& r : recs) sorted[fill[r.lm]++] = r;
    for (size_t l = 0; l < nl; ++l)
      std::sort(sorted.begin() + start[l], sorted.begin() + start[l + 1], byPoseCam);
    recs.swap(sorted);
  }
  i32[I_OLM].reserve(recs.size());
  i32[I_OPOSE].reserve(recs.size());
  i32[I_OEXT].reserve(recs.size());
  i32[I_OCAM].reserve(recs.size());
  f64[F_UV].reserve(2 * recs.size());
  f64[F_SW].reserve(recs.size());
  fw.obsHandle.clear();
  fw.obsHandle.reserve(recs.size());
  for (const Rec& r : recs) {
    fw.obsHandle.push_back(r.handle);
    i32[I_OLM].push_back(r.lm);
    i32[I_OPOSE].push_back(r.pose);
    i32[I_OEXT].push_back(r.ext);
    i32[I_OCAM].push_back(r.cam);
    f64[F_UV].push_back(r.u);
    f64[F_UV].push_back(r.v);
    f64[F_SW].push_back(r.sw);
  }
  // IMU factors
  for (int fi : sel.imu) {
    const ImuFactor& f = imuFactors_[fi];
    const int p0 = fw.poseMap[f.pose0Block], s0 = fw.sbMap[f.sb0Block], p1 = fw.poseMap[f.pose1Block],
              s1 = fw.sbMap[f.sb1Block];
    if (p0 < 0 || s0 < 0 || p1 < 0 || s1 < 0) throw Exception("flatten: IMU factor refers to a block outside the window");
    i32[I_IP0].push_back(p0);
    i32[I_IS0].push_back(s0);
    i32[I_IP1].push_back(p1);
    i32[I_IS1].push_back(s1);
    i64[0].push_back(f.t0);
    i64[1].push_back(f.t1);
    i32[I_SB].push_back((int)i64[2].size());
    i32[I_SC].push_back((int)f.meas.size());
    f64[F_SBREF].insert(f64[F_SBREF].end(), f.sbRef.begin(), f.sbRef.end());
    u8[2].push_back(f.hasRef ? (f.hasCache ? 2 : 1) : 0);
    if (f.hasCache) f64[F_ICACHE].insert(f64[F_ICACHE].end(), f.cache.begin(), f.cache.end());
    else f64[F_ICACHE].resize(f64[F_ICACHE].size() + OKVIS_BA_IMU_CACHE_DOUBLES, 0.0);
    for (const ImuMeasurement& m : f.meas) {
      i64[2].push_back(m.t_ns);
      f64[F_GYR].insert(f64[F_GYR].end(), m.gyr.begin(), m.gyr.end());
      f64[F_ACC].insert(f64[F_ACC].end(), m.acc.begin(), m.acc.end());
    }
  }
  for (int k : sel.pprior) {
    const PosePrior& p = posePriors_[k];
    i32[I_PPP].push_back(fw.poseMap[p.block]);
    f64[F_PPM].insert(f64[F_PPM].end(), p.meas.begin(), p.meas.end());
    f64[F_PPS].insert(f64[F_PPS].end(), p.sqrtInfo.begin(), p.sqrtInfo.end());
  }
  for (int k : sel.sbprior) {
    const SbPrior& p = sbPriors_[k];
    i32[I_SBP].push_back(fw.sbMap[p.block]);
    f64[F_SBM].insert(f64[F_SBM].end(), p.meas.begin(), p.meas.end());
    f64[F_SBS].insert(f64[F_SBS].end(), p.sqrtInfo.begin(), p.sqrtInfo.end());
  }
  for (int k : sel.rel) {
    const RelPose& r = relPoses_[k];
    i32[I_R0].push_back(fw.poseMap[r.block0]);
    i32[I_R1].push_back(fw.poseMap[r.block1]);
    f64[F_RELS].insert(f64[F_RELS].end(), r.sqrtInfo.begin(), r.sqrtInfo.end());
  }
  w.n_pose = (int)u8[0].size(); w.pose = f64[F_POSE].data(); w.pose_fixed = u8[0].data();
  w.n_sb = (int)u8[1].size(); w.sb = f64[F_SB].data(); w.sb_fixed = u8[1].data();
  w.n_lm = (int)nLandmarks; w.lm = f64[F_LM].data();
  w.n_cam = (int)ncam; w.cam_intr = f64[F_INTR].data(); w.cam_model = i32[I_MODEL].data();
  w.n_obs = (int)recs.size();
  w.obs_lm = i32[I_OLM].data(); w.obs_pose = i32[I_OPOSE].data(); w.obs_ext = i32[I_OEXT].data(); w.obs_cam = i32[I_OCAM].data();
  w.obs_uv = f64[F_UV].data(); w.obs_sqrtw = f64[F_SW].data();
  w.cauchy_b = 1.0;  // cauchyLossFunctionPtr_(new ::ceres::CauchyLoss(1)), Estimator.cpp:60
  w.n_imu = (int)i32[I_IP0].size();
  w.imu_pose0 = i32[I_IP0].data(); w.imu_sb0 = i32[I_IS0].data(); w.imu_pose1 = i32[I_IP1].data(); w.imu_sb1 = i32[I_IS1].data();
  w.imu_t0 = i64[0].data(); w.imu_t1 = i64[1].data(); w.imu_s_begin = i32[I_SB].data(); w.imu_s_count = i32[I_SC].data();
  w.n_imu_samples = (int)i64[2].size(); w.imu_s_t = i64[2].data(); w.imu_s_gyr = f64[F_GYR].data(); w.imu_s_acc = f64[F_ACC].data();
  w.imu_sb_ref = f64[F_SBREF].data(); w.imu_sb_ref_valid = u8[2].data(); w.imu_cache = f64[F_ICACHE].data();
  if (!imuParametersVec_.empty()) {
    const ImuParameters& ip = imuParametersVec_[0];
    w.imu_params = okvis_ba_imu_params{ip.sigma_g_c, ip.sigma_a_c, ip.sigma_gw_c, ip.sigma_aw_c, ip.g, ip.g_max, ip.a_max};
  }
  w.n_pprior = (int)i32[I_PPP].size(); w.pprior_pose = i32[I_PPP].data(); w.pprior_meas = f64[F_PPM].data(); w.pprior_sqrtinfo = f64[F_PPS].data();
  w.n_sbprior = (int)i32[I_SBP].size(); w.sbprior_sb = i32[I_SBP].data(); w.sbprior_meas = f64[F_SBM].data(); w.sbprior_sqrtinfo = f64[F_SBS].data();
  w.n_relpose = (int)i32[I_R0].size(); w.rel_pose0 = i32[I_R0].data(); w.rel_pose1 = i32[I_R1].data(); w.rel_sqrtinfo = f64[F_RELS].data();
  w.marg_dim = 0;
  if (sel.withPrior && prior_.dim > 0) {  // the MarginalizationError residual block (Estimator.cpp:750-759)
    int off = 0;
    for (size_t k = 0; k < prior_.block.size(); ++k) {
      const bool pose = prior_.type[k] == OKVIS_BA_BLOCK_POSE;
      const int wi = pose ? fw.poseMap[prior_.block[k]] : fw.sbMap[prior_.block[k]];
      if (wi < 0) throw Exception("flatten: marginalisation prior refers to a removed block");
      i32[I_MT].push_back(prior_.type[k]);
      i32[I_MI].push_back(wi);
      i32[I_MO].push_back(off);
      off += pose ? 6 : 9;
      f64[F_ML].insert(f64[F_ML].end(), prior_.lin[k].begin(), prior_.lin[k].end());
    }
    f64[F_MJ] = prior_.J;
    f64[F_ME] = prior_.e0;
    w.marg_dim = prior_.dim;
    w.marg_nblocks = (int)prior_.block.size();
    w.marg_block_type = i32[I_MT].data(); w.marg_block_idx = i32[I_MI].data(); w.marg_block_off = i32[I_MO].data();
    w.marg_J = f64[F_MJ].data(); w.marg_e0 = f64[F_ME].data(); w.marg_lin = f64[F_ML].data();
  }
}

// ---------------------------------------------------------------------------------------------------
// optimize (Estimator.cpp:843-906)
// ---------------------------------------------------------------------------------------------------
void Estimator::optimize(size_t numIter, size_t /*numThreads*/, bool verbose) {
  if (states_.empty()) return;
  typedef std::chrono::steady_clock clk;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const bool trace = diag_.trace, checkPatch = diag_.checkPatch;
  const auto t0 = clk::now();
  if (!dry_) check(okvis_ba_set_options(solver_, &options_), "set_options");
  // ---- the window: the edits since the last call as one patch of the window the solver holds, else flatten + upload ----
  patchSplit_ = {0.0, 0.0};
  try {
    lastWasPatch_ = usePatch_ && patchWindow();
  } catch (...) {
    // (the description of the edits may have been half consumed — e.g. the numbers of a pending marginalisation failed to
    //  arrive in the middle of it: the next call describes the window from scratch)
    invalidateSynced();
    throw;
  }
  if (!lastWasPatch_) {
    FlatWindow fw;
    uploadWindow(fw);
    if (trace)
      for (size_t a = 0; a < fw.f64.size(); ++a) {
        size_t bad = 0;
        for (double v : fw.f64[a]) bad += !std::isfinite(v);
        if (bad) std::printf("okvis_amd::Estimator::optimize: input array %zu holds %zu non-finite values of %zu\n", a, bad, fw.f64[a].size());
      }
  }
  const bool syncAfter = diag_.syncAfterHandover;   // (diagnostics: the enqueued copies are charged to the hand-over)
  if (syncAfter && !dry_) check(okvis_ba_synchronize(solver_), "synchronize");
  const auto t2 = clk::now();
  const SyncedWindow& S = synced_;
  if (checkPatch || dry_) {
    const std::string diff = debugCheckWindow();
    if (!diff.empty()) throw Exception("okvis_amd::Estimator::optimize: the window the solver holds differs from the estimator's: " + diff);
  }
  if (dry_) {   // book-keeping only: nothing is computed (estimator.hpp)
    timings_ = {patchSplit_[0], patchSplit_[1], 0.0, 0.0};
    return;
  }
  okvis_ba_window view;
  if (windowObserver_ || verbose || trace) currentWindowView(&view);
  if (windowObserver_) windowObserver_(&view, 0, windowObserverUser_);
  if (trace) std::printf("window %s: describe %.4f ms, hand-over %.4f ms\n", lastWasPatch_ ? "patched" : "uploaded", patchSplit_[0], patchSplit_[1]);
  if (hasTimeLimit_)  // CeresIterationCallback semantics (CeresIterationCallback.hpp:77-86)
    check(okvis_ba_optimize_timed(solver_, (int)numIter, minIterations_, timeLimit_, &summary_), "optimize");
  else
    check(okvis_ba_optimize(solver_, (int)numIter, &summary_), "optimize");
  const auto t3 = clk::now();
  if (verbose || trace)  // the reference prints summary.FullReport() when verbose (Estimator.cpp:870-872)
    std::printf("okvis_amd::Estimator::optimize: %d poses, %d speed/bias, %d landmarks, %d observations, %d IMU terms, prior %d | "
                "iterations %d (%d successful), cost %.9g -> %.9g, termination %d, radius %.3g\n",
                view.n_pose, view.n_sb, view.n_lm, view.n_obs, view.n_imu, view.marg_dim, summary_.iterations,
                summary_.successful_steps, summary_.initial_cost, summary_.final_cost, summary_.termination, summary_.final_radius);
  // copy the estimates back (the reference's parameter blocks are updated in place by Ceres)
  const size_t nl = S.lm.size();
  std::vector<double>& pose = resPose_; std::vector<double>& sb = resSb_; std::vector<double>& lm = resLm_; std::vector<double>& q = resQ_; std::vector<double>& ref = resRef_;
  pose.resize(7 * S.pose.size()), sb.resize(9 * S.sb.size()), lm.resize(4 * nl), q.resize(nl), ref.resize(9 * S.imu.size());
  check(okvis_ba_fetch_results(solver_, 0, pose.data(), sb.data(), lm.data(), q.empty() ? nullptr : q.data(),
                               ref.empty() ? nullptr : ref.data()), "fetch_results");
  if (windowObserver_) {
    currentWindowView(&view);   // (the container has taken the results over meanwhile; the arrays below are the same numbers)
    okvis_ba_window after = view;
    after.pose = pose.data(), after.sb = sb.data(), after.lm = lm.data();
    windowObserver_(&after, 1, windowObserverUser_);
  }
  for (size_t i = 0; i < S.pose.size(); ++i)
    std::copy(pose.begin() + 7 * i, pose.begin() + 7 * i + 7, poseBlocks_[S.pose[i]].x.begin());
  for (size_t i = 0; i < S.sb.size(); ++i)
    std::copy(sb.begin() + 9 * i, sb.begin() + 9 * i + 9, sbBlocks_[S.sb[i]].x.begin());
  // the ImuError caches live on: remember the bias each one was (re)built at (window order = order of imuFactors_)
  resCache_.resize((size_t)OKVIS_BA_IMU_CACHE_DOUBLES * S.imu.size());
  if (!S.imu.empty()) check(okvis_ba_fetch_imu_caches(solver_, 0, resCache_.data()), "fetch_imu_caches");
  for (size_t i = 0; i < S.imu.size(); ++i) {
    ImuFactor& f = imuFactors_[i];
    std::copy(ref.begin() + 9 * i, ref.begin() + 9 * i + 9, f.sbRef.begin());
    f.hasRef = true;
    const double* c = resCache_.data() + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * i;
    int32_t valid;
    std::memcpy(&valid, c + OKVIS_BA_IMU_CACHE_DOUBLES - 1, 4);   // (the record's flag word: 1 = evaluated since the upload)
    f.hasCache = valid == 1;
    if (f.hasCache) f.cache.assign(c, c + OKVIS_BA_IMU_CACHE_DOUBLES);
  }
  {
    // update landmarks: quality = sqrt(lambda_min)/sqrt(lambda_max) of the un-robustified H_l and the
    // estimate (Estimator.cpp:880-900); a landmark nobody observes is not part of the problem and gets the quality 0 the
    // reference derives from its zero H (Estimator.cpp:890-893)
    std::lock_guard<std::mutex> l(statesMutex_);
    for (auto& kv : landmarksMap_)
      if (kv.second.observations.empty()) kv.second.quality = 0.0;
    for (size_t i = 0; i < nl; ++i) {
      MapPoint& mp = *S.lm[i];
      mp.quality = q[i];
      std::copy(lm.begin() + 4 * i, lm.begin() + 4 * i + 4, mp.point.begin());
    }
  }
  timings_ = {patchSplit_[0], patchSplit_[1] + ms(t0, t2) - patchSplit_[0] - patchSplit_[1], ms(t2, t3), ms(t3, clk::now())};
}

// ---------------------------------------------------------------------------------------------------
// the window between two optimize() calls (Map.cpp:292-565 as edits of the window the solver holds)
// ---------------------------------------------------------------------------------------------------
void Estimator::forgetLandmark(MapPoint& mp) {
  if (mp.winIdx >= 0 && synced_.valid && (size_t)mp.winIdx < synced_.lm.size() && synced_.lm[mp.winIdx] == &mp) {
    synced_.lm[mp.winIdx] = nullptr;
    erasedWinLm_.push_back(mp.winIdx);
  }
  if (mp.touched)
    for (MapPoint*& t : touchedLm_)
      if (t == &mp) t = nullptr;
  if (mp.pendingAdds > 0)
    for (PendingObs& a : obsAdded_)
      if (a.lm == &mp) a.lm = nullptr;
}

// an observation leaves: one that the window holds is logged with its landmark's window index, one that was added since the last
// hand-over just disappears from the log of additions (handles are handed out in increasing order)
void Estimator::noteObservationRemoved(MapPoint& mp, uint64_t handle) {
  touch(mp);
  if (!synced_.valid) return;
  if (handle >= firstPendingHandle_) {
    for (size_t k = obsAdded_.size(); k-- > 0;)
      if (obsAdded_[k].handle == handle) {
        obsAdded_[k].lm = nullptr;
        mp.pendingAdds--;
        break;
      }
  } else if (mp.winIdx >= 0) {
    obsRemoved_.push_back(RemovedObs{mp.winIdx, handle});
  }
}

void Estimator::invalidateSynced() {
  synced_.valid = false;
  touchedLm_.clear();
  erasedWinLm_.clear();
  obsAdded_.clear();
  obsRemoved_.clear();
}

void Estimator::currentWindowView(okvis_ba_window* out) {
  if (dry_) {
    if (!dryStore_) throw Exception("no window yet");
    check(okvis_ba_store_view(dryStore_, out), "okvis_ba_store_view");
  } else {
    check(okvis_ba_patched_view(solver_, 0, out), "okvis_ba_patched_view");
  }
}

void Estimator::uploadWindow(FlatWindow& fw) {
  typedef std::chrono::steady_clock clk;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = clk::now();
  invalidateSynced();
  const WindowSel sel = selectAll();
  flatten(sel, fw);
  const auto t1 = clk::now();
  if (dry_) {
    if (dryStore_) okvis_ba_store_destroy(dryStore_);
    dryStore_ = nullptr;
    check(okvis_ba_store_create(&fw.w, &dryStore_), "okvis_ba_store_create");
  } else {
    check(okvis_ba_upload(solver_, 1, &fw.w), "upload");
  }
  // ---- what the solver holds now, in the estimator's terms ----
  SyncedWindow& S = synced_;
  S.pose = sel.pose;
  S.sb = sel.sb;
  S.poseWin = fw.poseMap;
  S.sbWin = fw.sbMap;
  S.poseFixed.assign(fw.u8[0].begin(), fw.u8[0].end());
  S.sbFixed.assign(fw.u8[1].begin(), fw.u8[1].end());
  for (auto& kv : landmarksMap_) kv.second.winIdx = -1, kv.second.touched = kv.second.valueSet = false, kv.second.pendingAdds = 0;
  firstPendingHandle_ = nextHandle_;
  const size_t nl = sel.landmarks.size();
  S.lm.assign(nl, nullptr);
  S.lmObs.assign(nl, {});
  for (size_t n = 0; n < nl; ++n) {
    MapPoint* mp = const_cast<MapPoint*>(sel.lmPtr[n]);
    mp->winIdx = (int)n;
    S.lm[n] = mp;
  }
  for (int o = 0; o < fw.w.n_obs; ++o)
    S.lmObs[fw.w.obs_lm[o]].push_back(SyncedObs{fw.obsHandle[o], S.pose[fw.w.obs_pose[o]], fw.w.obs_cam[o]});
  S.imu.clear();
  for (const ImuFactor& f : imuFactors_) S.imu.push_back(f.uid);
  S.camIntr.assign(fw.w.cam_intr, fw.w.cam_intr + 12 * (size_t)fw.w.n_cam);
  S.camModel.assign(fw.w.cam_model, fw.w.cam_model + fw.w.n_cam);
  S.nPoseBlocks = poseBlocks_.size();
  S.nSbBlocks = sbBlocks_.size();
  S.nObs = (size_t)fw.w.n_obs;
  S.valid = true;
  poseValueSet_.clear();
  sbValueSet_.clear();
  familiesChanged_ = 0;
  patchSplit_ = {ms(t0, t1), ms(t1, clk::now())};
}

bool Estimator::patchWindow() {
  typedef std::chrono::steady_clock clk;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = clk::now();
  SyncedWindow& S = synced_;
  if (!S.valid || states_.empty()) return false;
  if (dry_ ? dryStore_ == nullptr : false) return false;
  {  // the cameras are part of the window description: new intrinsics start the window over
    const MultiFramePtr& mf = multiFramePtrMap_.at(states_.back().id);
    if (mf->geometry.size() != S.camModel.size()) return false;
    for (size_t c = 0; c < mf->geometry.size(); ++c)
      if (mf->geometry[c].model != S.camModel[c] || std::memcmp(mf->geometry[c].intr.data(), &S.camIntr[12 * c], 12 * sizeof(double)) != 0)
        return false;
  }
  PatchBuffers& B = patchBuf_;
  B.clear();
  // ---- parameter blocks: the marginalised ones leave, the ones created since the last hand-over are appended ----
  std::vector<int>& pose2 = B.pose2;
  std::vector<int>& sb2 = B.sb2;
  std::vector<int>& poseWin2 = B.poseWin2;
  std::vector<int>& sbWin2 = B.sbWin2;
  poseWin2.assign(poseBlocks_.size(), -1);
  sbWin2.assign(sbBlocks_.size(), -1);
  for (size_t wi = 0; wi < S.pose.size(); ++wi) {
    const int b = S.pose[wi];
    if (!poseBlocks_[b].alive) {
      B.remPose.push_back((int32_t)wi);
      continue;
    }
    if ((poseBlocks_[b].fixed ? 1 : 0) != S.poseFixed[wi]) return false;
    poseWin2[b] = (int)pose2.size();
    pose2.push_back(b);
    B.poseFixed2.push_back(S.poseFixed[wi]);
  }
  for (size_t b = S.nPoseBlocks; b < poseBlocks_.size(); ++b)
    if (poseBlocks_[b].alive) {
      poseWin2[b] = (int)pose2.size();
      pose2.push_back((int)b);
      B.poseFixed2.push_back(poseBlocks_[b].fixed ? 1 : 0);
      B.addPose.insert(B.addPose.end(), poseBlocks_[b].x.begin(), poseBlocks_[b].x.end());
      B.addPoseFixed.push_back(poseBlocks_[b].fixed ? 1 : 0);
    }
  for (size_t wi = 0; wi < S.sb.size(); ++wi) {
    const int b = S.sb[wi];
    if (!sbBlocks_[b].alive) {
      B.remSb.push_back((int32_t)wi);
      continue;
    }
    if ((sbBlocks_[b].fixed ? 1 : 0) != S.sbFixed[wi]) return false;
    sbWin2[b] = (int)sb2.size();
    sb2.push_back(b);
    B.sbFixed2.push_back(S.sbFixed[wi]);
  }
  for (size_t b = S.nSbBlocks; b < sbBlocks_.size(); ++b)
    if (sbBlocks_[b].alive) {
      sbWin2[b] = (int)sb2.size();
      sb2.push_back((int)b);
      B.sbFixed2.push_back(sbBlocks_[b].fixed ? 1 : 0);
      B.addSb.insert(B.addSb.end(), sbBlocks_[b].x.begin(), sbBlocks_[b].x.end());
      B.addSbFixed.push_back(sbBlocks_[b].fixed ? 1 : 0);
    }
  // ---- landmarks: erased ones and the ones that lost their last observation leave; observed ones that are not part of the
  //      window yet take their places by id among the ones that stay (add_lm_before): the window keeps the order of
  //      landmarksMap_, the order flatten() gives it ----
  std::vector<MapPoint*>& added = B.addedLm;
  for (int wi : erasedWinLm_) B.remLm.push_back(wi);
  for (MapPoint* mp : touchedLm_) {
    if (!mp) continue;
    if (mp->winIdx >= 0 && mp->observations.empty()) B.remLm.push_back(mp->winIdx);
    if (mp->winIdx < 0 && !mp->observations.empty()) added.push_back(mp);
  }
  std::sort(B.remLm.begin(), B.remLm.end());
  std::sort(added.begin(), added.end(), [](const MapPoint* a, const MapPoint* b) { return a->id < b->id; });
  std::vector<int>& lmWin2 = B.lmWin2;   // old window index -> new
  lmWin2.assign(S.lm.size(), 0);
  for (int wi : B.remLm) lmWin2[wi] = -1;
  int nl2 = 0;
  {
    size_t a = 0;   // new landmarks placed so far
    int kept = 0;
    for (size_t wi = 0; wi < S.lm.size(); ++wi) {
      if (lmWin2[wi] < 0) continue;
      while (a < added.size() && added[a]->id < S.lm[wi]->id) {
        B.addLmBefore.push_back(kept);
        B.addLmIdx.push_back(kept + (int)a);
        ++a;
      }
      lmWin2[wi] = kept + (int)a;
      ++kept;
    }
    for (; a < added.size(); ++a) {
      B.addLmBefore.push_back(kept);
      B.addLmIdx.push_back(kept + (int)a);
    }
    nl2 = kept + (int)added.size();
  }
  for (MapPoint* mp : added) B.addLm.insert(B.addLm.end(), mp->point.begin(), mp->point.end());
  // ---- observations: what left and what came since the last hand-over, from the two logs (no look-up, no comparison of lists) ----
  // From here on the description is edited in place.  Whenever this function gives up (return false) optimize() describes the
  // window from scratch (uploadWindow), so a half-edited description is never used.
  std::vector<size_t>& obsBegin = B.obsBegin;   // first observation of each landmark in the window as it stands
  obsBegin.assign(S.lm.size() + 1, 0);
  for (size_t wi = 0; wi < S.lm.size(); ++wi) obsBegin[wi + 1] = obsBegin[wi] + S.lmObs[wi].size();
  size_t nObs2 = S.nObs;
  for (int wi : B.remLm) nObs2 -= S.lmObs[wi].size();
  std::vector<int>& edited = B.editedLm;   // old window indices of the landmarks whose list changes
  for (const RemovedObs& r : obsRemoved_) {
    if (lmWin2[r.lmWin] < 0) continue;   // (the landmark leaves with everything it has)
    std::vector<SyncedObs>& list = S.lmObs[r.lmWin];
    size_t k = 0;
    while (k < list.size() && list[k].handle != r.handle) ++k;
    if (k == list.size()) continue;      // (an observation of a camera the window does not have)
    list[k].handle = 0;                  // (handles start at 1) taken out below, once every index has been computed
    B.remObs.push_back((int32_t)(obsBegin[r.lmWin] + k));
    edited.push_back(r.lmWin);
    --nObs2;
  }
  std::sort(B.remObs.begin(), B.remObs.end());
  for (int wi : edited) {
    std::vector<SyncedObs>& list = S.lmObs[wi];
    list.erase(std::remove_if(list.begin(), list.end(), [](const SyncedObs& o) { return o.handle == 0; }), list.end());
  }
  edited.clear();
  for (size_t a = 0; a < added.size(); ++a) added[a]->winIdx = -2 - (int)a;   // (their final index: B.addLmIdx[a])
  B.addedLists.resize(added.size());
  for (const PendingObs& o : obsAdded_) {
    if (!o.lm) continue;                               // removed again, or its landmark erased, before it ever reached the window
    if ((size_t)o.cam >= S.camModel.size()) continue;  // (flatten() skips the observations of cameras the window does not have)
    MapPoint& mp = *o.lm;
    const int ip = poseWin2[o.poseBlock], ie = poseWin2[o.extBlock];
    if (ip < 0 || ie < 0) return false;                // (an observation of a block outside the window: flatten() reports it)
    int lmNew;
    std::vector<SyncedObs>* list;
    if (mp.winIdx >= 0) {
      lmNew = lmWin2[mp.winIdx];
      list = &S.lmObs[mp.winIdx];
      edited.push_back(mp.winIdx);
    } else if (mp.winIdx <= -2) {
      lmNew = B.addLmIdx[-2 - mp.winIdx];
      list = &B.addedLists[-2 - mp.winIdx];
    } else {
      return false;   // (cannot happen: a landmark with an observation is part of the new window)
    }
    if (lmNew < 0) return false;
    B.aoLm.push_back(lmNew), B.aoPose.push_back(ip), B.aoExt.push_back(ie), B.aoCam.push_back(o.cam);
    B.aoUv.push_back(o.u), B.aoUv.push_back(o.v), B.aoSw.push_back(o.sqrtw);
    list->push_back(SyncedObs{o.handle, o.poseBlock, o.cam});
    ++nObs2;
  }
  // the lists stay sorted by (pose, camera), what came behind what was there among equal keys (WindowStore::apply, ba_store.hpp).
  // The usual edit, observations of the newest frame, lands at the end and in order already.
  auto restore = [&](std::vector<SyncedObs>& list) {
    auto less = [&](const SyncedObs& a, const SyncedObs& b) {
      const int pa = poseWin2[a.poseBlock], pb = poseWin2[b.poseBlock];
      return pa != pb ? pa < pb : a.cam < b.cam;
    };
    if (!std::is_sorted(list.begin(), list.end(), less)) std::stable_sort(list.begin(), list.end(), less);
  };
  for (int wi : edited) restore(S.lmObs[wi]);
  for (auto& list : B.addedLists) restore(list);
  for (MapPoint* mp : touchedLm_)
    if (mp && mp->valueSet && mp->winIdx >= 0 && lmWin2[mp->winIdx] >= 0) {
      B.setLmIdx.push_back(lmWin2[mp->winIdx]);
      B.setLm.insert(B.setLm.end(), mp->point.begin(), mp->point.end());
    }
  // ---- IMU terms (both lists are in time order) ----
  {
    size_t j = 0;
    for (size_t i = 0; i < S.imu.size(); ++i) {
      if (j < imuFactors_.size() && imuFactors_[j].uid == S.imu[i]) {
        ++j;
      } else {
        B.remImu.push_back((int32_t)i);
      }
    }
    for (size_t i = 0, k = 0; i < S.imu.size(); ++i) {   // (what stays must be the front of imuFactors_, in order)
      if (!B.remImu.empty() && std::binary_search(B.remImu.begin(), B.remImu.end(), (int32_t)i)) continue;
      if (k >= imuFactors_.size() || imuFactors_[k].uid != S.imu[i]) return false;
      ++k;
    }
    for (; j < imuFactors_.size(); ++j) {
      const ImuFactor& f = imuFactors_[j];
      const int p0 = poseWin2[f.pose0Block], s0 = sbWin2[f.sb0Block], p1 = poseWin2[f.pose1Block], s1 = sbWin2[f.sb1Block];
      if (p0 < 0 || s0 < 0 || p1 < 0 || s1 < 0) return false;
      B.aiP0.push_back(p0), B.aiS0.push_back(s0), B.aiP1.push_back(p1), B.aiS1.push_back(s1);
      B.aiT0.push_back(f.t0), B.aiT1.push_back(f.t1);
      B.aiBegin.push_back((int32_t)B.aiSt.size()), B.aiCount.push_back((int32_t)f.meas.size());
      for (const ImuMeasurement& m : f.meas) {
        B.aiSt.push_back(m.t_ns);
        B.aiGyr.insert(B.aiGyr.end(), m.gyr.begin(), m.gyr.end());
        B.aiAcc.insert(B.aiAcc.end(), m.acc.begin(), m.acc.end());
      }
    }
  }
  // ---- the patch ----
  okvis_ba_patch P;
  std::memset(&P, 0, sizeof(P));
  P.n_remove_obs = (int32_t)B.remObs.size(), P.remove_obs = B.remObs.data();
  P.n_remove_lm = (int32_t)B.remLm.size(), P.remove_lm = B.remLm.data();
  P.n_remove_pose = (int32_t)B.remPose.size(), P.remove_pose = B.remPose.data();
  P.n_remove_sb = (int32_t)B.remSb.size(), P.remove_sb = B.remSb.data();
  P.n_remove_imu = (int32_t)B.remImu.size(), P.remove_imu = B.remImu.data();
  P.n_add_pose = (int32_t)B.addPoseFixed.size(), P.add_pose = B.addPose.data(), P.add_pose_fixed = B.addPoseFixed.data();
  P.n_add_sb = (int32_t)B.addSbFixed.size(), P.add_sb = B.addSb.data(), P.add_sb_fixed = B.addSbFixed.data();
  P.n_add_lm = (int32_t)added.size(), P.add_lm = B.addLm.data(), P.add_lm_before = B.addLmBefore.data();
  P.n_add_obs = (int32_t)B.aoLm.size();
  P.add_obs_lm = B.aoLm.data(), P.add_obs_pose = B.aoPose.data(), P.add_obs_ext = B.aoExt.data(), P.add_obs_cam = B.aoCam.data();
  P.add_obs_uv = B.aoUv.data(), P.add_obs_sqrtw = B.aoSw.data();
  P.n_add_imu = (int32_t)B.aiP0.size();
  P.add_imu_pose0 = B.aiP0.data(), P.add_imu_sb0 = B.aiS0.data(), P.add_imu_pose1 = B.aiP1.data(), P.add_imu_sb1 = B.aiS1.data();
  P.add_imu_t0 = B.aiT0.data(), P.add_imu_t1 = B.aiT1.data(), P.add_imu_s_begin = B.aiBegin.data(), P.add_imu_s_count = B.aiCount.data();
  P.n_add_imu_samples = (int32_t)B.aiSt.size();
  P.add_imu_s_t = B.aiSt.data(), P.add_imu_s_gyr = B.aiGyr.data(), P.add_imu_s_acc = B.aiAcc.data();
  P.replace = familiesChanged_;
  if (familiesChanged_ & OKVIS_BA_PATCH_POSE_PRIORS)
    for (const PosePrior& pp : posePriors_) {
      if (poseWin2[pp.block] < 0) return false;
      B.ppPose.push_back(poseWin2[pp.block]);
      B.ppMeas.insert(B.ppMeas.end(), pp.meas.begin(), pp.meas.end());
      B.ppSi.insert(B.ppSi.end(), pp.sqrtInfo.begin(), pp.sqrtInfo.end());
    }
  if (familiesChanged_ & OKVIS_BA_PATCH_SB_PRIORS)
    for (const SbPrior& sp : sbPriors_) {
      if (sbWin2[sp.block] < 0) return false;
      B.spSb.push_back(sbWin2[sp.block]);
      B.spMeas.insert(B.spMeas.end(), sp.meas.begin(), sp.meas.end());
      B.spSi.insert(B.spSi.end(), sp.sqrtInfo.begin(), sp.sqrtInfo.end());
    }
  if (familiesChanged_ & OKVIS_BA_PATCH_RELPOSE)
    for (const RelPose& r : relPoses_) {
      if (poseWin2[r.block0] < 0 || poseWin2[r.block1] < 0) return false;
      B.rp0.push_back(poseWin2[r.block0]), B.rp1.push_back(poseWin2[r.block1]);
      B.rpSi.insert(B.rpSi.end(), r.sqrtInfo.begin(), r.sqrtInfo.end());
    }
  P.n_pprior = (int32_t)B.ppPose.size(), P.pprior_pose = B.ppPose.data(), P.pprior_meas = B.ppMeas.data(), P.pprior_sqrtinfo = B.ppSi.data();
  P.n_sbprior = (int32_t)B.spSb.size(), P.sbprior_sb = B.spSb.data(), P.sbprior_meas = B.spMeas.data(), P.sbprior_sqrtinfo = B.spSi.data();
  P.n_relpose = (int32_t)B.rp0.size(), P.rel_pose0 = B.rp0.data(), P.rel_pose1 = B.rp1.data(), P.rel_sqrtinfo = B.rpSi.data();
  // The numbers of a prior that is still being computed (okvis_ba_marginalize_begin) are not waited for here: the patch carries the
  // prior's blocks and linearisation points with stand-in numbers (the result buffers as they are), the solver edits its container
  // and builds its index lists meanwhile, and the numbers follow through okvis_ba_set_marg_prior_values at the end of this function.
  const bool lateNumbers = priorPending_ && !dry_ && (familiesChanged_ & OKVIS_BA_PATCH_MARG_PRIOR) && prior_.dim > 0 &&
                           margJ_.size() >= (size_t)prior_.dim * prior_.dim && margE_.size() >= (size_t)prior_.dim;
  if ((familiesChanged_ & OKVIS_BA_PATCH_MARG_PRIOR) && !lateNumbers) resolvePrior();
  if ((familiesChanged_ & OKVIS_BA_PATCH_MARG_PRIOR) && prior_.dim > 0) {   // the MarginalizationError residual block (Estimator.cpp:750-759)
    int off = 0;
    for (size_t k = 0; k < prior_.block.size(); ++k) {
      const bool isPose = prior_.type[k] == OKVIS_BA_BLOCK_POSE;
      const int wi = isPose ? poseWin2[prior_.block[k]] : sbWin2[prior_.block[k]];
      if (wi < 0) return false;
      B.mType.push_back(prior_.type[k]), B.mIdx.push_back(wi), B.mOff.push_back(off);
      off += isPose ? 6 : 9;
      B.mLin.insert(B.mLin.end(), prior_.lin[k].begin(), prior_.lin[k].end());
    }
    P.marg_dim = prior_.dim;
    P.marg_nblocks = (int32_t)prior_.block.size();
    P.marg_block_type = B.mType.data(), P.marg_block_idx = B.mIdx.data(), P.marg_block_off = B.mOff.data();
    P.marg_J = lateNumbers ? margJ_.data() : prior_.J.data();
    P.marg_e0 = lateNumbers ? margE_.data() : prior_.e0.data();
    P.marg_lin = B.mLin.data();
  }
  // values the caller has set since (blocks that were part of the window already; new ones carry theirs)
  for (int b : poseValueSet_)
    if ((size_t)b < S.nPoseBlocks && poseWin2[b] >= 0) {
      B.setPoseIdx.push_back(poseWin2[b]);
      B.setPose.insert(B.setPose.end(), poseBlocks_[b].x.begin(), poseBlocks_[b].x.end());
    }
  for (int b : sbValueSet_)
    if ((size_t)b < S.nSbBlocks && sbWin2[b] >= 0) {
      B.setSbIdx.push_back(sbWin2[b]);
      B.setSb.insert(B.setSb.end(), sbBlocks_[b].x.begin(), sbBlocks_[b].x.end());
    }
  P.n_set_pose = (int32_t)B.setPoseIdx.size(), P.set_pose_idx = B.setPoseIdx.data(), P.set_pose = B.setPose.data();
  P.n_set_sb = (int32_t)B.setSbIdx.size(), P.set_sb_idx = B.setSbIdx.data(), P.set_sb = B.setSb.data();
  P.n_set_lm = (int32_t)B.setLmIdx.size(), P.set_lm_idx = B.setLmIdx.data(), P.set_lm = B.setLm.data();
  const auto t1 = clk::now();
  const int rc = dry_ ? okvis_ba_store_patch(dryStore_, &P) : okvis_ba_patch_window(solver_, 0, &P);
  if (rc != OKVIS_BA_OK) {
    // (all or nothing on the solver's side: the old window is still there; optimize() describes the new one from scratch)
    if (diag_.trace) std::printf("okvis_amd::Estimator: patch refused (%s), uploading the window instead\n", okvis_ba_error_string(rc));
    return false;
  }
  // ---- the description follows the container ----
  for (MapPoint* mp : touchedLm_)
    if (mp) mp->touched = mp->valueSet = false, mp->pendingAdds = 0;
  {
    std::vector<MapPoint*> lm2((size_t)nl2, nullptr);
    std::vector<std::vector<SyncedObs>> obs2((size_t)nl2);
    for (size_t wi = 0; wi < S.lm.size(); ++wi) {
      if (lmWin2[wi] < 0) {
        if (S.lm[wi]) S.lm[wi]->winIdx = -1;
        continue;
      }
      lm2[lmWin2[wi]] = S.lm[wi];
      S.lm[wi]->winIdx = lmWin2[wi];
      obs2[lmWin2[wi]] = std::move(S.lmObs[wi]);
    }
    for (size_t a = 0; a < added.size(); ++a) {
      lm2[B.addLmIdx[a]] = added[a];
      added[a]->winIdx = B.addLmIdx[a];
      obs2[B.addLmIdx[a]] = std::move(B.addedLists[a]);
    }
    S.lm.swap(lm2);
    S.lmObs.swap(obs2);
  }
  S.pose.swap(pose2);
  S.sb.swap(sb2);
  S.poseWin.swap(poseWin2);
  S.sbWin.swap(sbWin2);
  S.poseFixed.swap(B.poseFixed2);
  S.sbFixed.swap(B.sbFixed2);
  S.imu.clear();
  for (const ImuFactor& f : imuFactors_) S.imu.push_back(f.uid);
  S.nPoseBlocks = poseBlocks_.size();
  S.nSbBlocks = sbBlocks_.size();
  S.nObs = nObs2;
  touchedLm_.clear();
  erasedWinLm_.clear();
  obsAdded_.clear();
  obsRemoved_.clear();
  firstPendingHandle_ = nextHandle_;
  poseValueSet_.clear();
  sbValueSet_.clear();
  familiesChanged_ = 0;
  if (lateNumbers) {
    // the solver holds the window with stand-in numbers in its prior: wait for the real ones (the device has had the
    // container edit, the index build and the copies of the hand-over to finish them) and set them before anything is computed
    resolvePrior();
    check(okvis_ba_set_marg_prior_values(solver_, 0, prior_.J.data(), prior_.e0.data()), "okvis_ba_set_marg_prior_values");
  }
  patchSplit_ = {ms(t0, t1), ms(t1, clk::now())};
  return true;
}

std::string Estimator::debugCheckWindow() {
  okvis_ba_window v;
  currentWindowView(&v);
  const WindowSel sel = selectAll();
  FlatWindow fw;
  flatten(sel, fw);
  const okvis_ba_window& f = fw.w;
  const SyncedWindow& S = synced_;
  auto num = [](const char* what, long a, long b) { return std::string(what) + " " + std::to_string(a) + " vs " + std::to_string(b); };
  if (!S.valid) return "no window description";
  if (v.n_pose != f.n_pose) return num("n_pose", v.n_pose, f.n_pose);
  if (v.n_sb != f.n_sb) return num("n_sb", v.n_sb, f.n_sb);
  if (v.n_lm != f.n_lm) return num("n_lm", v.n_lm, f.n_lm);
  if (v.n_obs != f.n_obs) return num("n_obs", v.n_obs, f.n_obs);
  if (v.n_imu != f.n_imu) return num("n_imu", v.n_imu, f.n_imu);
  if (v.n_cam != f.n_cam) return num("n_cam", v.n_cam, f.n_cam);
  if (v.n_pprior != f.n_pprior || v.n_sbprior != f.n_sbprior || v.n_relpose != f.n_relpose) return "prior counts";
  if (v.marg_dim != f.marg_dim || (v.marg_dim > 0 && v.marg_nblocks != f.marg_nblocks)) return num("marg_dim", v.marg_dim, f.marg_dim);
  if ((size_t)v.n_pose != S.pose.size() || (size_t)v.n_sb != S.sb.size() || (size_t)v.n_lm != S.lm.size() || (size_t)v.n_obs != S.nObs ||
      (size_t)v.n_imu != S.imu.size())
    return "the description's sizes differ from the container's";
  auto same = [](const void* a, const void* b, size_t n) { return n == 0 || std::memcmp(a, b, n) == 0; };
  // blocks: same order in both (blocks are created in time order and appended)
  if (!same(v.pose_fixed, f.pose_fixed, (size_t)v.n_pose) || !same(v.sb_fixed, f.sb_fixed, (size_t)v.n_sb)) return "fixed flags";
  if (!dry_) {
    // (with a device the container carries what the DEVICE holds: the estimator wrote the same numbers back)
  }
  if (!same(v.pose, f.pose, 56 * (size_t)v.n_pose)) return "pose values";
  if (!same(v.sb, f.sb, 72 * (size_t)v.n_sb)) return "speed/bias values";
  for (int i = 0; i < v.n_pose; ++i)
    if (S.pose[i] != sel.pose[i]) return "pose order";
  for (int i = 0; i < v.n_sb; ++i)
    if (S.sb[i] != sel.sb[i]) return "speed/bias order";
  if (!same(v.cam_intr, f.cam_intr, 96 * (size_t)v.n_cam) || !same(v.cam_model, f.cam_model, 4 * (size_t)v.n_cam)) return "cameras";
  // landmarks: any order; observation lists per landmark as sorted tuples
  std::vector<int> fBegin((size_t)f.n_lm + 1, 0), vBegin((size_t)v.n_lm + 1, 0);
  for (int o = 0; o < f.n_obs; ++o) fBegin[f.obs_lm[o] + 1]++;
  for (int o = 0; o < v.n_obs; ++o) vBegin[v.obs_lm[o] + 1]++;
  for (int l = 0; l < f.n_lm; ++l) fBegin[l + 1] += fBegin[l], vBegin[l + 1] += vBegin[l];
  for (int o = 1; o < v.n_obs; ++o) {
    if (v.obs_lm[o - 1] > v.obs_lm[o] || (v.obs_lm[o - 1] == v.obs_lm[o] && (v.obs_pose[o - 1] > v.obs_pose[o] ||
        (v.obs_pose[o - 1] == v.obs_pose[o] && v.obs_cam[o - 1] > v.obs_cam[o]))))
      return num("container observations unsorted at", o, v.n_obs);
  }
  typedef std::array<double, 6> Tup;
  std::vector<Tup> a, b;
  for (int l = 0; l < f.n_lm; ++l) {
    const MapPoint* mp = sel.lmPtr[l];
    const int wi = mp->winIdx;
    if (wi != l) return num("landmark order: window index vs place in the map,", wi, l);
    if (wi < 0 || wi >= v.n_lm || S.lm[wi] != mp) return num("landmark not in the description:", (long)mp->id, wi);
    if (!same(v.lm + 4 * (size_t)wi, f.lm + 4 * (size_t)l, 32)) return num("landmark value", (long)mp->id, wi);
    a.clear(), b.clear();
    for (int o = fBegin[l]; o < fBegin[l + 1]; ++o)
      a.push_back(Tup{(double)f.obs_pose[o], (double)f.obs_ext[o], (double)f.obs_cam[o], f.obs_uv[2 * o], f.obs_uv[2 * o + 1], f.obs_sqrtw[o]});
    for (int o = vBegin[wi]; o < vBegin[wi + 1]; ++o)
      b.push_back(Tup{(double)v.obs_pose[o], (double)v.obs_ext[o], (double)v.obs_cam[o], v.obs_uv[2 * o], v.obs_uv[2 * o + 1], v.obs_sqrtw[o]});
    std::sort(a.begin(), a.end()), std::sort(b.begin(), b.end());
    if (a != b) return num("observations of landmark", (long)mp->id, (long)a.size() * 1000 + (long)b.size());
    // the description's own list (what the next patch computes its removal indices from) is the container's, entry by entry
    if (S.lmObs[wi].size() != b.size()) return num("described observation count of landmark", (long)mp->id, (long)S.lmObs[wi].size());
    for (size_t k = 0; k < S.lmObs[wi].size(); ++k) {
      const int o = vBegin[wi] + (int)k;
      const Observation* it = observations_.find(S.lmObs[wi][k].handle);
      if (!it) return num("described observation gone, landmark", (long)mp->id, (long)k);
      if (S.poseWin[it->poseBlock] != v.obs_pose[o] || (int)it->camIdx != v.obs_cam[o] || it->u != v.obs_uv[2 * o] ||
          it->v != v.obs_uv[2 * o + 1])
        return num("described observation order, landmark", (long)mp->id, (long)k);
    }
  }
  // IMU terms, priors: same order
  if (!same(v.imu_pose0, f.imu_pose0, 4 * (size_t)v.n_imu) || !same(v.imu_sb0, f.imu_sb0, 4 * (size_t)v.n_imu) ||
      !same(v.imu_pose1, f.imu_pose1, 4 * (size_t)v.n_imu) || !same(v.imu_sb1, f.imu_sb1, 4 * (size_t)v.n_imu) ||
      !same(v.imu_t0, f.imu_t0, 8 * (size_t)v.n_imu) || !same(v.imu_t1, f.imu_t1, 8 * (size_t)v.n_imu) ||
      !same(v.imu_s_count, f.imu_s_count, 4 * (size_t)v.n_imu) || v.n_imu_samples != f.n_imu_samples ||
      !same(v.imu_s_t, f.imu_s_t, 8 * (size_t)v.n_imu_samples) || !same(v.imu_s_gyr, f.imu_s_gyr, 24 * (size_t)v.n_imu_samples) ||
      !same(v.imu_s_acc, f.imu_s_acc, 24 * (size_t)v.n_imu_samples))
    return "IMU terms";
  // what the terms carry from the last optimisation: the flag, the reference bias and (flag 2) the preintegration record
  for (int i = 0; i < v.n_imu; ++i) {
    const int fv = v.imu_sb_ref_valid ? v.imu_sb_ref_valid[i] : 0, ff = f.imu_sb_ref_valid ? f.imu_sb_ref_valid[i] : 0;
    if (fv != ff) return num("flag of an IMU term's preintegration", fv, ff);
    if (fv && !same(v.imu_sb_ref + 9 * (size_t)i, f.imu_sb_ref + 9 * (size_t)i, 72)) return "reference bias of an IMU term";
    if (fv == 2 && !same(v.imu_cache + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * i, f.imu_cache + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * i,
                         8 * (size_t)OKVIS_BA_IMU_CACHE_DOUBLES))
      return "preintegration record of an IMU term";
  }
  if (!same(v.pprior_pose, f.pprior_pose, 4 * (size_t)v.n_pprior) || !same(v.pprior_meas, f.pprior_meas, 56 * (size_t)v.n_pprior) ||
      !same(v.pprior_sqrtinfo, f.pprior_sqrtinfo, 288 * (size_t)v.n_pprior))
    return "pose priors";
  if (!same(v.sbprior_sb, f.sbprior_sb, 4 * (size_t)v.n_sbprior) || !same(v.sbprior_meas, f.sbprior_meas, 72 * (size_t)v.n_sbprior) ||
      !same(v.sbprior_sqrtinfo, f.sbprior_sqrtinfo, 648 * (size_t)v.n_sbprior))
    return "speed/bias priors";
  if (!same(v.rel_pose0, f.rel_pose0, 4 * (size_t)v.n_relpose) || !same(v.rel_pose1, f.rel_pose1, 4 * (size_t)v.n_relpose) ||
      !same(v.rel_sqrtinfo, f.rel_sqrtinfo, 288 * (size_t)v.n_relpose))
    return "relative-pose terms";
  if (v.marg_dim > 0) {
    const size_t d = (size_t)v.marg_dim, nb = (size_t)v.marg_nblocks;
    if (!same(v.marg_block_type, f.marg_block_type, 4 * nb) || !same(v.marg_block_idx, f.marg_block_idx, 4 * nb) ||
        !same(v.marg_block_off, f.marg_block_off, 4 * nb) || !same(v.marg_J, f.marg_J, 8 * d * d) || !same(v.marg_e0, f.marg_e0, 8 * d) ||
        !same(v.marg_lin, f.marg_lin, 72 * nb))
      return "marginalisation prior";
  }
  return std::string();
}

// ---------------------------------------------------------------------------------------------------
// applyMarginalizationStrategy (Estimator.cpp:434-773).  The decisions (which frames / blocks / landmarks /
// observations go) are book-keeping and stay on the host like in the reference; everything the reference's
// MarginalizationError computes (linearisation of the selected residuals at their linearisation points,
// landmark and dense Schur complements with pseudo-inverses, eigen-decomposition into J, e0) runs on the GPU
// through okvis_ba_marginalize.
// ---------------------------------------------------------------------------------------------------
struct Estimator::MargUndo {
  // decisions and deletions of Estimator.cpp:485-725 are interleaved, so they are applied as the reference applies them and
  // logged; the log is replayed backwards if the numerics (upload / okvis_ba_marginalize) throw
  // (the removed observations have a list of their own — a frame's worth per call, and an Op carries a whole MapPoint; on the way
  //  back the landmarks return first, then the observations, whose landmarks are all there again by then)
  struct Op {
    enum Kind { SB_CLEARED, LANDMARK_ERASED } kind;
    size_t stateIdx = 0;
    int sbBlock = -1;
    MapPoint landmark;
    bool initialized = false;
  };
  std::vector<Op> ops;
  std::vector<Observation> removedObs;
  size_t removedSize = 0;
  // the prior as it stood before the call replaced it (with its numbers: whatever throws behind the replacement — an allocation in
  // the compactions, say — must not leave a prior with blocks but without J / e0, which every later optimize() would be refused)
  bool priorSaved = false;
  MargPrior prior;
  uint32_t familiesChanged = 0;
};

bool Estimator::applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, MapPointVector& removedLandmarks) {
  resolvePrior();   // (the previous prior's numbers go into this marginalisation)
  MargUndo undo;
  undo.removedSize = removedLandmarks.size();
  // the logs of window edits as they stand: a failed call takes its entries back (possible unless it may have dropped entries of
  // the log of additions, which cannot be brought back; then the next optimize() describes the window from scratch)
  const size_t nRemovedLog = obsRemoved_.size(), nErasedLog = erasedWinLm_.size();
  const bool logsRestorable = synced_.valid && obsAdded_.empty();
  try {
    return applyMarginalizationStrategyImpl(numKeyframes, numImuFrames, removedLandmarks, undo);
  } catch (...) {
    if (priorPending_) {   // (something threw behind okvis_ba_marginalize_begin: its numbers are not wanted any more)
      priorPending_ = false;
      (void)okvis_ba_marginalize_end(margSolver_, &margRes_);
    }
    std::lock_guard<std::mutex> l(statesMutex_);
    if (undo.priorSaved) {   // the old prior, numbers included, is the window's prior again
      prior_ = std::move(undo.prior);
      familiesChanged_ = undo.familiesChanged | OKVIS_BA_PATCH_MARG_PRIOR;   // (the solver may have seen the replacement's blocks)
    }
    if (logsRestorable) {
      obsRemoved_.resize(nRemovedLog);
      erasedWinLm_.resize(nErasedLog);
    } else {
      invalidateSynced();
    }
    for (size_t k = undo.ops.size(); k-- > 0;) {
      MargUndo::Op& op = undo.ops[k];
      if (op.kind == MargUndo::Op::SB_CLEARED) {
        states_[op.stateIdx].sbBlock = op.sbBlock;
      } else {
        landmarkInitialized_[op.landmark.id] = op.initialized;
        const uint64_t id = op.landmark.id;
        MapPoint& node = landmarksMap_[id];
        node = std::move(op.landmark);
        if (synced_.valid) {   // the landmark is back as a new map node: the window description points at it again
          if (node.winIdx >= 0 && (size_t)node.winIdx < synced_.lm.size()) synced_.lm[node.winIdx] = &node;
          if (node.touched) touchedLm_.push_back(&node);
        } else {
          node.winIdx = -1;
          node.touched = node.valueSet = false;
          node.pendingAdds = 0;
        }
      }
    }
    for (size_t k = undo.removedObs.size(); k-- > 0;) {
      const Observation& o = undo.removedObs[k];
      observations_.insert(o);
      landmarksMap_.at(o.landmarkId).observations[KeypointIdentifier{o.poseId, o.camIdx, o.keypointIdx}] = o.handle;
    }
    removedLandmarks.resize(undo.removedSize);
    throw;
  }
}

bool Estimator::applyMarginalizationStrategyImpl(size_t numKeyframes, size_t numImuFrames, MapPointVector& removedLandmarks,
                                                 MargUndo& undo) {
  auto logLandmarkErase = [&](const MapPoint& mp) {
    MargUndo::Op op;
    op.kind = MargUndo::Op::LANDMARK_ERASED;
    op.landmark = mp;
    auto it = landmarkInitialized_.find(mp.id);
    op.initialized = it != landmarkInitialized_.end() && it->second;
    undo.ops.push_back(std::move(op));
  };
  // (the caller holds the landmark and the observation's key: one look-up in observations_, one erase by key in the landmark's own
  //  map — removeObservation(handle) would find the landmark in landmarksMap_ and scan its observations for the handle)
  auto removeObservationLogged = [&](MapPoint& mp, uint64_t hnd, const KeypointIdentifier& kid) {
    const Observation* it = observations_.find(hnd);
    if (!it) return;
    undo.removedObs.push_back(*it);
    mp.observations.erase(kid);
    noteObservationRemoved(mp, hnd);
    observations_.erase(hnd);
  };
  // keep the newest numImuFrames (:439-446)
  if (states_.size() <= numImuFrames) return true;
  const size_t nOlder = states_.size() - numImuFrames;  // states_[0 .. nOlder-1], visited newest -> oldest

  // distinguish if we marginalize everything or everything but pose (:468-483)
  std::vector<uint64_t> removeFrames, allLinearizedFrames;
  size_t countedKeyframes = 0;
  for (size_t k = nOlder; k-- > 0;) {
    const State& st = states_[k];
    if (!st.isKeyframe || countedKeyframes >= numKeyframes)
      removeFrames.push_back(st.id);
    else
      countedKeyframes++;
    allLinearizedFrames.push_back(st.id);
  }
  auto contains = [](const std::vector<uint64_t>& v, uint64_t id) { return std::find(v.begin(), v.end(), id) != v.end(); };

  WindowSel sel;  // the residuals that get linearised into the prior (MarginalizationError::addResidualBlock)
  std::vector<char> imuSel(imuFactors_.size(), 0), ppSel(posePriors_.size(), 0), ppDrop(posePriors_.size(), 0),
      sbpSel(sbPriors_.size(), 0), relSel(relPoses_.size(), 0);
  std::vector<int> margPose, margSb;
  std::vector<uint64_t> margLandmarks;
  auto selectPoseResiduals = [&](int block, bool isT_WS, bool* sawPoseError) {
    for (size_t i = 0; i < posePriors_.size(); ++i)
      if (posePriors_[i].block == block && !ppSel[i] && !ppDrop[i]) {
        if (isT_WS) {  // "avoids linearising initial pose error" (:569-574)
          ppDrop[i] = 1;
          if (sawPoseError) *sawPoseError = true;
        } else {
          ppSel[i] = 1;
        }
      }
    for (size_t i = 0; i < imuFactors_.size(); ++i)
      if (imuFactors_[i].pose0Block == block || imuFactors_[i].pose1Block == block) imuSel[i] = 1;
    for (size_t i = 0; i < relPoses_.size(); ++i)
      if (relPoses_[i].block0 == block || relPoses_[i].block1 == block) relSel[i] = 1;
  };

  // marginalize everything but pose (:485-554): the speed/bias block of every older state
  for (size_t k = nOlder; k-- > 0;) {
    State& st = states_[k];
    if (st.sbBlock < 0 || sbBlocks_[st.sbBlock].fixed) continue;
    const int b = st.sbBlock;
    {
      MargUndo::Op op;
      op.kind = MargUndo::Op::SB_CLEARED;
      op.stateIdx = k;
      op.sbBlock = b;
      undo.ops.push_back(std::move(op));
    }
    st.sbBlock = -1;  // "remember we removed"
    margSb.push_back(b);
    for (size_t i = 0; i < imuFactors_.size(); ++i)
      if (imuFactors_[i].sb0Block == b || imuFactors_[i].sb1Block == b) imuSel[i] = 1;
    for (size_t i = 0; i < sbPriors_.size(); ++i)
      if (sbPriors_[i].block == b) sbpSel[i] = 1;
  }

  // marginalize ONLY pose now (:556-733)
  bool reDoFixation = false;
  const uint64_t currentKfId = allLinearizedFrames.at(0);
  std::vector<uint64_t> selObs;
  std::unordered_set<uint64_t> selObsSet;  // membership test of selObs
  std::unordered_set<uint64_t> margLandmarkSet;   // membership test of margLandmarks (only consulted for a second removed frame)
  // (one buffer for every landmark of the loop below; the frame id of an observation is part of its key in the landmark's own
  // observation map, so the loop needs no look-up in observations_)
  struct Residual {
    uint64_t handle, poseId;
    size_t cam, kp;
  };
  uint64_t newestRemoved = 0;
  for (uint64_t id : removeFrames) newestRemoved = std::max(newestRemoved, id);
  std::vector<Residual> residuals;
  for (size_t rf = 0; rf < removeFrames.size(); ++rf) {
    size_t k = 0;
    while (states_[k].id != removeFrames[rf]) ++k;
    State& st = states_[k];
    margPose.push_back(st.poseBlock);
    selectPoseResiduals(st.poseBlock, true, &reDoFixation);
    // the camera extrinsics of this frame, if they are not shared with the next frame (:587-617)
    for (size_t j = 0; j < st.extBlocks.size(); ++j) {
      const int b = st.extBlocks[j];
      if (b < 0 || poseBlocks_[b].fixed) continue;
      if (k + 1 < states_.size() && states_[k + 1].extBlocks.at(j) == b) continue;
      if (std::find(margPose.begin(), margPose.end(), b) != margPose.end()) continue;
      margPose.push_back(b);
      selectPoseResiduals(b, false, nullptr);
    }
    // now finally we treat all the observations (:620-725)
    for (auto pit = landmarksMap_.begin(); pit != landmarksMap_.end();) {
      MapPoint& mp = pit->second;
      if (rf > 0 && margLandmarkSet.count(pit->first)) {
        ++pit;  // already scheduled (the reference erased it from landmarksMap_ at that point, :715-719)
        continue;
      }
      if (selObsSet.empty() && !mp.observations.empty()) {
        // most landmarks have nothing to do with the frames that leave: their observations are ordered by frame, so the ones of
        // the removed frames, if any, come first (the same outcome as the full pass below: skipLandmark)
        bool touchesRemoved = false;
        for (const auto& ob : mp.observations) {
          if (ob.first.frameId > newestRemoved) break;
          if (contains(removeFrames, ob.first.frameId)) {
            touchesRemoved = true;
            break;
          }
        }
        if (!touchesRemoved) {
          ++pit;
          continue;
        }
      }
      residuals.clear();  // reprojection residuals still in the map
      for (const auto& ob : mp.observations)
        if (selObsSet.empty() || !selObsSet.count(ob.second))
          residuals.push_back(Residual{ob.second, ob.first.frameId, ob.first.cameraIndex, ob.first.keypointIndex});
      bool skipLandmark = true, hasNewObservations = false, justDelete = false, marginalize = true, errorTermAdded = false;
      size_t obsCount = 0;
      for (const Residual& res : residuals) {
        const uint64_t poseId = res.poseId;
        if (contains(removeFrames, poseId)) skipLandmark = false;
        if (poseId >= currentKfId) {
          marginalize = false;
          hasNewObservations = true;
        }
        if (contains(allLinearizedFrames, poseId)) obsCount++;
      }
      if (residuals.empty()) {  // :663-668
        removedLandmarks.push_back(mp);
        logLandmarkErase(mp);
        landmarkInitialized_.erase(pit->first);
        forgetLandmark(mp);
        pit = landmarksMap_.erase(pit);
        continue;
      }
      if (skipLandmark) {
        ++pit;
        continue;
      }
      for (size_t r = 0; r < residuals.size(); ++r) {
        const uint64_t hnd = residuals[r].handle;
        const uint64_t poseId = residuals[r].poseId;
        if ((contains(removeFrames, poseId) && hasNewObservations) ||
            (!contains(allLinearizedFrames, poseId) && marginalize)) {
          removeObservationLogged(mp, hnd, KeypointIdentifier{poseId, residuals[r].cam, residuals[r].kp});  // ok, let's ignore the observation
          residuals.erase(residuals.begin() + r);
          r--;
        } else if (marginalize && contains(allLinearizedFrames, poseId)) {
          if (obsCount < 2) {
            removeObservationLogged(mp, hnd, KeypointIdentifier{poseId, residuals[r].cam, residuals[r].kp});
            residuals.erase(residuals.begin() + r);
            r--;
          } else {
            errorTermAdded = true;  // add information to be considered in marginalization later
            selObs.push_back(hnd);
            selObsSet.insert(hnd);
          }
        }
        if (residuals.empty()) {
          justDelete = true;
          marginalize = false;
        }
      }
      if (justDelete) {
        removedLandmarks.push_back(mp);
        logLandmarkErase(mp);
        landmarkInitialized_.erase(pit->first);
        forgetLandmark(mp);
        pit = landmarksMap_.erase(pit);
        continue;
      }
      if (marginalize && errorTermAdded) {
        margLandmarks.push_back(pit->first);
        margLandmarkSet.insert(pit->first);
        removedLandmarks.push_back(mp);
        ++pit;  // the block itself is erased after the numerics below (its value is the linearisation point)
        continue;
      }
      ++pit;
    }
  }

  // ---- now apply the actual marginalization (:735-744) ----
  const bool anything = !margPose.empty() || !margSb.empty() || !margLandmarks.empty();
  if (anything) {
    // every block a selected residual or the previous prior touches
    std::vector<char> needPose(poseBlocks_.size(), 0), needSb(sbBlocks_.size(), 0);
    for (size_t i = 0; i < imuFactors_.size(); ++i)
      if (imuSel[i]) {
        sel.imu.push_back((int)i);
        needPose[imuFactors_[i].pose0Block] = needPose[imuFactors_[i].pose1Block] = 1;
        needSb[imuFactors_[i].sb0Block] = needSb[imuFactors_[i].sb1Block] = 1;
      }
    for (size_t i = 0; i < posePriors_.size(); ++i)
      if (ppSel[i]) {
        sel.pprior.push_back((int)i);
        needPose[posePriors_[i].block] = 1;
      }
    for (size_t i = 0; i < sbPriors_.size(); ++i)
      if (sbpSel[i]) {
        sel.sbprior.push_back((int)i);
        needSb[sbPriors_[i].block] = 1;
      }
    for (size_t i = 0; i < relPoses_.size(); ++i)
      if (relSel[i]) {
        sel.rel.push_back((int)i);
        needPose[relPoses_[i].block0] = needPose[relPoses_[i].block1] = 1;
      }
    for (uint64_t hnd : selObs) {
      const Observation& o = observations_.at(hnd);
      const State* st = findState(o.poseId);
      needPose[st->poseBlock] = needPose[st->extBlocks.at(o.camIdx)] = 1;
    }
    for (int b : margPose) needPose[b] = 1;
    for (int b : margSb) needSb[b] = 1;
    for (size_t k = 0; k < prior_.block.size(); ++k)
      (prior_.type[k] == OKVIS_BA_BLOCK_POSE ? needPose : needSb)[prior_.block[k]] = 1;
    for (size_t i = 0; i < poseBlocks_.size(); ++i)
      if (needPose[i]) sel.pose.push_back((int)i);
    for (size_t i = 0; i < sbBlocks_.size(); ++i)
      if (needSb[i]) sel.sb.push_back((int)i);
    sel.landmarks = margLandmarks;
    sel.obs = selObs;
    sel.withPrior = false;
    sel.atLinearizationPoint = true;  // first-estimate Jacobians (MarginalizationError.cpp:292-310)
    typedef std::chrono::steady_clock clk;
    auto msf = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto tm0 = clk::now();
    FlatWindow fw;
    flatten(sel, fw);
    const auto tm1 = clk::now();

    std::vector<uint8_t> pm(std::max<size_t>(1, sel.pose.size()), 0), sm(std::max<size_t>(1, sel.sb.size()), 0);
    for (int b : margPose) pm[fw.poseMap[b]] = 1;
    for (int b : margSb) sm[fw.sbMap[b]] = 1;
    okvis_ba_marg_spec spec;
    std::memset(&spec, 0, sizeof(spec));
    spec.pose_marg = pm.data();
    spec.sb_marg = sm.data();
    std::vector<int32_t> pt, pi, po;
    if (prior_.dim > 0) {
      int off = 0;
      for (size_t k = 0; k < prior_.block.size(); ++k) {
        const bool pose = prior_.type[k] == OKVIS_BA_BLOCK_POSE;
        pt.push_back(prior_.type[k]);
        pi.push_back(pose ? fw.poseMap[prior_.block[k]] : fw.sbMap[prior_.block[k]]);
        po.push_back(off);
        off += pose ? 6 : 9;
      }
      spec.prior_dim = prior_.dim;
      spec.prior_nblocks = (int)pt.size();
      spec.prior_block_type = pt.data();
      spec.prior_block_idx = pi.data();
      spec.prior_block_off = po.data();
      spec.prior_H = prior_.H.data();
      spec.prior_b0 = prior_.b0.data();
    }
    const int cap = 6 * (int)sel.pose.size() + 9 * (int)sel.sb.size(), capb = (int)(sel.pose.size() + sel.sb.size());
    // (the result's arrays outlive this call — the numbers arrive in resolvePrior() — and keep their size from frame to frame)
    std::vector<int32_t>&bt = margBt_, &bi = margBi_, &bo = margBo_;
    std::vector<double>&Hn = margH_, &bn = margB_, &Jn = margJ_, &en = margE_;
    auto grow = [](auto& v, size_t n) {
      if (v.size() < n) v.resize(n);
    };
    grow(bt, (size_t)std::max(1, capb)), grow(bi, (size_t)std::max(1, capb)), grow(bo, (size_t)std::max(1, capb));
    grow(Hn, (size_t)std::max(1, cap * cap)), grow(bn, (size_t)std::max(1, cap)), grow(Jn, (size_t)std::max(1, cap * cap)), grow(en, (size_t)std::max(1, cap));
    okvis_ba_marg_result& res = margRes_;
    std::memset(&res, 0, sizeof(res));
    res.capacity_dim = cap;
    res.capacity_blocks = capb;
    res.block_type = bt.data(); res.block_idx = bi.data(); res.block_off = bo.data();
    res.H = Hn.data(); res.b0 = bn.data(); res.J = Jn.data(); res.e0 = en.data();
    if (debugFailMarg_) {
      debugFailMarg_ = false;
      throw Exception("applyMarginalizationStrategy: injected failure (debugFailNextMarginalization)");
    }
    auto tm2 = clk::now();
    if (dry_) {
      // book-keeping only: the blocks the prior will connect are known without the numbers (every free block of the
      // sub-window that is not eliminated); a unit prior at the current values stands in
      int nb = 0, off = 0;
      auto keep = [&](int type, int wi, int dim) {
        bt[nb] = type, bi[nb] = wi, bo[nb] = off;
        ++nb;
        off += dim;
      };
      for (size_t i = 0; i < sel.pose.size(); ++i)
        if (!pm[i] && !poseBlocks_[sel.pose[i]].fixed) keep(OKVIS_BA_BLOCK_POSE, (int)i, 6);
      for (size_t i = 0; i < sel.sb.size(); ++i)
        if (!sm[i] && !sbBlocks_[sel.sb[i]].fixed) keep(OKVIS_BA_BLOCK_SPEEDBIAS, (int)i, 9);
      res.dim = off;
      res.nblocks = nb;
      std::fill(Hn.begin(), Hn.begin() + (size_t)off * off, 0.0);
      std::fill(Jn.begin(), Jn.begin() + (size_t)off * off, 0.0);
      std::fill(bn.begin(), bn.begin() + off, 0.0);
      std::fill(en.begin(), en.begin() + off, 0.0);
      for (int i = 0; i < off; ++i) Hn[(size_t)i * off + i] = Jn[(size_t)i * off + i] = 1.0;
    } else {
      // the sub-window has a solver of its own: solver_ keeps the window optimize() works on between the calls
      if (!margSolver_) check(okvis_ba_create(&margSolver_, device_), "okvis_ba_create (marginalisation)");
      check(okvis_ba_set_options(margSolver_, &options_), "set_options");
      check(okvis_ba_upload(margSolver_, 1, &fw.w), "upload (marginalisation window)");
      tm2 = clk::now();
      // everything is enqueued; the numbers are waited for where they are read next (resolvePrior)
      check(okvis_ba_marginalize_begin(margSolver_, 0, &spec, &res), "marginalize");
      priorPending_ = true;
    }
    margInfo_ = {msf(tm0, tm1), msf(tm1, tm2), msf(tm2, clk::now()), (double)res.sweeps[0], (double)res.sweeps[1],
                 (double)(6 * sel.pose.size() + 9 * sel.sb.size())};

    // the new prior over the remaining connected blocks; blocks that were connected before keep their
    // linearisation point, newly connected ones are linearised at the current estimate
    MargPrior np;
    np.dim = res.dim;
    for (int k = 0; k < res.nblocks; ++k) {
      const bool pose = res.block_type[k] == OKVIS_BA_BLOCK_POSE;
      const int blk = pose ? sel.pose[res.block_idx[k]] : sel.sb[res.block_idx[k]];
      np.type.push_back(res.block_type[k]);
      np.block.push_back(blk);
      std::array<double, 9> lin{};
      const double* x = (pose ? fw.f64[0].data() + 7 * res.block_idx[k] : fw.f64[1].data() + 9 * res.block_idx[k]);
      std::copy(x, x + (pose ? 7 : 9), lin.begin());
      np.lin.push_back(lin);
    }
    const size_t n = (size_t)res.dim;
    if (!priorPending_) {   // (book-keeping only: the stand-in numbers are there already)
      np.H.assign(Hn.begin(), Hn.begin() + n * n);
      np.b0.assign(bn.begin(), bn.begin() + n);
      np.J.assign(Jn.begin(), Jn.begin() + n * n);
      np.e0.assign(en.begin(), en.begin() + n);
    }
    if (np.dim == 0) np = MargPrior();  // "if(marginalizationErrorPtr_->num_residuals()==0) reset" (:747-749)
    undo.prior = std::move(prior_);
    undo.familiesChanged = familiesChanged_;
    undo.priorSaved = true;
    prior_ = np;
    familiesChanged_ |= OKVIS_BA_PATCH_MARG_PRIOR;
  }

  // ---- remove what was linearised / marginalised from the graph (Map::removeResidualBlock / removeParameterBlock)
  auto compact = [](auto& vec, const std::vector<char>& drop) {
    size_t o = 0;
    for (size_t i = 0; i < vec.size(); ++i)
      if (!drop[i]) {
        if (o != i) vec[o] = std::move(vec[i]);
        ++o;
      }
    vec.resize(o);
  };
  compact(imuFactors_, imuSel);
  for (size_t i = 0; i < posePriors_.size(); ++i) ppSel[i] = ppSel[i] || ppDrop[i];
  auto any = [](const std::vector<char>& v) { return std::find(v.begin(), v.end(), (char)1) != v.end(); };
  if (any(ppSel)) familiesChanged_ |= OKVIS_BA_PATCH_POSE_PRIORS;
  if (any(sbpSel)) familiesChanged_ |= OKVIS_BA_PATCH_SB_PRIORS;
  if (any(relSel)) familiesChanged_ |= OKVIS_BA_PATCH_RELPOSE;
  compact(posePriors_, ppSel);
  compact(sbPriors_, sbpSel);
  compact(relPoses_, relSel);
  for (uint64_t hnd : selObs) observations_.erase(hnd);
  {
    std::lock_guard<std::mutex> l(statesMutex_);
    for (uint64_t id : margLandmarks) {
      auto it = landmarksMap_.find(id);
      if (it != landmarksMap_.end()) {
        forgetLandmark(it->second);
        landmarksMap_.erase(it);
      }
      landmarkInitialized_.erase(id);
    }
  }
  for (int b : margPose) poseBlocks_[b].alive = false;
  for (int b : margSb) sbBlocks_[b].alive = false;
  // update book-keeping (:727-732)
  for (uint64_t id : removeFrames) {
    multiFramePtrMap_.erase(id);
    for (size_t k = 0; k < states_.size(); ++k)
      if (states_[k].id == id) {
        states_.erase(states_.begin() + k);
        break;
      }
  }

  if (reDoFixation && !states_.empty()) {  // finally fix the first pose properly (:761-770)
    std::array<double, 36> info{}, si{};
    info[0] = info[7] = info[14] = 1.0e14;
    info[35] = 1.0e14;
    sqrtInformation(info, 6, si);
    const int b = states_.front().poseBlock;
    posePriors_.push_back(PosePrior{b, poseBlocks_[b].x, si});
    familiesChanged_ |= OKVIS_BA_PATCH_POSE_PRIORS;
  }
  return true;
}

// ---- getters / setters -------------------------------------------------------------------------------
bool Estimator::isLandmarkInitialized(uint64_t id) const {
  auto it = landmarkInitialized_.find(id);
  if (it == landmarkInitialized_.end()) throw Exception("landmark not added");
  return it->second;
}
bool Estimator::getLandmark(uint64_t id, MapPoint& mapPoint) const {
  std::lock_guard<std::mutex> l(statesMutex_);
  auto it = landmarksMap_.find(id);
  if (it == landmarksMap_.end()) throw Exception("landmark with id = " + std::to_string(id) + " does not exist.");
  mapPoint = it->second;
  return true;
}
size_t Estimator::getLandmarks(PointMap& landmarks) const {
  std::lock_guard<std::mutex> l(statesMutex_);
  landmarks = landmarksMap_;
  return landmarksMap_.size();
}
size_t Estimator::getLandmarks(MapPointVector& landmarks) const {
  std::lock_guard<std::mutex> l(statesMutex_);
  landmarks.clear();
  landmarks.reserve(landmarksMap_.size());
  for (const auto& kv : landmarksMap_) landmarks.push_back(kv.second);
  return landmarksMap_.size();
}
void Estimator::printStates(uint64_t poseId, std::ostream& buffer) const {
  const State* st = findState(poseId);
  if (!st) throw Exception("Requested state does not exist in estimator.");   // statesMap_.at(poseId) throws in the reference
  auto item = [&](uint64_t id, bool fixed, const char* type) {
    if (fixed) buffer << "(";
    buffer << "id=" << id << ":" << type;
    if (fixed) buffer << ")";
    buffer << ", ";
  };
  buffer << "GLOBAL: ";
  if (st->poseBlock >= 0 && poseBlocks_[st->poseBlock].alive)
    item(poseBlocks_[st->poseBlock].id, poseBlocks_[st->poseBlock].fixed, "PoseParameterBlock");
  buffer << "SENSOR: ";
  for (int e : st->extBlocks)
    if (e >= 0 && poseBlocks_[e].alive) item(poseBlocks_[e].id, poseBlocks_[e].fixed, "PoseParameterBlock");
  if (st->sbBlock >= 0 && sbBlocks_[st->sbBlock].alive)
    item(sbBlocks_[st->sbBlock].id, sbBlocks_[st->sbBlock].fixed, "SpeedAndBiasParameterBlock");
  buffer << std::endl;
}
MultiFramePtr Estimator::multiFrame(uint64_t frameId) const {
  auto it = multiFramePtrMap_.find(frameId);
  if (it == multiFramePtrMap_.end()) throw Exception("Requested multi-frame does not exist in estimator.");
  return it->second;
}
bool Estimator::get_T_WS(uint64_t poseId, Transformation& T_WS) const {
  const State* s = findState(poseId);
  if (!s) return false;
  T_WS.p = poseBlocks_[s->poseBlock].x;
  return true;
}
bool Estimator::getSpeedAndBias(uint64_t poseId, uint64_t, SpeedAndBias& sb) const {
  const State* s = findState(poseId);
  if (!s || s->sbBlock < 0) return false;
  sb = sbBlocks_[s->sbBlock].x;
  return true;
}
bool Estimator::getCameraSensorStates(uint64_t poseId, size_t cameraIdx, Transformation& T_SCi) const {
  const State* s = findState(poseId);
  if (!s || cameraIdx >= s->extBlocks.size()) return false;
  T_SCi.p = poseBlocks_[s->extBlocks[cameraIdx]].x;
  return true;
}
uint64_t Estimator::currentKeyframeId() const {
  for (auto it = states_.rbegin(); it != states_.rend(); ++it)
    if (it->isKeyframe) return it->id;
  throw Exception("no keyframes existing...");
}
uint64_t Estimator::frameIdByAge(size_t age) const {
  if (age >= states_.size()) throw Exception("requested age " + std::to_string(age) + " out of range.");
  return states_[states_.size() - 1 - age].id;
}
uint64_t Estimator::currentFrameId() const {
  if (states_.empty()) throw Exception("no frames added yet.");
  return states_.back().id;
}
bool Estimator::isKeyframe(uint64_t frameId) const {
  const State* s = findState(frameId);
  if (!s) throw Exception("frame does not exist");
  return s->isKeyframe;
}
bool Estimator::isInImuWindow(uint64_t frameId) const {
  const State* s = findState(frameId);
  if (!s) return false;
  return s->sbBlock >= 0;
}
int64_t Estimator::timestamp(uint64_t frameId) const {
  const State* s = findState(frameId);
  if (!s) throw Exception("frame does not exist");
  return s->t_ns;
}
bool Estimator::set_T_WS(uint64_t poseId, const Transformation& T_WS) {
  State* s = findState(poseId);
  if (!s) return false;
  poseBlocks_[s->poseBlock].x = T_WS.p;
  poseValueSet_.push_back(s->poseBlock);
  return true;
}
bool Estimator::setSpeedAndBias(uint64_t poseId, size_t, const SpeedAndBias& sb) {
  State* s = findState(poseId);
  if (!s || s->sbBlock < 0) return false;
  sbBlocks_[s->sbBlock].x = sb;
  sbValueSet_.push_back(s->sbBlock);
  return true;
}
bool Estimator::setCameraSensorStates(uint64_t poseId, size_t cameraIdx, const Transformation& T_SCi) {
  State* s = findState(poseId);
  if (!s || cameraIdx >= s->extBlocks.size()) return false;
  poseBlocks_[s->extBlocks[cameraIdx]].x = T_SCi.p;
  poseValueSet_.push_back(s->extBlocks[cameraIdx]);
  return true;
}
bool Estimator::setLandmark(uint64_t landmarkId, const std::array<double, 4>& landmark) {
  auto it = landmarksMap_.find(landmarkId);
  if (it == landmarksMap_.end()) return false;
  it->second.point = landmark;
  it->second.valueSet = true;
  touch(it->second);
  return true;
}
void Estimator::setLandmarkInitialized(uint64_t landmarkId, bool initialized) {
  if (!landmarksMap_.count(landmarkId)) throw Exception("landmark not added");
  landmarkInitialized_[landmarkId] = initialized;
}
void Estimator::setKeyframe(uint64_t frameId, bool isKeyframe) {
  State* s = findState(frameId);
  if (s) s->isKeyframe = isKeyframe;
}

}  // namespace okvis_amd
