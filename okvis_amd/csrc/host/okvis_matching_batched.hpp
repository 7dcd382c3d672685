// The frontend's keyframe-window matching with its reprojection-heavy inner loops on the GPU — the reference-side binding
// INTEGRATION.md describes, as a class.
//
// okvis::VioKeyframeWindowMatchingAlgorithm<G> (okvis_frontend/include/okvis/VioKeyframeWindowMatchingAlgorithm.hpp:66-255,
// okvis_frontend/src/VioKeyframeWindowMatchingAlgorithm.cpp) is what okvis::Frontend hands to okvis::DenseMatcher::match
// (Frontend.cpp: matchToKeyframes / matchStereo).  The matcher calls doSetup() once, then distance(a, b) for every pair of
// keypoints — which verifies a pair whose descriptors are close by a stereo triangulation (2D-2D, :310-317) or a chi-square gate
// of the landmark's projection (3D-2D, :320-337), ONE pair per call — and finally setBestMatch() for the mutual best pairs,
// which triangulates again, with uncertainty (:376-392), or gates again (:494-512).
//
// This class has the same interface (DenseMatcher::match<okvis_amd::BatchedKeyframeWindowMatching<G>> compiles and runs
// unchanged) and the same book-keeping on okvis::Estimator, but doSetup() does the geometry of ALL pairs that can come up in
// three batched calls of include/okvis_amd_frontend.h:
//   okvis_fe_project_landmarks   the projection loop of doSetup (:165-213),
//   okvis_fe_gate_3d2d           the gates of verifyMatch and setBestMatch for every pair within the descriptor threshold,
//   okvis_fe_stereo_triangulate  stereoTriangulate + getUncertainty for every such pair (ProbabilisticStereoTriangulator.cpp:178-355)
// and verifyMatch / setBestMatch read their pair's result from a table.  Compiled only where the OKVIS headers exist
// (oracle/ref/matcher_runtime.cpp runs it next to the reference's class on the same frames: tests/test_gpu_matcher_binding.py).
#pragma once
#if __has_include(<okvis/MatchingAlgorithm.hpp>) && __has_include(<okvis/Estimator.hpp>)
#define OKVIS_AMD_HAVE_MATCHING 1

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include <okvis/Estimator.hpp>
#include <okvis/FrameTypedefs.hpp>
#include <okvis/IdProvider.hpp>
#include <okvis/MatchingAlgorithm.hpp>
#include <okvis/MultiFrame.hpp>

#include "../../../include/okvis_amd_frontend.h"

namespace okvis_amd {

template <class CAMERA_GEOMETRY_T>
class BatchedKeyframeWindowMatching : public okvis::MatchingAlgorithm {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  typedef CAMERA_GEOMETRY_T camera_geometry_t;
  enum MatchingTypes { Match3D2D = 1, Match2D2D = 2 };   // (VioKeyframeWindowMatchingAlgorithm.hpp:74-77)

  // descriptorBytes: 48 for BRISK (what specificDescriptorDistance compares, VioKeyframeWindowMatchingAlgorithm.hpp:245-253)
  BatchedKeyframeWindowMatching(okvis::Estimator& estimator, int matchingType, float distanceThreshold, bool usePoseUncertainty = true,
                                int device = 0, int descriptorBytes = 48)
      : est_(&estimator), type_(matchingType), threshold_(distanceThreshold), usePoseUncertainty_(usePoseUncertainty),
        descBytes_(descriptorBytes) {
    const int rc = okvis_fe_create(&fe_, device);
    if (rc != OKVIS_BA_OK) throw std::runtime_error(std::string("okvis_fe_create: ") + okvis_ba_error_string(rc));
  }
  ~BatchedKeyframeWindowMatching() override { okvis_fe_destroy(fe_); }
  BatchedKeyframeWindowMatching(const BatchedKeyframeWindowMatching&) = delete;
  BatchedKeyframeWindowMatching& operator=(const BatchedKeyframeWindowMatching&) = delete;

  // (setFrames, :76-111)
  void setFrames(uint64_t mfIdA, uint64_t mfIdB, size_t camIdA, size_t camIdB) {
    if (mfIdA == mfIdB && camIdA == camIdB) throw std::runtime_error("trying to match identical frames.");
    idA_ = mfIdA, idB_ = mfIdB, camA_ = camIdA, camB_ = camIdB;
    A_ = est_->multiFrame(idA_);
    B_ = est_->multiFrame(idB_);
    fA_ = A_->template geometryAs<CAMERA_GEOMETRY_T>(camA_)->focalLengthU();
    fB_ = B_->template geometryAs<CAMERA_GEOMETRY_T>(camB_)->focalLengthU();
    okvis::kinematics::Transformation T_WSa, T_WSb;
    est_->getCameraSensorStates(idA_, camA_, T_SaCa_);
    est_->getCameraSensorStates(idB_, camB_, T_SbCb_);
    est_->get_T_WS(idA_, T_WSa);
    est_->get_T_WS(idB_, T_WSb);
    T_WCa_ = T_WSa * T_SaCa_;
    T_WCb_ = T_WSb * T_SbCb_;
    T_CbW_ = T_WCb_.inverse();
    T_CaCb_ = T_WCa_.inverse() * T_WCb_;
  }
  void setMatchingType(int matchingType) { type_ = matchingType; }

  size_t sizeA() const override { return A_->numKeypoints(camA_); }
  size_t sizeB() const override { return B_->numKeypoints(camB_); }
  float distanceThreshold() const override { return threshold_; }
  void setDistanceThreshold(float t) { threshold_ = t; }
  bool skipA(size_t a) const override { return skipA_[a]; }
  bool skipB(size_t b) const override { return skipB_[b]; }
  void reserveMatches(size_t) override {}
  size_t numMatches() { return numMatches_; }
  size_t numUncertainMatches() { return numUncertain_; }

  // (:132-146) the descriptor distance, and the pair's geometry from the table
  float distance(size_t a, size_t b) const override {
    const float d = (float)hamming(A_->keypointDescriptor(camA_, a), B_->keypointDescriptor(camB_, b));
    if (d < threshold_ && verifyMatch(a, b)) return d;
    return std::numeric_limits<float>::max();
  }
  bool verifyMatch(size_t a, size_t b) const {
    const auto it = pair_.find(key(a, b));
    if (it == pair_.end()) return false;   // (not within the descriptor threshold when doSetup looked: distance() never asks)
    return type_ == Match2D2D ? (flags_[it->second] & OKVIS_FE_TRI_VALID) != 0 : (flags_[it->second] & OKVIS_FE_GATE_VERIFIED) != 0;
  }

  // (doSetup, :121-262): the per-keypoint decisions as the reference takes them, the geometry in batches
  void doSetup() override {
    if (usePoseUncertainty_) throw std::runtime_error("No pose uncertainty use currently supported");
    // relative pose uncertainty (:131-146)
    double UO[36] = {0};
    for (int i = 0; i < 6; ++i) UO[7 * i] = i < 3 ? 1.0 : 1e-8;
    const uint64_t cur = est_->currentFrameId();
    double p_scale = 4e-8;
    if (est_->isInImuWindow(cur) && idA_ != idB_) {
      okvis::SpeedAndBias sb;
      est_->getSpeedAndBias(cur, 0, sb);
      const double s = std::max(1.0, sb.template head<3>().norm());
      p_scale = s * s * 1.0e-2;
    }
    for (int i = 0; i < 3; ++i) UO[7 * i] *= p_scale;
    numMatches_ = numUncertain_ = 0;
    const size_t nA = sizeA(), nB = sizeB();
    skipA_.assign(nA, false);
    skipB_.assign(nB, false);
    sigA_.resize(nA);
    sigB_.resize(nB);
    kpA_.resize(3 * nA);
    kpB_.resize(3 * nB);
    auto keypoints = [](okvis::MultiFrame& f, size_t cam, size_t n, std::vector<float>& kp, std::vector<double>& sig, double focal) {
      for (size_t k = 0; k < n; ++k) {
        Eigen::Vector2d uv;
        double size;
        f.getKeypoint(cam, k, uv);
        f.getKeypointSize(cam, k, size);
        kp[3 * k] = (float)uv[0], kp[3 * k + 1] = (float)uv[1], kp[3 * k + 2] = (float)size;
        sig[k] = std::sqrt(std::sqrt(2.0)) * (0.8 * size / 12.0) / focal;   // (:207-210)
      }
    };
    keypoints(*A_, camA_, nA, kpA_, sigA_, fA_);
    keypoints(*B_, camB_, nB, kpB_, sigB_, fB_);
    const okvis_fe_camera cA = camera(*A_, camA_), cB = camera(*B_, camB_);
    uv_.assign(2 * nA, 0.0);
    U_.assign(4 * nA, 0.0);
    if (type_ == Match3D2D) {
      // ---- the landmarks of A's keypoints into B (:165-213): one batched projection
      std::vector<double> hp(4 * nA, 0.0);
      std::vector<int> rows;   // keypoints that have a landmark to project
      for (size_t k = 0; k < nA; ++k) {
        const uint64_t lm = A_->landmarkId(camA_, k);
        if (lm == 0 || !est_->isLandmarkAdded(lm) || !est_->isLandmarkInitialized(lm)) {
          skipA_[k] = true;
          continue;
        }
        okvis::MapPoint mp;
        est_->getLandmark(lm, mp);
        for (int c = 0; c < 4; ++c) hp[4 * rows.size() + c] = mp.point[c];
        rows.push_back((int)k);
      }
      const int n = (int)rows.size();
      std::vector<double> uv(2 * (size_t)n), U(4 * (size_t)n);
      std::vector<uint8_t> st((size_t)n);
      const double P3[9] = {UO[0], 0, 0, 0, UO[7], 0, 0, 0, UO[14]};
      double T[7];
      pose7(T_CbW_, T);
      if (n > 0) check(okvis_fe_project_landmarks(fe_, &cB, T, P3, n, hp.data(), uv.data(), U.data(), st.data()), "okvis_fe_project_landmarks");
      for (int r = 0; r < n; ++r) {
        const size_t k = (size_t)rows[r];
        if (st[r] != OKVIS_FE_PROJ_SUCCESSFUL) {
          skipA_[k] = true;
          continue;
        }
        // (a landmark seen less than twice is not trusted for 3D-2D matching, :190-194)
        okvis::MapPoint mp;
        est_->getLandmark(A_->landmarkId(camA_, k), mp);
        if (mp.observations.size() < 2) {
          est_->setLandmarkInitialized(A_->landmarkId(camA_, k), false);
          skipA_[k] = true;
          continue;
        }
        uv_[2 * k] = uv[2 * r], uv_[2 * k + 1] = uv[2 * r + 1];
        std::memcpy(&U_[4 * k], &U[4 * r], 4 * sizeof(double));
      }
      for (size_t k = 0; k < nB; ++k) {   // (:241-252) a keypoint of B that already observes its landmark
        const uint64_t lm = B_->landmarkId(camB_, k);
        if (lm != 0 && est_->isLandmarkAdded(lm)) {
          okvis::MapPoint mp;
          est_->getLandmark(lm, mp);
          skipB_[k] = mp.observations.find(okvis::KeypointIdentifier(idB_, camB_, k)) != mp.observations.end();
        }
      }
    } else {
      for (size_t k = 0; k < nA; ++k) {   // (:215-229) initialised landmarks are not triangulated again
        const uint64_t lm = A_->landmarkId(camA_, k);
        if (lm != 0 && est_->isLandmarkAdded(lm) && est_->isLandmarkInitialized(lm)) skipA_[k] = true;
      }
      for (size_t k = 0; k < nB; ++k) {   // (:253-261)
        const uint64_t lm = B_->landmarkId(camB_, k);
        if (lm != 0 && est_->isLandmarkAdded(lm)) skipB_[k] = est_->isLandmarkInitialized(lm);
      }
    }
    // ---- every pair the matcher can ask about: both keypoints in play, descriptors within the threshold
    pairs_.clear();
    pair_.clear();
    for (size_t a = 0; a < nA; ++a) {
      if (skipA_[a]) continue;
      const unsigned char* da = A_->keypointDescriptor(camA_, a);
      for (size_t b = 0; b < nB; ++b) {
        if (skipB_[b]) continue;
        if ((float)hamming(da, B_->keypointDescriptor(camB_, b)) < threshold_) {
          pair_[key(a, b)] = (int)(pairs_.size() / 2);
          pairs_.push_back((int32_t)a);
          pairs_.push_back((int32_t)b);
        }
      }
    }
    const int np = (int)(pairs_.size() / 2);
    flags_.assign((size_t)np, 0);
    hp_.assign(4 * (size_t)np, 0.0);
    if (np == 0) return;
    if (type_ == Match3D2D) {
      std::vector<double> chi2((size_t)np);
      check(okvis_fe_gate_3d2d(fe_, (int)nA, uv_.data(), U_.data(), (int)nB, kpB_.data(), np, pairs_.data(), chi2.data(), flags_.data()),
            "okvis_fe_gate_3d2d");
    } else {
      // stereoTriangulate(a, b, ., ., max(sigma_a, sigma_b)) and getUncertainty for all of them (:310-317, :376-392)
      std::vector<double> sigma((size_t)np), cov(9 * (size_t)np);
      for (int i = 0; i < np; ++i) sigma[i] = std::max(sigA_[pairs_[2 * i]], sigB_[pairs_[2 * i + 1]]);
      double T[7];
      pose7(T_CaCb_, T);
      check(okvis_fe_stereo_triangulate(fe_, &cA, &cB, T, UO, (int)nA, kpA_.data(), (int)nB, kpB_.data(), np, pairs_.data(), sigma.data(),
                                        1, hp_.data(), cov.data(), flags_.data()),
            "okvis_fe_stereo_triangulate");
    }
  }

  // (setBestMatch, :362-527): the book-keeping on the estimator and the frames, the geometry from the table
  void setBestMatch(size_t a, size_t b, double /*distance*/) override {
    const auto it = pair_.find(key(a, b));
    if (it == pair_.end()) return;
    const int i = it->second;
    uint64_t lmA = A_->landmarkId(camA_, a), lmB = B_->landmarkId(camB_, b);
    if (type_ == Match2D2D) {
      if (lmA != 0 && lmB != 0) return;                      // both assigned already
      if (!(flags_[i] & OKVIS_FE_TRI_VALID)) return;
      const bool canInit = (flags_[i] & OKVIS_FE_TRI_CAN_INIT) != 0;
      const Eigen::Vector4d hP_Ca(hp_[4 * i], hp_[4 * i + 1], hp_[4 * i + 2], hp_[4 * i + 3]);
      const Eigen::Vector4d hP_W = T_WCa_ * hP_Ca;
      bool obsA = lmA == 0, obsB = lmB == 0, newBlock = false;
      uint64_t lm = 0;
      if (obsA && obsB) {
        lm = okvis::IdProvider::instance().newId();
        A_->setLandmarkId(camA_, a, lm);
        B_->setLandmarkId(camB_, b, lm);
        newBlock = true;
      } else {
        if (!obsA) {
          lm = lmA;
          if (!est_->isLandmarkAdded(lm)) newBlock = obsA = true;
        }
        if (!obsB) {
          lm = lmB;
          if (!est_->isLandmarkAdded(lm)) newBlock = obsB = true;
        }
      }
      if (newBlock) {
        est_->addLandmark(lm, hP_W);
        est_->setLandmarkInitialized(lm, canInit);
      } else if (canInit) {
        est_->setLandmarkInitialized(lm, true);
        est_->setLandmark(lm, hP_W);
      }
      if (obsA) {
        A_->setLandmarkId(camA_, a, lm);
        est_->template addObservation<camera_geometry_t>(lm, idA_, camA_, a);
      }
      if (obsB) {
        B_->setLandmarkId(camB_, b, lm);
        est_->template addObservation<camera_geometry_t>(lm, idB_, camB_, b);
      }
      if (canInit) est_->setLandmark(lm, hP_W);
    } else {
      if (!(flags_[i] & OKVIS_FE_GATE_ACCEPTED)) return;
      if (flags_[i] & OKVIS_FE_GATE_UNCERTAIN) ++numUncertain_;
      B_->setLandmarkId(camB_, b, lmA);
      okvis::MapPoint mp;
      est_->getLandmark(lmA, mp);
      if (mp.observations.find(okvis::KeypointIdentifier(idB_, camB_, b)) == mp.observations.end())
        est_->template addObservation<camera_geometry_t>(lmA, idB_, camB_, b);
    }
    ++numMatches_;
  }

  size_t candidatePairs() const { return pairs_.size() / 2; }

 private:
  static uint64_t key(size_t a, size_t b) { return ((uint64_t)a << 32) | (uint64_t)b; }
  static void check(int rc, const char* what) {
    if (rc != OKVIS_BA_OK) throw std::runtime_error(std::string(what) + ": " + okvis_ba_error_string(rc));
  }
  static void pose7(const okvis::kinematics::Transformation& T, double* out) {
    const Eigen::Vector3d r = T.r();
    const Eigen::Quaterniond q = T.q();
    out[0] = r[0], out[1] = r[1], out[2] = r[2], out[3] = q.x(), out[4] = q.y(), out[5] = q.z(), out[6] = q.w();
  }
  // popcount of the XOR of two descriptors (what brisk::Hamming::PopcntofXORed computes for BRISK's 48 bytes)
  uint32_t hamming(const unsigned char* x, const unsigned char* y) const {
    uint32_t n = 0;
    for (int i = 0; i < descBytes_; ++i) n += (uint32_t)__builtin_popcount((unsigned)(x[i] ^ y[i]));
    return n;
  }
  static okvis_fe_camera camera(okvis::MultiFrame& f, size_t cam) {
    okvis_fe_camera c;
    std::memset(&c, 0, sizeof(c));
    Eigen::VectorXd intr;
    f.geometry(cam)->getIntrinsics(intr);
    for (int i = 0; i < intr.size() && i < 12; ++i) c.intr[i] = intr[i];
    const std::string d = f.geometry(cam)->distortionType();
    c.model = d == "RadialTangentialDistortion"    ? OKVIS_BA_DIST_RADTAN
              : d == "EquidistantDistortion"       ? OKVIS_BA_DIST_EQUIDISTANT
              : d == "RadialTangentialDistortion8" ? OKVIS_BA_DIST_RADTAN8
                                                   : OKVIS_BA_DIST_NONE;
    c.width = (int32_t)f.geometry(cam)->imageWidth();
    c.height = (int32_t)f.geometry(cam)->imageHeight();
    return c;
  }

  okvis::Estimator* est_;
  okvis_fe_context* fe_ = nullptr;
  int type_;
  float threshold_;
  bool usePoseUncertainty_;
  int descBytes_;
  uint64_t idA_ = 0, idB_ = 0;
  size_t camA_ = 0, camB_ = 0;
  std::shared_ptr<okvis::MultiFrame> A_, B_;
  double fA_ = 0, fB_ = 0;
  okvis::kinematics::Transformation T_SaCa_, T_SbCb_, T_WCa_, T_WCb_, T_CbW_, T_CaCb_;
  std::vector<bool> skipA_, skipB_;
  std::vector<double> sigA_, sigB_, uv_, U_, hp_;
  std::vector<float> kpA_, kpB_;
  std::vector<int32_t> pairs_;
  std::vector<uint8_t> flags_;
  std::unordered_map<uint64_t, int> pair_;
  size_t numMatches_ = 0, numUncertain_ = 0;
};

}  // namespace okvis_amd
#endif
