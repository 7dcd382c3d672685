// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// The frontend binding EXECUTED (VERDICT r5, item 8 ii): the reference's own matching code —
// okvis_frontend/src/VioKeyframeWindowMatchingAlgorithm.cpp and okvis_matcher/src/{DenseMatcher, MatchingAlgorithm, ThreadPool}.cpp,
// compiled where they lie, unmodified, with <okvis/Estimator.hpp> resolved to the product's drop-in — matches two synthetic
// multi-frames the way okvis::Frontend does (Frontend.cpp: matchToKeyframes 3D-2D then 2D-2D, matchStereo), twice:
//   R  okvis::VioKeyframeWindowMatchingAlgorithm<G> as written: one ProbabilisticStereoTriangulator call / one chi-square gate per
//      candidate pair, on the CPU (the reference's triangulator, also compiled unmodified);
//   B  okvis_amd::BatchedKeyframeWindowMatching<G> (okvis_amd/csrc/host/okvis_matching_batched.hpp): the same interface and
//      book-keeping, the geometry of all candidate pairs in three okvis_fe_* calls on the GPU (include/okvis_amd_frontend.h);
// each on an estimator of its own, built from the same seed.  Compared: after every matching step the landmark every keypoint of
// both frames was assigned to, the numbers of matches and of uncertain matches; at the end every landmark's point, whether it
// counts as initialised, and its observations.  Exit code 0 = identical match sets and decisions, points within 1e-9.
// Built by oracle/ref/Makefile into oracle/_ref/matcher_runtime; run on the GPU by tests/test_gpu_matcher_binding.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <vector>

#include <okvis/Estimator.hpp>   // (the product's adapter: oracle/ref/product_shadow)
#ifndef OKVIS_AMD_HAVE_OKVIS
#error "the okvis headers were not found"
#endif
#include <okvis/DenseMatcher.hpp>
#include <okvis/VioKeyframeWindowMatchingAlgorithm.hpp>
#include <okvis/cameras/NCameraSystem.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>

#include "okvis_matching_batched.hpp"
#ifndef OKVIS_AMD_HAVE_MATCHING
#error "okvis_matching_batched.hpp did not find the okvis headers"
#endif

namespace google {
int eshim_log_warnings = 0;
}
// BRISK is not installed: the one function of it the reference's matching algorithm calls (hamming.h: the popcount of the XOR of
// two descriptors of numberOf128BitWords x 16 bytes), as a stand-in
namespace brisk {
unsigned int Hamming::PopcntofXORed(const unsigned char* a, const unsigned char* b, const int numberOf128BitWords) {
  unsigned int n = 0;
  for (int i = 0; i < 16 * numberOf128BitWords; ++i) n += (unsigned)__builtin_popcount((unsigned)(a[i] ^ b[i]));
  return n;
}
}  // namespace brisk

typedef okvis::cameras::PinholeCamera<okvis::cameras::RadialTangentialDistortion> Camera;
typedef okvis::VioKeyframeWindowMatchingAlgorithm<Camera> RefAlgorithm;
typedef okvis_amd::BatchedKeyframeWindowMatching<Camera> BatchedAlgorithm;

struct Scene {
  okvis::cameras::NCameraSystem ncs;
  std::shared_ptr<const Camera> geometry;
  std::vector<std::shared_ptr<const okvis::kinematics::Transformation> > T_SC;
  std::shared_ptr<okvis::Estimator> estimator;
  okvis::MultiFramePtr mf[2];
  // what each keypoint is: [frame][cam][k] = index of the 3D point, or -1 (a distractor)
  std::vector<int> truth[2][2];
};

static const uint64_t FRAME_ID[2] = {101, 102};

// the same scene from the same seed, every time
static bool build(Scene& S, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  std::normal_distribution<double> nrm(0.0, 1.0);
  S.geometry.reset(new Camera(752, 480, 458.654, 457.296, 367.215, 248.375,
                              okvis::cameras::RadialTangentialDistortion(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)));
  for (int i = 0; i < 2; ++i) {
    S.T_SC.push_back(std::shared_ptr<const okvis::kinematics::Transformation>(
        new okvis::kinematics::Transformation(Eigen::Vector3d(0, 0.11 * i, 0), Eigen::Quaterniond(1, 0, 0, 0))));
    S.ncs.addCamera(S.T_SC[i], S.geometry, okvis::cameras::NCameraSystem::RadialTangential, false);
  }
  S.estimator.reset(new okvis::Estimator());
  okvis::Estimator& est = *S.estimator;
  okvis::ExtrinsicsEstimationParameters ext(0, 0, 0, 0);
  est.addCamera(ext);
  est.addCamera(ext);
  okvis::ImuParameters imu;
  imu.a_max = 1000.0, imu.g_max = 1000.0, imu.sigma_g_c = 6.0e-4, imu.sigma_a_c = 2.0e-3, imu.sigma_bg = 0.03;
  imu.sigma_ba = 0.1, imu.sigma_gw_c = 3.0e-6, imu.sigma_aw_c = 2.0e-5, imu.tau = 3600.0, imu.g = 9.81;
  imu.a0 = Eigen::Vector3d(0, 0, 0);
  imu.rate = 100;
  est.addImu(imu);
  const double DT = 0.01, FRAME_DT = 0.5;
  okvis::ImuMeasurementDeque stream;
  for (int i = 0; i < 60; ++i)
    stream.push_back(okvis::ImuMeasurement(okvis::Time(1, 0) + okvis::Duration((i - 2) * DT),
                                           okvis::ImuSensorReadings(Eigen::Vector3d(0, 0, 0), Eigen::Vector3d(0, 0, imu.g))));
  for (int f = 0; f < 2; ++f) {
    const okvis::Time t = okvis::Time(1, 0) + okvis::Duration(f * FRAME_DT);
    S.mf[f].reset(new okvis::MultiFrame(S.ncs, t, FRAME_ID[f]));
    okvis::ImuMeasurementDeque d;
    for (const okvis::ImuMeasurement& m : stream)
      if (m.timeStamp >= (f ? t - okvis::Duration(FRAME_DT + 0.02) : t - okvis::Duration(0.02)) && m.timeStamp <= t + okvis::Duration(0.03))
        d.push_back(m);
    if (!est.addStates(S.mf[f], d, f == 0)) return false;
  }
  // the second frame 0.35 m further, a few degrees turned (the estimator's own propagation stood still: no velocity yet)
  okvis::kinematics::Transformation T_WS0;
  est.get_T_WS(FRAME_ID[0], T_WS0);
  {
    Eigen::Quaterniond dq(Eigen::AngleAxisd(0.05, Eigen::Vector3d(0.2, 1.0, -0.3).normalized()));
    okvis::kinematics::Transformation T_S0S1(Eigen::Vector3d(0.3, 0.15, 0.1), dq);
    est.set_T_WS(FRAME_ID[1], T_WS0 * T_S0S1);
    okvis::SpeedAndBias sb = okvis::SpeedAndBias::Zero();
    sb[0] = 0.7, sb[1] = 0.3;
    est.setSpeedAndBias(FRAME_ID[1], 0, sb);
  }
  // 3D points in front of the first camera, and what each image sees of them (+ distractors), in shuffled order
  const int NP = 260, ND = 40;
  std::vector<Eigen::Vector4d> pts;
  std::vector<std::vector<unsigned char> > desc(NP, std::vector<unsigned char>(48));
  const okvis::kinematics::Transformation T_WC0 = T_WS0 * (*S.T_SC[0]);
  for (int j = 0; j < NP; ++j) {
    const double z = 2.0 + 5.0 * (uni(rng) + 1.0);
    const Eigen::Vector4d p_C(0.55 * z * uni(rng), 0.38 * z * uni(rng), z, 1.0);
    pts.push_back(T_WC0 * p_C);
    for (int b = 0; b < 48; ++b) desc[j][b] = (unsigned char)(rng() & 0xFF);
  }
  for (int f = 0; f < 2; ++f) {
    okvis::kinematics::Transformation T_WS;
    est.get_T_WS(FRAME_ID[f], T_WS);
    for (size_t c = 0; c < 2; ++c) {
      const okvis::kinematics::Transformation T_CW = (T_WS * (*S.T_SC[c])).inverse();
      struct Kp {
        cv::KeyPoint kp;
        std::vector<unsigned char> d;
        int what;
      };
      std::vector<Kp> all;
      for (int j = 0; j < NP; ++j) {
        const Eigen::Vector4d p_C = T_CW * pts[j];
        Eigen::Vector2d uv;
        if (p_C[2] < 0.3 || S.geometry->project(p_C.head<3>(), &uv) != okvis::cameras::CameraBase::ProjectionStatus::Successful) continue;
        if ((rng() & 15) == 0) continue;   // not every point is detected in every image
        Kp k;
        k.kp = cv::KeyPoint((float)(uv[0] + 0.3 * nrm(rng)), (float)(uv[1] + 0.3 * nrm(rng)), 8.0f + 4.0f * (float)(rng() & 1));
        k.d = desc[j];
        for (int flips = 0; flips < 4; ++flips) k.d[rng() % 48] ^= (unsigned char)(1u << (rng() & 7));
        k.what = j;
        all.push_back(k);
      }
      for (int j = 0; j < ND; ++j) {
        Kp k;
        k.kp = cv::KeyPoint((float)(376 + 370 * uni(rng)), (float)(240 + 235 * uni(rng)), 8.0f);
        k.d.resize(48);
        for (int b = 0; b < 48; ++b) k.d[b] = (unsigned char)(rng() & 0xFF);
        if (j < ND / 2) {   // half of the distractors look like a real point (the geometry has to turn them down)
          k.d = desc[(size_t)(rng() % NP)];
          for (int flips = 0; flips < 6; ++flips) k.d[rng() % 48] ^= (unsigned char)(1u << (rng() & 7));
        }
        k.what = -1;
        all.push_back(k);
      }
      std::shuffle(all.begin(), all.end(), rng);
      std::vector<cv::KeyPoint> kps;
      cv::Mat D((int)all.size(), 48, CV_8UC1);
      S.truth[f][c].clear();
      for (size_t i = 0; i < all.size(); ++i) {
        kps.push_back(all[i].kp);
        for (int b = 0; b < 48; ++b) D.at<unsigned char>((int)i, b) = all[i].d[b];
        S.truth[f][c].push_back(all[i].what);
      }
      S.mf[f]->resetKeypoints(c, kps);
      S.mf[f]->resetDescriptors(c, D);
    }
  }
  return true;
}

// what a run leaves behind, in terms that do not depend on the ids the IdProvider happened to hand out
struct Record {
  std::vector<size_t> matches, uncertain;                       // per step
  std::vector<std::vector<long long> > assigned;                // per step: for every keypoint of every image, the landmark's signature
  struct Lm {
    Eigen::Vector4d point;
    bool initialised;
    std::set<long long> obs;
  };
  std::map<long long, Lm> landmarks;                            // by signature
};
static long long kpkey(int f, size_t c, size_t k) { return ((long long)f * 2 + (long long)c) * 100000 + (long long)k; }

template <class ALGORITHM>
static bool run(Scene& S, Record& R, const char* name) {
  okvis::Estimator& est = *S.estimator;
  okvis::DenseMatcher matcher(4);
  ALGORITHM algo(est, ALGORITHM::Match2D2D, 60.0f, false);
  struct Step {
    int fa, fb;
    size_t ca, cb;
    int type;
  };
  const Step steps[] = {{0, 0, 0, 1, ALGORITHM::Match2D2D},    // stereo in the first frame: landmarks come into being
                        {0, 1, 0, 0, ALGORITHM::Match3D2D},    // the new frame against the keyframe: landmarks to keypoints,
                        {0, 1, 1, 1, ALGORITHM::Match3D2D},
                        {0, 1, 0, 0, ALGORITHM::Match2D2D},    // then keypoints to keypoints for what is left (Frontend.cpp: matchToKeyframes)
                        {0, 1, 1, 1, ALGORITHM::Match2D2D},
                        {1, 1, 0, 1, ALGORITHM::Match3D2D},    // stereo in the new frame (matchStereo: 3D-2D first, then 2D-2D)
                        {1, 1, 0, 1, ALGORITHM::Match2D2D}};
  // signature of a landmark: its first observation in (frame, camera, keypoint) order — the same in both runs if they agree
  auto signature = [&](uint64_t lm) -> long long {
    okvis::MapPoint mp;
    if (lm == 0 || !est.isLandmarkAdded(lm) || !est.getLandmark(lm, mp)) return -1;
    long long best = -1;
    for (const auto& o : mp.observations) {
      const int f = o.first.frameId == FRAME_ID[0] ? 0 : 1;
      const long long k = kpkey(f, o.first.cameraIndex, o.first.keypointIndex);
      if (best < 0 || k < best) best = k;
    }
    return best;
  };
  for (const Step& st : steps) {
    algo.setFrames(FRAME_ID[st.fa], FRAME_ID[st.fb], st.ca, st.cb);
    algo.setMatchingType(st.type);
    matcher.match<ALGORITHM>(algo);
    R.matches.push_back(algo.numMatches());
    R.uncertain.push_back(algo.numUncertainMatches());
    std::vector<long long> as;
    for (int f = 0; f < 2; ++f)
      for (size_t c = 0; c < 2; ++c)
        for (size_t k = 0; k < S.mf[f]->numKeypoints(c); ++k) as.push_back(signature(S.mf[f]->landmarkId(c, k)));
    R.assigned.push_back(as);
    std::printf("%s step (%d cam %zu -> %d cam %zu, %s): %zu matches, %zu uncertain, %zu landmarks\n", name, st.fa, st.ca, st.fb, st.cb,
                st.type == ALGORITHM::Match2D2D ? "2D-2D" : "3D-2D", algo.numMatches(), algo.numUncertainMatches(), est.numLandmarks());
  }
  okvis::PointMap all;
  est.getLandmarks(all);
  for (const auto& kv : all) {
    Record::Lm L;
    L.point = kv.second.point;
    L.initialised = est.isLandmarkInitialized(kv.first);
    for (const auto& o : kv.second.observations)
      L.obs.insert(kpkey(o.first.frameId == FRAME_ID[0] ? 0 : 1, o.first.cameraIndex, o.first.keypointIndex));
    R.landmarks[signature(kv.first)] = L;
  }
  return true;
}

int main() {
  Scene SR, SB;
  if (!build(SR, 11) || !build(SB, 11)) return std::printf("scene: addStates failed\n"), 2;
  Record RR, RB;
  run<RefAlgorithm>(SR, RR, "reference");
  run<BatchedAlgorithm>(SB, RB, "batched  ");
  // how good the matching is against the truth (both runs must agree; this only says the scenario means something)
  size_t right = 0, wrong = 0;
  for (const auto& kv : RR.landmarks) {
    std::set<int> what;
    for (long long o : kv.second.obs) {
      const int f = (int)(o / 200000), c = (int)((o / 100000) % 2), k = (int)(o % 100000);
      what.insert(SR.truth[f][c][(size_t)k]);
    }
    if (what.size() == 1 && *what.begin() >= 0) ++right;
    else ++wrong;
  }
  std::printf("reference run: %zu landmarks whose observations all show one 3D point, %zu that mix points or distractors\n", right, wrong);
  int bad = 0;
  if (RR.matches != RB.matches || RR.uncertain != RB.uncertain) bad |= 1, std::printf("DIFFERENT: numbers of matches\n");
  for (size_t s = 0; s < RR.assigned.size(); ++s)
    if (RR.assigned[s] != RB.assigned[s]) {
      size_t n = 0;
      for (size_t i = 0; i < RR.assigned[s].size(); ++i) n += RR.assigned[s][i] != RB.assigned[s][i];
      bad |= 2, std::printf("DIFFERENT: step %zu, %zu keypoints assigned differently\n", s, n);
    }
  if (RR.landmarks.size() != RB.landmarks.size()) bad |= 4, std::printf("DIFFERENT: %zu vs %zu landmarks\n", RR.landmarks.size(), RB.landmarks.size());
  double worst = 0.0;
  size_t n_init = 0;
  for (const auto& kv : RR.landmarks) {
    const auto it = RB.landmarks.find(kv.first);
    if (it == RB.landmarks.end()) {
      bad |= 4;
      continue;
    }
    if (kv.second.initialised != it->second.initialised) bad |= 8, std::printf("DIFFERENT: landmark %lld initialised %d vs %d\n", kv.first, (int)kv.second.initialised, (int)it->second.initialised);
    if (kv.second.obs != it->second.obs) bad |= 16, std::printf("DIFFERENT: landmark %lld observations\n", kv.first);
    n_init += kv.second.initialised;
    const double e = (kv.second.point - it->second.point).norm() / std::max(1e-12, kv.second.point.norm());
    worst = std::max(worst, e);
  }
  if (worst > 1e-9) bad |= 32;
  std::printf("landmarks: %zu (%zu initialised), largest relative difference of a point %.2e\n", RR.landmarks.size(), n_init, worst);
  if (RR.landmarks.size() < 150 || n_init < 100) return std::printf("the scenario did not produce enough landmarks to mean anything\n"), 3;
  if (bad) return std::printf("MATCHER BINDING DIFFERS (code %d)\n", bad), 1;
  std::printf("MATCHER BINDING OK\n");
  return 0;
}
