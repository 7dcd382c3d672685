// Flat C wrapper of okvis_amd::Estimator so that the pytest suite (ctypes) can re-state the reference's
// integration test (okvis_ceres/test/TestEstimator.cpp).  Exceptions never cross the boundary: every
// function returns a status (>= 0 ok, -1 exception: message via okvis_est_last_error).
#include <cstring>
#include <string>

#include "estimator.hpp"
#include "okvis_config.hpp"
#include "replay.hpp"

using namespace okvis_amd;

namespace {
thread_local std::string g_err;
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
}  // namespace

extern "C" {

const char* okvis_est_last_error() { return g_err.c_str(); }

void* okvis_est_create(int device) {
  try {
    return new Estimator(device);
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void okvis_est_destroy(void* h) { delete static_cast<Estimator*>(h); }

int okvis_est_add_camera(void* h, const double sigmas[4]) {
  return guarded([&] {
    ExtrinsicsEstimationParameters p;
    p.sigma_absolute_translation = sigmas[0];
    p.sigma_absolute_orientation = sigmas[1];
    p.sigma_c_relative_translation = sigmas[2];
    p.sigma_c_relative_orientation = sigmas[3];
    return static_cast<Estimator*>(h)->addCamera(p);
  });
}
// prm = a_max g_max sigma_g_c sigma_a_c sigma_bg sigma_ba sigma_gw_c sigma_aw_c tau g a0x a0y a0z
int okvis_est_add_imu(void* h, const double prm[13]) {
  return guarded([&] {
    ImuParameters p;
    p.a_max = prm[0]; p.g_max = prm[1]; p.sigma_g_c = prm[2]; p.sigma_a_c = prm[3]; p.sigma_bg = prm[4];
    p.sigma_ba = prm[5]; p.sigma_gw_c = prm[6]; p.sigma_aw_c = prm[7]; p.tau = prm[8]; p.g = prm[9];
    p.a0 = {{prm[10], prm[11], prm[12]}};
    return static_cast<Estimator*>(h)->addImu(p);
  });
}

// a multiframe is created first, keypoints are appended, then it is handed to addStates
void* okvis_est_frame_create(uint64_t id, int64_t t_ns, int ncam, const double* T_SC /*[ncam][7]*/,
                             const double* intr /*[ncam][12]*/, const int* models) {
  MultiFramePtr* mf = new MultiFramePtr(new MultiFrame);
  (*mf)->id = id;
  (*mf)->t_ns = t_ns;
  for (int c = 0; c < ncam; ++c) {
    Transformation T;
    std::memcpy(T.p.data(), T_SC + 7 * c, 56);
    (*mf)->T_SC.push_back(T);
    CameraGeometry g;
    std::memcpy(g.intr.data(), intr + 12 * c, 96);
    g.model = models[c];
    (*mf)->geometry.push_back(g);
    (*mf)->keypoints.emplace_back();
  }
  return mf;
}
void okvis_est_frame_destroy(void* f) { delete static_cast<MultiFramePtr*>(f); }
int okvis_est_frame_add_keypoint(void* f, int cam, float x, float y, float size) {
  MultiFramePtr& mf = *static_cast<MultiFramePtr*>(f);
  mf->keypoints[cam].push_back(Keypoint{x, y, size});
  return (int)mf->keypoints[cam].size() - 1;
}

int okvis_est_add_states(void* h, void* frame, int n_imu, const int64_t* t, const double* gyr, const double* acc,
                         int asKeyframe) {
  return guarded([&] {
    ImuMeasurementDeque d(n_imu);
    for (int i = 0; i < n_imu; ++i) {
      d[i].t_ns = t[i];
      for (int c = 0; c < 3; ++c) {
        d[i].gyr[c] = gyr[3 * i + c];
        d[i].acc[c] = acc[3 * i + c];
      }
    }
    Estimator* e = static_cast<Estimator*>(h);
    if (e->addStates(*static_cast<MultiFramePtr*>(frame), d, asKeyframe != 0)) return 1;
    g_err = e->lastRefusal();   // (okvis_est_last_error)
    if (g_err.empty()) g_err = "addStates returned false without a reason (" + std::to_string(e->numFrames()) + " frames in the window)";
    return 0;
  });
}
int okvis_est_add_landmark(void* h, uint64_t id, const double hp[4]) {
  return guarded([&] { return static_cast<Estimator*>(h)->addLandmark(id, {{hp[0], hp[1], hp[2], hp[3]}}) ? 1 : 0; });
}
// returns 1 added, 0 duplicate, -1 exception
int okvis_est_add_observation(void* h, uint64_t lm, uint64_t pose, int cam, int kp, uint64_t* handle) {
  return guarded([&] {
    const uint64_t r = static_cast<Estimator*>(h)->addObservation(lm, pose, (size_t)cam, (size_t)kp);
    if (handle) *handle = r;
    return r ? 1 : 0;
  });
}
int okvis_est_remove_observation(void* h, uint64_t lm, uint64_t pose, int cam, int kp) {
  return guarded([&] { return static_cast<Estimator*>(h)->removeObservation(lm, pose, (size_t)cam, (size_t)kp) ? 1 : 0; });
}
int okvis_est_optimize(void* h, int numIter, int numThreads, int verbose, okvis_ba_summary* summary) {
  return guarded([&] {
    Estimator* e = static_cast<Estimator*>(h);
    e->optimize((size_t)numIter, (size_t)numThreads, verbose != 0);
    if (summary) *summary = e->summary();
    return 0;
  });
}
int okvis_est_set_time_limit(void* h, double limit, int minIter) {
  return guarded([&] { return static_cast<Estimator*>(h)->setOptimizationTimeLimit(limit, minIter) ? 1 : 0; });
}
int okvis_est_apply_marginalization(void* h, int numKeyframes, int numImuFrames) {
  return guarded([&] {
    MapPointVector removed;
    return static_cast<Estimator*>(h)->applyMarginalizationStrategy((size_t)numKeyframes, (size_t)numImuFrames, removed) ? 1 : 0;
  });
}
// test hook (tests/test_gpu_estimator.py): the next applyMarginalizationStrategy throws where its GPU call would be
int okvis_est_debug_fail_next_marginalization(void* h) {
  return guarded([&] {
    static_cast<Estimator*>(h)->debugFailNextMarginalization();
    return 1;
  });
}
// test hook: the marginalisation that is on its way fails where its numbers are waited for
int okvis_est_debug_fail_pending_marginalization(void* h) {
  return guarded([&] {
    static_cast<Estimator*>(h)->debugFailPendingMarginalization();
    return 1;
  });
}
// diagnostics hook (Estimator::setWindowObserver): cb(window, user) sees each window optimize() flattens
int okvis_est_set_window_observer(void* h, void (*cb)(const okvis_ba_window*, int, void*), void* user) {
  return guarded([&] {
    static_cast<Estimator*>(h)->setWindowObserver(cb, user);
    return 1;
  });
}
int okvis_est_get_options(void* h, okvis_ba_options* out) {
  return guarded([&] {
    *out = static_cast<Estimator*>(h)->options();
    return 1;
  });
}
int okvis_est_get_T_WS(void* h, uint64_t id, double out[7]) {
  return guarded([&] {
    Transformation T;
    if (!static_cast<Estimator*>(h)->get_T_WS(id, T)) return 0;
    std::memcpy(out, T.p.data(), 56);
    return 1;
  });
}
int okvis_est_get_speed_and_bias(void* h, uint64_t id, double out[9]) {
  return guarded([&] {
    SpeedAndBias sb;
    if (!static_cast<Estimator*>(h)->getSpeedAndBias(id, 0, sb)) return 0;
    std::memcpy(out, sb.data(), 72);
    return 1;
  });
}
int okvis_est_get_extrinsics(void* h, uint64_t id, int cam, double out[7]) {
  return guarded([&] {
    Transformation T;
    if (!static_cast<Estimator*>(h)->getCameraSensorStates(id, (size_t)cam, T)) return 0;
    std::memcpy(out, T.p.data(), 56);
    return 1;
  });
}
int okvis_est_get_landmark(void* h, uint64_t id, double point[4], double* quality, int* n_obs) {
  return guarded([&] {
    MapPoint mp;
    static_cast<Estimator*>(h)->getLandmark(id, mp);
    std::memcpy(point, mp.point.data(), 32);
    if (quality) *quality = mp.quality;
    if (n_obs) *n_obs = (int)mp.observations.size();
    return 1;
  });
}
// applyMarginalizationStrategy returning the removed landmark ids (the MapPointVector of the reference call)
int okvis_est_apply_marginalization2(void* h, int numKeyframes, int numImuFrames, int* n_removed, uint64_t* removed_ids,
                                     int capacity) {
  return guarded([&] {
    MapPointVector removed;
    const bool ok = static_cast<Estimator*>(h)->applyMarginalizationStrategy((size_t)numKeyframes, (size_t)numImuFrames, removed);
    if (n_removed) *n_removed = (int)removed.size();
    for (int i = 0; i < capacity && i < (int)removed.size(); ++i) removed_ids[i] = removed[i].id;
    return ok ? 1 : 0;
  });
}
int okvis_est_prior_info(void* h, int* dim, int* nblocks) {
  return guarded([&] {
    *dim = static_cast<Estimator*>(h)->priorDimension();
    *nblocks = (int)static_cast<Estimator*>(h)->priorNumBlocks();
    return 1;
  });
}
int okvis_est_frame_id_by_age(void* h, int age, uint64_t* id) {
  return guarded([&] {
    *id = static_cast<Estimator*>(h)->frameIdByAge((size_t)age);
    return 1;
  });
}
int okvis_est_is_keyframe(void* h, uint64_t id) {
  return guarded([&] { return static_cast<Estimator*>(h)->isKeyframe(id) ? 1 : 0; });
}
int okvis_est_is_in_imu_window(void* h, uint64_t id) {
  return guarded([&] { return static_cast<Estimator*>(h)->isInImuWindow(id) ? 1 : 0; });
}
int okvis_est_last_marg_info(void* h, double out[6]) {
  return guarded([&] {
    const auto& t = static_cast<Estimator*>(h)->lastMarginalizationInfo();
    for (int i = 0; i < 6; ++i) out[i] = t[i];
    return 1;
  });
}
int okvis_est_last_timings(void* h, double out[4]) {
  return guarded([&] {
    const auto& t = static_cast<Estimator*>(h)->lastOptimizeTimings();
    for (int i = 0; i < 4; ++i) out[i] = t[i];
    return 1;
  });
}
// solver options of the backend (okvis_ba_options): launch mode of the iteration loop
int okvis_est_set_use_graph(void* h, int use_graph) {
  return guarded([&] {
    static_cast<Estimator*>(h)->options().use_graph = use_graph;
    return 1;
  });
}
// setters (Estimator.hpp:366-411): 1 = done, 0 = unknown id
int okvis_est_set_T_WS(void* h, uint64_t id, const double T[7]) {
  return guarded([&] {
    Transformation t;
    std::copy(T, T + 7, t.p.begin());
    return static_cast<Estimator*>(h)->set_T_WS(id, t) ? 1 : 0;
  });
}
int okvis_est_set_speed_and_bias(void* h, uint64_t id, const double sb[9]) {
  return guarded([&] {
    SpeedAndBias v;
    std::copy(sb, sb + 9, v.begin());
    return static_cast<Estimator*>(h)->setSpeedAndBias(id, 0, v) ? 1 : 0;
  });
}
int okvis_est_set_extrinsics(void* h, uint64_t id, int cam, const double T[7]) {
  return guarded([&] {
    Transformation t;
    std::copy(T, T + 7, t.p.begin());
    return static_cast<Estimator*>(h)->setCameraSensorStates(id, (size_t)cam, t) ? 1 : 0;
  });
}
int okvis_est_set_landmark(void* h, uint64_t id, const double hp[4]) {
  return guarded([&] {
    std::array<double, 4> p{{hp[0], hp[1], hp[2], hp[3]}};
    return static_cast<Estimator*>(h)->setLandmark(id, p) ? 1 : 0;
  });
}
// 1 = optimize() sends the edits since the last call as a patch of the window the solver holds (default), 0 = flatten + upload
int okvis_est_set_use_patch(void* h, int on) {
  return guarded([&] {
    static_cast<Estimator*>(h)->setUsePatch(on != 0);
    return 1;
  });
}
int okvis_est_last_was_patch(void* h) {
  return guarded([&] { return static_cast<Estimator*>(h)->lastOptimizeWasPatch() ? 1 : 0; });
}
// 1 = the window the solver holds equals a freshly flattened one; 0 = it differs (what: okvis_est_last_error)
int okvis_est_debug_check_window(void* h) {
  return guarded([&] {
    const std::string d = static_cast<Estimator*>(h)->debugCheckWindow();
    if (!d.empty()) g_err = d;
    return d.empty() ? 1 : 0;
  });
}
int okvis_est_num_frames(void* h) { return (int)static_cast<Estimator*>(h)->numFrames(); }
int okvis_est_num_landmarks(void* h) { return (int)static_cast<Estimator*>(h)->numLandmarks(); }
long long okvis_est_debug_obs_slots(void* h) { return (long long)static_cast<Estimator*>(h)->debugObservationSlots(); }
int okvis_est_current_frame_id(void* h, uint64_t* id) {
  return guarded([&] {
    *id = static_cast<Estimator*>(h)->currentFrameId();
    return 1;
  });
}
// static helpers
int okvis_est_init_pose_from_imu(int n, const double* acc, double out[7]) {
  ImuMeasurementDeque d(n);
  for (int i = 0; i < n; ++i) {
    d[i].t_ns = i;
    d[i].gyr = {{0, 0, 0}};
    d[i].acc = {{acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]}};
  }
  Transformation T;
  const bool ok = Estimator::initPoseFromImu(d, T);
  std::memcpy(out, T.p.data(), 56);
  return ok ? 1 : 0;
}
int okvis_est_propagation(int n, const int64_t* t, const double* gyr, const double* acc, const double prm[13],
                          double T_WS[7], double sb[9], int64_t t_start, int64_t t_end) {
  ImuMeasurementDeque d(n);
  for (int i = 0; i < n; ++i) {
    d[i].t_ns = t[i];
    for (int c = 0; c < 3; ++c) {
      d[i].gyr[c] = gyr[3 * i + c];
      d[i].acc[c] = acc[3 * i + c];
    }
  }
  ImuParameters p;
  p.a_max = prm[0]; p.g_max = prm[1]; p.sigma_g_c = prm[2]; p.sigma_a_c = prm[3]; p.sigma_bg = prm[4];
  p.sigma_ba = prm[5]; p.sigma_gw_c = prm[6]; p.sigma_aw_c = prm[7]; p.tau = prm[8]; p.g = prm[9];
  Transformation T;
  std::memcpy(T.p.data(), T_WS, 56);
  SpeedAndBias s;
  std::memcpy(s.data(), sb, 72);
  const int r = Estimator::propagation(d, p, T, s, t_start, t_end);
  std::memcpy(T_WS, T.p.data(), 56);
  std::memcpy(sb, s.data(), 72);
  return r;
}

}  // extern "C"

// ---- dataset replay (replay.hpp): reads an ASL folder + recorded tracks and drives a fresh Estimator over it -------------
// opts: numKeyframes, numImuFrames, numIterations, numThreads, maxFrames, minObservationsPerLandmark, imuAsFloat (ints);
// imuOverlap (s).  stats[8]: frames, landmarks removed, has ground truth, rms position, final position, final rotation,
// mean ms optimize, mean ms marginalize.  trajectory_csv may be null.
extern "C" int okvis_replay_run(const char* path, int device, const int* opts, double imuOverlap, const char* trajectory_csv,
                                double* stats) {
  return guarded([&] {
    okvis_amd::ReplayOptions o;
    o.numKeyframes = opts[0], o.numImuFrames = opts[1], o.numIterations = opts[2], o.numThreads = opts[3];
    o.maxFrames = opts[4], o.minObservationsPerLandmark = opts[5];
    o.imuOverlap = imuOverlap;
    const okvis_amd::Recording rec = okvis_amd::readRecording(path, opts[6] != 0);
    Estimator est(device);
    const okvis_amd::ReplayResult r = okvis_amd::replay(rec, o, est);
    if (trajectory_csv && trajectory_csv[0]) okvis_amd::writeTrajectoryCsv(trajectory_csv, r);
    double mo = 0, mm = 0;
    for (const auto& f : r.frames) mo += f.msOptimize, mm += f.msMarginalize;
    const double n = r.frames.empty() ? 1.0 : (double)r.frames.size();
    stats[0] = (double)r.frames.size(), stats[1] = (double)r.landmarksRemoved, stats[2] = r.hasGroundTruth ? 1 : 0;
    stats[3] = r.rmsPosition, stats[4] = r.finalPosition, stats[5] = r.finalRotation, stats[6] = mo / n, stats[7] = mm / n;
    return 1;
  });
}
// readers alone (no GPU): counts of what a folder holds, for the CPU tests.  counts[6]: imu samples, cameras, ground-truth
// rows, frames, observations, landmarks
extern "C" int okvis_replay_probe(const char* path, int imuAsFloat, long long* counts, double* first_imu7, double* cam0_T_SC7,
                                  double* cam0_intr12, int* cam0_model, double* imu_params4) {
  return guarded([&] {
    const okvis_amd::Recording rec = okvis_amd::readRecording(path, imuAsFloat != 0);
    counts[0] = (long long)rec.imu.size(), counts[1] = (long long)rec.cameras.size(), counts[2] = (long long)rec.groundTruth.size();
    counts[3] = (long long)rec.frames.size(), counts[4] = (long long)rec.observations.size(), counts[5] = (long long)rec.landmarks.size();
    first_imu7[0] = (double)rec.imu[0].t_ns;
    for (int k = 0; k < 3; ++k) first_imu7[1 + k] = rec.imu[0].gyr[k], first_imu7[4 + k] = rec.imu[0].acc[k];
    const okvis_amd::Transformation T = rec.cameras[0].T_SC();
    for (int k = 0; k < 7; ++k) cam0_T_SC7[k] = T.p[k];
    for (int k = 0; k < 12; ++k) cam0_intr12[k] = rec.cameras[0].geometry.intr[k];
    *cam0_model = rec.cameras[0].geometry.model;
    imu_params4[0] = rec.imuParameters.sigma_g_c, imu_params4[1] = rec.imuParameters.sigma_gw_c;
    imu_params4[2] = rec.imuParameters.sigma_a_c, imu_params4[3] = rec.imuParameters.sigma_aw_c;
    return 1;
  });
}
// ---- the reference's configuration file (okvis_config.hpp), for the CPU tests -------------------------------------------------
// the parsed document as JSON text (scalars as strings with their cv::FileNode type: {"i": "5"}, {"r": "0.035"}, {"s": "abc"}),
// so that a test can hold the parser against another YAML implementation.  Returns the length needed (without the 0).
static void yamlToJson(const okvis_amd::YamlNode& n, std::string& o) {
  auto esc = [&](const std::string& t) {
    o += '"';
    for (char ch : t) {
      if (ch == '"' || ch == '\\') o += '\\';
      o += ch;
    }
    o += '"';
  };
  switch (n.kind) {
    case okvis_amd::YamlNode::NONE: o += "null"; break;
    case okvis_amd::YamlNode::SCALAR:
      o += n.isInt() ? "{\"i\": " : n.isReal() ? "{\"r\": " : "{\"s\": ";
      esc(n.scalar);
      o += "}";
      break;
    case okvis_amd::YamlNode::SEQ:
      o += "[";
      for (size_t k = 0; k < n.seq.size(); ++k) {
        if (k) o += ", ";
        yamlToJson(n.seq[k], o);
      }
      o += "]";
      break;
    case okvis_amd::YamlNode::MAP:
      o += "{\"m\": {";
      for (size_t k = 0; k < n.map.size(); ++k) {
        if (k) o += ", ";
        esc(n.map[k].first);
        o += ": ";
        yamlToJson(n.map[k].second, o);
      }
      o += "}}";
      break;
  }
}
extern "C" int okvis_yaml_to_json(const char* text, char* out, int capacity) {
  return guarded([&] {
    std::string o;
    yamlToJson(okvis_amd::parseYaml(text, "<text>"), o);
    if (out && capacity > 0) {
      const size_t n = std::min<size_t>(o.size(), (size_t)capacity - 1);
      std::memcpy(out, o.data(), n);
      out[n] = 0;
    }
    return (int)o.size();
  });
}
// ints[8 + 3 * cameras]: numKeyframes, numImuFrames, minIterations, maxIterations, cameraRate, imu rate, cameras, 0, then per camera
// width, height, model.  reals[36 + 19 * cameras]: timeLimit, imageDelay, timestampTolerance, the 4 extrinsics sigmas, the 13 IMU
// parameters (order of okvis_est_add_imu), T_BS (16), then per camera T_SC as r, q(xyzw) (7) and the 12 intrinsics.
// Returns the number of cameras (at most max_cameras are written).
extern "C" int okvis_config_read(const char* file, int max_cameras, int* ints, double* reals) {
  return guarded([&] {
    const okvis_amd::OkvisConfig c = okvis_amd::readOkvisConfig(file);
    ints[0] = c.numKeyframes, ints[1] = c.numImuFrames, ints[2] = c.minIterations, ints[3] = c.maxIterations, ints[4] = c.cameraRate;
    ints[5] = c.imu.rate, ints[6] = (int)c.cameras.size(), ints[7] = 0;
    const ImuParameters& p = c.imu;
    const double head[20] = {c.timeLimit, c.imageDelay, c.timestampTolerance, c.extrinsics.sigma_absolute_translation,
                             c.extrinsics.sigma_absolute_orientation, c.extrinsics.sigma_c_relative_translation,
                             c.extrinsics.sigma_c_relative_orientation, p.a_max, p.g_max, p.sigma_g_c, p.sigma_a_c, p.sigma_bg, p.sigma_ba,
                             p.sigma_gw_c, p.sigma_aw_c, p.tau, p.g, p.a0[0], p.a0[1], p.a0[2]};
    std::memcpy(reals, head, sizeof(head));
    std::memcpy(reals + 20, c.T_BS, sizeof(c.T_BS));
    for (size_t k = 0; k < c.cameras.size() && (int)k < max_cameras; ++k) {
      ints[8 + 3 * k] = c.cameras[k].width, ints[9 + 3 * k] = c.cameras[k].height, ints[10 + 3 * k] = c.cameras[k].geometry.model;
      const okvis_amd::Transformation T = c.cameras[k].T_SC();
      for (int e = 0; e < 7; ++e) reals[36 + 19 * k + e] = T.p[e];
      for (int e = 0; e < 12; ++e) reals[36 + 19 * k + 7 + e] = c.cameras[k].geometry.intr[e];
    }
    return (int)c.cameras.size();
  });
}
// cam<i>/data of an ASL folder as okvis_app_synchronous enumerates it; returns the number of images (t_ns: the first `capacity`)
extern "C" int okvis_asl_list_images(const char* path, int cam, long long* t_ns, int capacity) {
  return guarded([&] {
    const std::vector<okvis_amd::AslImage> v = okvis_amd::listAslImages(path, cam);
    for (size_t k = 0; k < v.size() && (int)k < capacity; ++k) t_ns[k] = v[k].t_ns;
    return (int)v.size();
  });
}
extern "C" int okvis_asl_read_image_csv(const char* file, long long* t_ns, int capacity) {
  return guarded([&] {
    const std::vector<okvis_amd::AslImage> v = okvis_amd::readAslImageCsv(file);
    for (size_t k = 0; k < v.size() && (int)k < capacity; ++k) t_ns[k] = v[k].t_ns;
    return (int)v.size();
  });
}
// okvis_replay_probe / okvis_replay_run with the calibration and the estimator parameters of a configuration file
// (`okvis_app_synchronous <config> <dataset folder>`); opts as okvis_replay_run, entries < 0 keep the file's value;
// use_time_limit: the file's ceres_options timeLimit / minIterations bound every optimize() (the non-blocking mode of ThreadedKFVio)
extern "C" int okvis_replay_probe_config(const char* path, const char* config, int imuAsFloat, long long* counts, double* cam0_T_SC7,
                                         double* cam0_intr12, int* cam0_model, double* imu_params13, double* extrinsics4) {
  return guarded([&] {
    const okvis_amd::OkvisConfig c = okvis_amd::readOkvisConfig(config);
    const okvis_amd::Recording rec = okvis_amd::readRecording(path, c, imuAsFloat != 0);
    counts[0] = (long long)rec.imu.size(), counts[1] = (long long)rec.cameras.size(), counts[2] = (long long)rec.groundTruth.size();
    counts[3] = (long long)rec.frames.size(), counts[4] = (long long)rec.observations.size(), counts[5] = (long long)rec.landmarks.size();
    const okvis_amd::Transformation T = rec.cameras[0].T_SC();
    for (int k = 0; k < 7; ++k) cam0_T_SC7[k] = T.p[k];
    for (int k = 0; k < 12; ++k) cam0_intr12[k] = rec.cameras[0].geometry.intr[k];
    *cam0_model = rec.cameras[0].geometry.model;
    const ImuParameters& p = rec.imuParameters;
    const double prm[13] = {p.a_max, p.g_max, p.sigma_g_c, p.sigma_a_c, p.sigma_bg, p.sigma_ba, p.sigma_gw_c, p.sigma_aw_c,
                            p.tau, p.g, p.a0[0], p.a0[1], p.a0[2]};
    std::memcpy(imu_params13, prm, sizeof(prm));
    extrinsics4[0] = rec.extrinsics.sigma_absolute_translation, extrinsics4[1] = rec.extrinsics.sigma_absolute_orientation;
    extrinsics4[2] = rec.extrinsics.sigma_c_relative_translation, extrinsics4[3] = rec.extrinsics.sigma_c_relative_orientation;
    return 1;
  });
}
extern "C" int okvis_replay_run_config(const char* path, const char* config, int device, const int* opts, double imuOverlap,
                                       int use_time_limit, const char* trajectory_csv, double* stats) {
  return guarded([&] {
    const okvis_amd::OkvisConfig c = okvis_amd::readOkvisConfig(config);
    okvis_amd::ReplayOptions o = okvis_amd::replayOptionsFrom(c);
    if (opts[0] >= 0) o.numKeyframes = opts[0];
    if (opts[1] >= 0) o.numImuFrames = opts[1];
    if (opts[2] >= 0) o.numIterations = opts[2];
    if (opts[3] >= 0) o.numThreads = opts[3];
    o.maxFrames = std::max(0, opts[4]), o.minObservationsPerLandmark = std::max(0, opts[5]);
    o.imuOverlap = imuOverlap;
    if (use_time_limit) o.timeLimit = c.timeLimit;
    const okvis_amd::Recording rec = okvis_amd::readRecording(path, c, opts[6] != 0);
    Estimator est(device);
    const okvis_amd::ReplayResult r = okvis_amd::replay(rec, o, est);
    if (trajectory_csv && trajectory_csv[0]) okvis_amd::writeTrajectoryCsv(trajectory_csv, r);
    double mo = 0, mm = 0;
    for (const auto& f : r.frames) mo += f.msOptimize, mm += f.msMarginalize;
    const double n = r.frames.empty() ? 1.0 : (double)r.frames.size();
    stats[0] = (double)r.frames.size(), stats[1] = (double)r.landmarksRemoved, stats[2] = r.hasGroundTruth ? 1 : 0;
    stats[3] = r.rmsPosition, stats[4] = r.finalPosition, stats[5] = r.finalRotation, stats[6] = mo / n, stats[7] = mm / n;
    return 1;
  });
}
// the whole recording as flat arrays (sizes from okvis_replay_probe): lets a test hand exactly what readRecording() read to
// another Estimator implementation.  imu_t[n], imu_ga[n][6] (gyr, acc); cam_T_SC[c][7], cam_intr[c][12], cam_model[c];
// imu_params13 as okvis_est_add_imu takes them; frames[n][3] = t_ns, id, keyframe; obs_i[n][3] = t_ns, cam, landmark,
// obs_f[n][3] = u, v, size (time order as replay() walks them); lm_i[n][2] = id, t_ns, lm_hp[n][4]
extern "C" int okvis_replay_read(const char* path, int imuAsFloat, long long* imu_t, double* imu_ga, double* cam_T_SC,
                                 double* cam_intr, int* cam_model, double* imu_params13, long long* frames, long long* obs_i,
                                 float* obs_f, long long* lm_i, double* lm_hp) {
  return guarded([&] {
    const okvis_amd::Recording rec = okvis_amd::readRecording(path, imuAsFloat != 0);
    for (size_t i = 0; i < rec.imu.size(); ++i) {
      imu_t[i] = rec.imu[i].t_ns;
      for (int k = 0; k < 3; ++k) imu_ga[6 * i + k] = rec.imu[i].gyr[k], imu_ga[6 * i + 3 + k] = rec.imu[i].acc[k];
    }
    for (size_t c = 0; c < rec.cameras.size(); ++c) {
      const okvis_amd::Transformation T = rec.cameras[c].T_SC();
      for (int k = 0; k < 7; ++k) cam_T_SC[7 * c + k] = T.p[k];
      for (int k = 0; k < 12; ++k) cam_intr[12 * c + k] = rec.cameras[c].geometry.intr[k];
      cam_model[c] = rec.cameras[c].geometry.model;
    }
    const ImuParameters& p = rec.imuParameters;
    const double prm[13] = {p.a_max, p.g_max, p.sigma_g_c, p.sigma_a_c, p.sigma_bg, p.sigma_ba, p.sigma_gw_c, p.sigma_aw_c,
                            p.tau, p.g, p.a0[0], p.a0[1], p.a0[2]};
    std::memcpy(imu_params13, prm, sizeof(prm));
    for (size_t i = 0; i < rec.frames.size(); ++i)
      frames[3 * i] = rec.frames[i].t_ns, frames[3 * i + 1] = (long long)rec.frames[i].id, frames[3 * i + 2] = rec.frames[i].keyframe;
    for (size_t i = 0; i < rec.observations.size(); ++i) {
      const okvis_amd::RecordedObservation& o = rec.observations[i];
      obs_i[3 * i] = o.t_ns, obs_i[3 * i + 1] = o.cam, obs_i[3 * i + 2] = (long long)o.landmark;
      obs_f[3 * i] = o.u, obs_f[3 * i + 1] = o.v, obs_f[3 * i + 2] = o.size;
    }
    for (size_t i = 0; i < rec.landmarks.size(); ++i) {
      lm_i[2 * i] = (long long)rec.landmarks[i].id, lm_i[2 * i + 1] = rec.landmarks[i].t_ns;
      for (int k = 0; k < 4; ++k) lm_hp[4 * i + k] = rec.landmarks[i].hp_S[k];
    }
    return 1;
  });
}
