// See replay.hpp.  Host code only; every numeric step of the backend runs behind okvis_amd::Estimator.
#include "replay.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

namespace okvis_amd {
namespace {

[[noreturn]] void fail(const std::string& file, size_t line, const std::string& what) {
  std::ostringstream s;
  s << file;
  if (line) s << ":" << line;
  s << ": " << what;
  throw std::runtime_error(s.str());
}

std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// one CSV file -> rows of trimmed fields; '#' comments and empty lines skipped; `line_no` of every kept row
struct Csv {
  std::string file;
  std::vector<std::vector<std::string>> rows;
  std::vector<size_t> line_no;
};
Csv readCsv(const std::string& file, size_t min_fields) {
  std::ifstream in(file);
  if (!in.good()) fail(file, 0, "cannot open");
  Csv c;
  c.file = file;
  std::string line;
  size_t n = 0;
  while (std::getline(in, line)) {
    ++n;
    const std::string t = trim(line);
    if (t.empty() || t[0] == '#') continue;
    std::vector<std::string> f;
    std::stringstream ss(t);
    std::string cell;
    while (std::getline(ss, cell, ',')) f.push_back(trim(cell));
    if (f.size() < min_fields) fail(file, n, "expected at least " + std::to_string(min_fields) + " fields, found " + std::to_string(f.size()));
    c.rows.push_back(f);
    c.line_no.push_back(n);
  }
  return c;
}
int64_t toInt64(const Csv& c, size_t r, size_t k) {
  const std::string& s = c.rows[r][k];
  char* end = nullptr;
  const long long v = std::strtoll(s.c_str(), &end, 10);
  if (s.empty() || *end != 0) fail(c.file, c.line_no[r], "not an integer: '" + s + "'");
  return (int64_t)v;
}
double toDouble(const Csv& c, size_t r, size_t k) {
  const std::string& s = c.rows[r][k];
  char* end = nullptr;
  const double v = std::strtod(s.c_str(), &end);
  if (s.empty() || *end != 0 || !std::isfinite(v)) fail(c.file, c.line_no[r], "not a finite number: '" + s + "'");
  return v;
}

// ---- the subset of YAML the ASL sensor files use: "key: scalar", "key: [a, b, ...]" (possibly over several lines), nested
// "T_BS:" / "data: [...]" --------------------------------------------------------------------------------------------
struct Yaml {
  std::string file;
  std::map<std::string, std::string> scalar;
  std::map<std::string, std::vector<double>> list;
};
Yaml readYaml(const std::string& file) {
  std::ifstream in(file);
  if (!in.good()) fail(file, 0, "cannot open");
  Yaml y;
  y.file = file;
  std::string line, parent;
  size_t n = 0;
  while (std::getline(in, line)) {
    ++n;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    if (trim(line).empty() || trim(line)[0] == '%') continue;
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    const bool nested = line[0] == ' ' || line[0] == '\t';
    std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
    if (!nested) parent.clear();
    if (val.empty()) {  // "T_BS:" opens a mapping
      if (!nested) parent = key;
      continue;
    }
    if (nested && !parent.empty()) key = parent + "." + key;
    if (val[0] == '[') {
      std::string all = val;
      while (all.find(']') == std::string::npos) {
        if (!std::getline(in, line)) fail(file, n, "unterminated list for key '" + key + "'");
        ++n;
        all += " " + trim(line);
      }
      all = all.substr(1, all.find(']') - 1);
      std::vector<double> v;
      std::stringstream ss(all);
      std::string cell;
      while (std::getline(ss, cell, ',')) {
        cell = trim(cell);
        if (cell.empty()) continue;
        char* end = nullptr;
        const double d = std::strtod(cell.c_str(), &end);
        if (*end != 0) fail(file, n, "not a number in list '" + key + "': '" + cell + "'");
        v.push_back(d);
      }
      y.list[key] = v;
    } else {
      y.scalar[key] = val;
    }
  }
  return y;
}
const std::vector<double>& need(const Yaml& y, const std::string& key, size_t n) {
  auto it = y.list.find(key);
  if (it == y.list.end()) fail(y.file, 0, "missing list '" + key + "'");
  if (it->second.size() < n) fail(y.file, 0, "list '" + key + "' has " + std::to_string(it->second.size()) + " entries, " + std::to_string(n) + " needed");
  return it->second;
}

// rotation matrix (row-major 3x3 inside a 4x4) -> unit quaternion xyzw
void rotToQuat(const double* T, double q[4]) {
  const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
  const double tr = m00 + m11 + m22;
  double x, y, z, w;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s, x = (m21 - m12) / s, y = (m02 - m20) / s, z = (m10 - m01) / s;
  } else if (m00 > m11 && m00 > m22) {
    const double s = std::sqrt(1.0 + m00 - m11 - m22) * 2;
    w = (m21 - m12) / s, x = 0.25 * s, y = (m01 + m10) / s, z = (m02 + m20) / s;
  } else if (m11 > m22) {
    const double s = std::sqrt(1.0 + m11 - m00 - m22) * 2;
    w = (m02 - m20) / s, x = (m01 + m10) / s, y = 0.25 * s, z = (m12 + m21) / s;
  } else {
    const double s = std::sqrt(1.0 + m22 - m00 - m11) * 2;
    w = (m10 - m01) / s, x = (m02 + m20) / s, y = (m12 + m21) / s, z = 0.25 * s;
  }
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  q[0] = x / n, q[1] = y / n, q[2] = z / n, q[3] = w / n;
}
void quatRotate(const double q[4], const double v[3], double out[3]) {  // q = xyzw
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
  out[0] = v[0] + w * tx + (y * tz - z * ty);
  out[1] = v[1] + w * ty + (z * tx - x * tz);
  out[2] = v[2] + w * tz + (x * ty - y * tx);
}
void quatMul(const double a[4], const double b[4], double o[4]) {  // xyzw
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}

}  // namespace

Transformation AslCamera::T_SC() const {
  Transformation T;
  T.p[0] = T_BS[3], T.p[1] = T_BS[7], T.p[2] = T_BS[11];
  double q[4];
  rotToQuat(T_BS, q);
  for (int k = 0; k < 4; ++k) T.p[3 + k] = q[k];
  return T;
}

std::vector<ImuMeasurement> readAslImuCsv(const std::string& file, bool as_float) {
  const Csv c = readCsv(file, 7);
  std::vector<ImuMeasurement> out;
  out.reserve(c.rows.size());
  for (size_t r = 0; r < c.rows.size(); ++r) {
    ImuMeasurement m;
    m.t_ns = toInt64(c, r, 0);
    if (!out.empty() && m.t_ns <= out.back().t_ns) fail(file, c.line_no[r], "IMU timestamps must increase");
    for (int k = 0; k < 3; ++k) {
      const double g = toDouble(c, r, 1 + k), a = toDouble(c, r, 4 + k);
      m.gyr[k] = as_float ? (double)(float)g : g;  // std::stof in okvis_app_synchronous.cpp:337-349
      m.acc[k] = as_float ? (double)(float)a : a;
    }
    out.push_back(m);
  }
  if (out.empty()) fail(file, 0, "no imu messages present");  // okvis_app_synchronous.cpp:250-253
  return out;
}

std::vector<AslGroundTruth> readAslGroundTruthCsv(const std::string& file) {
  const Csv c = readCsv(file, 17);
  std::vector<AslGroundTruth> out;
  for (size_t r = 0; r < c.rows.size(); ++r) {
    AslGroundTruth g;
    g.t_ns = toInt64(c, r, 0);
    if (!out.empty() && g.t_ns <= out.back().t_ns) fail(file, c.line_no[r], "ground-truth timestamps must increase");
    for (int k = 0; k < 3; ++k) g.p[k] = toDouble(c, r, 1 + k);
    for (int k = 0; k < 4; ++k) g.q_wxyz[k] = toDouble(c, r, 4 + k);
    for (int k = 0; k < 3; ++k) g.v[k] = toDouble(c, r, 8 + k);
    for (int k = 0; k < 3; ++k) g.bg[k] = toDouble(c, r, 11 + k);
    for (int k = 0; k < 3; ++k) g.ba[k] = toDouble(c, r, 14 + k);
    out.push_back(g);
  }
  return out;
}

AslCamera readAslCameraYaml(const std::string& file) {
  const Yaml y = readYaml(file);
  AslCamera cam;
  const std::vector<double>& T = need(y, "T_BS.data", 16);
  std::copy(T.begin(), T.begin() + 16, cam.T_BS);
  const std::vector<double>& in = need(y, "intrinsics", 4);
  for (int k = 0; k < 4; ++k) cam.geometry.intr[k] = in[k];
  auto res = y.list.find("resolution");
  if (res != y.list.end() && res->second.size() >= 2) cam.width = (int)res->second[0], cam.height = (int)res->second[1];
  auto dm = y.scalar.find("distortion_model");
  const std::string model = dm == y.scalar.end() ? "radial-tangential" : dm->second;
  auto dc = y.list.find("distortion_coefficients");
  const std::vector<double> d = dc == y.list.end() ? std::vector<double>() : dc->second;
  // the ASL name ("radial-tangential") and the names the reference's YAML reader understands (okvis_common/src/VioParametersReader.cpp:320-366)
  if (model == "radial-tangential" || model == "radialtangential" || model == "plumb_bob") {
    cam.geometry.model = d.size() > 4 ? OKVIS_BA_DIST_RADTAN8 : OKVIS_BA_DIST_RADTAN;
  } else if (model == "radial-tangential8" || model == "radialtangential8" || model == "plumb_bob8") {
    cam.geometry.model = OKVIS_BA_DIST_RADTAN8;
  } else if (model == "equidistant" || model == "equdistant") {
    cam.geometry.model = OKVIS_BA_DIST_EQUIDISTANT;
  } else if (model == "none") {
    cam.geometry.model = OKVIS_BA_DIST_NONE;
  } else {
    fail(file, 0, "unknown distortion_model '" + model + "'");
  }
  for (size_t k = 0; k < d.size() && k < 8; ++k) cam.geometry.intr[4 + k] = d[k];
  return cam;
}

ImuParameters readAslImuYaml(const std::string& file, const ImuParameters& base) {
  const Yaml y = readYaml(file);
  ImuParameters p = base;
  auto get = [&](const char* key, double& dst) {
    auto it = y.scalar.find(key);
    if (it == y.scalar.end()) return;
    char* end = nullptr;
    const double v = std::strtod(it->second.c_str(), &end);
    if (*end != 0) fail(file, 0, std::string("not a number for '") + key + "': '" + it->second + "'");
    dst = v;
  };
  get("gyroscope_noise_density", p.sigma_g_c);
  get("gyroscope_random_walk", p.sigma_gw_c);
  get("accelerometer_noise_density", p.sigma_a_c);
  get("accelerometer_random_walk", p.sigma_aw_c);
  double rate = p.rate;
  get("rate_hz", rate);
  p.rate = (int)rate;
  return p;
}

Recording readRecording(const std::string& path, bool imu_as_float) {
  Recording rec;
  rec.imu = readAslImuCsv(path + "/imu0/data.csv", imu_as_float);
  {
    std::ifstream probe(path + "/imu0/sensor.yaml");
    if (probe.good()) rec.imuParameters = readAslImuYaml(path + "/imu0/sensor.yaml");
  }
  for (int i = 0;; ++i) {
    const std::string f = path + "/cam" + std::to_string(i) + "/sensor.yaml";
    std::ifstream probe(f);
    if (!probe.good()) break;
    rec.cameras.push_back(readAslCameraYaml(f));
  }
  if (rec.cameras.empty()) fail(path + "/cam0/sensor.yaml", 0, "no camera calibration found");
  {
    const std::string f = path + "/state_groundtruth_estimate0/data.csv";
    std::ifstream probe(f);
    if (probe.good()) rec.groundTruth = readAslGroundTruthCsv(f);
  }
  readRecordedTracks(path, rec);
  return rec;
}

void readRecordedTracks(const std::string& path, Recording& rec) {
  if (rec.cameras.empty()) fail(path, 0, "readRecordedTracks: the recording has no cameras yet");
  {
    const std::string f = path + "/okvis_amd_tracks/frames.csv";
    const Csv c = readCsv(f, 3);
    std::set<uint64_t> ids;
    for (size_t r = 0; r < c.rows.size(); ++r) {
      RecordedFrame fr{toInt64(c, r, 0), (uint64_t)toInt64(c, r, 1), toInt64(c, r, 2) != 0};
      if (!rec.frames.empty() && fr.t_ns <= rec.frames.back().t_ns) fail(f, c.line_no[r], "frame timestamps must increase");
      if (!rec.frames.empty() && fr.id <= rec.frames.back().id) fail(f, c.line_no[r], "frame ids must increase");
      rec.frames.push_back(fr);
    }
    if (rec.frames.empty()) fail(f, 0, "no frames");
  }
  {
    const std::string f = path + "/okvis_amd_tracks/landmarks.csv";
    const Csv c = readCsv(f, 6);
    std::set<uint64_t> ids;
    std::set<int64_t> times;
    for (const RecordedFrame& fr : rec.frames) times.insert(fr.t_ns);
    for (size_t r = 0; r < c.rows.size(); ++r) {
      RecordedLandmark l;
      l.id = (uint64_t)toInt64(c, r, 0);
      if (!ids.insert(l.id).second) fail(f, c.line_no[r], "landmark id appears twice");
      l.t_ns = toInt64(c, r, 1);
      if (!times.count(l.t_ns)) fail(f, c.line_no[r], "landmark expressed in a frame that does not exist");
      for (int k = 0; k < 4; ++k) l.hp_S[k] = toDouble(c, r, 2 + k);
      rec.landmarks.push_back(l);
    }
  }
  {
    const std::string f = path + "/okvis_amd_tracks/observations.csv";
    const Csv c = readCsv(f, 6);
    std::set<int64_t> times;
    for (const RecordedFrame& fr : rec.frames) times.insert(fr.t_ns);
    std::set<uint64_t> lms;
    for (const RecordedLandmark& l : rec.landmarks) lms.insert(l.id);
    for (size_t r = 0; r < c.rows.size(); ++r) {
      RecordedObservation o;
      o.t_ns = toInt64(c, r, 0);
      o.cam = (int)toInt64(c, r, 1);
      o.u = (float)toDouble(c, r, 2), o.v = (float)toDouble(c, r, 3), o.size = (float)toDouble(c, r, 4);
      o.landmark = (uint64_t)toInt64(c, r, 5);
      if (!times.count(o.t_ns)) fail(f, c.line_no[r], "observation at a time that is no frame");
      if (o.cam < 0 || o.cam >= (int)rec.cameras.size()) fail(f, c.line_no[r], "camera index out of range");
      if (!lms.count(o.landmark)) fail(f, c.line_no[r], "observation of an unknown landmark");
      rec.observations.push_back(o);
    }
    std::stable_sort(rec.observations.begin(), rec.observations.end(),
                     [](const RecordedObservation& a, const RecordedObservation& b) { return a.t_ns < b.t_ns; });
  }
}

ReplayResult replay(const Recording& rec, const ReplayOptions& opt, Estimator& est) {
  typedef std::chrono::steady_clock clk;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (size_t i = 0; i < rec.cameras.size(); ++i) est.addCamera(rec.extrinsics);  // all zero = fixed extrinsics (EuRoC config)
  est.addImu(rec.imuParameters);
  std::map<uint64_t, const RecordedLandmark*> landmarkOf;
  for (const RecordedLandmark& l : rec.landmarks) landmarkOf[l.id] = &l;
  std::map<uint64_t, int> tracked;  // recorded observations per landmark
  for (const RecordedObservation& o : rec.observations) tracked[o.landmark]++;
  std::set<uint64_t> gone;
  std::map<int64_t, uint64_t> frameIdAt;
  for (const RecordedFrame& fr : rec.frames) frameIdAt[fr.t_ns] = fr.id;

  ReplayResult out;
  const int64_t overlap = (int64_t)std::llround(opt.imuOverlap * 1e9);
  size_t obsAt = 0, imuBegin = 0;
  int64_t lastT = 0;
  const size_t nFrames = opt.maxFrames > 0 ? std::min<size_t>(opt.maxFrames, rec.frames.size()) : rec.frames.size();
  for (size_t k = 0; k < nFrames; ++k) {
    const RecordedFrame& fr = rec.frames[k];
    MultiFramePtr mf(new MultiFrame);
    mf->id = fr.id;
    mf->t_ns = fr.t_ns;
    for (const AslCamera& c : rec.cameras) {
      mf->T_SC.push_back(c.T_SC());
      mf->geometry.push_back(c.geometry);
    }
    mf->keypoints.resize(rec.cameras.size());
    // IMU measurements between the previous state - overlap and this frame + overlap (ThreadedKFVio.cpp:472-474; for the
    // first frame the deque only has to cover the frame time, Estimator.cpp:121 initPoseFromImu)
    const int64_t t0 = (k ? lastT : fr.t_ns) - overlap, t1 = fr.t_ns + overlap;
    while (imuBegin < rec.imu.size() && rec.imu[imuBegin].t_ns < t0) ++imuBegin;
    ImuMeasurementDeque deque;
    for (size_t i = imuBegin; i < rec.imu.size() && rec.imu[i].t_ns <= t1; ++i) deque.push_back(rec.imu[i]);
    if (!est.addStates(mf, deque, fr.keyframe))
      throw std::runtime_error("replay: addStates failed at frame " + std::to_string(fr.id) + " (" + std::to_string(deque.size()) +
                               " IMU measurements)");
    lastT = fr.t_ns;
    int nObs = 0;
    for (; obsAt < rec.observations.size() && rec.observations[obsAt].t_ns <= fr.t_ns; ++obsAt) {
      const RecordedObservation& o = rec.observations[obsAt];
      if (o.t_ns != fr.t_ns || gone.count(o.landmark)) continue;
      if (tracked[o.landmark] < opt.minObservationsPerLandmark) continue;
      if (!est.isLandmarkAdded(o.landmark)) {
        // hp_W = T_WS(frame of the triangulation, current estimate) hp_S
        const RecordedLandmark* l = landmarkOf[o.landmark];
        Transformation T;
        if (!est.get_T_WS(frameIdAt[l->t_ns], T)) continue;  // its frame has left the window: the track is not started
        double pw[3];
        quatRotate(T.p.data() + 3, l->hp_S, pw);
        est.addLandmark(l->id, {{pw[0] + l->hp_S[3] * T.p[0], pw[1] + l->hp_S[3] * T.p[1], pw[2] + l->hp_S[3] * T.p[2], l->hp_S[3]}});
      }
      std::vector<Keypoint>& kps = mf->keypoints[o.cam];
      kps.push_back(Keypoint{o.u, o.v, o.size});
      if (est.addObservation(o.landmark, fr.id, (size_t)o.cam, kps.size() - 1) != 0) ++nObs;
    }
    // ThreadedKFVio.cpp:313-318 (blocking: no limit, max_iterations as the minimum) / :526-530 (the budget left of timeLimit)
    if (opt.timeLimit >= 0) est.setOptimizationTimeLimit(opt.timeLimit, opt.minIterations);
    const auto a = clk::now();
    est.optimize((size_t)opt.numIterations, (size_t)opt.numThreads, false);
    const auto b = clk::now();
    const std::array<double, 4> optT = est.lastOptimizeTimings();
    MapPointVector removed;
    est.applyMarginalizationStrategy((size_t)opt.numKeyframes, (size_t)opt.numImuFrames, removed);
    const auto c = clk::now();
    for (const MapPoint& mp : removed) gone.insert(mp.id);
    out.landmarksRemoved += removed.size();
    ReplayFrameResult r;
    r.t_ns = fr.t_ns;
    r.id = fr.id;
    est.get_T_WS(fr.id, r.T_WS);
    est.getSpeedAndBias(fr.id, 0, r.speedAndBias);
    r.observations = nObs;
    r.landmarksInWindow = (int)est.numLandmarks();
    r.framesInWindow = (int)est.numFrames();
    r.iterations = est.summary().iterations;
    r.initialCost = est.summary().initial_cost;
    r.finalCost = est.summary().final_cost;
    r.msOptimize = ms(a, b);
    r.msMarginalize = ms(b, c);
    r.msFlatten = optT[0], r.msUpload = optT[1], r.msIterations = optT[2], r.msDownload = optT[3];
    {
      const std::array<double, 6>& mi = est.lastMarginalizationInfo();
      r.msMargFlatten = mi[0], r.msMargUpload = mi[1], r.msMargCall = mi[2];
    }
    out.frames.push_back(r);
  }

  if (!rec.groundTruth.empty() && !out.frames.empty()) {
    // the estimator's world frame is gravity aligned with free yaw and origin at the first pose: align it to the ground truth
    // with the 4-DoF-free rigid transform that maps the first estimated pose onto the first ground-truth pose
    auto gtAt = [&](int64_t t, double p[3], double q[4]) {  // nearest sample; q xyzw
      auto it = std::lower_bound(rec.groundTruth.begin(), rec.groundTruth.end(), t,
                                 [](const AslGroundTruth& g, int64_t tt) { return g.t_ns < tt; });
      if (it == rec.groundTruth.end()) --it;
      if (it != rec.groundTruth.begin() && std::llabs((it - 1)->t_ns - t) < std::llabs(it->t_ns - t)) --it;
      for (int k = 0; k < 3; ++k) p[k] = it->p[k];
      q[0] = it->q_wxyz[1], q[1] = it->q_wxyz[2], q[2] = it->q_wxyz[3], q[3] = it->q_wxyz[0];
    };
    double p0g[3], q0g[4];
    gtAt(out.frames[0].t_ns, p0g, q0g);
    const double* e0 = out.frames[0].T_WS.p.data();
    const double q0e_inv[4] = {-e0[3], -e0[4], -e0[5], e0[6]};
    double qA[4];
    quatMul(q0g, q0e_inv, qA);  // rotation estimator-world -> ground-truth-world
    double se = 0;
    for (const ReplayFrameResult& r : out.frames) {
      double pg[3], qg[4], d[3] = {r.T_WS.p[0] - e0[0], r.T_WS.p[1] - e0[1], r.T_WS.p[2] - e0[2]}, pa[3];
      gtAt(r.t_ns, pg, qg);
      quatRotate(qA, d, pa);
      double e2 = 0;
      for (int k = 0; k < 3; ++k) e2 += (pa[k] + p0g[k] - pg[k]) * (pa[k] + p0g[k] - pg[k]);
      se += e2;
      out.finalPosition = std::sqrt(e2);
      double qe[4], qerr[4];
      quatMul(qA, r.T_WS.p.data() + 3, qe);
      const double qg_inv[4] = {-qg[0], -qg[1], -qg[2], qg[3]};
      quatMul(qe, qg_inv, qerr);
      out.finalRotation = 2 * std::sqrt(qerr[0] * qerr[0] + qerr[1] * qerr[1] + qerr[2] * qerr[2]);
    }
    out.rmsPosition = std::sqrt(se / out.frames.size());
    out.hasGroundTruth = true;
  }
  return out;
}

void writeTrajectoryCsv(const std::string& file, const ReplayResult& r) {
  std::ofstream o(file);
  if (!o.good()) fail(file, 0, "cannot open for writing");
  o << "#timestamp [ns],p_WS_W_x,p_WS_W_y,p_WS_W_z,q_WS_x,q_WS_y,q_WS_z,q_WS_w,v_WS_W_x,v_WS_W_y,v_WS_W_z,b_g_x,b_g_y,b_g_z,"
       "b_a_x,b_a_y,b_a_z,frames,landmarks,observations,iterations,initial_cost,final_cost,ms_optimize,ms_marginalize\n";
  o.precision(17);
  for (const ReplayFrameResult& f : r.frames) {
    o << f.t_ns;
    for (int k = 0; k < 7; ++k) o << "," << f.T_WS.p[k];
    for (int k = 0; k < 9; ++k) o << "," << f.speedAndBias[k];
    o << "," << f.framesInWindow << "," << f.landmarksInWindow << "," << f.observations << "," << f.iterations << ","
      << f.initialCost << "," << f.finalCost << "," << f.msOptimize << "," << f.msMarginalize << "\n";
  }
}

}  // namespace okvis_amd
