// see okvis_config.hpp
#include "okvis_config.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace okvis_amd {
namespace {

[[noreturn]] void failAt(const std::string& file, int line, const std::string& what) {
  std::ostringstream o;
  o << file;
  if (line > 0) o << ":" << line;
  o << ": " << what;
  throw std::runtime_error(o.str());
}

const YamlNode kNone;

// ---- the parser: one cursor over the whole text (comments and the %YAML / --- lines blanked out beforehand, so that a flow
// collection can be read across line ends without looking at them again) ---------------------------------------------------
struct Parser {
  std::string s, file;
  size_t i = 0;

  int lineAt(size_t pos) const { return 1 + (int)std::count(s.begin(), s.begin() + std::min(pos, s.size()), '\n'); }
  [[noreturn]] void fail(const std::string& what) const { failAt(file, lineAt(i), what); }
  size_t columnOf(size_t pos) const {
    if (pos == 0) return 0;
    const size_t b = s.rfind('\n', pos - 1);
    return b == std::string::npos ? pos : pos - b - 1;
  }
  // the next character that is not white space (line ends included); false at the end of the text
  bool skipSpace() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
    return i < s.size();
  }
  void skipBlanksInLine() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r')) ++i;
  }
  bool atSequenceDash() const { return i < s.size() && s[i] == '-' && (i + 1 >= s.size() || s[i + 1] == ' ' || s[i + 1] == '\n' || s[i + 1] == '\r'); }

  static std::string trimmed(const std::string& t) {
    size_t a = 0, b = t.size();
    while (a < b && std::isspace((unsigned char)t[a])) ++a;
    while (b > a && std::isspace((unsigned char)t[b - 1])) --b;
    return t.substr(a, b - a);
  }
  YamlNode scalarNode(std::string text, int line) const {
    YamlNode n;
    n.kind = YamlNode::SCALAR;
    n.line = line;
    text = trimmed(text);
    if (text.size() >= 2 && ((text.front() == '"' && text.back() == '"') || (text.front() == '\'' && text.back() == '\''))) {
      text = text.substr(1, text.size() - 2);
      n.quoted = true;
    }
    n.scalar = text;
    return n;
  }

  // "[a, b, [c, d]]" / "{k: v, k2: [..]}" / a bare scalar that ends at one of `stops` (never consumed)
  YamlNode flow(const char* stops) {
    if (!skipSpace()) fail("the document ends inside a flow collection");
    const int line = lineAt(i);
    if (s[i] == '[') {
      ++i;
      YamlNode n;
      n.kind = YamlNode::SEQ;
      n.line = line;
      for (;;) {
        if (!skipSpace()) fail("unterminated '['");
        if (s[i] == ']') {
          ++i;
          return n;
        }
        n.seq.push_back(flow(",]"));
        if (!skipSpace()) fail("unterminated '['");
        if (s[i] == ',') ++i;
        else if (s[i] != ']') fail("',' or ']' expected in a flow sequence");
      }
    }
    if (s[i] == '{') {
      ++i;
      YamlNode n;
      n.kind = YamlNode::MAP;
      n.line = line;
      for (;;) {
        if (!skipSpace()) fail("unterminated '{'");
        if (s[i] == '}') {
          ++i;
          return n;
        }
        const size_t k0 = i;
        while (i < s.size() && s[i] != ':' && s[i] != ',' && s[i] != '}' && s[i] != '\n') ++i;
        if (i >= s.size() || s[i] != ':') fail("':' expected after a key of a flow mapping");
        const std::string key = scalarNode(s.substr(k0, i - k0), line).scalar;
        if (key.empty()) fail("empty key in a flow mapping");
        ++i;
        n.map.emplace_back(key, flow(",}"));
        if (!skipSpace()) fail("unterminated '{'");
        if (s[i] == ',') ++i;
        else if (s[i] != '}') fail("',' or '}' expected in a flow mapping");
      }
    }
    const size_t a = i;
    while (i < s.size() && !std::strchr(stops, s[i]) && s[i] != '\n') ++i;
    return scalarNode(s.substr(a, i - a), line);
  }

  // a value that starts at the cursor (somewhere inside a line): a flow collection, or a scalar up to the end of the line
  YamlNode inlineValue() {
    if (s[i] == '[' || s[i] == '{') {
      YamlNode n = flow("");
      skipBlanksInLine();
      if (i < s.size() && s[i] != '\n') fail("unexpected text after a flow collection");
      return n;
    }
    const size_t a = i;
    const int line = lineAt(i);
    while (i < s.size() && s[i] != '\n') ++i;
    return scalarNode(s.substr(a, i - a), line);
  }

  // the block node whose first character is at the cursor, in column `col`; everything of it sits in columns >= col
  YamlNode block(size_t col) {
    YamlNode n;
    n.line = lineAt(i);
    if (s[i] == '[' || s[i] == '{') return inlineValue();
    if (atSequenceDash()) {
      n.kind = YamlNode::SEQ;
      for (;;) {
        ++i;  // '-'
        skipBlanksInLine();
        if (i >= s.size() || s[i] == '\n') {  // "-" alone: the item is the block below
          if (!skipSpace() || columnOf(i) <= col) fail("empty sequence item");
          n.seq.push_back(block(columnOf(i)));
        } else {
          n.seq.push_back(block(columnOf(i)));  // "- {..}", "- scalar" or "- key: value" (a mapping in the dash's column + 2)
        }
        const size_t save = i;
        if (!skipSpace()) return n;
        const size_t c = columnOf(i);
        if (c == col && atSequenceDash()) continue;
        if (c > col) fail("bad indentation inside a sequence");
        i = save;
        return n;
      }
    }
    // a mapping ("key: ...") or a lone scalar
    {
      size_t e = i;
      while (e < s.size() && s[e] != '\n' && s[e] != ':') ++e;
      if (e >= s.size() || s[e] != ':' || !(e + 1 >= s.size() || s[e + 1] == ' ' || s[e + 1] == '\n' || s[e + 1] == '\r' || s[e + 1] == '\t')) return inlineValue();
    }
    n.kind = YamlNode::MAP;
    for (;;) {
      const size_t k0 = i;
      while (i < s.size() && s[i] != ':' && s[i] != '\n') ++i;
      if (i >= s.size() || s[i] != ':') {
        i = k0;
        fail("'key: value' expected");
      }
      const std::string key = scalarNode(s.substr(k0, i - k0), 0).scalar;
      if (key.empty()) fail("empty key");
      for (const auto& kv : n.map)
        if (kv.first == key) fail("key '" + key + "' appears twice");
      ++i;
      skipBlanksInLine();
      if (i < s.size() && s[i] != '\n') {
        n.map.emplace_back(key, inlineValue());
      } else {
        // the value is below: a deeper block, a sequence in the same column ("key:\n- a"), or nothing
        const size_t save = i;
        if (skipSpace() && (columnOf(i) > col || (columnOf(i) == col && atSequenceDash()))) {
          n.map.emplace_back(key, block(columnOf(i)));
        } else {
          i = save;
          n.map.emplace_back(key, YamlNode());
        }
      }
      const size_t save = i;
      if (!skipSpace()) return n;
      const size_t c = columnOf(i);
      if (c == col) continue;
      if (c > col) fail("bad indentation inside a mapping");
      i = save;
      return n;
    }
  }
};

bool parsesAsInt(const std::string& t, long long* v) {
  if (t.empty()) return false;
  errno = 0;
  char* end = nullptr;
  const long long x = std::strtoll(t.c_str(), &end, 10);
  if (*end != 0 || errno) return false;
  if (v) *v = x;
  return true;
}
bool parsesAsDouble(const std::string& t, double* v) {
  if (t.empty()) return false;
  // cv::FileStorage: .inf / .nan spellings aside, a number is what strtod takes completely; "0x10" and the like are strings
  for (char ch : t)
    if (!(std::isdigit((unsigned char)ch) || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E')) return false;
  char* end = nullptr;
  const double x = std::strtod(t.c_str(), &end);
  if (*end != 0 || end == t.c_str()) return false;
  if (v) *v = x;
  return true;
}

}  // namespace

const YamlNode& YamlNode::operator[](const std::string& key) const {
  if (kind == MAP)
    for (const auto& kv : map)
      if (kv.first == key) return kv.second;
  return kNone;
}
const YamlNode& YamlNode::operator[](size_t k) const { return kind == SEQ && k < seq.size() ? seq[k] : kNone; }
bool YamlNode::isInt() const { return kind == SCALAR && !quoted && parsesAsInt(scalar, nullptr); }
bool YamlNode::isReal() const { return kind == SCALAR && !quoted && !parsesAsInt(scalar, nullptr) && parsesAsDouble(scalar, nullptr); }
bool YamlNode::isString() const { return kind == SCALAR && !isInt() && !isReal(); }
long long YamlNode::asInt() const {
  long long v = 0;
  if (kind == SCALAR && parsesAsInt(scalar, &v)) return v;
  double d = 0;
  if (kind == SCALAR && parsesAsDouble(scalar, &d)) return (long long)d;
  throw std::runtime_error("YamlNode::asInt: not a number: '" + scalar + "'");
}
double YamlNode::asReal() const {
  double d = 0;
  if (kind == SCALAR && parsesAsDouble(scalar, &d)) return d;
  throw std::runtime_error("YamlNode::asReal: not a number: '" + scalar + "'");
}

YamlNode parseYaml(const std::string& text, const std::string& fileForMessages) {
  Parser p;
  p.file = fileForMessages;
  p.s = text;
  // blank out comments ('#' at the start of a line or after white space, outside quotes), the "%YAML:1.0" directive and "---"
  bool lineStart = true, inS = false, inD = false;
  for (size_t k = 0; k < p.s.size(); ++k) {
    char& ch = p.s[k];
    if (ch == '\n') {
      lineStart = true, inS = inD = false;
      continue;
    }
    if (ch == '\t') ch = ' ';
    if (lineStart && (ch == '%' || (ch == '-' && p.s.compare(k, 3, "---") == 0 && (k + 3 >= p.s.size() || p.s[k + 3] == '\n' || p.s[k + 3] == '\r')) ||
                      (ch == '.' && p.s.compare(k, 3, "...") == 0))) {
      while (k < p.s.size() && p.s[k] != '\n') p.s[k++] = ' ';
      --k;
      continue;
    }
    if (ch == '"' && !inS) inD = !inD;
    else if (ch == '\'' && !inD) inS = !inS;
    else if (ch == '#' && !inS && !inD && (lineStart || p.s[k - 1] == ' ')) {
      while (k < p.s.size() && p.s[k] != '\n') p.s[k++] = ' ';
      --k;
      continue;
    }
    if (ch != ' ' && ch != '\r') lineStart = false;
  }
  if (!p.skipSpace()) return YamlNode();
  YamlNode root = p.block(p.columnOf(p.i));
  if (p.skipSpace()) p.fail("unexpected text after the document (indentation?)");
  return root;
}

YamlNode readYamlFile(const std::string& file) {
  std::ifstream in(file);
  if (!in.good()) failAt(file, 0, "Could not open config file");  // VioParametersReader.cpp:83
  std::stringstream ss;
  ss << in.rdbuf();
  return parseYaml(ss.str(), file);
}

OkvisConfig okvisConfigFromYaml(const YamlNode& file, const std::string& name) {
  OkvisConfig c;
  auto need = [&](bool ok, const std::string& what) {
    if (!ok) failAt(name, 0, what);
  };
  // VioParametersReader.cpp:88-128: optional with defaults
  if (file["numKeyframes"].isInt()) c.numKeyframes = (int)file["numKeyframes"].asInt();
  else c.numKeyframes = 5;
  if (file["numImuFrames"].isInt()) c.numImuFrames = (int)file["numImuFrames"].asInt();
  else c.numImuFrames = 2;
  const YamlNode& co = file["ceres_options"];
  c.minIterations = co["minIterations"].isInt() ? (int)co["minIterations"].asInt() : 1;
  c.maxIterations = co["maxIterations"].isInt() ? (int)co["maxIterations"].asInt() : 10;
  c.timeLimit = co["timeLimit"].isReal() ? co["timeLimit"].asReal() : -1.0;
  // :131-167 useDriver / displayImages / detection_options: frontend and application, not kept.  The reference asserts their
  // presence; a backend replay does not need them, so their absence is no error here.
  // :169-202
  need(file["imageDelay"].isReal(), "'imageDelay' parameter missing in configuration file.");
  c.imageDelay = file["imageDelay"].asReal();
  const YamlNode& cp = file["camera_params"];
  need(cp["camera_rate"].isInt(), "'camera_params: camera_rate' parameter missing in configuration file.");
  c.cameraRate = (int)cp["camera_rate"].asInt();
  need(c.cameraRate > 0, "'camera_params: camera_rate' must be positive.");
  if (cp["timestamp_tolerance"].isReal()) {
    c.timestampTolerance = cp["timestamp_tolerance"].asReal();
    need(c.timestampTolerance < 0.5 / c.cameraRate, "Timestamp tolerance for stereo frames is larger than half the time between frames.");
    need(c.timestampTolerance >= 0.0, "Timestamp tolerance is smaller than 0");
  } else {
    c.timestampTolerance = 0.2 / c.cameraRate;
  }
  // :205-236 (0.0 when absent)
  auto sigma = [&](const char* key) { return cp[key].isReal() ? cp[key].asReal() : 0.0; };
  c.extrinsics.sigma_absolute_translation = sigma("sigma_absolute_translation");
  c.extrinsics.sigma_absolute_orientation = sigma("sigma_absolute_orientation");
  c.extrinsics.sigma_c_relative_translation = sigma("sigma_c_relative_translation");
  c.extrinsics.sigma_c_relative_orientation = sigma("sigma_c_relative_orientation");

  // cameras (:520-575).  An incomplete entry makes the reference drop the whole calibration ("Did not find any calibration!")
  const YamlNode& cams = file["cameras"];
  need(cams.isSeq() && cams.size() > 0, "Did not find any calibration!");
  for (size_t k = 0; k < cams.size(); ++k) {
    const YamlNode& cam = cams[k];
    const bool complete = cam.isMap() && cam["T_SC"].isSeq() && cam["image_dimension"].isSeq() && cam["image_dimension"].size() == 2 &&
                          cam["distortion_coefficients"].isSeq() && cam["distortion_coefficients"].size() >= 4 &&
                          cam["distortion_type"].isString() && cam["focal_length"].isSeq() && cam["focal_length"].size() == 2 &&
                          cam["principal_point"].isSeq() && cam["principal_point"].size() == 2;
    need(complete, "Found incomplete calibration in configuration file for camera " + std::to_string(k) + ". Did not find any calibration!");
    need(cam["T_SC"].size() == 16, "camera " + std::to_string(k) + ": T_SC needs 16 entries");
    AslCamera a;
    for (int e = 0; e < 16; ++e) a.T_BS[e] = cam["T_SC"][e].asReal();
    a.width = (int)cam["image_dimension"][0].asInt(), a.height = (int)cam["image_dimension"][1].asInt();
    a.geometry.intr[0] = cam["focal_length"][0].asReal(), a.geometry.intr[1] = cam["focal_length"][1].asReal();
    a.geometry.intr[2] = cam["principal_point"][0].asReal(), a.geometry.intr[3] = cam["principal_point"][1].asReal();
    const std::string type = cam["distortion_type"].scalar;
    const size_t nd = cam["distortion_coefficients"].size();
    // :320-390: four coefficients for equidistant and radialtangential, eight for radialtangential8
    size_t use = 4;
    if (type == "equidistant") a.geometry.model = OKVIS_BA_DIST_EQUIDISTANT;
    else if (type == "radialtangential" || type == "plumb_bob") a.geometry.model = OKVIS_BA_DIST_RADTAN;
    else if (type == "radialtangential8" || type == "plumb_bob8") {
      a.geometry.model = OKVIS_BA_DIST_RADTAN8, use = 8;
      need(nd >= 8, "camera " + std::to_string(k) + ": distortion_type " + type + " needs 8 distortion_coefficients");
    } else {
      need(false, "unrecognized distortion type " + type);  // :386-389 LOG(ERROR)
    }
    for (size_t e = 0; e < use; ++e) a.geometry.intr[4 + e] = cam["distortion_coefficients"][e].asReal();
    c.cameras.push_back(a);
  }

  // IMU (:398-460)
  const YamlNode& ip = file["imu_params"];
  need(ip["T_BS"].isSeq(), "'T_BS' parameter missing in the configuration file or in the wrong format.");
  need(ip["T_BS"].size() == 16, "'imu_params: T_BS' needs 16 entries");
  for (int e = 0; e < 16; ++e) c.T_BS[e] = ip["T_BS"][e].asReal();
  struct {
    const char* key;
    double* dst;
  } reals[] = {{"a_max", &c.imu.a_max},           {"g_max", &c.imu.g_max},           {"sigma_g_c", &c.imu.sigma_g_c}, {"sigma_a_c", &c.imu.sigma_a_c},
               {"sigma_bg", &c.imu.sigma_bg},     {"sigma_ba", &c.imu.sigma_ba},     {"sigma_gw_c", &c.imu.sigma_gw_c}, {"tau", &c.imu.tau},
               {"g", &c.imu.g}};
  for (const auto& r : reals) {
    need(ip[r.key].isReal(), std::string("'imu_params: ") + r.key + "' parameter missing in configuration file.");
    *r.dst = ip[r.key].asReal();
  }
  // sigma_aw_c is read without an assertion of its own (:457; the reference asserts sigma_g_c twice instead): a file without it
  // leaves the reference with an unset value, here it is an error
  need(ip["sigma_aw_c"].isReal() || ip["sigma_aw_c"].isInt(), "'imu_params: sigma_aw_c' parameter missing in configuration file.");
  c.imu.sigma_aw_c = ip["sigma_aw_c"].asReal();
  need(ip["a0"].isSeq() && ip["a0"].size() == 3, "'imu_params: a0' parameter missing in configuration file.");
  for (int e = 0; e < 3; ++e) c.imu.a0[e] = ip["a0"][e].asReal();
  need(ip["imu_rate"].isInt(), "'imu_params: imu_rate' parameter missing in configuration file.");
  c.imu.rate = (int)ip["imu_rate"].asInt();
  return c;
}

OkvisConfig readOkvisConfig(const std::string& file) { return okvisConfigFromYaml(readYamlFile(file), file); }

std::vector<AslImage> listAslImages(const std::string& path, int cam) {
  const std::string folder = path + "/cam" + std::to_string(cam) + "/data";
  DIR* d = opendir(folder.c_str());
  if (!d) failAt(folder, 0, "cannot open the image folder");
  std::vector<std::string> names;
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n == "." || n == "..") continue;
    struct stat st;
    if (stat((folder + "/" + n).c_str(), &st) == 0 && S_ISDIR(st.st_mode)) continue;  // okvis_app_synchronous.cpp:268
    names.push_back(n);
  }
  closedir(d);
  std::sort(names.begin(), names.end());  // :283
  std::vector<AslImage> out;
  for (const std::string& n : names) {
    // :314-318: seconds = name[0 .. size-13), nanoseconds = the 9 characters after that
    if (n.size() < 14) failAt(folder + "/" + n, 0, "an image name must be <seconds><9 digits of nanoseconds>.<3-letter extension>");
    const std::string sec = n.substr(0, n.size() - 13), nsec = n.substr(n.size() - 13, 9);
    long long s = 0, ns = 0;
    if (!parsesAsInt(sec, &s) || !parsesAsInt(nsec, &ns) || s < 0 || ns < 0)
      failAt(folder + "/" + n, 0, "an image name must be <seconds><9 digits of nanoseconds>.<3-letter extension>");
    out.push_back(AslImage{(int64_t)(s * 1000000000LL + ns), n});
  }
  return out;
}

std::vector<AslImage> readAslImageCsv(const std::string& file) {
  std::ifstream in(file);
  if (!in.good()) failAt(file, 0, "cannot open");
  std::vector<AslImage> out;
  std::string line;
  int n = 0;
  while (std::getline(in, line)) {
    ++n;
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    const size_t comma = line.find(',');
    if (comma == std::string::npos) failAt(file, n, "'timestamp,filename' expected");
    long long t = 0;
    if (!parsesAsInt(line.substr(0, comma), &t)) failAt(file, n, "not an integer timestamp: '" + line.substr(0, comma) + "'");
    if (!out.empty() && t <= out.back().t_ns) failAt(file, n, "image timestamps must increase");
    std::string name = line.substr(comma + 1);
    while (!name.empty() && name.front() == ' ') name.erase(name.begin());
    out.push_back(AslImage{(int64_t)t, name});
  }
  return out;
}

Recording readRecording(const std::string& path, const OkvisConfig& config, bool imu_as_float) {
  Recording rec;
  rec.imu = readAslImuCsv(path + "/imu0/data.csv", imu_as_float);
  rec.imuParameters = config.imu;
  rec.extrinsics = config.extrinsics;
  rec.cameras = config.cameras;
  {
    const std::string f = path + "/state_groundtruth_estimate0/data.csv";
    std::ifstream probe(f);
    if (probe.good()) rec.groundTruth = readAslGroundTruthCsv(f);
  }
  readRecordedTracks(path, rec);
  // the recorded frames against the images of the dataset, when they are there: a frame is the multiframe of one image per
  // camera, stamped with camera 0's time (frame synchroniser, timestamp_tolerance)
  const int64_t tol = (int64_t)std::llround(config.timestampTolerance * 1e9);
  for (size_t camIdx = 0; camIdx < rec.cameras.size(); ++camIdx) {
    const std::string folder = path + "/cam" + std::to_string(camIdx) + "/data";
    struct stat st;
    if (stat(folder.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) continue;
    const std::vector<AslImage> images = listAslImages(path, (int)camIdx);
    if (images.empty()) failAt(folder, 0, "no images at " + folder);  // okvis_app_synchronous.cpp:276
    for (const RecordedFrame& fr : rec.frames) {
      auto it = std::lower_bound(images.begin(), images.end(), fr.t_ns - tol, [](const AslImage& a, int64_t t) { return a.t_ns < t; });
      if (it == images.end() || it->t_ns > fr.t_ns + tol)
        failAt(path + "/okvis_amd_tracks/frames.csv", 0,
               "frame " + std::to_string(fr.id) + " at " + std::to_string(fr.t_ns) + " ns has no image of camera " + std::to_string(camIdx) + " within the timestamp tolerance");
    }
  }
  return rec;
}

ReplayOptions replayOptionsFrom(const OkvisConfig& config) {
  ReplayOptions o;
  o.numKeyframes = config.numKeyframes;
  o.numImuFrames = config.numImuFrames;
  o.numIterations = config.maxIterations;
  o.minIterations = config.minIterations;
  o.timeLimit = -1.0;  // blocking, as okvis_app_synchronous runs (setBlocking(true)); the caller sets config.timeLimit for the real-time budget
  return o;
}

}  // namespace okvis_amd
