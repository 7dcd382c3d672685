// The configuration file of the reference's applications (`okvis_app_synchronous <config.yaml> <dataset folder>`,
// reference config/config_fpga_p2_euroc.yaml), read the way okvis::VioParametersReader reads it
// (reference okvis_common/src/VioParametersReader.cpp:75-300 readConfigFile, :395-460 IMU, :500-575 getCalibrationViaConfig):
// the same keys, the same defaults where the reference has a default, the same refusal where it asserts.
//
// Only the keys that reach the backend are kept; the detection / display / publishing options of the
// frontend and of the ROS node are parsed (the file must be well formed) and ignored.
//
// The file is OpenCV-FileStorage YAML 1.0: block mappings by indentation, block sequences ("- item"), flow sequences and flow
// mappings that may span lines ("- {T_SC: [ ... ], image_dimension: [752, 480], ...}"), '#' comments, a "%YAML:1.0" header.
// YamlNode is that subset as a tree with the type tests of cv::FileNode the reference's reader relies on (isInt / isReal / isSeq
// / isMap / isString: "176.0" is a real, "176" is an int and NOT a real, exactly as in OpenCV).
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "replay.hpp"

namespace okvis_amd {

struct YamlNode {
  enum Kind { NONE, SCALAR, SEQ, MAP };
  Kind kind = NONE;
  std::string scalar;
  bool quoted = false;
  std::vector<YamlNode> seq;
  std::vector<std::pair<std::string, YamlNode>> map;
  int line = 0;

  const YamlNode& operator[](const std::string& key) const;  // NONE node when absent or not a mapping
  const YamlNode& operator[](size_t i) const;                // NONE node when out of range or not a sequence
  size_t size() const { return kind == SEQ ? seq.size() : kind == MAP ? map.size() : 0; }
  bool isNone() const { return kind == NONE; }
  bool isSeq() const { return kind == SEQ; }
  bool isMap() const { return kind == MAP; }
  bool isInt() const;
  bool isReal() const;
  bool isString() const;
  long long asInt() const;   // of an int or a real (truncated, like cv::FileNode's conversion)
  double asReal() const;     // of an int or a real
};
// throws std::runtime_error "<file>:<line>: ..." on a malformed document
YamlNode parseYaml(const std::string& text, const std::string& fileForMessages = "<text>");
YamlNode readYamlFile(const std::string& file);

struct OkvisConfig {
  // optimization (VioParametersReader.cpp:88-128)
  int numKeyframes = 5, numImuFrames = 2, minIterations = 1, maxIterations = 10;
  double timeLimit = -1.0;  // [s], negative: none
  // sensors_information (:169-202)
  double imageDelay = 0;
  int cameraRate = 0;
  double timestampTolerance = 0;
  // camera_extrinsics (:205-236)
  ExtrinsicsEstimationParameters extrinsics;
  // imu (:395-460)
  ImuParameters imu;
  double T_BS[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // nCameraSystem (:500-575, :309-390)
  std::vector<AslCamera> cameras;
};
// Errors carry the reference's wording where it asserts ("'imu_params: a_max' parameter missing in configuration file.").
OkvisConfig readOkvisConfig(const std::string& file);
OkvisConfig okvisConfigFromYaml(const YamlNode& root, const std::string& fileForMessages);

// <path>/cam<i>/data/<t_ns>.png as okvis_app_synchronous.cpp:264-318 enumerates it: the regular files of the folder, sorted by
// name, the timestamp taken from the name (seconds = all but the last 13 characters, nanoseconds = the 9 characters before
// ".png").  Throws when the folder does not exist; an empty vector is the reference's "no images at <folder>".
struct AslImage {
  int64_t t_ns;
  std::string file;  // name inside <path>/cam<i>/data
};
std::vector<AslImage> listAslImages(const std::string& path, int cam);
// <path>/cam<i>/data.csv of the ASL format: "#timestamp [ns],filename"
std::vector<AslImage> readAslImageCsv(const std::string& file);

// The recording of `path` with cameras, IMU parameters and extrinsics uncertainty taken from the configuration file instead of
// <path>/cam<i>/sensor.yaml and <path>/imu0/sensor.yaml (okvis_app_synchronous never reads those: its calibration is the
// config file's).  When <path>/cam<i>/data exists, every recorded frame must be one of its images (within timestampTolerance).
Recording readRecording(const std::string& path, const OkvisConfig& config, bool imu_as_float = true);
// numKeyframes, numImuFrames, maxIterations (-> numIterations), minIterations and timeLimit of the file
ReplayOptions replayOptionsFrom(const OkvisConfig& config);

}  // namespace okvis_amd
