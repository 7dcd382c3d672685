// Dataset side of the backend: readers for the ASL / EuRoC folder layout that okvis_app_synchronous consumes
// (reference okvis_apps/src/okvis_app_synchronous.cpp:233-379: <path>/imu0/data.csv, <path>/cam<i>/data/<t_ns>.png) and a
// replay loop that drives okvis_amd::Estimator the way ThreadedKFVio drives okvis::Estimator per frame
// (okvis_multisensor_processing/src/ThreadedKFVio.cpp:501-533, 736-765).
//
// The frontend (BRISK detection, matching, RANSAC, triangulation, keyframe selection) needs OpenCV and is out of scope
// (SURVEY.md section 8f rank 4).  What it would hand to the backend is therefore read from a RECORDING next to the ASL files:
//
//   <path>/imu0/data.csv                          ASL: #timestamp [ns],w_x,w_y,w_z [rad/s],a_x,a_y,a_z [m/s^2]
//   <path>/imu0/sensor.yaml                       ASL (optional): noise densities / random walks / rate_hz
//   <path>/cam<i>/sensor.yaml                     ASL: T_BS (4x4 row-major), intrinsics [fu,fv,cu,cv], distortion_model,
//                                                 distortion_coefficients, resolution
//   <path>/state_groundtruth_estimate0/data.csv   ASL (optional): #timestamp,p(3),q(w,x,y,z),v(3),b_w(3),b_a(3)
//   <path>/okvis_amd_tracks/frames.csv            #timestamp [ns],frame_id,is_keyframe      (Frontend::doWeNeedANewKeyframe)
//   <path>/okvis_amd_tracks/landmarks.csv         #landmark_id,timestamp [ns],x,y,z,w       (homogeneous point in the SENSOR frame S
//                                                  of the frame at that time, as the stereo triangulation of the frontend
//                                                  yields it; the replay moves it to W with the pose estimate of that frame,
//                                                  VioKeyframeWindowMatchingAlgorithm.cpp:430 `addLandmark(lmId, T_WCa_ * hP_Ca)`)
//   <path>/okvis_amd_tracks/observations.csv      #timestamp [ns],cam,u,v,size,landmark_id  (matched keypoints, any order
//                                                  inside a frame; the keypoint index is the order of appearance)
//
// Lines starting with '#' and empty lines are skipped everywhere.  Errors (missing file, malformed line, non-monotonic
// timestamps) throw std::runtime_error naming the file and line.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "estimator.hpp"

namespace okvis_amd {

struct AslGroundTruth {
  int64_t t_ns;
  double p[3], q_wxyz[4], v[3], bg[3], ba[3];
};
struct AslCamera {
  double T_BS[16];  // row-major 4x4 (sensor.yaml T_BS.data) = T_SC of the reference
  int width = 0, height = 0;
  CameraGeometry geometry;
  Transformation T_SC() const;  // r, q(xyzw)
};

// okvis_app_synchronous.cpp:337-349 parses the IMU values with std::stof: every value goes through float.  `as_float`
// reproduces that (default), false keeps the full double of the file.
std::vector<ImuMeasurement> readAslImuCsv(const std::string& file, bool as_float = true);
std::vector<AslGroundTruth> readAslGroundTruthCsv(const std::string& file);
AslCamera readAslCameraYaml(const std::string& file);
// noise parameters present in the file override `base` (config_fpga_p2_euroc.yaml values are the defaults of ImuParameters)
ImuParameters readAslImuYaml(const std::string& file, const ImuParameters& base = ImuParameters());

struct RecordedFrame {
  int64_t t_ns;
  uint64_t id;
  bool keyframe;
};
struct RecordedObservation {
  int64_t t_ns;
  int cam;
  float u, v, size;
  uint64_t landmark;
};
struct RecordedLandmark {
  uint64_t id;
  int64_t t_ns;   // the frame in whose sensor frame hp_S is expressed
  double hp_S[4];
};
struct Recording {
  std::vector<ImuMeasurement> imu;
  std::vector<AslCamera> cameras;
  ImuParameters imuParameters;
  ExtrinsicsEstimationParameters extrinsics;  // camera_params sigma_* of the configuration file (all 0: fixed extrinsics)
  std::vector<AslGroundTruth> groundTruth;  // may be empty
  std::vector<RecordedFrame> frames;        // by time
  std::vector<RecordedObservation> observations;  // sorted by frame time (stable)
  std::vector<RecordedLandmark> landmarks;
};
Recording readRecording(const std::string& path, bool imu_as_float = true);
// the okvis_amd_tracks/ part alone, into a recording whose cameras are already there (okvis_config.hpp uses it)
void readRecordedTracks(const std::string& path, Recording& rec);

struct ReplayOptions {
  int numKeyframes = 5, numImuFrames = 3;  // config_fpga_p2_euroc.yaml
  int numIterations = 10, numThreads = 2;  // max_iterations / ThreadedKFVio.cpp:736
  int minIterations = 1;                   // ceres_options minIterations, used with timeLimit
  double timeLimit = -1.0;                 // [s] per optimize(); negative: none (okvis_app_synchronous runs blocking, ThreadedKFVio.cpp:313-318)
  double imuOverlap = 0.02;                // temporal_imu_data_overlap
  int maxFrames = 0;                       // 0 = all
  int minObservationsPerLandmark = 0;      // landmarks with fewer recorded observations are never added (0 = keep all)
};
struct ReplayFrameResult {
  int64_t t_ns;
  uint64_t id;
  Transformation T_WS;
  SpeedAndBias speedAndBias;
  int observations, landmarksInWindow, framesInWindow, iterations;
  double initialCost, finalCost, msOptimize, msMarginalize;
  double msFlatten, msUpload, msIterations, msDownload;   // split of msOptimize (Estimator::lastOptimizeTimings)
  double msMargFlatten = 0, msMargUpload = 0, msMargCall = 0;   // part of msMarginalize (Estimator::lastMarginalizationInfo)
};
struct ReplayResult {
  std::vector<ReplayFrameResult> frames;
  size_t landmarksRemoved = 0;
  // against the ground truth when the recording has one: RMS position error after aligning the first pose (m), final errors
  bool hasGroundTruth = false;
  double rmsPosition = 0, finalPosition = 0, finalRotation = 0;
};
// Drives `estimator` (cameras / IMU are added here from the recording) frame by frame.
ReplayResult replay(const Recording& rec, const ReplayOptions& opt, Estimator& estimator);
// "timestamp, p_WS_W_x, p_WS_W_y, p_WS_W_z, q_WS_x, q_WS_y, q_WS_z, q_WS_w, v_WS_W_x, ..., b_a_z" as the reference's CSV
// output of the full state (okvis_app_synchronous / ThreadedKFVio::csvSaveFullStateAsCallback)
void writeTrajectoryCsv(const std::string& file, const ReplayResult& r);

}  // namespace okvis_amd
