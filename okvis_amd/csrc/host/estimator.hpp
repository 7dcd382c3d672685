// okvis_amd::Estimator — the host side of the drop-in boundary, in C++ like the reference.
//
// Mirrors the public surface of okvis::Estimator (reference okvis_ceres/include/okvis/Estimator.hpp:77-581,
// implementing okvis::VioBackendInterface, okvis_common/include/okvis/VioBackendInterface.hpp:67-336):
// same method names, argument meaning, return values and error behaviour, but with plain-old-data types
// because Eigen / OpenCV / glog are not available in this environment.  okvis_estimator_adapter.hpp wraps
// this class into a source-compatible `okvis::Estimator` where those libraries exist.
//
// The graph book-keeping the reference keeps in okvis::ceres::Map (hash maps of shared_ptr parameter and
// residual blocks, Map.hpp:348-402) is replaced by id->index maps over flat arrays; optimize() hands the
// flat window to the GPU through the C-ABI of include/okvis_amd_ba.h and copies the estimates back.  No
// numeric part of optimize() runs on the CPU.
#pragma once
#include <array>
#include <deque>
#include <cstdint>
#include <map>
#include <ostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/okvis_amd_ba.h"

namespace okvis_amd {

// ---- POD stand-ins of the OKVIS types that cross the boundary ----
struct Transformation {  // okvis::kinematics::Transformation: r, q(xyzw)
  std::array<double, 7> p{{0, 0, 0, 0, 0, 0, 1}};
};
typedef std::array<double, 9> SpeedAndBias;  // okvis::SpeedAndBias (v_W, b_g, b_a)

struct ExtrinsicsEstimationParameters {  // okvis_common Parameters.hpp
  double sigma_absolute_translation = 0, sigma_absolute_orientation = 0;
  double sigma_c_relative_translation = 0, sigma_c_relative_orientation = 0;
};
struct ImuParameters {  // okvis_common Parameters.hpp ImuParameters
  double a_max = 176, g_max = 7.8, sigma_g_c = 12e-4, sigma_a_c = 8e-3, sigma_bg = 0.03, sigma_ba = 0.1;
  double sigma_gw_c = 4e-6, sigma_aw_c = 4e-5, tau = 3600, g = 9.81007;
  std::array<double, 3> a0{{0, 0, 0}};
  int rate = 200;
};
struct ImuMeasurement {  // okvis::ImuMeasurement (Measurements.hpp)
  int64_t t_ns;
  std::array<double, 3> gyr, acc;
};
typedef std::vector<ImuMeasurement> ImuMeasurementDeque;

struct CameraGeometry {  // PinholeCamera<D> intrinsics
  std::array<double, 12> intr{};  // fu fv cu cv d0..d7
  int model = OKVIS_BA_DIST_RADTAN;
};
struct Keypoint {  // cv::KeyPoint subset (floats, implementation/Frame.hpp:210-242)
  float x, y, size;
};
struct MultiFrame {  // okvis::MultiFrame subset the backend reads
  uint64_t id = 0;
  int64_t t_ns = 0;
  std::vector<Transformation> T_SC;
  std::vector<CameraGeometry> geometry;
  std::vector<std::vector<Keypoint>> keypoints;
  size_t numFrames() const { return geometry.size(); }
};
typedef std::shared_ptr<MultiFrame> MultiFramePtr;

struct KeypointIdentifier {  // okvis::KeypointIdentifier (FrameTypedefs.hpp)
  uint64_t frameId;
  size_t cameraIndex, keypointIndex;
  bool operator<(const KeypointIdentifier& o) const {
    if (frameId != o.frameId) return frameId < o.frameId;
    if (cameraIndex != o.cameraIndex) return cameraIndex < o.cameraIndex;
    return keypointIndex < o.keypointIndex;
  }
};
struct MapPoint {  // okvis::MapPoint (FrameTypedefs.hpp)
  uint64_t id = 0;
  std::array<double, 4> point{{0, 0, 0, 1}};
  double quality = 0, distance = 0;
  std::map<KeypointIdentifier, uint64_t> observations;  // value: opaque residual handle
  // book-keeping of okvis_amd::Estimator (not part of okvis::MapPoint): the landmark's index in the window the solver holds
  // (-1 = not part of it), and whether its observations / its value changed since that window was brought up to date
  int winIdx = -1;
  bool touched = false, valueSet = false;
  int pendingAdds = 0;   // observations added since then
};
typedef std::vector<MapPoint> MapPointVector;
typedef std::map<uint64_t, MapPoint> PointMap;

class Estimator {
 public:
  typedef std::runtime_error Exception;  // OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error), Estimator.hpp:80

  // device < 0: book-keeping only (no solver is created; optimize() brings the window description up to date, checks it against a
  // freshly flattened one and computes NOTHING, applyMarginalizationStrategy() keeps the structure decisions and leaves a unit prior):
  // what the CPU tests drive the window edits with.  Every numeric result needs a device.
  explicit Estimator(int device = 0);
  ~Estimator();
  Estimator(const Estimator&) = delete;
  Estimator& operator=(const Estimator&) = delete;

  // ---- sensor configuration (Estimator.hpp:102-121) ----
  int addCamera(const ExtrinsicsEstimationParameters& p);
  int addImu(const ImuParameters& p);
  void clearCameras();
  void clearImus();

  // ---- window growth (Estimator.hpp:132-172) ----
  bool addStates(MultiFramePtr multiFrame, const ImuMeasurementDeque& imuMeasurements, bool asKeyframe);
  bool addLandmark(uint64_t landmarkId, const std::array<double, 4>& landmark);
  // returns the opaque residual handle (reference: ::ceres::ResidualBlockId); 0 for a duplicate
  uint64_t addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx);
  bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx);
  bool removeObservation(uint64_t residualHandle);

  // ---- the hot path (Estimator.hpp:183-211) ----
  void optimize(size_t numIter, size_t numThreads = 1, bool verbose = false);
  bool setOptimizationTimeLimit(double timeLimit, int minIterations);
  // Strong guarantee: when the GPU part fails (exception), the book-keeping is rolled back to the state before the call.
  bool applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, MapPointVector& removedLandmarks);
  // test hook: the next applyMarginalizationStrategy throws where the GPU call would be (exercises the roll-back)
  void debugFailNextMarginalization() { debugFailMarg_ = true; }
  // test hook: the numbers of the marginalisation that is on its way are treated as failed where they are waited for
  // (exercises the late failure path of resolvePrior: the prior is dropped, the estimator stays usable)
  void debugFailPendingMarginalization() { debugFailPending_ = true; }
  const std::string& lastRefusal() const { return lastRefusal_; }
  // diagnostics hook: called by optimize() with the flattened window it is about to upload (stage 0) and again with the
  // same window carrying the optimised pose / sb / lm arrays (stage 1); pointers are valid during the call only.  Lets a
  // test hand the very same problem to another solver and compare in the window's own indexing; not used by the product.
  typedef void (*WindowObserver)(const okvis_ba_window* window, int stage, void* user);
  void setWindowObserver(WindowObserver f, void* user) { windowObserver_ = f, windowObserverUser_ = user; }
  static bool initPoseFromImu(const ImuMeasurementDeque& imuMeasurements, Transformation& T_WS);

  // ---- getters (Estimator.hpp:218-354) ----
  bool isLandmarkAdded(uint64_t id) const { return landmarksMap_.count(id) != 0; }
  bool isLandmarkInitialized(uint64_t id) const;
  bool getLandmark(uint64_t id, MapPoint& mapPoint) const;
  size_t getLandmarks(PointMap& landmarks) const;
  size_t getLandmarks(MapPointVector& landmarks) const;
  MultiFramePtr multiFrame(uint64_t frameId) const;   // throws for an unknown id, like Estimator.cpp:1114-1120
  /// Estimator::printStates (Estimator.hpp:141, Estimator.cpp:776-809): the blocks of one state, fixed ones in parentheses
  void printStates(uint64_t poseId, std::ostream& buffer) const;
  bool hasFrame(uint64_t frameId) const { return multiFramePtrMap_.count(frameId) != 0; }   // non-throwing query
  bool get_T_WS(uint64_t poseId, Transformation& T_WS) const;
  bool getSpeedAndBias(uint64_t poseId, uint64_t imuIdx, SpeedAndBias& sb) const;
  bool getCameraSensorStates(uint64_t poseId, size_t cameraIdx, Transformation& T_SCi) const;
  size_t numFrames() const { return states_.size(); }
  size_t numLandmarks() const { return landmarksMap_.size(); }
  size_t debugObservationSlots() const { return observations_.slots(); }   // entries of memory the observation table holds (tests)
  uint64_t currentKeyframeId() const;
  uint64_t frameIdByAge(size_t age) const;
  uint64_t currentFrameId() const;
  bool isKeyframe(uint64_t frameId) const;
  bool isInImuWindow(uint64_t frameId) const;
  int64_t timestamp(uint64_t frameId) const;

  // ---- setters (Estimator.hpp:366-411) ----
  bool set_T_WS(uint64_t poseId, const Transformation& T_WS);
  bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const SpeedAndBias& sb);
  bool setCameraSensorStates(uint64_t poseId, size_t cameraIdx, const Transformation& T_SCi);
  bool setLandmark(uint64_t landmarkId, const std::array<double, 4>& landmark);
  void setLandmarkInitialized(uint64_t landmarkId, bool initialized);
  void setKeyframe(uint64_t frameId, bool isKeyframe);

  // wall-clock split of the last optimize() in ms: window description (edits since the last call, or a full flatten), hand-over
  // to the solver (okvis_ba_patch_window, or okvis_ba_upload), iterations, downloads
  const std::array<double, 4>& lastOptimizeTimings() const { return timings_; }
  // The solver keeps the window between calls (okvis_ba_set_patchable): optimize() sends it the edits since the last call
  // (addStates, addObservation, removeObservation, applyMarginalizationStrategy, the setters) as one okvis_ba_patch instead of
  // flattening and uploading everything again.  false = flatten + upload every time (the round-3 route; A/B switch).
  void setUsePatch(bool on) { usePatch_ = on; }
  // Print / cross-check diagnostics; none changes a result.  trace: non-finite inputs and refused patches are printed;
  // checkPatch: every hand-over is compared with a freshly flattened window (debugCheckWindow(), throws on a difference);
  // syncAfterHandover: optimize() waits for the enqueued copies before it starts the iterations' clock.  The constructor also
  // takes them from the ONE environment variable this class reads, OKVIS_AMD_DEBUG, a comma-separated list of
  // "trace", "check_patch", "sync_after_handover" (so that a whole test run can be cross-checked: scripts/).
  struct Diagnostics {
    bool trace = false, checkPatch = false, syncAfterHandover = false;
  };
  void setDiagnostics(const Diagnostics& d) { diag_ = d; }
  const Diagnostics& diagnostics() const { return diag_; }
  bool lastOptimizeWasPatch() const { return lastWasPatch_; }
  // diagnostics: compares the window the solver holds with a freshly flattened one (landmark by landmark, any landmark order);
  // returns an empty string when they agree.  optimize() runs it after every hand-over when Diagnostics::checkPatch is set.
  std::string debugCheckWindow();

  // last applyMarginalizationStrategy(): ms flatten / upload / okvis_ba_marginalize, Jacobi sweeps of the two
  // decompositions (0 = Cholesky fast path), reduced dimension of the marginalisation window
  const std::array<double, 6>& lastMarginalizationInfo() const { return margInfo_; }

  // ---- diagnostics of the marginalisation prior (MarginalizationError::num_residuals etc.) ----
  int priorDimension() const { return prior_.dim; }
  size_t priorNumBlocks() const { return prior_.block.size(); }

  // ---- diagnostics of the last optimize() (what ::ceres::Solver::Summary exposes via Map::summary) ----
  const okvis_ba_summary& summary() const { return summary_; }
  okvis_ba_options& options() { return options_; }

  // static ImuError::propagation (ImuError.cpp:287-504): used by addStates and, in OKVIS, by the frontend
  // (Frontend.cpp:287, ThreadedKFVio.cpp:416).  Host code like in the reference; not part of optimize().
  static int propagation(const ImuMeasurementDeque& imuMeasurements, const ImuParameters& imuParams,
                         Transformation& T_WS, SpeedAndBias& speedAndBiases, int64_t t_start, int64_t t_end);

 private:
  struct State {  // one entry of the reference's statesMap_ (Estimator.hpp:434-503)
    uint64_t id;
    int64_t t_ns;
    bool isKeyframe;
    int poseBlock;                // index into poseBlocks_
    std::vector<int> extBlocks;   // per camera: index into poseBlocks_
    int sbBlock;                  // index into sbBlocks_, -1 once marginalised
  };
  struct PoseBlock {
    std::array<double, 7> x;
    bool fixed;
    uint64_t id;
    bool alive = true;  // false once marginalised out (Map::removeParameterBlock)
  };
  struct SbBlock {
    SpeedAndBias x;
    bool fixed;
    uint64_t id;
    bool alive = true;
  };
  struct Observation {
    uint64_t handle = 0, landmarkId = 0, poseId = 0;
    size_t camIdx, keypointIdx;
    double u, v, sqrtw;
    int poseBlock, extBlock;  // T_WS and T_SCi blocks of the observing frame (cached at addObservation)
  };
  struct ImuFactor {
    uint64_t uid;  // identity of the term between windows (the vector is compacted when terms are marginalised)
    int pose0Block, sb0Block, pose1Block, sb1Block;  // indices into poseBlocks_ / sbBlocks_
    int64_t t0, t1;
    ImuMeasurementDeque meas;  // the deque is COPIED into the factor (ImuError.hpp:151-153)
    // reference bias of the factor's preintegration cache after the last optimize() (speedAndBiases_ref_)
    std::array<double, 9> sbRef{};
    bool hasRef = false;
    // ... and the preintegration itself (okvis_ba_fetch_imu_caches): like the reference's ImuError object, the term keeps it between
    // optimize() calls and hands it to every window it is part of (the optimised one and the marginalisation sub-window), so that
    // nothing is re-preintegrated unless the bias moves past the threshold (ImuError.cpp:549)
    std::vector<double> cache;
    bool hasCache = false;
  };
  struct PosePrior {
    int block;
    std::array<double, 7> meas;
    std::array<double, 36> sqrtInfo;
  };
  struct SbPrior {
    int block;
    SpeedAndBias meas;
    std::array<double, 81> sqrtInfo;
  };
  struct RelPose {
    int block0, block1;
    std::array<double, 36> sqrtInfo;
  };

  // The reference's MarginalizationError object (MarginalizationError.hpp:303-327): H_, b0_ persist between
  // calls; J_, e0_ are what optimize() evaluates; one linearisation point per connected block.
  struct MargPrior {
    int dim = 0;
    std::vector<int> type, block;               // OKVIS_BA_BLOCK_POSE / _SPEEDBIAS, index into poseBlocks_ / sbBlocks_
    std::vector<std::array<double, 9>> lin;     // linearisation point (7 or 9 entries used)
    std::vector<double> H, b0, J, e0;
  };
  // which blocks / factors go into one flat window
  struct WindowSel {
    std::vector<int> pose, sb;                  // block indices, window order
    std::vector<uint64_t> landmarks;            // ids, window order
    std::vector<uint64_t> obs;                  // observation handles
    // selectAll(): every observation of the listed landmarks is part of the window; flatten() then walks each landmark's own
    // observation map (already in (frame, camera, keypoint) order) instead of looking every observation's landmark up
    bool allObservations = false;
    std::vector<const MapPoint*> lmPtr;         // the listed landmarks (valid until the maps change)
    std::vector<int> imu, pprior, sbprior, rel; // indices into the factor vectors
    bool withPrior = false;                     // attach prior_ as marg_* (optimize) or not (marginalisation)
    bool atLinearizationPoint = false;          // values of prior-connected blocks = their linearisation point
  };
  struct FlatWindow {
    std::vector<std::vector<double>> f64;
    std::vector<std::vector<int32_t>> i32;
    std::vector<std::vector<int64_t>> i64;
    std::vector<std::vector<uint8_t>> u8;
    std::vector<int> poseMap, sbMap;            // block index -> window index (-1 = not in the window)
    std::vector<uint64_t> obsHandle;            // per observation of the window: its handle
    okvis_ba_window w;
  };
  // ---- the window the solver holds (Map::addParameterBlock / addResidualBlock / remove*, Map.cpp:292-565, as edits of it) ----
  struct SyncedObs {
    uint64_t handle;
    int poseBlock, cam;
  };
  struct SyncedWindow {
    bool valid = false;
    std::vector<int> pose, sb;                    // window index -> block
    std::vector<int> poseWin, sbWin;              // block -> window index (-1 = not in the window)
    std::vector<uint8_t> poseFixed, sbFixed;      // (a block that changes this flag makes the window start over)
    std::vector<MapPoint*> lm;                    // window index -> landmark (nodes of landmarksMap_ do not move); nullptr = erased
    std::vector<std::vector<SyncedObs>> lmObs;    // per window landmark: its observations in window order
    std::vector<uint64_t> imu;                    // window index -> ImuFactor::uid
    std::vector<double> camIntr;
    std::vector<int32_t> camModel;
    size_t nPoseBlocks = 0, nSbBlocks = 0;        // poseBlocks_.size() / sbBlocks_.size() at that time: later blocks are new
    size_t nObs = 0;
  };
  SyncedWindow synced_;
  // scratch of patchWindow(), kept between calls (no allocation in the steady state)
  struct PatchBuffers {
    std::vector<int32_t> remObs, remLm, remPose, remSb, remImu;
    std::vector<double> addPose, addSb, addLm, aoUv, aoSw, aiGyr, aiAcc, ppMeas, ppSi, spMeas, spSi, rpSi, mLin, setPose, setSb, setLm;
    std::vector<uint8_t> addPoseFixed, addSbFixed, poseFixed2, sbFixed2;
    std::vector<int32_t> aoLm, aoPose, aoExt, aoCam, aiP0, aiS0, aiP1, aiS1, aiBegin, aiCount, ppPose, spSb, rp0, rp1, mType, mIdx, mOff,
        setPoseIdx, setSbIdx, setLmIdx;
    std::vector<int64_t> aiT0, aiT1, aiSt;
    std::vector<int> pose2, sb2, poseWin2, sbWin2, lmWin2, addLmIdx;
    std::vector<int32_t> addLmBefore;
    std::vector<size_t> obsBegin;
    std::vector<char> found;
    std::vector<uint64_t> fresh;
    std::vector<MapPoint*> addedLm;
    std::vector<int> editedLm;
    std::vector<std::vector<SyncedObs>> addedLists;
    void clear() {
      for (auto* v : {&remObs, &remLm, &remPose, &remSb, &remImu, &aoLm, &aoPose, &aoExt, &aoCam, &aiP0, &aiS0, &aiP1, &aiS1, &aiBegin, &aiCount,
                      &ppPose, &spSb, &rp0, &rp1, &mType, &mIdx, &mOff, &setPoseIdx, &setSbIdx, &setLmIdx})
        v->clear();
      for (auto* v : {&addPose, &addSb, &addLm, &aoUv, &aoSw, &aiGyr, &aiAcc, &ppMeas, &ppSi, &spMeas, &spSi, &rpSi, &mLin, &setPose, &setSb, &setLm})
        v->clear();
      for (auto* v : {&addPoseFixed, &addSbFixed, &poseFixed2, &sbFixed2}) v->clear();
      for (auto* v : {&aiT0, &aiT1, &aiSt}) v->clear();
      for (auto* v : {&pose2, &sb2, &poseWin2, &sbWin2, &lmWin2, &addLmIdx, &addLmBefore}) v->clear();
      obsBegin.clear(), found.clear(), fresh.clear(), addedLm.clear(), editedLm.clear(), addedLists.clear();
    }
  };
  PatchBuffers patchBuf_;
  std::array<double, 2> patchSplit_{};            // ms: describing the window (patch or flatten), handing it over
  std::vector<double> resPose_, resSb_, resLm_, resQ_, resRef_, resCache_;   // results of the last optimize() (kept: no allocation per frame)
  std::vector<MapPoint*> touchedLm_;              // landmarks with MapPoint::touched (nullptr = erased meanwhile)
  std::vector<int> erasedWinLm_;                  // window indices of landmarks erased from the map
  // the two logs of observation edits since the last hand-over (addObservation / removeObservation write them)
  struct PendingObs {
    MapPoint* lm;                                 // nullptr = removed again (or its landmark erased) before it reached the window
    uint64_t handle;
    int poseBlock, extBlock, cam;
    double u, v, sqrtw;
  };
  struct RemovedObs {
    int lmWin;                                    // window index of its landmark
    uint64_t handle;
  };
  std::vector<PendingObs> obsAdded_;
  std::vector<RemovedObs> obsRemoved_;
  uint64_t firstPendingHandle_ = 1;               // handles from here on were handed out after the last hand-over
  void noteObservationRemoved(MapPoint& mp, uint64_t handle);
  std::vector<int> poseValueSet_, sbValueSet_;    // blocks whose value the caller has set
  int familiesChanged_ = 0;                       // OKVIS_BA_PATCH_* bits: prior families edited since the last hand-over
  bool usePatch_ = true, lastWasPatch_ = false;
  Diagnostics diag_;
  void touch(MapPoint& mp) {
    if (!mp.touched) {
      mp.touched = true;
      touchedLm_.push_back(&mp);
    }
  }
  void forgetLandmark(MapPoint& mp);              // before the landmark is erased from landmarksMap_
  void invalidateSynced();                        // the next optimize() flattens and uploads
  bool patchWindow();                             // edits since the last hand-over -> okvis_ba_patch_window; false = not possible
  void uploadWindow(FlatWindow& fw);              // flatten + okvis_ba_upload, synced_ rebuilt
  void currentWindowView(okvis_ba_window* out);   // the container the solver (or the book-keeping-only store) holds

  const State* findState(uint64_t id) const;
  State* findState(uint64_t id);
  // why the last addStates returned false (the reference logs the reason and returns false, Estimator.cpp:121-163)
  bool refuse(const std::string& why) {
    lastRefusal_ = why;
    return false;
  }
  std::string lastRefusal_;
  WindowSel selectAll() const;
  void flatten(const WindowSel& sel, FlatWindow& fw) const;

  int device_;
  okvis_ba_solver* solver_ = nullptr;
  okvis_ba_solver* margSolver_ = nullptr;          // the marginalisation sub-window has a solver of its own: solver_ keeps its window
  okvis_ba_window_store* dryStore_ = nullptr;      // book-keeping-only instance (device < 0): the container, no solver
  bool dry_ = false;
  okvis_ba_options options_;
  okvis_ba_summary summary_;
  double timeLimit_ = -1.0;
  int minIterations_ = 0;
  bool hasTimeLimit_ = false;

  std::vector<ExtrinsicsEstimationParameters> extrinsicsEstimationParametersVec_;
  std::vector<ImuParameters> imuParametersVec_;
  std::vector<State> states_;  // ordered by insertion (= by id/time like the reference's std::map)
  std::map<uint64_t, MultiFramePtr> multiFramePtrMap_;
  std::vector<PoseBlock> poseBlocks_;
  std::vector<SbBlock> sbBlocks_;
  PointMap landmarksMap_;
  std::map<uint64_t, bool> landmarkInitialized_;
  // Observations by handle.  Handles are handed out in increasing order and observations leave roughly in that order too (the
  // oldest frames go first), so the table is a deque indexed by handle - base with dead slots in between: no hashing on the paths
  // that touch every observation of a frame (addObservation, the marginalisation's removals, flatten).
  class ObsTable {
   public:
    Observation* find(uint64_t h) {
      if (h >= base_ && h - base_ < slots_.size()) {
        Observation& o = slots_[h - base_];
        return o.handle == h ? &o : nullptr;
      }
      if (!aged_.empty() && h < base_) {
        auto it = aged_.find(h);
        return it == aged_.end() ? nullptr : &it->second;
      }
      return nullptr;
    }
    const Observation* find(uint64_t h) const { return const_cast<ObsTable*>(this)->find(h); }
    const Observation& at(uint64_t h) const {
      const Observation* o = find(h);
      if (!o) throw std::out_of_range("observation handle");
      return *o;
    }
    void insert(const Observation& o) {   // (a handle below the range comes back in a roll-back: the range grows downwards)
      if (o.handle < base_ && (!aged_.empty() || (!slots_.empty() && base_ - o.handle > kAgedGap))) {
        if (aged_.emplace(o.handle, o).second) ++live_;   // (far below the dense range, or the range has shed old entries: the side map)
        else aged_[o.handle] = o;
        return;
      }
      if (slots_.empty()) base_ = o.handle;
      while (o.handle < base_) {
        slots_.push_front(Observation{});
        --base_;
      }
      while (o.handle - base_ >= slots_.size()) slots_.push_back(Observation{});
      Observation& slot = slots_[o.handle - base_];
      if (slot.handle != o.handle) {
        ++live_;
        ++dense_live_;
      }
      slot = o;
    }
    bool erase(uint64_t h) {
      if (h >= base_ && h - base_ < slots_.size()) {
        Observation& o = slots_[h - base_];
        if (o.handle != h) return false;
        o.handle = 0;   // (handles start at 1)
        --live_;
        --dense_live_;
        while (!slots_.empty() && slots_.front().handle == 0) {
          slots_.pop_front();
          ++base_;
        }
        shed();
        return true;
      }
      if (!aged_.empty() && aged_.erase(h)) {
        --live_;
        return true;
      }
      return false;
    }
    size_t size() const { return live_; }
    size_t slots() const { return slots_.size() + aged_.size(); }   // memory held, in entries (tests)

   private:
    // Handles grow with time and the dead slots are only dropped at the front: a few long-lived observations (a camera standing
    // still keeps its old keyframes, and their observations, while those of every passing frame come and go) would pin an ever
    // longer range of dead slots.  When the live entries are few in a long range, the oldest part of the range is dropped and
    // its live entries move to a side map — the range stays within a constant factor of what is alive.
    static constexpr size_t kShedMin = 4096, kAgedGap = 1u << 20;
    void shed() {
      if (slots_.size() < kShedMin || dense_live_ * 8 >= slots_.size()) return;
      while (slots_.size() > kShedMin / 4 && dense_live_ * 2 < slots_.size()) {
        Observation& f = slots_.front();
        if (f.handle != 0) {
          aged_.emplace(f.handle, f);
          --dense_live_;
        }
        slots_.pop_front();
        ++base_;
      }
      while (!slots_.empty() && slots_.front().handle == 0) {
        slots_.pop_front();
        ++base_;
      }
    }
    std::deque<Observation> slots_;
    std::unordered_map<uint64_t, Observation> aged_;
    uint64_t base_ = 1;
    size_t live_ = 0, dense_live_ = 0;
  };
  ObsTable observations_;
  std::vector<ImuFactor> imuFactors_;
  std::vector<PosePrior> posePriors_;
  std::vector<SbPrior> sbPriors_;
  std::vector<RelPose> relPoses_;
  mutable MargPrior prior_;   // (mutable: resolvePrior() fills in the numbers of a pending marginalisation)
  std::array<double, 4> timings_{};
  bool debugFailMarg_ = false;
  mutable bool debugFailPending_ = false;
  WindowObserver windowObserver_ = nullptr;
  void* windowObserverUser_ = nullptr;
  struct MargUndo;  // what applyMarginalizationStrategy changed before its GPU call (estimator.cpp)
  bool applyMarginalizationStrategyImpl(size_t numKeyframes, size_t numImuFrames, MapPointVector& removedLandmarks, MargUndo& undo);
  // last marginalisation: ms flatten, upload, marginalize (the call that enqueues it + the wait for its numbers, wherever that
  // wait took place); Jacobi sweeps (2); sub-window D
  mutable std::array<double, 6> margInfo_{};
  // A marginalisation whose numbers are still on their way (okvis_ba_marginalize_begin): prior_ already has its blocks and
  // linearisation points — all the deletions and the book-keeping need — and resolvePrior() waits for H, b0, J, e0 where they are
  // read next: the next window description / flatten, the next marginalisation, the destructor.  The device computes while
  // applyMarginalizationStrategy deletes what was marginalised and the caller adds the next frame.  A numeric failure then
  // surfaces at that later point and cannot be rolled back any more (the reference's marginalisation has no way back either).
  mutable bool priorPending_ = false;
  mutable std::vector<int32_t> margBt_, margBi_, margBo_;
  mutable std::vector<double> margH_, margB_, margJ_, margE_;
  mutable okvis_ba_marg_result margRes_{};
  void resolvePrior() const;
  uint64_t nextId_ = 1ULL << 40;    // IdProvider::instance().newId() stand-in for internal blocks
  uint64_t nextHandle_ = 1;
  uint64_t nextImuUid_ = 1;
  mutable std::mutex statesMutex_;  // guards getLandmark(s) like Estimator.cpp:936,956,965
};

}  // namespace okvis_amd
