// okvis_amd_replay <dataset folder> [trajectory.csv] [--config <okvis config.yaml>] [--time-limit] [--keyframes N] [--imu-frames N]
//                  [--iterations N] [--max-frames N] [--no-patch] [--device N]
// okvis_amd_replay <okvis config.yaml> <dataset folder> [trajectory.csv] [...]        (the argument order of okvis_app_synchronous)
//
// --config: cameras, IMU parameters, extrinsics uncertainty, numKeyframes, numImuFrames and ceres_options of the reference's
// configuration file (config/config_fpga_p2_euroc.yaml; okvis_config.hpp) instead of the ASL sensor.yaml files and the defaults;
// options given after it override single values.  --time-limit: every optimize() is bound by the file's ceres_options timeLimit /
// minIterations (the non-blocking mode of ThreadedKFVio.cpp:526-530) instead of running maxIterations (okvis_app_synchronous).
//
// --no-patch: every optimize() flattens and uploads its window (the round-3 route) instead of patching the window the solver holds.
//
// The backend-side counterpart of `okvis_app_synchronous <config> <dataset folder>` (reference
// okvis_apps/src/okvis_app_synchronous.cpp): reads the ASL folder plus the recorded tracks (replay.hpp) and runs the per-frame
// loop of ThreadedKFVio on okvis_amd::Estimator.  Prints one line per 20 frames and a summary.
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "okvis_config.hpp"
#include "replay.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <dataset folder> [trajectory.csv] [--keyframes N] [--imu-frames N] [--iterations N] [--max-frames N]\n", argv[0]);
    return -1;  // okvis_app_synchronous.cpp:208-212
  }
  okvis_amd::ReplayOptions opt;
  std::string out, config, dataset = argv[1];
  bool usePatch = true, useTimeLimit = false;
  int device = 0, first = 2;
  auto endsWith = [](const std::string& s, const char* e) { return s.size() >= std::strlen(e) && s.compare(s.size() - std::strlen(e), std::string::npos, e) == 0; };
  if (endsWith(dataset, ".yaml") || endsWith(dataset, ".yml")) {   // okvis_app_synchronous <config> <dataset folder>
    if (argc < 3) {
      std::fprintf(stderr, "usage: %s <okvis config.yaml> <dataset folder> [trajectory.csv] [...]\n", argv[0]);
      return -1;
    }
    config = dataset, dataset = argv[2], first = 3;
  }
  // first pass: the configuration file, so that the options after it override its values whatever their order
  for (int i = first; i + 1 < argc; ++i)
    if (!std::strcmp(argv[i], "--config")) config = argv[i + 1];
  try {
    okvis_amd::OkvisConfig cfg;
    if (!config.empty()) {
      cfg = okvis_amd::readOkvisConfig(config);
      opt = okvis_amd::replayOptionsFrom(cfg);
    }
    for (int i = first; i < argc; ++i) {
      auto val = [&](int& dst) {
        if (i + 1 >= argc) {
          std::fprintf(stderr, "%s needs a value\n", argv[i]);
          std::exit(-1);
        }
        dst = std::atoi(argv[++i]);
      };
      if (!std::strcmp(argv[i], "--keyframes")) val(opt.numKeyframes);
      else if (!std::strcmp(argv[i], "--imu-frames")) val(opt.numImuFrames);
      else if (!std::strcmp(argv[i], "--iterations")) val(opt.numIterations);
      else if (!std::strcmp(argv[i], "--max-frames")) val(opt.maxFrames);
      else if (!std::strcmp(argv[i], "--no-patch")) usePatch = false;
      else if (!std::strcmp(argv[i], "--time-limit")) useTimeLimit = true;
      else if (!std::strcmp(argv[i], "--config")) ++i;
      else if (!std::strcmp(argv[i], "--device")) val(device);   // (-1: book-keeping only, nothing is computed: host timings without a GPU)
      else out = argv[i];
    }
    if (useTimeLimit) {
      if (config.empty()) throw std::runtime_error("--time-limit needs --config (ceres_options timeLimit, minIterations)");
      opt.timeLimit = cfg.timeLimit;
    }
    const okvis_amd::Recording rec = config.empty() ? okvis_amd::readRecording(dataset) : okvis_amd::readRecording(dataset, cfg);
    if (!config.empty())
      std::printf("configuration %s: %zu cameras, numKeyframes %d, numImuFrames %d, iterations %d..%d, timeLimit %g s (%s)\n", config.c_str(),
                  cfg.cameras.size(), opt.numKeyframes, opt.numImuFrames, opt.minIterations, opt.numIterations, cfg.timeLimit,
                  opt.timeLimit >= 0 ? "applied" : "not applied: blocking");
    std::printf("No. IMU measurements: %zu\n", rec.imu.size());  // okvis_app_synchronous.cpp:249
    std::printf("No. frames: %zu, cameras: %zu, recorded observations: %zu, landmarks: %zu\n", rec.frames.size(), rec.cameras.size(),
                rec.observations.size(), rec.landmarks.size());
    okvis_amd::Estimator estimator(device);
    if (!usePatch) estimator.setUsePatch(false);
    const okvis_amd::ReplayResult r = okvis_amd::replay(rec, opt, estimator);
    double mo = 0, mm = 0, t4[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < r.frames.size(); ++k) {
      const okvis_amd::ReplayFrameResult& f = r.frames[k];
      mo += f.msOptimize, mm += f.msMarginalize;
      t4[0] += f.msFlatten, t4[1] += f.msUpload, t4[2] += f.msIterations, t4[3] += f.msDownload;
      if (k % 20 == 0)
        std::printf("frame %4zu  window %d frames / %d landmarks / %d new observations  cost %.4g -> %.4g (%d it)  %.2f + %.2f ms\n", k,
                    f.framesInWindow, f.landmarksInWindow, f.observations, f.initialCost, f.finalCost, f.iterations, f.msOptimize,
                    f.msMarginalize);
    }
    const double n = r.frames.empty() ? 1.0 : (double)r.frames.size();
    std::printf("Finished: %zu frames, %zu landmarks marginalised or dropped, optimize %.3f ms + marginalise %.3f ms per frame\n",
                r.frames.size(), r.landmarksRemoved, mo / n, mm / n);
    // window = describing it (the edits since the last frame as one patch; --no-patch: a full flatten), hand-over = giving it to the
    // solver (okvis_ba_patch_window; --no-patch: okvis_ba_upload)
    std::printf("window route: %s\n", usePatch ? "patch (edits of the window the solver holds)" : "flatten + upload");
    std::printf("optimize split per frame: window %.3f + hand-over %.3f + iterations %.3f + download %.3f ms\n", t4[0] / n, t4[1] / n, t4[2] / n,
                t4[3] / n);
    {   // medians: the means above carry the first frames' device allocations and page-locking
      auto median = [&](auto get) {
        std::vector<double> v;
        for (const okvis_amd::ReplayFrameResult& f : r.frames) v.push_back(get(f));
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      std::printf("medians per frame: optimize %.3f (window %.3f + hand-over %.3f + iterations %.3f + download %.3f) + marginalise %.3f ms\n",
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msOptimize; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msFlatten; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msUpload; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msIterations; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msDownload; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msMarginalize; }));
    }
    {
      auto median = [&](auto get) {
        std::vector<double> v;
        for (const okvis_amd::ReplayFrameResult& f : r.frames) v.push_back(get(f));
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      std::printf("marginalise, medians per frame: sub-window flatten %.3f + upload %.3f + okvis_ba_marginalize_begin %.3f ms (the rest: the decisions and deletions of "
                  "applyMarginalizationStrategy; the device computes meanwhile and the numbers are waited for in the next frame's window description)\n",
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msMargFlatten; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msMargUpload; }),
                  median([](const okvis_amd::ReplayFrameResult& f) { return f.msMargCall; }));
    }
    if (r.hasGroundTruth)
      std::printf("against the ground truth (first pose aligned): rms position %.4f m, final position %.4f m, final rotation %.5f rad\n",
                  r.rmsPosition, r.finalPosition, r.finalRotation);
    if (!out.empty()) okvis_amd::writeTrajectoryCsv(out, r);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
