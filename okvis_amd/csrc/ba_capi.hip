// C-ABI of the MI355X sliding-window BA backend (include/okvis_amd_ba.h): host-side structure building
// (the index arrays that replace okvis::ceres::Map's pointer graph), the HBM arena, kernel launches and
// hipGraph capture of the iteration sequence.  There is NO CPU compute path in this file: every numeric
// entry point ends in a kernel launch and fails with OKVIS_BA_ERR_NO_DEVICE without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/okvis_amd_ba.h"
#include "ba_imu.hpp"
#include "ba_chol_tiles.hpp"
#include "ba_linearize.hpp"
#include "ba_linearize2.hpp"
#include "ba_schur2.hpp"
#include "ba_marg.hpp"
#include "ba_marg_tiles.hpp"
#include "ba_schur.hpp"
#include "ba_solve.hpp"
#include "ba_store.hpp"

using namespace ba;

// One translation unit, in parts:
#include "capi_solver.inc"        // records: HostWin, Arena, okvis_ba_solver, DebugWord
#include "capi_index_build.inc"   // build_window and its helpers (the index build)
#include "capi_launch.inc"        // LDS sizes, launches of one iteration, sub-batch fork / join
// below: life cycle, options, upload / patch, state, begin / iterate / finish, queries and downloads, measurement hooks; then
// capi_standalone.inc and capi_marginalize.inc

// =====================================================================================================
extern "C" {

int okvis_ba_abi_version(void) { return OKVIS_BA_ABI_VERSION; }

void okvis_ba_get_limits(okvis_ba_limits* out) {
  if (!out) return;
  out->max_obs_per_lm = GROUP_OBS;
  out->max_reduced_dim = MAX_D;
  out->max_marg_dim = MAX_MARG_DIM;
  out->max_imu_samples_per_factor = MAX_IMU_SAMPLES;
}

void okvis_ba_default_options(okvis_ba_options* o) {
  if (!o) return;
  // Ceres 1.9 defaults restated from its documentation (not in the reference tree)
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->use_graph = 1;
  o->schur_lm_per_block = 0;
  o->debug_arrays = 0;
  o->gauss_newton = 0;
  o->n_streams = 0;
  o->fp32_linearize = 0;
  o->strategy = OKVIS_BA_STRATEGY_DOGLEG;   // what the reference configures (Estimator.cpp:858)
  o->jacobi_scaling = 1;
  o->max_consecutive_invalid_steps = 5;
  o->reserved0 = 0;
  std::memset(&o->tuning, 0, sizeof(o->tuning));
}

const char* okvis_ba_error_string(int status) {
  switch (status) {
    case OKVIS_BA_OK: return "ok";
    case OKVIS_BA_ERR_ARG: return "invalid argument (null pointer, index out of range, unsorted/duplicate observations)";
    case OKVIS_BA_ERR_STATE: return "call order violated";
    case OKVIS_BA_ERR_UNSUPPORTED: return "structure exceeds a documented limit (okvis_ba_get_limits)";
    case OKVIS_BA_ERR_NO_DEVICE: return "no HIP device visible: this backend has no CPU path";
    case OKVIS_BA_ERR_NUMERIC: return "numeric failure";
  }
  if (status >= OKVIS_BA_HIP_ERROR_BASE) return hipGetErrorString((hipError_t)(status - OKVIS_BA_HIP_ERROR_BASE));
  return "unknown status";
}

int okvis_ba_create(okvis_ba_solver** out, int device) {
  if (!out) return OKVIS_BA_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OKVIS_BA_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return OKVIS_BA_ERR_ARG;
  okvis_ba_solver* s = new okvis_ba_solver();
  s->device = device;
  okvis_ba_default_options(&s->opt);
  s->stagger_ticks = 10 * 100;   // start offset of the sub-batch streams (us; okvis_ba_tuning::stagger_us; round 2: 20 / 45 / 70 us all lock the fast interleaving; round 6: 5 and 10 us do too)
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&s->ev0);
  if (e == hipSuccess) e = hipEventCreate(&s->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
  // the option record and the window records share one allocation — [OptD, padded | WinPtrs x capacity] — so that an upload
  // refreshes both with ONE copy
  if (e == hipSuccess) e = hipMalloc(&s->d_opt, records_bytes(1));
  if (e == hipSuccess) {
    s->d_wins = reinterpret_cast<WinPtrs*>(reinterpret_cast<unsigned char*>(s->d_opt) + OPT_PAD);
    s->d_ctrl = reinterpret_cast<CtrlSlot*>(reinterpret_cast<unsigned char*>(s->d_opt) + ctrl_off(1));
    s->wins_capacity = 1;
  }
  // kernels may use more than the default 64 KB of dynamic LDS
  auto lds = [&](const void* f, size_t bytes) {
    if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  };
  lds(reinterpret_cast<const void*>(&linearize_kernel<true, double, false>), std::max(lin_smem(true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<true, double, true>), std::max(lin_smem(true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<true, float, false>), std::max(lin_smem(true, true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<true, float, true>), std::max(lin_smem(true, true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<false, float, false>), std::max(lin_smem(false, true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<false, float, true>), std::max(lin_smem(false, true), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<false, double, false>), std::max(lin_smem(false), small_smem()));
  lds(reinterpret_cast<const void*>(&linearize_kernel<false, double, true>), std::max(lin_smem(false), small_smem()));
  {
    const size_t l2d = std::max(lin2_smem(MAX_D, true, false), small_smem()), l2f = std::max(lin2_smem(MAX_D, true, true), small_smem());
    lds(reinterpret_cast<const void*>(&linearize2_kernel<double, false, false, 3>), l2d);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<double, false, false, 4, 14>), l2d);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<double, false, true>), l2d);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<double, true, false>), l2d);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<double, true, true>), l2d);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<float, false, false, 3>), l2f);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<float, false, false, 4, 14>), l2f);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<float, false, true>), l2f);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<float, true, false>), l2f);
    lds(reinterpret_cast<const void*>(&linearize2_kernel<float, true, true>), l2f);
    lds(reinterpret_cast<const void*>(&small_kernel), small_smem());
    lds(reinterpret_cast<const void*>(&small_prepare_kernel), small_smem());
  }
  lds(reinterpret_cast<const void*>(&schur_ride_kernel<3>), std::max((size_t)sch2_tile_doubles(TILE_DIM, sch2_nlb(TILE_DIM, 9216)) * sizeof(double), small_eval_smem()));
  lds(reinterpret_cast<const void*>(&schur_mfma_kernel<3>), (size_t)sch2_tile_doubles(TILE_DIM, sch2_nlb(TILE_DIM, 9216)) * sizeof(double));
  lds(reinterpret_cast<const void*>(&schur_mfma_kernel<9>), (size_t)sch2_tile_doubles(TILE_DIM, sch2_nlb(TILE_DIM, 9216)) * sizeof(double));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&schur_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(2 * SCHUR_LM_BATCH * TILE_DIM * 3 * sizeof(double)));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            std::max((int)solve_smem(((MAX_D_LDS + 5) / 6) * 6, false), SOLVE_LDS_LIMIT));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            std::max((int)solve_smem(((MAX_D_LDS + 5) / 6) * 6, false), SOLVE_LDS_LIMIT));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_LDS_LIMIT_CHAIN);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_LDS_LIMIT_CHAIN);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)solve_smem(((MAX_D + 5) / 6) * 6, true));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_tiles_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            CT_SMEM_DOUBLES * 8);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&marg_dense_kernel<MAX_D_LDS, MARG_SMALL_PRIOR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, MARG_LDS_DOUBLES * 8);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&marg_dense_kernel<MAX_D, MAX_MARG_DIM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, MARG_LDS_DOUBLES_LARGE * 8);

  if (e != hipSuccess) {
    int code = OKVIS_BA_HIP_ERROR_BASE + (int)e;
    okvis_ba_destroy(s);
    return code;
  }
  *out = s;
  return OKVIS_BA_OK;
}

int okvis_ba_destroy(okvis_ba_solver* s) {
  if (!s) return OKVIS_BA_OK;
  (void)hipSetDevice(s->device);
  destroy_graphs(s);
  if (s->d_arena) (void)hipFree(s->d_arena);
  if (s->d_pre) (void)hipFree(s->d_pre);
  if (s->ev_marg_vals) (void)hipEventDestroy(s->ev_marg_vals);
  if (s->d_opt) (void)hipFree(s->d_opt);   // (d_wins lives in the same allocation)
  if (s->h_ctrl_stage) (void)hipHostFree(s->h_ctrl_stage);
  if (s->d_ctrl_stage) (void)hipFree(s->d_ctrl_stage);
  if (s->marg_scratch) (void)hipFree(s->marg_scratch);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  for (auto st : s->sub_streams)
    if (st != s->stream) (void)hipStreamDestroy(st);
  for (auto ev : s->sub_events) (void)hipEventDestroy(ev);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return OKVIS_BA_OK;
}

int okvis_ba_set_options(okvis_ba_solver* s, const okvis_ba_options* opt) {
  if (!s || !opt) return OKVIS_BA_ERR_ARG;
  if (!(opt->initial_radius > 0) || !(opt->min_lm_diagonal > 0) || !(opt->max_lm_diagonal >= opt->min_lm_diagonal))
    return OKVIS_BA_ERR_ARG;
  if (s->uploaded && (opt->debug_arrays != s->opt.debug_arrays || opt->schur_lm_per_block != s->opt.schur_lm_per_block ||
                      opt->n_streams != s->opt.n_streams))
    return OKVIS_BA_ERR_STATE;  // these shape the arena: set them before upload
  if (s->uploaded) {   // ... and so does the tuning record, but for the switches that are read per call and the stream stagger
    const uint32_t per_call = OKVIS_BA_TUNE_NO_MARG_TILES | OKVIS_BA_TUNE_NO_EARLY_PREINTEGRATION | OKVIS_BA_TUNE_H0_ON_HOST;
    okvis_ba_tuning a = opt->tuning, b = s->opt.tuning;
    a.flags &= ~per_call; b.flags &= ~per_call;
    a.stagger_us = b.stagger_us = 0;
    if (std::memcmp(&a, &b, sizeof(a)) != 0) return OKVIS_BA_ERR_STATE;
  }
  // unchanged options (the host class sets them before every upload): nothing to do - every upload writes the device copy
  if (std::memcmp(opt, &s->opt, sizeof(*opt)) == 0) return OKVIS_BA_OK;
  // captured graphs name the kernels and the launch sequence of the options they were captured under: another linearise
  // kernel (fp32) or another trust-region strategy (the DOGLEG graphs carry the iteration-budget kernel) invalidates them
  if (opt->fp32_linearize != s->opt.fp32_linearize || opt->strategy != s->opt.strategy || opt->gauss_newton != s->opt.gauss_newton)
    destroy_graphs(s);
  s->opt = *opt;
  s->stagger_ticks = (long long)(opt->tuning.stagger_us > 0 ? opt->tuning.stagger_us : opt->tuning.stagger_us < 0 ? 0 : 10) * 100;   // (round 6: 5 / 10 us 479 k, 20 us 476 k, 30 us 474 k, none 465 k at 64 windows)
  HIP_TRY(hipSetDevice(s->device));
  OptD d = make_optd(s->opt, (int)s->wins.size());
  HIP_TRY(hipMemcpyAsync(s->d_opt, &d, sizeof(d), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return OKVIS_BA_OK;
}

static int upload_impl(okvis_ba_solver* s, int n_windows, const okvis_ba_window* windows);

int okvis_ba_upload(okvis_ba_solver* s, int n_windows, const okvis_ba_window* windows) {
  if (!s || n_windows <= 0 || !windows) return OKVIS_BA_ERR_ARG;
  const int rc = upload_impl(s, n_windows, windows);
  s->mirror_fresh = false;
  if (rc != OKVIS_BA_OK || !s->patchable) {
    s->mirrors.clear();
    return rc;
  }
  const auto t_c0 = std::chrono::steady_clock::now();
  struct Note {
    std::chrono::steady_clock::time_point t0;
    ~Note() {
      if (g_build_times.on) g_build_times.ms["upload: copy into the container"] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  } note{t_c0};
  try {   // a patchable solver keeps what it was given (ba_store.hpp)
    s->mirrors.resize((size_t)n_windows);
    for (int i = 0; i < n_windows; ++i)
      if (int rs = s->mirrors[i].assign(windows[i])) {
        s->mirrors.clear();
        return rs;
      }
  } catch (const std::bad_alloc&) {
    s->mirrors.clear();
    return OKVIS_BA_ERR_ARG;
  }
  s->mirror_fresh = true;
  return OKVIS_BA_OK;
}

// IMU terms that arrive without a preintegration (flag 0) get theirs started now, at the bias their first evaluation will see (the
// uploaded value of their first speed/bias block), so that it runs while the host builds the index lists (imu_pre_kernel,
// ba_linearize2.hpp).  A handful of terms only — the new term of a sliding window; a batch of fresh windows re-preintegrates inside
// its first linearise launch as before (hundreds of terms fill the device either way).  Returns the number started; where[k] =
// (window, term) and the device records wait at *src for imu_pre_place_kernel.  OKVIS_BA_TUNE_NO_EARLY_PREINTEGRATION switches it off (A/B).
constexpr int PRE_MAX_TERMS = 8;
static int pre_launch(okvis_ba_solver* s, int n_windows, const okvis_ba_window* windows, const int2** where_dev, const ImuCacheD** src_dev) {
  if (s->opt.tuning.flags & OKVIS_BA_TUNE_NO_EARLY_PREINTEGRATION) return 0;
  struct Item {
    int w, f;
  };
  Item items[PRE_MAX_TERMS];
  int K = 0;
  size_t samples = 0;
  for (int i = 0; i < n_windows; ++i) {
    const okvis_ba_window& w = windows[i];
    if (w.n_imu <= 0) continue;
    if (!w.imu_pose0 || !w.imu_sb0 || !w.imu_t0 || !w.imu_t1 || !w.imu_s_begin || !w.imu_s_count || !w.imu_s_t || !w.imu_s_gyr || !w.imu_s_acc || !w.sb)
      return 0;   // (the index build reports it)
    for (int f = 0; f < w.n_imu; ++f) {
      if (w.imu_sb_ref && w.imu_sb_ref_valid && w.imu_sb_ref_valid[f]) continue;
      const int64_t b = w.imu_s_begin[f], c = w.imu_s_count[f];
      if (w.imu_sb0[f] < 0 || w.imu_sb0[f] >= w.n_sb || b < 0 || c < 2 || c > MAX_IMU_SAMPLES || b + c > (int64_t)w.n_imu_samples) return 0;
      if (!(w.imu_s_t[b + c - 1] >= w.imu_t1[f])) return 0;   // (ImuError::redoPreintegration's -1: the upload refuses the window)
      if (K == PRE_MAX_TERMS) return 0;
      items[K++] = Item{i, f};
      samples += (size_t)c;
    }
  }
  if (K == 0) return 0;
  // the staged block: stand-in window records | biases | (window, term) | t0 | t1 | sample begin (0) | sample count | samples | records
  auto up8 = [](size_t x) { return (x + 7) & ~size_t(7); };
  const size_t o_mini = 0, o_sb = o_mini + sizeof(WinPtrs) * K, o_where = o_sb + 72 * (size_t)K, o_t0 = o_where + sizeof(int2) * K,
               o_t1 = o_t0 + 8 * (size_t)K, o_beg = o_t1 + 8 * (size_t)K, o_cnt = up8(o_beg + 4 * (size_t)K), o_st = up8(o_cnt + 4 * (size_t)K),
               o_gyr = o_st + 8 * samples, o_acc = o_gyr + 24 * samples, total = o_acc + 24 * samples;
  // The device reads the block where the host writes it: page-locked host memory is mapped into the device's address space, the
  // few KB cross PCIe once, coalesced, and the host saves the copy call.  Only the records live in device memory (imu_redo writes
  // every field but the counter of re-preintegrations, which it counts up: imu_pre_place_kernel sets that to 1).
  if (!s->d_pre) {
    if (hipMalloc(&s->d_pre, sizeof(ImuCacheD) * PRE_MAX_TERMS) != hipSuccess) return 0;
  }
  if (s->stage_pre.size() < total) s->stage_pre.resize(total + total / 2);
  if (!stage_is_pinned(s->stage_pre)) return 0;   // (pageable memory: the device cannot read it in place)
  unsigned char* h = s->stage_pre.data();
  unsigned char* d = h;                // (inputs: the staging block itself)
  unsigned char* rec = s->d_pre;       // (outputs)
  std::memset(h, 0, o_st);             // (the fixed-size head; the samples are written in full below)
  size_t at = 0;   // samples placed so far
  for (int k = 0; k < K; ++k) {
    const okvis_ba_window& w = windows[items[k].w];
    const int f = items[k].f, b = w.imu_s_begin[f], c = w.imu_s_count[f];
    WinPtrs P;
    std::memset(&P, 0, sizeof(P));
    P.n_imu = 1;
    P.imu.sigma_g_c = w.imu_params.sigma_g_c; P.imu.sigma_a_c = w.imu_params.sigma_a_c;
    P.imu.sigma_gw_c = w.imu_params.sigma_gw_c; P.imu.sigma_aw_c = w.imu_params.sigma_aw_c;
    P.imu.g = w.imu_params.g; P.imu.g_max = w.imu_params.g_max; P.imu.a_max = w.imu_params.a_max;
    OFF(imu_t0, (size_t)(uintptr_t)(d + o_t0 + 8 * (size_t)k));
    OFF(imu_t1, (size_t)(uintptr_t)(d + o_t1 + 8 * (size_t)k));
    OFF(imu_s_begin, (size_t)(uintptr_t)(d + o_beg + 4 * (size_t)k));
    OFF(imu_s_count, (size_t)(uintptr_t)(d + o_cnt + 4 * (size_t)k));
    OFF(imu_s_t, (size_t)(uintptr_t)(d + o_st + 8 * at));
    OFF(imu_s_gyr, (size_t)(uintptr_t)(d + o_gyr + 24 * at));
    OFF(imu_s_acc, (size_t)(uintptr_t)(d + o_acc + 24 * at));
    OFF(imu_cache, (size_t)(uintptr_t)(rec + sizeof(ImuCacheD) * (size_t)k));
    std::memcpy(h + o_mini + sizeof(WinPtrs) * (size_t)k, &P, sizeof(P));
    std::memcpy(h + o_sb + 72 * (size_t)k, w.sb + 9 * (size_t)w.imu_sb0[f], 72);
    const int2 wf = make_int2(items[k].w, f);
    std::memcpy(h + o_where + sizeof(int2) * (size_t)k, &wf, sizeof(wf));
    const long long t0 = w.imu_t0[f], t1 = w.imu_t1[f];
    std::memcpy(h + o_t0 + 8 * (size_t)k, &t0, 8);
    std::memcpy(h + o_t1 + 8 * (size_t)k, &t1, 8);
    const int32_t cnt = c;
    std::memcpy(h + o_cnt + 4 * (size_t)k, &cnt, 4);
    for (int j = 0; j < c; ++j) {
      const long long t = w.imu_s_t[b + j];
      std::memcpy(h + o_st + 8 * (at + j), &t, 8);
    }
    std::memcpy(h + o_gyr + 24 * at, w.imu_s_gyr + 3 * (size_t)b, 24 * (size_t)c);
    std::memcpy(h + o_acc + 24 * at, w.imu_s_acc + 3 * (size_t)b, 24 * (size_t)c);
    at += (size_t)c;
  }
  // On the solver's own stream (idle here: upload_impl has waited for it, so the block of the previous call is no longer read).
  // A stream of their own with an event in front of imu_pre_place_kernel was measured too: the arena copy then no longer
  // queues up behind the recursion, but the cross-stream wait costs what that buys (replay optimize() 0.91 against 0.89 ms).
  hipLaunchKernelGGL(imu_pre_kernel, dim3((unsigned)K), dim3(IMU_THREADS), small_smem(), s->stream, reinterpret_cast<const WinPtrs*>(d + o_mini),
                     reinterpret_cast<const double*>(d + o_sb));
  if (hipGetLastError() != hipSuccess) return 0;
  *where_dev = reinterpret_cast<const int2*>(d + o_where);
  *src_dev = reinterpret_cast<const ImuCacheD*>(rec);
  return K;
}

static int upload_impl(okvis_ba_solver* s, int n_windows, const okvis_ba_window* windows) {
  if (s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first)
  const auto t_enter = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const auto t_synced = std::chrono::steady_clock::now();
  s->uploaded = false;
  s->begun = false;
  Arena A;
  A.host.swap(s->stage);   // page-locked, kept (with its size) from the previous upload (the stream is idle: see the sync above)
  struct GiveBack {
    StageVec& a;
    StageVec& b;
    ~GiveBack() { a.swap(b); }
  } give_back{A.host, s->stage};
  std::vector<HostWin> wins(n_windows);
  const int2* pre_where = nullptr;
  const ImuCacheD* pre_src = nullptr;
  const int n_pre = pre_launch(s, n_windows, windows, &pre_where, &pre_src);   // (runs on the device while the lists are built)
  const bool dbg_t = debug_word().upload;
  const auto t_u0 = std::chrono::steady_clock::now();
  // the piece path of the linearise launch (ba_linearize2.hpp) unless a window of the batch does not fit it (free extrinsics,
  // a landmark with more than LIN2_PIECES pieces) or options.reserved0 bit 3 asks for the staged kernel
  bool lin2 = !(s->opt.reserved0 & 8);
  // the reduced solve of the LDS-resident windows: the chain solver (ba_chain.hpp) when the options allow it and EVERY such
  // window of the batch fits it, else the dense LDL^T for all of them (one kernel instantiation per launch)
  int chain_min = 1;
  bool chain = want_chain(s->opt, &chain_min);
  for (int i = 0; i < n_windows; ++i) {
    int rc = build_window(windows[i], s->opt, A, wins[i], n_windows, lin2, chain);
    if (rc == BW_LIN2_UNFIT || rc == BW_CHAIN_UNFIT) {   // start over with the staged kernel's lists / the dense solver's layout for every window
      if (rc == BW_LIN2_UNFIT) lin2 = false;
      else chain = false;
      A.size = 0;
      A.zsize = 0;
      i = -1;
      continue;
    }
    if (rc != OKVIS_BA_OK) return rc;
  }
  s->lin2 = lin2;
  {
    // IMU / prior factors in a launch of their own when the batch fills the device (then four linearise workgroups share a
    // CU); one launch for everything when a few windows wait for one another's latency
    const int split_min = s->opt.tuning.split_small_min > 0 ? s->opt.tuning.split_small_min : SMALL_BATCH_WINDOWS;
    s->split_small = lin2 && n_windows >= split_min;
  }
  const auto t_u1 = std::chrono::steady_clock::now();
  // grow-only device allocations: the per-frame re-upload of okvis_amd::Estimator must not pay hipFree/hipMalloc
  if (A.host.size() < A.size) A.host.resize(A.size, 0);
  if (A.total() > s->arena_capacity) {
    if (s->d_arena) HIP_TRY(hipFree(s->d_arena));
    s->d_arena = nullptr;
    s->arena_capacity = 0;
    const size_t cap = A.total() + A.total() / 4;
    HIP_TRY(hipMalloc(&s->d_arena, cap));
    s->arena_capacity = cap;
  }
  s->arena_bytes = A.total();
  unsigned char* zbase = s->d_arena + A.data_bytes();
  if (A.zsize) HIP_TRY(hipMemsetAsync(zbase, 0, A.zsize, s->stream));   // overlaps with the copy below
  HIP_TRY(hipMemcpyAsync(s->d_arena, A.host.data(), A.size, hipMemcpyHostToDevice, s->stream));
  if ((size_t)n_windows > s->wins_capacity) {   // [OptD, padded | WinPtrs x n | CtrlSlot x n], see okvis_ba_create
    if (s->d_opt) HIP_TRY(hipFree(s->d_opt));
    s->d_opt = nullptr;
    s->d_wins = nullptr;
    s->d_ctrl = nullptr;
    s->wins_capacity = 0;
    HIP_TRY(hipMalloc(&s->d_opt, records_bytes((size_t)n_windows)));
    s->d_wins = reinterpret_cast<WinPtrs*>(reinterpret_cast<unsigned char*>(s->d_opt) + OPT_PAD);
    s->wins_capacity = (size_t)n_windows;
  }
  // (the control records follow the records of the windows that are there: n_windows of them, not the capacity)
  s->d_ctrl = reinterpret_cast<CtrlSlot*>(reinterpret_cast<unsigned char*>(s->d_opt) + ctrl_off((size_t)n_windows));
  std::vector<WinPtrs> ptrs(n_windows);
  s->max_group = s->max_imu = s->max_schur_blocks = s->max_lm = s->max_Dpad = s->max_Dp = s->max_spart_stride = 0;
  s->max_Dpad_small = s->max_Dpad_large = 0;
  s->max_chain_doubles = 0;
  s->chain = false;
  s->any_ext = false;
  s->group_chunks = true;
  s->spec_schur = true;
  s->fp32_at_upload = s->opt.fp32_linearize != 0;
  for (int i = 0; i < n_windows; ++i) {
    relocate(wins[i].ptrs, s->d_arena, zbase, s->opt.debug_arrays);
    wins[i].ptrs.ctrl = (decltype(wins[i].ptrs.ctrl))(&s->d_ctrl[i].c);   // (not the arena's slot: see CtrlSlot)
    ptrs[i] = wins[i].ptrs;
    const WinPtrs& P = ptrs[i];
    s->max_group = std::max(s->max_group, P.n_group);
    s->max_imu = std::max(s->max_imu, P.n_imu);
    s->max_schur_blocks = std::max(s->max_schur_blocks, P.n_chunk * (P.n_tile * (P.n_tile + 1) / 2));
    s->max_spart_stride = std::max(s->max_spart_stride, P.spart_stride);
    s->max_lm = std::max(s->max_lm, P.n_lm);
    s->max_Dpad = std::max(s->max_Dpad, ((P.D + 5) / 6) * 6);
    s->max_Dp = std::max(s->max_Dp, P.Dp);
    if (P.D <= MAX_D_LDS) {
      s->max_Dpad_small = std::max(s->max_Dpad_small, ((P.D + 5) / 6) * 6);
      if (P.chain) {
        s->chain = true;
        s->max_chain_doubles = std::max(s->max_chain_doubles, LChain::make(P.D, P.Dp).total);
      }
    } else
      s->max_Dpad_large = std::max(s->max_Dpad_large, ((P.D + 5) / 6) * 6);
    s->any_ext = s->any_ext || P.has_ext;
    s->group_chunks = s->group_chunks && wins[i].group_chunks;
    s->spec_schur = s->spec_schur && wins[i].spec_ok;
  }
  // a batch is fused as a whole or not at all; a batch that is not takes the decision-free Schur launch as a whole or not at all;
  // otherwise one set of partials for everybody
  if (s->group_chunks) s->spec_schur = false;
  if (!s->group_chunks)
    for (int i = 0; i < n_windows; ++i) ptrs[i].fuse_fast = 0;
  if (!s->group_chunks && !s->spec_schur)
    for (int i = 0; i < n_windows; ++i) ptrs[i].spart_buf_stride = 0;
  {
    // one copy: option record, window records and the (zeroed) control records behind them
    const size_t wb = sizeof(WinPtrs) * (size_t)n_windows, all = records_bytes((size_t)n_windows);
    s->stage_small.resize(all);
    std::memset(s->stage_small.data(), 0, all);
    const OptD d = make_optd(s->opt, n_windows);
    std::memcpy(s->stage_small.data(), &d, sizeof(d));
    std::memcpy(s->stage_small.data() + OPT_PAD, ptrs.data(), wb);
    HIP_TRY(hipMemcpyAsync(s->d_opt, s->stage_small.data(), all, hipMemcpyHostToDevice, s->stream));
  }
  if (n_pre > 0) {   // the records started before the index build take their places in the window
    hipLaunchKernelGGL(imu_pre_place_kernel, dim3((unsigned)n_pre), dim3(64), 0, s->stream, s->d_wins, pre_where, pre_src);
    HIP_TRY(hipGetLastError());
  }
  for (int i = 0; i < n_windows; ++i)
    if (wins[i].h0_on_device) {
      const int Dm = wins[i].marg_dim;
      hipLaunchKernelGGL(marg_h0_kernel, dim3((unsigned)(((size_t)Dm * Dm + 255) / 256), 1), dim3(256), 0, s->stream, s->d_wins, i);
      HIP_TRY(hipGetLastError());
    }
  s->wins.swap(wins);
  // ---- sub-batches: opt.n_streams (0 = auto).  Measured at 64 windows (scripts/sweep_streams.sh, r02): 1 stream 272 k,
  //      2: 300 k, 3: 325 k, 4: 198 k window-iterations/s — the streams of the process that have work, or ever had, must not
  //      exceed four (whatever GPU_MAX_HW_QUEUES and the stream priorities say: scripts/r06_streams.sh, r06_streams2.sh,
  //      tools/micro/stream_concurrency.hip) ----
  {
    // measured on MI355X / ROCm 7.2 (profiles/r01_notes.md): branches inside ONE captured graph are not
    // overlapped, but two independently replayed graphs on two streams are (+29 % at 64 windows); more
    // than two streams lose again
    // (round 6, profiles/r06_notes.md: from 128 windows on two streams are ahead again — 128: 613 k against 601 k, 256: 692 k
    //  against 659 k, 512: 727 k against 700 k window-iterations/s; 96 windows: three, 568 k against 549 k)
    int nsub = s->opt.n_streams > 0 ? s->opt.n_streams : (n_windows >= 128 ? 2 : (n_windows >= 56 ? 3 : (n_windows >= 8 ? 2 : 1)));   // (48 windows: 2 is better)
    nsub = std::max(1, std::min(nsub, n_windows));
    s->sub_begin.assign(nsub + 1, 0);
    for (int k = 0; k <= nsub; ++k) s->sub_begin[k] = (int)((int64_t)n_windows * k / nsub);
    const bool same = (nsub > 1 ? (int)s->sub_streams.size() == nsub : s->sub_streams.empty());
    if (!same) {
      for (auto st : s->sub_streams)
    if (st != s->stream) (void)hipStreamDestroy(st);
      for (auto ev : s->sub_events) (void)hipEventDestroy(ev);
      s->sub_streams.clear();
      s->sub_events.clear();
    }
    if (nsub > 1 && !same) {
      for (int k = 0; k < nsub; ++k) {
        hipStream_t st;
        hipEvent_t ev;
        // the first sub-batch runs on the solver's own stream (round 6): an idle stream still holds one of the process's hardware
        // queues, and the runtime runs four of them side by side, not more (profiles/r06_notes.md: 64 windows on 3 + 1 idle streams
        // 479 k, on the solver's stream + 2: 486 k window-iterations/s; a fifth active stream halves the rate)
        if (k == 0) st = s->stream;
        else HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s->sub_streams.push_back(st);
        s->sub_events.push_back(ev);
      }
    }
  }
  {
    // Captured graphs hold grid sizes, LDS sizes, kernel choices and the device addresses of the window / option records —
    // not the windows themselves.  A re-upload that leaves all of that unchanged (the same number of equally shaped windows:
    // the per-frame pattern of a batch service, the dogleg record of bench.py) keeps them; anything else drops them.
    std::vector<int64_t> sig = {n_windows, s->max_group, s->max_imu, s->max_schur_blocks, s->max_lm, s->max_Dpad, s->max_Dp,
                                s->max_Dpad_small, s->max_Dpad_large, s->max_spart_stride, s->any_ext, s->group_chunks, s->spec_schur, s->lin2, s->split_small, s->chain, s->max_chain_doubles,
                                s->fp32_at_upload, (int64_t)(intptr_t)s->d_wins, (int64_t)(intptr_t)s->d_opt,
                                (int64_t)s->sub_streams.size()};
    for (int b : s->sub_begin) sig.push_back(b);
    for (auto st : s->sub_streams) sig.push_back((int64_t)(intptr_t)st);
    if (sig != s->launch_sig) {
      destroy_graphs(s);
      s->launch_sig.swap(sig);
    }
  }
  s->uploaded = true;
  s->evaluated = false;
  s->res_staged = false;
  s->acc_fresh = true;   // Ctrl starts zeroed: accepted buffer 0, like HostWin::acc
  if (g_build_times.on) {
    const auto t_u2 = std::chrono::steady_clock::now();
    auto d = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    g_build_times.ms["upload_impl: wait for the stream"] += d(t_enter, t_synced);
    g_build_times.ms["upload_impl: index build (sum of the sections)"] += d(t_u0, t_u1);
    g_build_times.ms["upload_impl: staging + enqueue"] += d(t_u1, t_u2);
    g_build_times.ms["upload_impl calls"] += 1.0;
  }
  if (dbg_t) {
    const auto t_u2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "upload: sync %.3f ms, graphs %.3f ms, index build %.3f ms, staging + enqueue %.3f ms, arena %zu bytes data + %zu zero\n",
                 std::chrono::duration<double, std::milli>(t_synced - t_enter).count(),
                 std::chrono::duration<double, std::milli>(t_u0 - t_synced).count(),
                 std::chrono::duration<double, std::milli>(t_u1 - t_u0).count(),
                 std::chrono::duration<double, std::milli>(t_u2 - t_u1).count(), (size_t)A.size, (size_t)A.zsize);
  }
  return OKVIS_BA_OK;
}

int okvis_ba_check_window(const okvis_ba_window* w, const okvis_ba_options* opt, int64_t* stats) {
  if (!w) return OKVIS_BA_ERR_ARG;
  okvis_ba_options o;
  if (opt) o = *opt; else okvis_ba_default_options(&o);
  Arena A;
  HostWin H;
  // (the staging bytes are kept between calls like a solver keeps them between uploads — except for a dump, whose alignment
  // gaps must be zero)
  static thread_local StageVec kept;
  const bool dumping = !debug_word().arena.empty();
  if (!dumping) A.host.swap(kept);
  struct GiveBack {
    StageVec& a;
    StageVec& b;
    bool on;
    ~GiveBack() {
      if (on) a.swap(b);
    }
  } give_back{A.host, kept, !dumping};
  // the route okvis_ba_upload takes for a one-window batch: the piece path's lists unless the window does not fit them
  int chain_min = 1;
  bool chain = want_chain(o, &chain_min), lin2 = !(o.reserved0 & 8);
  int rc;
  for (;;) {   // (the route okvis_ba_upload takes for a one-window batch)
    A.size = 0;
    A.zsize = 0;
    rc = build_window(*w, o, A, H, 1, lin2, chain);
    if (rc == BW_LIN2_UNFIT) lin2 = false;
    else if (rc == BW_CHAIN_UNFIT) chain = false;
    else break;
  }
  if (rc != OKVIS_BA_OK) return rc;
  if (dumping) {   // diagnostics (OKVIS_BA_DEBUG=arena=<file>): the index build's output, byte for byte
    if (FILE* f = std::fopen(debug_word().arena.c_str(), "wb")) {
      std::fwrite(A.host.data(), 1, A.size, f);
      std::fwrite(&H.ptrs, 1, sizeof(H.ptrs), f);
      std::fclose(f);
    }
  }
  if (stats) {
    stats[0] = H.D; stats[1] = H.Dp; stats[2] = H.n_pair; stats[3] = H.n_group; stats[4] = H.n_chunk;
    stats[5] = H.ptrs.n_task; stats[6] = H.ptrs.gpart_size; stats[7] = (int64_t)A.total();
  }
  return OKVIS_BA_OK;
}

int okvis_ba_check_window_lists(const okvis_ba_window* w, const okvis_ba_options* opt, int32_t n_windows, int32_t which,
                                int32_t* out, int64_t capacity, int64_t* n) {
  if (!w || !n || n_windows <= 0 || capacity < 0 || (capacity > 0 && !out)) return OKVIS_BA_ERR_ARG;
  okvis_ba_options o;
  if (opt) o = *opt; else okvis_ba_default_options(&o);
  Arena A;
  HostWin H;
  int chain_min = 1;
  bool chain = want_chain(o, &chain_min), lin2 = !(o.reserved0 & 8);
  int rc;
  for (;;) {
    A.size = 0;
    A.zsize = 0;
    rc = build_window(*w, o, A, H, n_windows, lin2, chain);
    if (rc == BW_LIN2_UNFIT) lin2 = false;
    else if (rc == BW_CHAIN_UNFIT) chain = false;
    else break;
  }
  if (rc != OKVIS_BA_OK) return rc;
  const WinPtrs& P = H.ptrs;   // (the pointer members still hold offsets into the arena's data part)
  const unsigned char* base = A.host.data();
  auto at = [&](const void* field) { return base + reinterpret_cast<size_t>(field); };
  const int nb = P.Dp / 6;
  const void* src = nullptr;
  int64_t count = 0;
  int width = 4;   // bytes per entry
  switch (which) {
    case OKVIS_BA_LIST_GROUPS: src = at((const void*)P.groups), count = 16 * (int64_t)P.n_group; break;
    case OKVIS_BA_LIST_LM_OBS_BEGIN: src = at((const void*)P.lm_obs_begin), count = P.n_lm + 1; break;
    case OKVIS_BA_LIST_LM_PAIR_BEGIN: src = at((const void*)P.lm_pair_begin), count = P.n_lm + 1; break;
    case OKVIS_BA_LIST_PAIR_LM: src = at((const void*)P.pair_lm), count = P.n_pair; break;
    case OKVIS_BA_LIST_PAIR_BLOCK: src = at((const void*)P.pair_block), count = P.lin2 ? P.n_pair : 0; break;
    case OKVIS_BA_LIST_PAIR_OFF: src = at((const void*)P.pair_off), count = P.n_pair; break;
    case OKVIS_BA_LIST_PAIR_ROLE: src = at((const void*)P.pair_role), count = P.n_pair; break;
    case OKVIS_BA_LIST_LM_PIECE_BEGIN: src = at((const void*)P.lm_piece_begin), count = P.lin2 ? P.n_lm + 1 : 0; break;
    case OKVIS_BA_LIST_PAIR_PIECE: src = at((const void*)P.pair_piece), count = P.lin2 ? P.n_pair : 0; break;
    case OKVIS_BA_LIST_PAIR_LIST_BEGIN: src = at((const void*)P.pair_list_begin), count = P.n_pair + 1; break;
    case OKVIS_BA_LIST_PAIR_LIST:
      src = at((const void*)P.pair_list), width = 2;
      count = reinterpret_cast<const int*>(at((const void*)P.pair_list_begin))[P.n_pair];
      break;
    case OKVIS_BA_LIST_TASKS: src = at((const void*)P.tasks), count = 6 * (int64_t)P.n_task; break;
    case OKVIS_BA_LIST_TASK_LIST: {
      src = at((const void*)P.task_list), width = 2;
      const Group* g = reinterpret_cast<const Group*>(at((const void*)P.groups));
      count = P.n_group ? g[P.n_group - 1].tlist_end : 0;
      break;
    }
    case OKVIS_BA_LIST_CHUNKS: src = at((const void*)P.chunks), count = 2 * (int64_t)P.n_chunk; break;
    case OKVIS_BA_LIST_CHUNK_DIAG_BEGIN: src = at((const void*)P.chunk_diag_begin), count = (int64_t)P.n_chunk * nb + 1; break;
    case OKVIS_BA_LIST_CHUNK_DIAG_OUT:
      src = at((const void*)P.chunk_diag_out);
      count = reinterpret_cast<const int*>(at((const void*)P.chunk_diag_begin))[(size_t)P.n_chunk * nb];
      break;
    case OKVIS_BA_LIST_CHUNK_DESC: src = at((const void*)P.chunk_desc), count = (int64_t)P.n_chunk * SCHUR_DESC_INTS; break;
    case OKVIS_BA_LIST_PIECE_PATH: count = 1; break;
    case OKVIS_BA_LIST_LDL_COMP: count = 1; break;
    case OKVIS_BA_LIST_CHAIN: count = 1; break;
    default: return OKVIS_BA_ERR_ARG;
  }
  *n = count;
  if (capacity < count) return OKVIS_BA_ERR_ARG;
  if (which == OKVIS_BA_LIST_PIECE_PATH) {
    out[0] = P.lin2;
  } else if (which == OKVIS_BA_LIST_LDL_COMP) {
    out[0] = (int32_t)P.ldl_comp;
  } else if (which == OKVIS_BA_LIST_CHAIN) {
    out[0] = P.chain;
  } else if (width == 4) {
    if (count) std::memcpy(out, src, 4 * (size_t)count);
  } else {
    const uint16_t* s16 = static_cast<const uint16_t*>(src);
    for (int64_t i = 0; i < count; ++i) out[i] = s16[i];
  }
  return OKVIS_BA_OK;
}

int okvis_ba_set_state(okvis_ba_solver* s, int w, const double* pose, const double* sb, const double* lm) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return s && !s->uploaded ? OKVIS_BA_ERR_STATE : OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = refresh_acc(s, w)) return rc;
  HostWin& H = s->wins[w];
  if (pose) HIP_TRY(hipMemcpyAsync(H.ptrs.pose[H.acc], pose, 56 * (size_t)H.n_pose, hipMemcpyHostToDevice, s->stream));
  if (sb) HIP_TRY(hipMemcpyAsync(H.ptrs.sb[H.acc], sb, 72 * (size_t)H.n_sb, hipMemcpyHostToDevice, s->stream));
  if (lm) HIP_TRY(hipMemcpyAsync(H.ptrs.lm[H.acc], lm, 32 * (size_t)H.n_lm, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->begun = false;
  s->res_staged = s->mirror_fresh = false;
  return OKVIS_BA_OK;
}

int okvis_ba_get_state(okvis_ba_solver* s, int w, double* pose, double* sb, double* lm) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return s && !s->uploaded ? OKVIS_BA_ERR_STATE : OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = refresh_acc(s, w)) return rc;
  HostWin& H = s->wins[w];
  if (pose) HIP_TRY(hipMemcpyAsync(pose, H.ptrs.pose[H.acc], 56 * (size_t)H.n_pose, hipMemcpyDeviceToHost, s->stream));
  if (sb) HIP_TRY(hipMemcpyAsync(sb, H.ptrs.sb[H.acc], 72 * (size_t)H.n_sb, hipMemcpyDeviceToHost, s->stream));
  if (lm) HIP_TRY(hipMemcpyAsync(lm, H.ptrs.lm[H.acc], 32 * (size_t)H.n_lm, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return OKVIS_BA_OK;
}

// the packed results of window w (pose | speed/bias | landmarks | quality | IMU reference biases) in page-locked host memory
static int stage_results(okvis_ba_solver* s, int w, const unsigned char** rec) {
  HostWin& H = s->wins[w];
  *rec = nullptr;
  if (s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: its numbers wait in the same staging)
  const size_t total = results_bytes(H.n_pose, H.n_sb, H.n_lm, H.n_imu);
  if (total == 0) return OKVIS_BA_OK;
  if (s->res_staged && w == 0 && s->wins.size() == 1) {   // packed and copied by okvis_ba_finish already
    *rec = s->stage_res.data();
    return OKVIS_BA_OK;
  }
  if (int rc = refresh_acc(s, w)) return rc;
  s->stage_dl.resize(total);
  HIP_TRY(launch_imu_take_back(s, w, 1));
  hipLaunchKernelGGL(pack_results_kernel, dim3(8), dim3(256), 0, s->stream, s->d_wins + w, H.acc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->stage_dl.data(), H.ptrs.results, total, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *rec = s->stage_dl.data();
  return OKVIS_BA_OK;
}

int okvis_ba_set_patchable(okvis_ba_solver* s, int on) {
  if (!s) return OKVIS_BA_ERR_ARG;
  s->patchable = on != 0;
  if (!s->patchable) s->mirrors.clear();
  return OKVIS_BA_OK;
}

// the containers take over the values the device holds
static int refresh_mirrors(okvis_ba_solver* s) {
  if (s->mirror_fresh) return OKVIS_BA_OK;
  for (size_t i = 0; i < s->wins.size(); ++i) {
    const unsigned char* rec = nullptr;
    if (int rc = stage_results(s, (int)i, &rec)) return rc;
    if (rec) s->mirrors[i].take_results(rec, s->evaluated);
  }
  s->mirror_fresh = true;
  return OKVIS_BA_OK;
}

int okvis_ba_patch_window(okvis_ba_solver* s, int w, const okvis_ba_patch* p) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !p) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded || !s->patchable || s->mirrors.size() != s->wins.size()) return OKVIS_BA_ERR_STATE;
  if (w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  auto pt_prev = std::chrono::steady_clock::now();
  auto PT = [&](const char* name) {   // (OKVIS_BA_DEBUG=build: mean host time per section, printed at exit with build_window's)
    if (g_build_times.on) {
      const auto t_ = std::chrono::steady_clock::now();
      g_build_times.ms[name] += std::chrono::duration<double, std::milli>(t_ - pt_prev).count();
      pt_prev = t_;
    }
  };
  if (int rc = refresh_mirrors(s)) return rc;
  PT("patch: values from the device");
  // All or nothing.  The edit is applied to a COPY of window w's container; the solver's own container only changes (one swap,
  // which cannot throw) after the edited window has been indexed and uploaded.  Whatever fails before that — a rejected patch, a
  // structure limit, an allocation, the device — leaves the containers as they were; if the device no longer holds the old
  // windows (a failed upload has dropped them), they are uploaded again from the untouched containers.
  WindowStore& after = s->mirror_edit;   // (kept between calls: the copy reuses its storage)
  bool upload_started = false;
  int rc = OKVIS_BA_OK;
  try {
    after = s->mirrors[w];
    PT("patch: copy of the container");
    rc = after.apply(*p);   // (checks the whole patch before it touches `after`; `after` is discarded on failure anyway)
    PT("patch: edit");
    if (rc == OKVIS_BA_OK) {
      std::vector<okvis_ba_window> views(s->mirrors.size());
      for (size_t i = 0; i < s->mirrors.size(); ++i) ((int)i == w ? after : s->mirrors[i]).view(&views[i]);
      upload_started = true;
      rc = upload_impl(s, (int)views.size(), views.data());
      PT("patch: index + upload");
    }
  } catch (const std::bad_alloc&) {
    rc = OKVIS_BA_ERR_ARG;
  }
  if (rc != OKVIS_BA_OK) {
    if (upload_started && !s->uploaded) {   // the old windows back on the device (the containers still hold them)
      try {
        std::vector<okvis_ba_window> views(s->mirrors.size());
        for (size_t i = 0; i < s->mirrors.size(); ++i) s->mirrors[i].view(&views[i]);
        (void)upload_impl(s, (int)views.size(), views.data());
      } catch (const std::bad_alloc&) {
      }
    }
    // (after a successful re-upload the device holds exactly what the containers hold)
    s->mirror_fresh = s->uploaded;
    return rc;
  }
  std::swap(s->mirrors[w], after);   // (moves of vectors: cannot throw)
  s->mirror_fresh = true;   // the device holds exactly what the containers hold
  return OKVIS_BA_OK;
}

int okvis_ba_set_marg_prior_values(okvis_ba_solver* s, int w, const double* J, const double* e0) {
  if (!s || !J || !e0) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded || s->marg_pending.active) return OKVIS_BA_ERR_STATE;
  if (w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  HostWin& H = s->wins[w];
  const int Dm = H.marg_dim;
  if (Dm <= 0) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  // J | H0 | e0 sit behind one another in the arena (each on its 256-byte boundary): one staged image of that span, one copy
  const uintptr_t aJ = reinterpret_cast<uintptr_t>(H.ptrs.marg_J), aH0 = reinterpret_cast<uintptr_t>(H.ptrs.marg_H0),
                  ae0 = reinterpret_cast<uintptr_t>(H.ptrs.marg_e0);
  unsigned char* const dJ = reinterpret_cast<unsigned char*>(aJ);
  const size_t o_H0 = (size_t)(aH0 - aJ), o_e0 = (size_t)(ae0 - aJ);
  const size_t nJ = 8 * (size_t)Dm * Dm, span = o_e0 + 8 * (size_t)Dm;
  if (!(nJ <= o_H0 && o_H0 + nJ <= o_e0)) return OKVIS_BA_ERR_STATE;   // (the layout build_window gives them)
  try {
    if (s->stage_marg_vals.size() < span) s->stage_marg_vals.resize(span);
  } catch (const std::bad_alloc&) {
    return OKVIS_BA_ERR_ARG;
  }
  unsigned char* h = s->stage_marg_vals.data();
  // (the staging of the previous call must have been read: an event behind its copy, long reached in the steady state)
  if (!s->ev_marg_vals) HIP_TRY(hipEventCreateWithFlags(&s->ev_marg_vals, hipEventDisableTiming));
  else HIP_TRY(hipEventSynchronize(s->ev_marg_vals));
  std::memcpy(h, J, nJ);
  if (!H.h0_on_device) {
    double* H0 = reinterpret_cast<double*>(h + o_H0);
    std::memset(H0, 0, nJ);
    marg_h0_host(J, Dm, H0);
  }
  std::memcpy(h + o_e0, e0, 8 * (size_t)Dm);
  if (H.h0_on_device) {   // (H0 stays where it is and is formed again on the device behind the copy)
    HIP_TRY(hipMemcpyAsync(dJ, h, nJ, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(dJ + o_e0, h + o_e0, 8 * (size_t)Dm, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(marg_h0_kernel, dim3((unsigned)(((size_t)Dm * Dm + 255) / 256), 1), dim3(256), 0, s->stream, s->d_wins, w);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpyAsync(dJ, h, span, hipMemcpyHostToDevice, s->stream));
  }
  HIP_TRY(hipEventRecord(s->ev_marg_vals, s->stream));
  if (s->patchable && (size_t)w < s->mirrors.size()) {   // the container holds what the device holds
    try {
      s->mirrors[w].marg_J.assign(J, J + (size_t)Dm * Dm);
      s->mirrors[w].marg_e0.assign(e0, e0 + Dm);
    } catch (const std::bad_alloc&) {
      s->mirrors.clear();   // (no container any more: the next patch is refused and the caller uploads)
      s->mirror_fresh = false;
    }
  }
  s->begun = false;
  s->evaluated = false;
  s->res_staged = false;
  return OKVIS_BA_OK;
}

int okvis_ba_patched_view(okvis_ba_solver* s, int w, okvis_ba_window* out) {
  if (!s || !out) return OKVIS_BA_ERR_ARG;
  if (!s->patchable || s->mirrors.size() != s->wins.size() || s->mirrors.empty()) return OKVIS_BA_ERR_STATE;
  if (w < 0 || w >= (int)s->mirrors.size()) return OKVIS_BA_ERR_ARG;
  if (s->uploaded) {
    HIP_TRY(hipSetDevice(s->device));
    if (int rc = refresh_mirrors(s)) return rc;
  }
  s->mirrors[w].view(out);
  return OKVIS_BA_OK;
}

int okvis_ba_fetch_results(okvis_ba_solver* s, int w, double* pose, double* sb, double* lm, double* lm_quality,
                           double* imu_sb_ref) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return s && !s->uploaded ? OKVIS_BA_ERR_STATE : OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  HostWin& H = s->wins[w];
  // everything is gathered on the device into one contiguous record (pose | speed/bias | landmarks | quality | IMU reference
  // biases): one small kernel + ONE copy into page-locked staging instead of five copies (each costs ~15 us of its own)
  const size_t b_pose = 56 * (size_t)H.n_pose, b_sb = 72 * (size_t)H.n_sb, b_lm = 32 * (size_t)H.n_lm;
  const size_t b_q = 8 * (size_t)H.n_lm, b_ref = 72 * (size_t)H.n_imu;
  const size_t o_sb = b_pose, o_lm = o_sb + b_sb, o_q = o_lm + b_lm, o_ref = o_q + b_q, total = o_ref + b_ref;
  if (total == 0) return OKVIS_BA_OK;
  const unsigned char* st = nullptr;
  if (int rc = stage_results(s, w, &st)) return rc;
  if (pose && b_pose) std::memcpy(pose, st, b_pose);
  if (sb && b_sb) std::memcpy(sb, st + o_sb, b_sb);
  if (lm && b_lm) std::memcpy(lm, st + o_lm, b_lm);
  if (lm_quality && b_q) std::memcpy(lm_quality, st + o_q, b_q);
  if (imu_sb_ref && b_ref) std::memcpy(imu_sb_ref, st + o_ref, b_ref);
  return OKVIS_BA_OK;
}

int okvis_ba_fetch_imu_caches(okvis_ba_solver* s, int w, double* caches) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return s && !s->uploaded ? OKVIS_BA_ERR_STATE : OKVIS_BA_ERR_ARG;
  if (!caches) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  HostWin& H = s->wins[w];
  if (H.n_imu == 0) return OKVIS_BA_OK;
  const unsigned char* st = nullptr;
  if (int rc = stage_results(s, w, &st)) return rc;
  std::memcpy(caches, st + results_bytes(H.n_pose, H.n_sb, H.n_lm, 0) + 72 * (size_t)H.n_imu, sizeof(ImuCacheD) * (size_t)H.n_imu);
  return OKVIS_BA_OK;
}

int okvis_ba_begin(okvis_ba_solver* s) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (s) s->acc_fresh = s->res_staged = s->mirror_fresh = false;
  if (!s) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded) return OKVIS_BA_ERR_STATE;
  HIP_TRY(hipSetDevice(s->device));
  s->slots = 0;
  {
    const size_t n = s->wins.size();
    HIP_TRY(reserve_ctrl_stage(s, n));
    int* accs = reinterpret_cast<int*>(s->h_ctrl_stage);
    for (size_t i = 0; i < n; ++i) accs[i] = s->wins[i].acc;
    HIP_TRY(hipMemcpyAsync(s->d_ctrl_stage, accs, sizeof(int) * n, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(begin_kernel, dim3((unsigned)n), dim3(256), 0, s->stream, s->d_wins, reinterpret_cast<const int*>(s->d_ctrl_stage),
                       s->opt.initial_radius);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(launch_lin(s, whole(s), 1));
  s->begun = true;
  s->evaluated = true;
  return OKVIS_BA_OK;
}

int okvis_ba_iterate(okvis_ba_solver* s, int n) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (s) s->acc_fresh = false;
  if (!s || n < 0) return OKVIS_BA_ERR_ARG;
  if (!s->begun) return OKVIS_BA_ERR_STATE;
  if (n == 0) return OKVIS_BA_OK;
  HIP_TRY(hipSetDevice(s->device));
  s->slots += n;
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
  const int nsub = (int)s->sub_streams.size();
  if (s->opt.use_graph && nsub > 1) {
    // one graph per sub-batch, each replayed on its own stream (independent launches overlap; branches
    // inside ONE captured graph were measured not to)
    HIP_TRY(hipEventRecord(s->ev_fork, s->stream));
    for (int k = 0; k < nsub; ++k) {
      hipGraphExec_t exec = nullptr;
      auto it = s->sub_graphs.find({n, k});
      if (it == s->sub_graphs.end()) {
        hipGraph_t graph = nullptr;
        const Sub b{s->sub_streams[k], s->sub_begin[k], s->sub_begin[k + 1] - s->sub_begin[k]};
        HIP_TRY(hipStreamBeginCapture(b.st, hipStreamCaptureModeRelaxed));
        hipError_t e = launch_budget(s, b, n);
        for (int i = 0; i < n && e == hipSuccess; ++i) e = launch_iteration(s, b);
        hipError_t e2 = hipStreamEndCapture(b.st, &graph);
        if (e != hipSuccess) HIP_TRY(e);
        HIP_TRY(e2);
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphDestroy(graph));
        s->sub_graphs[{n, k}] = exec;
      } else {
        exec = it->second;
      }
      HIP_TRY(hipStreamWaitEvent(s->sub_streams[k], s->ev_fork, 0));
      // the sub-batches start staggered by a third (1 / nsub) of one iteration chain: started together, the streams lock into
      // one of two interleavings (0.162 or 0.179 ms per step at 64 windows, whole timed regions in either); the stagger
      // starts them in the pipelined pattern
      if (k > 0 && s->stagger_ticks > 0) {
        hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s->sub_streams[k], (long long)k * s->stagger_ticks);
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipGraphLaunch(exec, s->sub_streams[k]));
      HIP_TRY(hipEventRecord(s->sub_events[k], s->sub_streams[k]));
      HIP_TRY(hipStreamWaitEvent(s->stream, s->sub_events[k], 0));
    }
  } else if (s->opt.use_graph) {
    hipGraphExec_t exec = nullptr;
    auto it = s->graphs.find(n);
    if (it == s->graphs.end()) {
      hipGraph_t graph = nullptr;
      HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
      hipError_t e = launch_iterations_forked(s, n);
      hipError_t e2 = hipStreamEndCapture(s->stream, &graph);
      if (e != hipSuccess) HIP_TRY(e);
      HIP_TRY(e2);
      HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      HIP_TRY(hipGraphDestroy(graph));
      s->graphs[n] = exec;
      // the capture swallowed the first event record: re-record outside the graph
      HIP_TRY(hipEventRecord(s->ev0, s->stream));
    } else {
      exec = it->second;
    }
    HIP_TRY(hipGraphLaunch(exec, s->stream));
  } else {
    HIP_TRY(launch_iterations_forked(s, n));
  }
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  return OKVIS_BA_OK;
}

int okvis_ba_last_iterate_ms(okvis_ba_solver* s, float* total_ms) {
  if (!s || !total_ms) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(total_ms, s->ev0, s->ev1));
  return OKVIS_BA_OK;
}

int okvis_ba_finish(okvis_ba_solver* s, okvis_ba_summary* summaries) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s) return OKVIS_BA_ERR_ARG;
  if (!s->begun) return OKVIS_BA_ERR_STATE;
  HIP_TRY(hipSetDevice(s->device));
  // the final accept/reject needs the Schur partials of the buffer it may accept (gradient test)
  HIP_TRY(launch_schur(s, whole(s), 1));
  HIP_TRY(launch_solve(s, whole(s), 1));
  // landmark quality, the packed results of a one-window solver (the estimator's case: okvis_ba_fetch_results then costs no
  // launch, no copy and no synchronisation of its own) and the control records, all behind ONE synchronisation
  std::vector<Ctrl> cs;
  size_t res_bytes = 0;
  auto finalize = [&]() -> int {
    HIP_TRY(launch_imu_take_back(s, 0, (int)s->wins.size()));
    if (s->max_lm > 0) {
      hipLaunchKernelGGL(quality_kernel, dim3((s->max_lm + 255) / 256, (unsigned)s->wins.size()), dim3(256), 0, s->stream, s->d_wins);
      HIP_TRY(hipGetLastError());
    }
    if (s->wins.size() == 1) {
      const HostWin& H0 = s->wins[0];
      res_bytes = results_bytes(H0.n_pose, H0.n_sb, H0.n_lm, H0.n_imu);
      if (res_bytes) {
        s->stage_res.resize(res_bytes);
        hipLaunchKernelGGL(pack_results_kernel, dim3(8), dim3(256), 0, s->stream, s->d_wins, -1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(s->stage_res.data(), H0.ptrs.results, res_bytes, hipMemcpyDeviceToHost, s->stream));
      }
    }
    return fetch_ctrl(s, cs);
  };
  int rc = finalize();
  if (rc != OKVIS_BA_OK) return rc;
  if (s->opt.strategy == OKVIS_BA_STRATEGY_DOGLEG && !s->opt.gauss_newton) {
    // Dogleg: a launch slot that had to redo a mis-speculated Gauss-Newton trial as an explicit dogleg step did not
    // finish an iteration, and the decision just taken may itself ask for such a redo.  Windows that still owe
    // iterations of this call's budget get the missing slots (the others are stopped by the budget), then the final
    // decision (and what hangs on it) is taken again.  Rare: only when the Gauss-Newton point lies outside the trust region —
    // the common case pays one synchronisation for the whole finish.
    for (int round = 0; round < 64 && !s->skip_topup; ++round) {
      int need = 0;
      for (const Ctrl& c : cs) {
        if (c.done) continue;
        int k = c.max_iter - c.iter;
        if (c.explicit_next == 2) k += 1;   // the current iteration itself is unfinished
        need = std::max(need, k);
      }
      if (need <= 0) break;
      s->slots += need;
      HIP_TRY(launch_iterations_forked(s, need, 0));   // (on the sub-batch streams like every other iteration; no new budget)
      HIP_TRY(launch_schur(s, whole(s), 1));
      HIP_TRY(launch_solve(s, whole(s), 1));
      rc = finalize();
      if (rc != OKVIS_BA_OK) return rc;
    }
  }
  s->res_staged = res_bytes > 0;
  for (size_t i = 0; i < s->wins.size(); ++i) {
    s->wins[i].acc = cs[i].acc;
    if (summaries) {
      okvis_ba_summary& o = summaries[i];
      o.initial_cost = cs[i].initial_cost;
      o.final_cost = cs[i].cost;
      o.iterations = cs[i].iter;
      o.successful_steps = cs[i].successful;
      o.termination = cs[i].done ? (cs[i].done - 1 == 6 ? 0 : cs[i].done - 1) : 0;   // 6 = budget used up = max iterations
      o.reserved = cs[i].chol_fail;
      o.final_radius = cs[i].radius;
      o.gradient_max_norm = cs[i].grad_max;
    }
  }
  s->begun = false;
  s->acc_fresh = true;
  return OKVIS_BA_OK;
}

int okvis_ba_optimize(okvis_ba_solver* s, int num_iter, okvis_ba_summary* summaries) {
  int rc = okvis_ba_begin(s);
  if (rc != OKVIS_BA_OK) return rc;
  rc = okvis_ba_iterate(s, num_iter);
  if (rc != OKVIS_BA_OK) return rc;
  return okvis_ba_finish(s, summaries);
}

int okvis_ba_optimize_timed(okvis_ba_solver* s, int max_iter, int min_iter, double time_limit_s,
                            okvis_ba_summary* summaries) {
  if (!s || max_iter < 0 || min_iter < 0) return OKVIS_BA_ERR_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = okvis_ba_begin(s);
  if (rc != OKVIS_BA_OK) return rc;
  int done = 0;
  if (time_limit_s < 0) {  // Estimator::setOptimizationTimeLimit: no limit -> min iterations = max iterations
    rc = okvis_ba_iterate(s, max_iter);
    if (rc != OKVIS_BA_OK) return rc;
  } else {
    const int first = std::min(min_iter, max_iter);
    rc = okvis_ba_iterate(s, first);
    if (rc != OKVIS_BA_OK) return rc;
    done = first;
    // CeresIterationCallback.hpp:77-86: stop once iteration >= min and the time budget is exceeded
    while (done < max_iter) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (done >= min_iter && el > time_limit_s) {
        s->skip_topup = true;   // the time limit ends the call (CeresIterationCallback terminates the solve)
        break;
      }
      rc = okvis_ba_iterate(s, 1);
      if (rc != OKVIS_BA_OK) return rc;
      ++done;
    }
  }
  rc = okvis_ba_finish(s, summaries);
  s->skip_topup = false;
  return rc;
}

int okvis_ba_evaluate_cost(okvis_ba_solver* s, double* costs) {
  if (s) s->acc_fresh = false;
  if (!s || !costs) return OKVIS_BA_ERR_ARG;
  int rc = okvis_ba_begin(s);
  if (rc != OKVIS_BA_OK) return rc;
  std::vector<okvis_ba_summary> sum(s->wins.size());
  rc = okvis_ba_finish(s, sum.data());
  if (rc != OKVIS_BA_OK) return rc;
  for (size_t i = 0; i < sum.size(); ++i) costs[i] = sum[i].final_cost;
  return OKVIS_BA_OK;
}

int okvis_ba_reduced_dim(okvis_ba_solver* s, int w, int32_t* dim) {
  if (!s || !dim || w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  *dim = s->wins[w].D;
  return OKVIS_BA_OK;
}
int okvis_ba_helper_timeouts(okvis_ba_solver* s, int64_t* count) {
  if (!s || !count) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded) return OKVIS_BA_ERR_STATE;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *count = 0;
  for (const HostWin& H : s->wins) {
    int n = 0;
    HIP_TRY(hipMemcpy(&n, H.ptrs.sum_sync + 2, sizeof(int), hipMemcpyDeviceToHost));
    *count += n;
  }
  return OKVIS_BA_OK;
}
int okvis_ba_launch_route(okvis_ba_solver* s, int32_t* route) {
  if (!s || !route) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded) return OKVIS_BA_ERR_STATE;
  for (int i = 0; i < OKVIS_BA_ROUTE_COUNT; ++i) route[i] = 0;
  const int n = (int)s->wins.size(), nsub = std::max<int>(1, (int)s->sub_streams.size());
  int sub_max = n;
  if (s->sub_streams.size() > 1) {
    sub_max = 0;
    for (size_t k = 0; k + 1 < s->sub_begin.size(); ++k) sub_max = std::max(sub_max, s->sub_begin[k + 1] - s->sub_begin[k]);
  }
  route[OKVIS_BA_ROUTE_WINDOWS] = n;
  route[OKVIS_BA_ROUTE_FUSED] = fused(s) ? 1 : 0;
  route[OKVIS_BA_ROUTE_DECISION_FREE_SCHUR] = spec_schur_now(s) ? 1 : 0;
  route[OKVIS_BA_ROUTE_PIECE_PATH] = s->lin2 ? 1 : 0;
  route[OKVIS_BA_ROUTE_SPLIT_SMALL] = s->split_small ? 1 : 0;
  route[OKVIS_BA_ROUTE_SUB_BATCHES] = nsub;
  route[OKVIS_BA_ROUTE_SUB_BATCH_MAX_WINDOWS] = sub_max;
  {   // the kernel launch_schur picks (same conditions, nothing launched)
    const int trows = std::min(TILE_DIM, s->max_Dp);
    int k = 0;
    if (s->max_schur_blocks > 0 && !fused(s)) {
      if (!s->any_ext && !(s->opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_VALU) &&
          (trows + 1 <= SCH2_MAXT_SMALL_ROWS || (s->opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_MFMA_LARGE)))
        k = trows + 1 <= SCH2_MAXT_SMALL_ROWS ? 2 : 3;
      else
        k = 1;
    }
    route[OKVIS_BA_ROUTE_SCHUR_KERNEL] = k;
  }
  route[OKVIS_BA_ROUTE_SOLVE_DBUF] = (s->max_Dpad_small > 0 && (s->group_chunks || s->spec_schur)) ? 1 : 0;
  route[OKVIS_BA_ROUTE_SOLVE_TILED] = s->max_Dpad_large > 0 ? 1 : 0;
  route[OKVIS_BA_ROUTE_SOLVE_HELPERS] = sub_max <= SOLVE_HELPED_MAX_WINDOWS ? SOLVE_HELPERS : 0;
  route[OKVIS_BA_ROUTE_GRAPH] = s->opt.use_graph ? 1 : 0;
  int ch = 0;
  for (const HostWin& H : s->wins) ch = std::max(ch, H.n_chunk);
  route[OKVIS_BA_ROUTE_MAX_CHUNKS] = ch;
  route[OKVIS_BA_ROUTE_SLOTS] = (int32_t)std::min<long long>(s->slots, 0x7fffffff);
  route[OKVIS_BA_ROUTE_SMALL_RIDES] = small_rides(s) ? 1 : 0;
  route[OKVIS_BA_ROUTE_SOLVE_MODE] = s->max_Dpad_small > 0 ? (s->chain ? OKVIS_BA_SOLVE_CHAIN : OKVIS_BA_SOLVE_DENSE) : 0;
  return OKVIS_BA_OK;
}
int okvis_ba_pair_count(okvis_ba_solver* s, int w, int32_t* n_pair) {
  if (!s || !n_pair || w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  *n_pair = s->wins[w].n_pair;
  return OKVIS_BA_OK;
}
int okvis_ba_pairs(okvis_ba_solver* s, int w, int32_t* pair_lm, int32_t* pair_block) {
  if (!s || !pair_lm || !pair_block || w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  const HostWin& H = s->wins[w];
  for (int i = 0; i < H.n_pair; ++i) {
    pair_lm[i] = H.pair_lm[i];
    pair_block[i] = H.pair_block[i];
  }
  return OKVIS_BA_OK;
}

static int locate(okvis_ba_solver* s, int w, int which, const double** ptr, int64_t* n) {
  const HostWin& H = s->wins[w];
  const WinPtrs& P = H.ptrs;
  const int a = H.acc;
  switch (which) {
    case OKVIS_BA_ARR_POSE: *ptr = P.pose[a]; *n = 7 * (int64_t)H.n_pose; return 0;
    case OKVIS_BA_ARR_SB: *ptr = P.sb[a]; *n = 9 * (int64_t)H.n_sb; return 0;
    case OKVIS_BA_ARR_LM: *ptr = P.lm[a]; *n = 4 * (int64_t)H.n_lm; return 0;
    case OKVIS_BA_ARR_OBS_RESIDUAL: *ptr = P.obs_r[a]; *n = 2 * (int64_t)H.n_obs; return P.obs_r[a] ? 0 : OKVIS_BA_ERR_STATE;
    case OKVIS_BA_ARR_LM_V: *ptr = P.V[a]; *n = 6 * (int64_t)H.n_lm; return 0;
    case OKVIS_BA_ARR_LM_B: *ptr = P.bl[a]; *n = 3 * (int64_t)H.n_lm; return 0;
    case OKVIS_BA_ARR_LM_HQ: *ptr = P.Hq[a]; *n = 6 * (int64_t)H.n_lm; return 0;
    case OKVIS_BA_ARR_PAIR_W: *ptr = P.W[a]; *n = 18 * (int64_t)H.n_pair; return 0;
    case OKVIS_BA_ARR_REDUCED_S: *ptr = P.S; *n = (int64_t)H.D * H.D; return P.S ? 0 : OKVIS_BA_ERR_STATE;
    case OKVIS_BA_ARR_REDUCED_RHS: *ptr = P.rhs; *n = H.D; return P.rhs ? 0 : OKVIS_BA_ERR_STATE;
    case OKVIS_BA_ARR_STEP: *ptr = P.step; *n = H.D; return 0;
    case OKVIS_BA_ARR_LM_QUALITY: *ptr = P.quality; *n = H.n_lm; return 0;
    case OKVIS_BA_ARR_GRADIENT: *ptr = P.grad; *n = H.D; return 0;
    case OKVIS_BA_ARR_DAMPING: *ptr = P.Dp2; *n = H.D; return P.Dp2 ? 0 : OKVIS_BA_ERR_STATE;
    case 99: *ptr = P.prof; *n = 64 + 4 * 160; return P.prof ? 0 : OKVIS_BA_ERR_STATE;
    case 98: *ptr = nullptr; *n = H.n_imu; return 0;  // diagnostics: re-preintegration count per IMU factor
    case 97: *ptr = nullptr; *n = 24; return 0;       // diagnostics: trust-region control record
    case 96: *ptr = nullptr; *n = 1; return 0;        // diagnostics: launch slots since okvis_ba_begin
    case OKVIS_BA_ARR_IMU_SB_REF: *ptr = nullptr; *n = 9 * (int64_t)H.n_imu; return 0;
    case OKVIS_BA_ARR_IMU_RESIDUAL: *ptr = nullptr; *n = 15 * (int64_t)H.n_imu; return 0;
  }
  return OKVIS_BA_ERR_ARG;
}

int okvis_ba_array_size(okvis_ba_solver* s, int w, int which, int64_t* n_doubles) {
  if (!s || !n_doubles || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  const double* p;
  return locate(s, w, which, &p, n_doubles);
}

int okvis_ba_download(okvis_ba_solver* s, int w, int which, double* out, int64_t n_doubles) {
  if (s && s->marg_pending.active) return OKVIS_BA_ERR_STATE;   // (okvis_ba_marginalize_end first: between the two halves the solver takes no edits and hands out no results)
  if (!s || !out || !s->uploaded || w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  const double* p = nullptr;
  int64_t n = 0;
  if (int rc0 = refresh_acc(s, w)) return rc0;
  int rc = locate(s, w, which, &p, &n);
  if (rc != 0) return rc;
  if (n != n_doubles) return OKVIS_BA_ERR_ARG;
  if (which == OKVIS_BA_ARR_IMU_SB_REF) {
    const HostWin& H = s->wins[w];
    HIP_TRY(launch_imu_take_back(s, w, 1));
    HIP_TRY(hipStreamSynchronize(s->stream));
    // one strided copy gathers the reference biases out of the per-factor cache records
    if (H.n_imu > 0)
      HIP_TRY(hipMemcpy2D(out, 9 * sizeof(double), reinterpret_cast<const unsigned char*>(H.ptrs.imu_cache) + offsetof(ImuCacheD, sb_ref),
                          sizeof(ImuCacheD), 9 * sizeof(double), (size_t)H.n_imu, hipMemcpyDeviceToHost));
    return OKVIS_BA_OK;
  }
  if (which == 96) {   // diagnostics: launch slots since okvis_ba_begin (the same for every window of the batch)
    out[0] = (double)s->slots;
    return OKVIS_BA_OK;
  }
  if (which == 97) {
    Ctrl c;
    HIP_TRY(hipMemcpy(&c, s->wins[w].ptrs.ctrl, sizeof(c), hipMemcpyDeviceToHost));
    const double v[24] = {c.radius, c.mu, c.cost, c.cA, c.beta, c.dl_norm, c.pend_model, c.tot_A, c.tot_C, c.tot_E, c.gd_p,
                          c.ddd_p, c.last_rho, c.last_model_change, (double)c.iter, (double)c.successful, (double)c.tr_kind,
                          (double)c.explicit_next, (double)c.pending, (double)c.acc, (double)c.done, (double)c.max_iter,
                          (double)c.invalid_steps, (double)c.chol_fail};
    for (int i = 0; i < 24; ++i) out[i] = v[i];
    return OKVIS_BA_OK;
  }
  if (which == 98) {
    const HostWin& H = s->wins[w];
    for (int f = 0; f < H.n_imu; ++f) {
      ImuCacheD c;
      HIP_TRY(hipMemcpy(&c, H.ptrs.imu_cache + f, sizeof(c), hipMemcpyDeviceToHost));
      out[f] = c.redo_count;
    }
    return OKVIS_BA_OK;
  }
  if (which == OKVIS_BA_ARR_IMU_RESIDUAL) {
    const HostWin& H = s->wins[w];
    for (int f = 0; f < H.n_imu; ++f)
      HIP_TRY(hipMemcpy(out + 15 * f, H.ptrs.imu_lin[H.acc] + (size_t)f * IMU_LIN_STRIDE + IMU_R, 15 * 8, hipMemcpyDeviceToHost));
    return OKVIS_BA_OK;
  }
  if (n > 0) HIP_TRY(hipMemcpy(out, p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return OKVIS_BA_OK;
}

static int profile_impl(okvis_ba_solver* s, int n, float* ms4, float* per_launch) {
  if (s) s->acc_fresh = false;
  if (!s || n <= 0) return OKVIS_BA_ERR_ARG;
  if (!s->begun) return OKVIS_BA_ERR_STATE;
  HIP_TRY(hipSetDevice(s->device));
  // queue all n iterations with events between the kernels and synchronise ONCE: the kernels run
  // back-to-back on the GPU, so the event intervals are kernel durations, not host launch latency
  std::vector<hipEvent_t> ev(5 * (size_t)n);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  if (ms4)
    for (int k = 0; k < 4; ++k) ms4[k] = 0.f;
  HIP_TRY(launch_budget(s, whole(s), n));
  for (int i = 0; i < n; ++i) {
    hipEvent_t* e = &ev[5 * (size_t)i];
    HIP_TRY(hipEventRecord(e[0], s->stream));
    HIP_TRY(launch_schur(s, whole(s)));
    HIP_TRY(hipEventRecord(e[1], s->stream));
    HIP_TRY(launch_solve(s, whole(s), 0));
    HIP_TRY(hipEventRecord(e[2], s->stream));
    HIP_TRY(hipEventRecord(e[3], s->stream));   // (the small factors run inside the linearise launch)
    HIP_TRY(launch_lin(s, whole(s), 0));
    HIP_TRY(hipEventRecord(e[4], s->stream));
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 4; ++k) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, ev[5 * (size_t)i + k], ev[5 * (size_t)i + k + 1]));
      if (ms4) ms4[k] += t;
      if (per_launch) per_launch[4 * (size_t)i + k] = t;
    }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return OKVIS_BA_OK;
}
int okvis_ba_profile_iterations(okvis_ba_solver* s, int n, float* ms4) {
  if (!ms4) return OKVIS_BA_ERR_ARG;
  return profile_impl(s, n, ms4, nullptr);
}
int okvis_ba_profile_launches(okvis_ba_solver* s, int n, float* ms) {
  if (!ms) return OKVIS_BA_ERR_ARG;
  return profile_impl(s, n, nullptr, ms);
}

int okvis_ba_algorithmic_bytes(okvis_ba_solver* s, int64_t* lin, int64_t* schur, int64_t* solve, int64_t* small) {
  if (!s || !s->uploaded) return OKVIS_BA_ERR_ARG;
  int64_t a = 0, b = 0, c = 0, d = 0;
  for (auto& H : s->wins) {
    a += H.bytes_lin;
    b += H.bytes_schur;
    c += H.bytes_solve;
    d += H.bytes_small;
  }
  if (lin) *lin = a;
  if (schur) *schur = b;
  if (solve) *solve = c;
  if (small) *small = d;
  return OKVIS_BA_OK;
}

int okvis_ba_synchronize(okvis_ba_solver* s) {
  if (!s) return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return OKVIS_BA_OK;
}

#include "capi_standalone.inc"    // okvis_ba_shard, okvis_ba_batch_run, okvis_ba_dense_solve, okvis_ba_reduced_solve

#include "capi_marginalize.inc"   // okvis_ba_marginalize, _begin, _end

}  // extern "C"
