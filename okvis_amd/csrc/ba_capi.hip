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

namespace {

#define HIP_TRY(expr)                                              \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) {                                        \
      s->last_hip_error = (int)_e;                                 \
      return OKVIS_BA_HIP_ERROR_BASE + (int)_e;                    \
    }                                                              \
  } while (0)

struct HostWin {  // host copy of what the queries and downloads need
  int n_pose = 0, n_sb = 0, n_lm = 0, n_obs = 0, n_imu = 0, D = 0, Dp = 0, n_pair = 0, n_group = 0, n_chunk = 0;
  std::vector<int> pair_lm, pair_block;
  std::vector<int> pose_off, sb_off;  // reduced ordering (host copy)
  int marg_dim = 0;
  bool h0_on_device = false;   // H0 = J^T J of a large prior is formed by marg_h0_kernel after the upload, not by build_window
  bool group_chunks = false;   // one Schur chunk per linearise group (what the fused linearise + reduce launch needs)
  bool spec_ok = false;        // the window can take a decision-free Schur launch (one set of partials per linearisation buffer, see spec_schur)
  bool chain = false;          // laid out for the chain solver (ba_chain.hpp): WinPtrs::chain > 0
  WinPtrs ptrs;  // device pointers
  int acc = 0;
  int64_t bytes_lin = 0, bytes_schur = 0, bytes_solve = 0, bytes_small = 0;
};

// Device arena of a batch = [data part, built on the host and copied over PCIe | zero part, cleared on the device].
// Offsets into the zero part carry ARENA_ZFLAG until relocate() turns them into pointers: the linearisation buffers,
// Schur partials and work areas are more than half of a window's bytes and need not cross PCIe as zeros.
constexpr size_t ARENA_ZFLAG = size_t(1) << 62;
// Page-locked host memory for the staging buffers (H2D copies from pinned memory are asynchronous and about twice as
// fast as from pageable memory); plain malloc when there is no device (okvis_ba_check_window on a CPU-only host).
template <class T>
struct StageAlloc {
  typedef T value_type;
  StageAlloc() = default;
  template <class U>
  StageAlloc(const StageAlloc<U>&) {}
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T) + 16;
    void* p = nullptr;
    unsigned char tag = 1;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
      (void)hipGetLastError();
      p = std::malloc(bytes);
      tag = 0;
      if (!p) throw std::bad_alloc();
    }
    static_cast<unsigned char*>(p)[0] = tag;
    return reinterpret_cast<T*>(static_cast<unsigned char*>(p) + 16);
  }
  void deallocate(T* q, size_t) {
    unsigned char* p = reinterpret_cast<unsigned char*>(q) - 16;
    if (p[0]) (void)hipHostFree(p); else std::free(p);
  }
  template <class U>
  bool operator==(const StageAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const StageAlloc<U>&) const { return false; }
};
typedef std::vector<unsigned char, StageAlloc<unsigned char>> StageVec;
// page-locked (and so readable by the device in place) or the plain-malloc fall-back?  (the tag StageAlloc keeps in front of the block)
inline bool stage_is_pinned(const StageVec& v) { return !v.empty() && (v.data() - 16)[0] == 1; }
struct Arena {
  StageVec host;   // data part
  size_t size = 0;                   // bytes of the data part
  size_t zsize = 0;                  // bytes of the zero part
  size_t alloc(size_t bytes) {
    size_t off = (size + 255) & ~size_t(255);
    size = off + bytes;
    return off;
  }
  size_t zalloc(size_t bytes) {
    size_t off = (zsize + 255) & ~size_t(255);
    zsize = off + bytes;
    return off | ARENA_ZFLAG;
  }
  size_t data_bytes() const { return (size + 255) & ~size_t(255); }
  size_t total() const { return data_bytes() + zsize; }
};

}  // namespace

struct okvis_ba_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  // sub-batches of windows run on their own streams so that the (latency-bound) phases of different
  // windows overlap on the 256 CUs
  std::vector<hipStream_t> sub_streams;
  std::vector<hipEvent_t> sub_events;
  std::vector<int> sub_begin;  // [n_sub+1] window ranges
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr;
  okvis_ba_options opt;
  OptD* d_opt = nullptr;
  unsigned char* d_arena = nullptr;
  size_t arena_bytes = 0, arena_capacity = 0, wins_capacity = 0;
  WinPtrs* d_wins = nullptr;
  CtrlSlot* d_ctrl = nullptr;    // the control records of the uploaded windows (same allocation, behind the window records)
  StageVec stage_dl;             // pinned staging of result downloads (okvis_ba_marginalize)
  StageVec stage, stage_small;   // pinned staging of the arena's data part / of the WinPtrs + OptD records (kept across uploads)
  std::vector<HostWin> wins;
  bool uploaded = false, begun = false, any_ext = false;
  bool lin2 = false;          // the batch's index lists are those of the piece path (ba_linearize2.hpp)
  bool split_small = false;   // piece path: IMU / prior factors in a launch of their own (small_kernel), three linearise workgroups per CU
  bool group_chunks = false;   // every window of the batch has one Schur chunk per linearise group (see fused())
  // Batches that are not fused but run DOGLEG or fixed-radius iterations on windows the matrix-core Schur kernel serves: the Schur
  // launch takes no decision (schur_mfma_kernel, nodec) and reduces the trial buffer into that buffer's own set of partials, the
  // solve kernel decides (its DBUF instantiation, as in fused mode).  OKVIS_BA_TUNE_SCHUR_DECIDES keeps the decision in the Schur launch.
  bool spec_schur = false;
  bool chain = false;          // the LDS-resident windows of the batch are laid out for the chain solver (ba_chain.hpp): solve_kernel<.., CHAIN>
  int max_chain_doubles = 0;   // its matrix area (LChain::total), largest window
  bool fp32_at_upload = false;
  std::vector<int64_t> launch_sig;   // what the captured graphs depend on (see okvis_ba_upload)
  unsigned char* h_ctrl_stage = nullptr;   // pinned / device staging of per-window control data (begin, fetch_ctrl)
  unsigned char* d_ctrl_stage = nullptr;
  size_t ctrl_stage_bytes = 0;
  // incremental structure updates (okvis_ba_patch_window): the container of every uploaded window, kept only on request
  bool patchable = false;
  std::vector<WindowStore> mirrors;
  WindowStore mirror_edit;   // okvis_ba_patch_window edits a copy: this one
  long marg_tiles_fallbacks = 0;   // okvis_ba_marginalize calls whose tiled tail gave way to the single workgroup (diagnostics)
  bool evaluated = false;      // okvis_ba_begin ran since the last upload: every IMU term's cache has been (re)built
  bool mirror_fresh = false;   // the containers hold the values the device holds (nothing optimised / set since)
  bool res_staged = false;  // stage_res holds window 0's packed results as of the last okvis_ba_finish (single-window solvers)
  StageVec stage_res;
  StageVec stage_marg;           // host-written part of okvis_ba_marginalize's scratch block
  StageVec stage_marg_vals;      // okvis_ba_set_marg_prior_values: the staged J | H0 | e0 span ...
  hipEvent_t ev_marg_vals = nullptr;   // ... and the event behind its copy
  // okvis_ba_marginalize_begin without its _end yet: what _end needs to hand the numbers over (the kept blocks are known at begin)
  struct MargPending {
    bool active = false, synced = false;
    int w = 0, na = 0;
    size_t nn = 0, n1 = 0, out_bytes = 0;
    std::vector<int> bt, bi, bo;
  } marg_pending;
  StageVec stage_pre;            // first preintegrations started at upload (imu_pre_kernel): the staged block and its device copy
  unsigned char* d_pre = nullptr;   // (PRE_MAX_TERMS records)
  bool acc_fresh = false;   // HostWin::acc mirrors the device's accepted-buffer index (no kernel launched since it was read)
  int max_group = 0, max_imu = 0, max_schur_blocks = 0, max_lm = 0, max_Dpad = 0, max_Dp = 0, max_spart_stride = 0;
  int max_Dpad_small = 0, max_Dpad_large = 0;
  long long stagger_ticks = 0;   // start offset between consecutive sub-batch streams (wall_clock64 ticks, 100 MHz); okvis_ba_tuning::stagger_us
  bool skip_topup = false;   // okvis_ba_optimize_timed ran out of time: finish() must not grant the slots mis-speculated steps still owe
  long long slots = 0;   // launch slots (schur + solve + linearise triples) since okvis_ba_begin: diagnostics (array 96)
  std::map<int, hipGraphExec_t> graphs;
  std::map<std::pair<int, int>, hipGraphExec_t> sub_graphs;  // (n, sub) -> graph of that sub-batch's chain
  float last_iterate_ms = 0.f;
  int last_hip_error = 0;
  unsigned char* marg_scratch = nullptr;  // grow-only device scratch of okvis_ba_marginalize
  size_t marg_scratch_bytes = 0;
};

namespace {

size_t solve_smem(int Dpad, bool large);
size_t solve_smem_chain(int chain_doubles, int Dpad);

constexpr int FUSED_MAX_WINDOWS = 48;   // up to here the fused linearise + reduce launch beats the separate Schur launch (tools/gpu_fused_sweep.py:
                                        // 48 windows 148.5 vs 151.3 us per step, 64 windows 176 vs 163)
// (okvis_ba_tuning::fused_max_windows overrides it: A/B sweeps)
int fused_max_windows(const okvis_ba_options& o) {
  return o.tuning.fused_max_windows > 0 ? o.tuning.fused_max_windows : o.tuning.fused_max_windows < 0 ? 0 : FUSED_MAX_WINDOWS;
}
// Print-only diagnostics: the ONE environment variable the library reads, once per process.  OKVIS_BA_DEBUG is a comma-separated
// list of "build" (host time of build_window's sections, printed at exit), "upload" (sections of every upload), "marg" (ranks and
// bounds of every marginalisation), "arena=<file>" (okvis_ba_check_window dumps the index build's output).  Nothing here changes a
// result; everything that does is a field of okvis_ba_options::tuning.
struct DebugWord {
  bool build = false, upload = false, marg = false;
  std::string arena;
  DebugWord() {
    const char* e = std::getenv("OKVIS_BA_DEBUG");
    if (!e) return;
    std::string w(e);
    size_t at = 0;
    while (at <= w.size()) {
      size_t c = w.find(',', at);
      if (c == std::string::npos) c = w.size();
      const std::string tok = w.substr(at, c - at);
      if (tok == "build") build = true;
      else if (tok == "upload") upload = true;
      else if (tok == "marg") marg = true;
      else if (tok.rfind("arena=", 0) == 0) arena = tok.substr(6);
      at = c + 1;
    }
  }
};
const DebugWord& debug_word() {
  static const DebugWord w;
  return w;
}
constexpr size_t OPT_PAD = (sizeof(OptD) + 255) & ~size_t(255);   // the option record in front of the window records (one allocation, one copy)
// [OptD, padded | WinPtrs x n | (padded) CtrlSlot x n]: where the control records of n windows start / how long the block is
constexpr size_t ctrl_off(size_t n) { return (OPT_PAD + sizeof(WinPtrs) * n + 255) & ~size_t(255); }
constexpr size_t records_bytes(size_t n) { return ctrl_off(n) + sizeof(CtrlSlot) * n; }
constexpr int SMALL_BATCH_WINDOWS = 40;   // below: the device is not full - settings that shorten one window's chain win

OptD make_optd(const okvis_ba_options& o, int n_windows) {
  OptD d;
  d.initial_radius = o.initial_radius;
  d.max_radius = o.max_radius;
  d.min_radius = o.min_radius;
  d.min_lm_diag2 = o.min_lm_diagonal;   // Ceres clamps the squared column norm itself to [min_lm_diagonal, max_lm_diagonal]
  d.max_lm_diag2 = o.max_lm_diagonal;
  d.min_relative_decrease = o.min_relative_decrease;
  d.function_tolerance = o.function_tolerance;
  d.gradient_tolerance = o.gradient_tolerance;
  d.parameter_tolerance = o.parameter_tolerance;
  d.gauss_newton = o.gauss_newton;
  d.marg_mode = 0;
  d.dogleg = o.strategy == OKVIS_BA_STRATEGY_DOGLEG;
  d.jacobi_scaling = o.jacobi_scaling != 0;
  d.max_invalid = o.max_consecutive_invalid_steps > 0 ? o.max_consecutive_invalid_steps : 5;
  d.helper_polls = (o.reserved0 & 16) ? 0 : 1 << 22;   // (bit 4: the test of the time-out route gives up at once)
  return d;
}

void destroy_graphs(okvis_ba_solver* s) {
  for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
  s->graphs.clear();
  for (auto& kv : s->sub_graphs) (void)hipGraphExecDestroy(kv.second);
  s->sub_graphs.clear();
}

// ---------------------------------------------------------------------------------------------------
// structure building for one window; appends into the arena and fills HostWin/WinPtrs with OFFSETS
// (converted to device pointers after the arena is allocated).
// ---------------------------------------------------------------------------------------------------
template <class T>
size_t put(Arena& A, const std::vector<T>& v) {
  size_t off = A.alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  // (the staging buffer keeps its size from upload to upload: in steady state nothing is value-initialised here, the bytes are
  // written once by the copy below; the alignment gaps between arrays carry whatever they carried and are never read)
  if (A.host.size() < A.size) A.host.resize(A.size + A.size / 2, 0);
  if (!v.empty()) std::memcpy(A.host.data() + off, v.data(), v.size() * sizeof(T));
  else std::memset(A.host.data() + off, 0, sizeof(T));
  return off;
}
size_t put_zero(Arena& A, size_t bytes) { return A.zalloc(std::max<size_t>(bytes, 8)); }
// a caller's array straight into the arena (n elements; a null pointer or n = 0 leaves one zeroed element, like an empty vector)
template <class T>
size_t put_n(Arena& A, const T* p, size_t n) {
  if (!p) n = 0;
  size_t off = A.alloc(std::max<size_t>(n, 1) * sizeof(T));
  if (A.host.size() < A.size) A.host.resize(A.size + A.size / 2, 0);
  if (n) std::memcpy(A.host.data() + off, p, n * sizeof(T));
  else std::memset(A.host.data() + off, 0, sizeof(T));
  return off;
}

#define OFF(field, off) P.field = reinterpret_cast<std::remove_reference<decltype(P.field)>::type>(off)

// diagnostics: OKVIS_BA_DEBUG=build accumulates the host time of build_window's sections and prints them at exit
struct BuildTimes {
  bool on = debug_word().build;
  struct Acc {   // (`+=` keeps the call sites of the mean-only version)
    std::vector<double> v;
    Acc& operator+=(double x) {
      v.push_back(x);
      return *this;
    }
  };
  std::map<std::string, Acc> ms;
  long calls = 0;
  ~BuildTimes() {
    if (!on || !calls) return;
    std::fprintf(stderr, "build_window: %ld calls, median ms per section (number of samples):", calls);
    for (auto& kv : ms) {
      std::vector<double>& v = kv.second.v;
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      std::fprintf(stderr, "  %s %.4f (%zu)", kv.first.c_str(), v[v.size() / 2], v.size());
    }
    std::fprintf(stderr, "\n");
  }
};
BuildTimes g_build_times;

constexpr int GROUP_LM_DEFAULT = 32;   // landmarks the index build puts into one linearise group (see build_window) ...
constexpr int GROUP_LM_FEW = 16, GROUP_LM_FEW_WINDOWS = 8;   // ... and when at most this many windows share the device
constexpr int H0_DEVICE_MIN = 128;   // rows of a marginalisation prior from which H0 = J^T J is formed on the device

// H0 = J^T J of window blockIdx.y's prior, one entry per work-item, the terms of an entry added in row order and without
// contraction into fused multiply-adds: bit for bit what build_window computes on the host for the small priors
__global__ __launch_bounds__(256) void marg_h0_kernel(const WinPtrs* __restrict__ wins, int w0) {
  const WinPtrs& W = wins[w0 + blockIdx.y];
  const int Dm = W.marg_dim;
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (Dm <= H0_DEVICE_MIN || k >= (size_t)Dm * Dm) return;
  const int i = (int)(k / Dm), j = (int)(k - (size_t)i * Dm);
  const int lo = i < j ? i : j, hi = i < j ? j : i;   // (the host fills the upper triangle and mirrors it)
  const BA_G double* J = W.marg_J;
  double sacc = 0.0;
  {
#pragma clang fp contract(off)   // (the host's index build has no fused multiply-add: product and sum are rounded separately)
    for (int r = 0; r < Dm; ++r) {
      const double pr = J[(size_t)r * Dm + lo] * J[(size_t)r * Dm + hi];
      sacc = sacc + pr;
    }
  }
  const_cast<BA_G double*>(W.marg_H0)[k] = sacc;
}

// Work vectors of build_window, one set per host thread, kept between calls (capacity only: every call assigns what it reads)
struct BuildScratch {
  std::vector<int> pose_off, sb_off, role, lm_obs_begin, pair_lm, pair_block, pair_off, pair_role, lm_pair_begin, blocks, seen;
  std::vector<int> chunk_diag_begin, chunk_diag_out, chunk_cross_begin, chunk_cross, chunk_desc, blk_cursor;
  std::vector<int> pair_list_begin, blk_slot, touched, lm_piece_begin, pair_piece, blk_cnt;
  std::vector<uint16_t> pair_list, task_list;
  std::vector<Group> groups;
  std::vector<Task> tasks;
  std::vector<Chunk> chunks;
};
BuildScratch& build_scratch() {
  static thread_local BuildScratch S;
  return S;
}

// H0 = J^T J on the host (H0 zeroed by the caller): the upper triangle as a sum of row outer products, then mirrored
inline void marg_h0_host(const double* J, int Dm, double* H0) {
  for (int r = 0; r < Dm; ++r) {
    const double* Jr = J + (size_t)r * Dm;
    for (int i = 0; i < Dm; ++i) {
      const double a = Jr[i];
      if (a == 0.0) continue;   // (J of the reference's prior is upper triangular up to the rank: 0 * x adds nothing)
      double* Hi = H0 + (size_t)i * Dm;
      for (int j = i; j < Dm; ++j) Hi[j] += a * Jr[j];
    }
  }
  for (int i = 0; i < Dm; ++i)
    for (int j = i + 1; j < Dm; ++j) H0[(size_t)j * Dm + i] = H0[(size_t)i * Dm + j];
}

// Internal status of build_window(lin2 = true): the window does not fit the piece path of the linearise launch
// (ba_linearize2.hpp: free extrinsics, or one landmark with more than LIN2_PIECES pieces); the caller rebuilds the batch
// for ba_linearize.hpp.
constexpr int BW_LIN2_UNFIT = -1000;
// ... (chain = true): the window's speed/bias blocks do not form a chain in the order of the reduced system, or the chain solver
// does not pay for it (okvis_ba_upload then lays the whole batch out for the dense LDL^T)
constexpr int BW_CHAIN_UNFIT = -1001;
// OKVIS_BA_SOLVE_AUTO: the chain solver from this many speed/bias blocks on.  Measured (tools/gpu_chain_shapes.py, device ticks of
// one solve, chain / dense): 10 poses + 10 blocks 0.92, 12 + 10: 0.93, 10 + 8: 0.96, 10 + 6: 0.97, 10 + 5: 1.06, 10 + 3: 1.12,
// 8 + 3 (the sliding window of the replay): 1.32 — two sweeps, the pose system's update and the back-substitution are fixed costs
// that a short chain does not earn back.
constexpr int CHAIN_AUTO_MIN_BLOCKS = 8;
// the reduced solve the options ask for: 0 = dense, 1 = chain where every window has at least `*min_blocks` blocks in a chain
inline bool want_chain(const okvis_ba_options& o, int* min_blocks) {
  *min_blocks = o.tuning.solve_mode == OKVIS_BA_SOLVE_CHAIN ? 1 : CHAIN_AUTO_MIN_BLOCKS;
  return o.tuning.solve_mode != OKVIS_BA_SOLVE_DENSE;
}

int build_window(const okvis_ba_window& w, const okvis_ba_options& opt, Arena& A, HostWin& H, int n_windows_total = 1, bool lin2 = false,
                 bool chain = false) {
  auto bw_t0 = std::chrono::steady_clock::now();
  if (g_build_times.on) g_build_times.calls++;
#define BW_T(name)                                                                                       \
  do {                                                                                                   \
    if (g_build_times.on) {                                                                              \
      const auto t_ = std::chrono::steady_clock::now();                                                  \
      g_build_times.ms[name] += std::chrono::duration<double, std::milli>(t_ - bw_t0).count();           \
      bw_t0 = t_;                                                                                        \
    }                                                                                                    \
  } while (0)
  if (w.n_pose < 0 || w.n_sb < 0 || w.n_lm < 0 || w.n_obs < 0 || w.n_imu < 0 || w.n_cam < 0) return OKVIS_BA_ERR_ARG;
  if ((w.n_pose && (!w.pose || !w.pose_fixed)) || (w.n_sb && (!w.sb || !w.sb_fixed)) || (w.n_lm && !w.lm))
    return OKVIS_BA_ERR_ARG;
  if (w.n_obs && (!w.obs_lm || !w.obs_pose || !w.obs_ext || !w.obs_cam || !w.obs_uv || !w.obs_sqrtw || !w.cam_intr ||
                  !w.cam_model))
    return OKVIS_BA_ERR_ARG;
  if (w.n_pose > 65535 || w.n_lm >= (1 << 24) || w.n_cam > 255) return OKVIS_BA_ERR_UNSUPPORTED;
  const int npose = w.n_pose, nsb = w.n_sb, nlm = w.n_lm, nobs = w.n_obs;
  // (the index lists live in one set of vectors per host thread: a frame's upload allocates nothing once they have grown)
  BuildScratch& S = build_scratch();
  // ---- reduced ordering: free pose blocks (6 each) then free speed/bias blocks (9 each) ----
  std::vector<int>&pose_off = S.pose_off, &sb_off = S.sb_off;
  pose_off.assign(npose, -1);
  sb_off.assign(nsb, -1);
  int off = 0;
  for (int i = 0; i < npose; ++i)
    if (!w.pose_fixed[i]) {
      pose_off[i] = off;
      off += 6;
    }
  const int Dp = off;
  for (int i = 0; i < nsb; ++i)
    if (!w.sb_fixed[i]) {
      sb_off[i] = off;
      off += 9;
    }
  const int D = off;
  if (D > MAX_D || D == 0) return OKVIS_BA_ERR_UNSUPPORTED;
  // ---- one pass over the observations, landmark by landmark: validation, roles, observation records and the
  //      (landmark, free block) pairs ----
  std::vector<int>&role = S.role, &lm_obs_begin = S.lm_obs_begin;   // role: 0 pose role, 1 extrinsics role
  role.assign(npose, -1);
  lm_obs_begin.resize((size_t)nlm + 1);
  // The arena's first arrays have sizes known by now, so the observation records are written where they stay (the arena is not
  // touched again before the pass below ends; a failed or unfit window's bytes are discarded by the caller).
  WinPtrs& P = H.ptrs;
  std::memset(&P, 0, sizeof(P));
  for (int b = 0; b < 2; ++b) {
    OFF(pose[b], put_n(A, w.pose, 7 * (size_t)npose));
    OFF(sb[b], put_n(A, w.sb, 9 * (size_t)nsb));
    OFF(lm[b], put_n(A, w.lm, 4 * (size_t)nlm));
  }
  OFF(pose_off, put(A, pose_off));
  OFF(sb_off, put(A, sb_off));
  OFF(cam_intr, put_n(A, w.cam_intr, 12 * (size_t)w.n_cam));
  OFF(cam_model, put_n(A, w.cam_model, (size_t)w.n_cam));
  const size_t recs_off = put_n(A, (const ObsRec*)nullptr, 0);   // (an empty array's single zeroed element ...)
  if (nobs > 1) {                                                // (... grown to n_obs records)
    A.size = recs_off + (size_t)nobs * sizeof(ObsRec);
    if (A.host.size() < A.size) A.host.resize(A.size + A.size / 2, 0);
  }
  OFF(obs, recs_off);
  ObsRec* const recs = reinterpret_cast<ObsRec*>(A.host.data() + recs_off);
  std::vector<int>&pair_lm = S.pair_lm, &pair_block = S.pair_block, &pair_off = S.pair_off, &pair_role = S.pair_role, &lm_pair_begin = S.lm_pair_begin;
  int npair_run = 0;   // (the pair vectors are work space of at least 2 n_obs entries: the first npair_run are this window's)
  lm_pair_begin.resize((size_t)nlm + 1);
  bool has_ext = false;
  {
    const size_t guess = 2 * (size_t)nobs;
    if (pair_lm.size() < guess) pair_lm.resize(guess), pair_block.resize(guess), pair_off.resize(guess), pair_role.resize(guess);
    int *const pl = pair_lm.data(), *const pb = pair_block.data(), *const po = pair_off.data(), *const pr = pair_role.data();
    std::vector<int>&blocks = S.blocks, &seen = S.seen;   // a "seen for this landmark" stamp per block
    seen.assign(npose, -1);
    if ((int)blocks.size() < npose + 2) blocks.resize((size_t)npose + 2);   // (a landmark's blocks are distinct; one slot for the store below)
    int* const bl = blocks.data();
    // what the separate passes of the earlier versions reported after the whole observation list had been checked
    bool obs_over = false, pairs_over = false;
    const unsigned unpose = (unsigned)npose, uncam = (unsigned)w.n_cam;
    int o = 0;
    for (int l = 0; l < nlm; ++l) {
      const int o0 = o;
      lm_obs_begin[l] = o0;
      int nb = 0, last = -1;         // blocks of this landmark so far, the last one taken
      bool ascending = true;
      int prev_ip = -1, prev_c = -1;
      // (the tests that depend on the data — same pose as the observation before? a block not seen yet? — are arithmetic, not
      // branches: they fail to predict about once per observation)
      for (; o < nobs && w.obs_lm[o] == l; ++o) {
        const int ip = w.obs_pose[o], ie = w.obs_ext[o], c = w.obs_cam[o];
        if ((int)((unsigned)ip >= unpose) | (int)((unsigned)ie >= unpose) | (int)((unsigned)c >= uncam)) return OKVIS_BA_ERR_ARG;
        // sorted by (landmark, pose, cam); REPEATED (landmark, pose, cam) entries are legal: the reference adds one residual
        // block per matched keypoint (implementation/Estimator.hpp:52-56 only rejects an identical KeypointIdentifier)
        if ((int)(prev_ip > ip) | ((int)(prev_ip == ip) & (int)(prev_c > c))) return OKVIS_BA_ERR_ARG;  // unsorted
        if ((int)(role[ip] == 1) | (int)(role[ie] == 0) | (int)(ip == ie)) return OKVIS_BA_ERR_UNSUPPORTED;
        role[ip] = 0;
        role[ie] = 1;
        ObsRec& R = recs[o];
        R.lm_cam = (uint32_t)l | ((uint32_t)c << 24);
        R.pose = (uint16_t)ip;
        R.ext = (uint16_t)ie;
        R.u = w.obs_uv[2 * o];
        R.v = w.obs_uv[2 * o + 1];
        R.sw = w.obs_sqrtw[o];
        // a landmark's observations are sorted by pose and no block has both roles: a pose-role block is new exactly when the
        // pose index changes; an extrinsics block needs the stamp
        const int newp = (int)(ip != prev_ip) & (int)(pose_off[ip] >= 0);
        ascending &= !(newp & (int)(ip < last));
        bl[nb] = ip;
        nb += newp;
        last = newp ? ip : last;
        prev_ip = ip;
        prev_c = c;
        if (pose_off[ie] >= 0) {
          has_ext = true;
          if (seen[ie] != l) {
            seen[ie] = l;
            if (ie < last) ascending = false;
            bl[nb++] = ie;
            last = ie;
          }
        }
      }
      if (o - o0 > GROUP_OBS) obs_over = true;
      if (!ascending) std::sort(bl, bl + nb);   // (only extrinsics blocks can come out of order)
      lm_pair_begin[l] = npair_run;
      for (int k = 0; k < nb; ++k) {
        const int b = bl[k];
        pl[npair_run] = l;
        pb[npair_run] = b;
        po[npair_run] = pose_off[b];
        pr[npair_run] = role[b];
        ++npair_run;
      }
      // (the earlier version stopped at the first landmark with too many blocks; an unsorted or out-of-range observation
      // further on still comes first, see below)
      if (nb > GROUP_PAIRS) pairs_over = true;
    }
    lm_obs_begin[nlm] = o;
    // an observation the walk did not consume names a landmark out of range or before its predecessor's
    if (o < nobs) return OKVIS_BA_ERR_ARG;
    if (obs_over) return OKVIS_BA_ERR_UNSUPPORTED;
    if (lin2 && has_ext) return BW_LIN2_UNFIT;
    if (pairs_over) return OKVIS_BA_ERR_UNSUPPORTED;
  }
  lm_pair_begin[nlm] = npair_run;
  const int npair = npair_run;
  BW_T("observations + pairs");
  // ---- groups ----
  std::vector<Group>& groups = S.groups;
  groups.clear();
  // okvis_ba_tuning::group_work (sweeps): a group also closes when the block products of its landmark elimination, sum of
  // pairs (pairs + 1) / 2, reach this number.  Measured (profiles/r04_notes.md): a window that has the device to itself finishes
  // sooner with more, lighter groups (cap 250: replay 0.833 -> 0.805 ms per frame for the ten iterations, one configs[1] window
  // 73.1 -> 72.2 us per iteration), with more windows the additional workgroups cost more than they bring (4 windows: 74 -> 78 us per
  // step, 64: 451 k -> 358 k it/s).  Not the default: the other grouping moves the rounding of every single-window run, and one
  // of the ill-conditioned DOGLEG cases that sit at the 1e-6 bound (test_dogleg_rejected_steps) lands at 1.25e-6.
  const long group_work_cap = opt.tuning.group_work > 0 ? opt.tuning.group_work : 0L;
  // Landmarks per group: GROUP_LM (64) is what the kernels hold; the index build fills 32, and 16 when at most GROUP_LM_FEW_WINDOWS
  // windows share the device.  A group of 64 short tracks (landmarks that entered the window with the last frame or two: 2 - 4
  // observations each) is the slowest workgroup of its launch — the landmark elimination loops over the landmarks of the group —
  // and OKVIS hands its landmark ids out in increasing order, so a real window has its short tracks side by side at the end.
  // Measured (profiles/r04_notes.md; windows whose groups close at 256 observations first — configs[1]: 12 landmarks per group —
  // are not touched): one 8-frame window in age order 77.3 us per iteration with 64, 69.0 with 32 (= random order); batches of
  // short-track windows (8 frames, 430 landmarks, 8 observations each), us per step with 64 / 32 / 24 / 16 landmarks per group:
  // 1 window 68.7 / 68.9 / 65.5 / 64.3, 8: 72.1 / 72.3 / 69.0 / 68.6, 64: 124.8 / 125.5 / 115.1 / 121.1, 256: 302 / 304 / 304 / 331;
  // the replay's ten iterations per frame 0.830 (ids by first sighting) / 0.817 / 0.785 / 0.783 ms.  okvis_ba_tuning::group_lm overrides.
  const int group_lm_cap = std::max(1, std::min(opt.tuning.group_lm > 0 ? opt.tuning.group_lm : (n_windows_total <= GROUP_LM_FEW_WINDOWS ? GROUP_LM_FEW : GROUP_LM_DEFAULT), GROUP_LM));
  // piece path (ba_linearize2.hpp): pieces instead of per-observation lists.  A piece = one or two adjacent observations of the same
  // pose inside one row of 16 lanes, greedy from the start of the run (the rule of linearize2_kernel's phase B).  The pieces of a
  // landmark depend on the lane its first observation takes, i.e. on the group it joins, so they are laid out while the groups
  // are formed: lm_piece_begin[l] = the landmark's first piece in the window, pair_piece[p] = the pair's first piece in its group
  // | its number of pieces << 16, per group the pieces before waves 1..3.
  std::vector<int>&lm_piece_begin = S.lm_piece_begin, &pair_piece = S.pair_piece;
  if (lin2) {
    lm_piece_begin.resize((size_t)nlm + 1);
    pair_piece.resize((size_t)npair + 1);   // (one more: where the pieces of fixed poses count)
  }
  {
    int l = 0;
    int piece_total = 0;
    int* const ppc = pair_piece.data();
    while (l < nlm) {
      Group G;
      G.lm_begin = l;
      G.obs_begin = lm_obs_begin[l];
      G.pair_begin = lm_pair_begin[l];
      int no = 0, np = 0, nl = 0, npc = 0;
      int wave_count[4] = {0, 0, 0, 0};
      long work = 0;   // block products of the group's landmark elimination: sum of pairs (pairs + 1) / 2
      while (l < nlm) {
        const int o0 = lm_obs_begin[l], o1 = lm_obs_begin[l + 1], p0 = lm_pair_begin[l], p1 = lm_pair_begin[l + 1];
        const int lo = o1 - o0, lp = p1 - p0;
        if (group_work_cap > 0 && nl > 0 && work + (long)lp * (lp + 1) / 2 > group_work_cap) break;
        if (nl > 0 && (no + lo > GROUP_OBS || np + lp > GROUP_PAIRS || nl + 1 > group_lm_cap)) break;
        int lpc = 0;
        int wc[4] = {0, 0, 0, 0};
        if (lin2) {   // piece path: at most LIN2_PIECES pieces per group (a pair has at least one piece)
          // One walk over the landmark's observations; the tests that depend on the data are arithmetic.  An observation opens a
          // piece when it opens a run (new pose, or first lane of a row) or sits at an even place of its run.
          int pnext = p0, cur = npair, prev = -1, par = 0, local = npc;
          int v = 0;   // first piece | pieces << 16 of the pair the walk is in: kept in a register, stored (never re-read) every step
          int w0 = 0, w1 = 0, w2 = 0, w3 = 0;
          for (int o = o0; o < o1; ++o) {
            const int lane = no + (o - o0), pose = w.obs_pose[o];
            const int changed = pose != prev, freeb = pose_off[pose] >= 0;
            cur = changed ? (freeb ? pnext : npair) : cur;   // (the pairs of a landmark are its free poses in this order)
            pnext += changed & freeb;
            const int start = changed | (int)((lane & 15) == 0);
            par = start ? 0 : par ^ 1;
            const int newp = par == 0;
            v = changed ? 0 : v;   // (every pair is entered once: it starts empty)
            v = (newp & (int)((v >> 16) == 0)) ? local : v;
            v += newp << 16;
            ppc[cur] = v;
            const int wv = lane >> 6;
            w0 += newp & (int)(wv == 0), w1 += newp & (int)(wv == 1), w2 += newp & (int)(wv == 2), w3 += newp & (int)(wv == 3);
            local += newp;
            prev = pose;
          }
          wc[0] = w0, wc[1] = w1, wc[2] = w2, wc[3] = w3;
          lpc = local - npc;
          if (nl == 0 && lpc > LIN2_PIECES) return BW_LIN2_UNFIT;
          if (nl > 0 && npc + lpc > LIN2_PIECES) break;   // (the landmark opens the next group and is laid out again from lane 0)
          lm_piece_begin[l] = piece_total + npc;
          for (int k = 0; k < 4; ++k) wave_count[k] += wc[k];
        }
        work += (long)lp * (lp + 1) / 2;
        no += lo;
        np += lp;
        npc += lpc;
        ++nl;
        ++l;
      }
      G.lm_end = l;
      G.obs_end = lm_obs_begin[l];
      G.pair_end = lm_pair_begin[l];
      G.task_begin = G.task_end = 0;
      G.plist_begin = G.plist_end = G.tlist_begin = G.tlist_end = 0;
      G.piece_begin = piece_total;
      G.pw1 = wave_count[0];
      G.pw2 = wave_count[0] + wave_count[1];
      G.pw3 = wave_count[0] + wave_count[1] + wave_count[2];
      if (!lin2) G.piece_begin = 0;
      piece_total += npc;
      groups.push_back(G);
    }
    if (lin2) lm_piece_begin[nlm] = piece_total;
  }
  const int ngroup = (int)groups.size();
  BW_T("groups");
  // ---- per-pair observation lists, per-group tasks ----
  std::vector<int>& pair_list_begin = S.pair_list_begin;
  pair_list_begin.assign((size_t)npair + 1, 0);
  std::vector<uint16_t>&pair_list = S.pair_list, &task_list = S.task_list;
  std::vector<Task>& tasks = S.tasks;
  pair_list.clear(), task_list.clear(), tasks.clear();
  int gpart_size = 0;
  pair_list.reserve(2 * (size_t)nobs);
  task_list.reserve(3 * (size_t)nobs);
  std::vector<int>& blk_slot = S.blk_slot;              // scratch: block -> pair of the current landmark / task of the group
  blk_slot.assign(npose, -1);
  // scratch: block -> observations of the current group (own role).  Kept between calls (every list is left empty): the lists
  // grow to a few hundred entries each, once, instead of through ten reallocations per block in every upload
  static thread_local std::vector<std::vector<uint16_t>> blk_obs;
  if ((int)blk_obs.size() < npose) blk_obs.resize(npose);
  for (auto& v : blk_obs) v.clear();   // (whatever an interrupted call may have left)
  std::vector<int>& touched = S.touched;
  touched.clear();
  // piece path (ba_linearize2.hpp): pieces instead of per-observation lists
  if (lin2) {
    std::vector<int>& blk_cnt = S.blk_cnt;   // pairs of the current group per block, then the next slot of the block
    blk_cnt.assign(npose, 0);
    for (int g = 0; g < ngroup; ++g) {
      Group& G = groups[g];
      G.tlist_begin = (int)task_list.size();
      // tasks: one per free block seen by the group, ascending; its list = the group-local pairs of that block
      G.task_begin = (int)tasks.size();
      touched.clear();
      for (int p = G.pair_begin; p < G.pair_end; ++p) {
        const int b = pair_block[p];
        if (blk_cnt[b]++ == 0) touched.push_back(b);
      }
      std::sort(touched.begin(), touched.end());
      task_list.resize(task_list.size() + (size_t)(G.pair_end - G.pair_begin));
      int slot = 0;
      for (int b : touched) {
        Task T;
        T.type = 0;
        T.off_a = pose_off[b];
        T.off_b = -1;
        // task_list holds, per group-local pair, the SLOT of its block record: the records of one block are contiguous
        // [list_begin, list_end) in slot order (pairs ascending), which is the order they are summed in
        T.list_begin = G.tlist_begin + slot;
        const int n = blk_cnt[b];
        blk_cnt[b] = slot;   // from here on: the slot the block's next pair takes
        slot += n;
        T.list_end = G.tlist_begin + slot;
        T.out = gpart_size;
        gpart_size += 27;
        tasks.push_back(T);
      }
      for (int p = G.pair_begin; p < G.pair_end; ++p) task_list[(size_t)G.tlist_begin + (p - G.pair_begin)] = (uint16_t)blk_cnt[pair_block[p]]++;
      for (int b : touched) blk_cnt[b] = 0;
      G.task_end = (int)tasks.size();
      G.tlist_end = (int)task_list.size();
    }
  }
  for (int g = 0; g < ngroup && !lin2; ++g) {
    Group& G = groups[g];
    G.plist_begin = (int)pair_list.size();
    G.tlist_begin = (int)task_list.size();
    // per-pair observation lists: one pass over a landmark's observations (two-pass counting fill)
    for (int l = G.lm_begin; l < G.lm_end; ++l) {
      const int p0 = lm_pair_begin[l], p1 = lm_pair_begin[l + 1];
      for (int p = p0; p < p1; ++p) blk_slot[pair_block[p]] = p;
      for (int p = p0; p < p1; ++p) pair_list_begin[p] = 0;
      for (int o = lm_obs_begin[l]; o < lm_obs_begin[l + 1]; ++o) {
        const int pp = blk_slot[w.obs_pose[o]], pe = blk_slot[w.obs_ext[o]];
        if (pp >= 0) ++pair_list_begin[pp];
        if (pe >= 0) ++pair_list_begin[pe];
      }
      int run = (int)pair_list.size();
      for (int p = p0; p < p1; ++p) {
        const int c = pair_list_begin[p];
        pair_list_begin[p] = run;
        run += c;
      }
      const size_t base = pair_list.size();
      pair_list.resize((size_t)run);
      std::vector<int>& cur = touched;   // reuse as the per-pair fill cursor
      cur.assign(p1 - p0, 0);
      for (int o = lm_obs_begin[l]; o < lm_obs_begin[l + 1]; ++o) {
        const int cand[2] = {blk_slot[w.obs_pose[o]], blk_slot[w.obs_ext[o]]};
        for (int c = 0; c < 2; ++c)
          if (cand[c] >= 0) pair_list[(size_t)pair_list_begin[cand[c]] + cur[cand[c] - p0]++] = (uint16_t)(o - G.obs_begin);
      }
      (void)base;
      for (int p = p0; p < p1; ++p) blk_slot[pair_block[p]] = -1;
    }
    G.task_begin = (int)tasks.size();
    touched.clear();
    std::map<std::pair<int, int>, std::vector<uint16_t>> cross;  // (pose, ext) -> obs (only with free extrinsics)
    for (int o = G.obs_begin; o < G.obs_end; ++o) {
      const int ip = w.obs_pose[o], ie = w.obs_ext[o];
      const uint16_t lo = (uint16_t)(o - G.obs_begin);
      const int cand[2] = {ip, ie};
      for (int c = 0; c < 2; ++c)
        if (pose_off[cand[c]] >= 0) {
          if (blk_obs[cand[c]].empty()) touched.push_back(cand[c]);
          blk_obs[cand[c]].push_back(lo);
        }
      if (pose_off[ip] >= 0 && pose_off[ie] >= 0) cross[{ip, ie}].push_back(lo);
    }
    std::sort(touched.begin(), touched.end());   // ascending block index, as a std::map would iterate
    for (int b : touched) {
      Task T;
      T.type = role[b];
      T.off_a = pose_off[b];
      T.off_b = -1;
      T.list_begin = (int)task_list.size();
      task_list.insert(task_list.end(), blk_obs[b].begin(), blk_obs[b].end());
      T.list_end = (int)task_list.size();
      T.out = gpart_size;
      gpart_size += 27;
      tasks.push_back(T);
      blk_obs[b].clear();
    }
    for (auto& kv : cross) {
      Task T;
      T.type = 2;
      T.off_a = pose_off[kv.first.first];
      T.off_b = pose_off[kv.first.second];
      T.list_begin = (int)task_list.size();
      task_list.insert(task_list.end(), kv.second.begin(), kv.second.end());
      T.list_end = (int)task_list.size();
      T.out = gpart_size;
      gpart_size += 36;
      tasks.push_back(T);
    }
    G.task_end = (int)tasks.size();
    G.plist_end = (int)pair_list.size();
    G.tlist_end = (int)task_list.size();
  }
  pair_list_begin[npair] = (int)pair_list.size();
  BW_T("lists+tasks");
  // ---- chunks (Schur workgroups) ----
  std::vector<Chunk>& chunks = S.chunks;
  chunks.clear();
  {
    // landmarks per Schur workgroup: 48 (three staged batches of 16) keeps the workgroup count low when many windows share
    // the device; a few windows have the device to themselves and finish sooner with 32 (measured, tools/gpu_chunk_diag.py:
    // one window 114.7 vs 119.7 us per iteration, 64 windows 239 vs 217)
    int per = std::min(opt.schur_lm_per_block > 0 ? opt.schur_lm_per_block : (n_windows_total <= 8 ? 16 : n_windows_total < SMALL_BATCH_WINDOWS ? 32 : n_windows_total < 128 ? 48 : 64), SCHUR_CHUNK_LM_MAX);   // (round 4 sweep with the matrix-core kernel, 64 windows: 12: 382 k, 24: 436 k, 48: 448 k, 64: 448 k it/s; 256 windows: 48: 585 k, 64: 597 k)
    // fused mode (the linearise workgroup reduces its own group, no Schur launch: DOGLEG and fixed-radius runs): possible when
    // the reduced system is solved in LDS, the pose part is one Schur tile and the reduction's landmark tables fit the observation stage of the linearise kernel;
    // then chunk = group.  options.reserved0 bit 2 keeps the separate launch (A/B switch).
    const int stage = opt.fp32_linearize ? (has_ext ? LinCfg<true, float>::STAGE_DOUBLES : LinCfg<false, float>::STAGE_DOUBLES)
                                         : (has_ext ? LinCfg<true, double>::STAGE_DOUBLES : LinCfg<false, double>::STAGE_DOUBLES);
    H.group_chunks = !(opt.reserved0 & 4) && opt.schur_lm_per_block == 0 && D <= MAX_D_LDS && Dp <= TILE_DIM && n_windows_total <= fused_max_windows(opt) && 2 * SCHUR_LM_BATCH * 3 * Dp <= stage &&
                     (opt.strategy == OKVIS_BA_STRATEGY_DOGLEG || opt.gauss_newton);
    if (H.group_chunks) per = 1;
    {
      const bool no_spec = (opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_DECIDES) != 0, no_mfma = (opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_VALU) != 0;
      H.spec_ok = !no_spec && !no_mfma && opt.schur_lm_per_block == 0 && D <= MAX_D_LDS && !has_ext && std::min(TILE_DIM, Dp) + 1 <= SCH2_MAXT_SMALL_ROWS &&
                  (opt.strategy == OKVIS_BA_STRATEGY_DOGLEG || opt.gauss_newton);
    }
    int g = 0;
    while (g < ngroup) {
      Chunk C;
      C.group_begin = g;
      int nl = 0;
      while (g < ngroup && (nl == 0 || nl + (groups[g].lm_end - groups[g].lm_begin) <= per)) {
        nl += groups[g].lm_end - groups[g].lm_begin;
        ++g;
      }
      C.group_end = g;
      chunks.push_back(C);
    }
    if (chunks.empty()) {  // no landmarks: one empty chunk is not representable; handled by n_chunk = 0
    }
  }
  const int nchunk = (int)chunks.size();
  // ---- per-chunk lists: which per-group partials sum into which pose block / cross block ----
  const int npose_blk_c = Dp / 6;
  std::vector<int>&chunk_diag_begin = S.chunk_diag_begin, &chunk_diag_out = S.chunk_diag_out, &chunk_cross_begin = S.chunk_cross_begin,
                   &chunk_cross = S.chunk_cross;
  chunk_diag_begin.assign((size_t)nchunk * npose_blk_c + 1, 0);
  chunk_cross_begin.assign((size_t)nchunk + 1, 0);
  chunk_cross.clear();
  {
    // per chunk and pose block: the partials (Task::out) of that block in (group, task) order — counted, then placed
    int n_diag = 0;
    for (const Task& T : tasks) n_diag += T.type < 2;
    chunk_diag_out.resize((size_t)n_diag);
    std::vector<int>& cur = S.blk_cursor;
    cur.resize((size_t)npose_blk_c + 1);
    int run = 0;
    for (int c = 0; c < nchunk; ++c) {
      chunk_cross_begin[c] = (int)chunk_cross.size() / 3;
      int* const begin = chunk_diag_begin.data() + (size_t)c * npose_blk_c;   // (zeroed above: counts first)
      const int t0 = groups[chunks[c].group_begin].task_begin, t1 = groups[chunks[c].group_end - 1].task_end;   // (tasks are laid out group by group)
      for (int t = t0; t < t1; ++t) {
        const Task& T = tasks[t];
        if (T.type < 2) {
          ++begin[T.off_a / 6];
        } else {
          chunk_cross.push_back(T.off_a);
          chunk_cross.push_back(T.off_b);
          chunk_cross.push_back(T.out);
        }
      }
      for (int bkk = 0; bkk < npose_blk_c; ++bkk) {
        const int n = begin[bkk];
        begin[bkk] = cur[bkk] = run;
        run += n;
      }
      for (int t = t0; t < t1; ++t) {
        const Task& T = tasks[t];
        if (T.type < 2) chunk_diag_out[(size_t)cur[T.off_a / 6]++] = T.out;
      }
    }
  }
  chunk_diag_begin[(size_t)nchunk * npose_blk_c] = (int)chunk_diag_out.size();
  chunk_cross_begin[nchunk] = (int)chunk_cross.size() / 3;
  // chunk descriptors of the matrix-core Schur kernel: one record instead of the chain chunks -> groups -> lm_pair_begin
  std::vector<int>& chunk_desc = S.chunk_desc;
  chunk_desc.assign((size_t)nchunk * SCHUR_DESC_INTS, 0);
  for (int c = 0; c < nchunk; ++c) {
    int* d = chunk_desc.data() + (size_t)c * SCHUR_DESC_INTS;
    const int lb = groups[chunks[c].group_begin].lm_begin, le = groups[chunks[c].group_end - 1].lm_end;
    d[0] = lb;
    d[1] = le;
    for (int i = 0; i <= SCHUR_CHUNK_LM_MAX / 4; ++i) d[2 + i] = lm_pair_begin[std::min(lb + 4 * i, le)];
  }
  BW_T("chunks");
  // ---- greedy colouring of the IMU factors: factors of one colour share no parameter block ----
  std::vector<int> imu_color(w.n_imu, 0);
  int n_imu_color = 0;
  for (int f = 0; f < w.n_imu; ++f) {
    int col = 0;
    for (;; ++col) {
      bool clash = false;
      for (int g2 = 0; g2 < f && !clash; ++g2) {
        if (imu_color[g2] != col) continue;
        const int pa[2] = {w.imu_pose0[f], w.imu_pose1[f]}, pb[2] = {w.imu_pose0[g2], w.imu_pose1[g2]};
        const int sa[2] = {w.imu_sb0[f], w.imu_sb1[f]}, sbb[2] = {w.imu_sb0[g2], w.imu_sb1[g2]};
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j) clash = clash || pa[i] == pb[j] || sa[i] == sbb[j];
      }
      if (!clash) break;
    }
    imu_color[f] = col;
    n_imu_color = std::max(n_imu_color, col + 1);
  }
  std::vector<int> imu_order, imu_color_begin(n_imu_color + 1, 0), imu_coloff(30 * (size_t)w.n_imu, -1);
  for (int c = 0; c < n_imu_color; ++c) {
    imu_color_begin[c] = (int)imu_order.size();
    for (int f = 0; f < w.n_imu; ++f)
      if (imu_color[f] == c) imu_order.push_back(f);
  }
  imu_color_begin[n_imu_color] = (int)imu_order.size();
  for (int f = 0; f < w.n_imu; ++f) {
    const int offs[4] = {pose_off[w.imu_pose0[f]], sb_off[w.imu_sb0[f]], pose_off[w.imu_pose1[f]], sb_off[w.imu_sb1[f]]};
    const int start[4] = {0, 6, 15, 21}, dims[4] = {6, 9, 6, 9};
    for (int b = 0; b < 4; ++b)
      for (int k = 0; k < dims[b]; ++k) imu_coloff[30 * (size_t)f + start[b] + k] = offs[b] < 0 ? -1 : offs[b] + k;
  }
  BW_T("imu colouring");
  const int npose_blk = Dp / 6;
  const int spart_stride = npose_blk * (npose_blk + 1) / 2 * 36 + 3 * Dp;
  const int ntile = std::max(1, (Dp / 6 + SCHUR_TILE_BLOCKS - 1) / SCHUR_TILE_BLOCKS);
  // ---- IMU ----
  for (int f = 0; f < w.n_imu; ++f) {
    if (w.imu_pose0[f] < 0 || w.imu_pose0[f] >= npose || w.imu_pose1[f] < 0 || w.imu_pose1[f] >= npose ||
        w.imu_sb0[f] < 0 || w.imu_sb0[f] >= nsb || w.imu_sb1[f] < 0 || w.imu_sb1[f] >= nsb)
      return OKVIS_BA_ERR_ARG;
    if (w.imu_s_begin[f] < 0 || w.imu_s_count[f] < 2 || w.imu_s_begin[f] + w.imu_s_count[f] > w.n_imu_samples)
      return OKVIS_BA_ERR_ARG;
    if (w.imu_s_count[f] > MAX_IMU_SAMPLES) return OKVIS_BA_ERR_UNSUPPORTED;
    // ImuError::redoPreintegration returns -1 when the samples do not cover [t0,t1] (ImuError.cpp:87-89)
    if (!(w.imu_s_t[w.imu_s_begin[f] + w.imu_s_count[f] - 1] >= w.imu_t1[f])) return OKVIS_BA_ERR_ARG;
  }
  BW_T("imu checks");
  // ---- marginalisation prior: H0 = J^T J ----
  const int Dm = w.marg_dim;
  if (Dm < 0 || Dm > MAX_MARG_DIM) return OKVIS_BA_ERR_UNSUPPORTED;
  std::vector<double> H0((size_t)Dm * Dm, 0.0);
  if (Dm > 0) {
    if (!w.marg_J || !w.marg_e0 || !w.marg_lin || !w.marg_block_type || !w.marg_block_idx || !w.marg_block_off ||
        w.marg_nblocks <= 0)
      return OKVIS_BA_ERR_ARG;
    for (int b = 0; b < w.marg_nblocks; ++b) {
      const int lim = w.marg_block_type[b] == OKVIS_BA_BLOCK_POSE ? npose : nsb;
      if (w.marg_block_idx[b] < 0 || w.marg_block_idx[b] >= lim || w.marg_block_off[b] < 0 ||
          w.marg_block_off[b] + (w.marg_block_type[b] == OKVIS_BA_BLOCK_POSE ? 6 : 9) > Dm)
        return OKVIS_BA_ERR_ARG;
      if (b > 0 && w.marg_block_off[b] <= w.marg_block_off[b - 1]) return OKVIS_BA_ERR_ARG;
    }
    // a prior of more than H0_DEVICE_MIN rows: the O(Dm^3) product is left to the device (marg_h0_kernel, launched behind the
    // upload: the same sums in the same order)
    H.h0_on_device = Dm > H0_DEVICE_MIN && !(opt.tuning.flags & OKVIS_BA_TUNE_H0_ON_HOST);   // (the switch: A/B test of the two)
    // upper triangle as a sum of row outer products: every entry still adds its terms in row order (same value as the
    // column-by-column dot products), but the inner loop runs along a row of J (contiguous: 10 us -> 3 us at 45 rows)
    if (!H.h0_on_device) marg_h0_host(w.marg_J, Dm, H0.data());
  }
  const int nmb = Dm > 0 ? w.marg_nblocks : 0;

  // ---- fill sizes ----
  P.n_pose = npose; P.n_sb = nsb; P.n_lm = nlm; P.n_cam = w.n_cam; P.n_obs = nobs; P.n_imu = w.n_imu;
  P.n_pprior = w.n_pprior; P.n_sbprior = w.n_sbprior; P.n_rel = w.n_relpose;
  P.marg_dim = Dm; P.marg_nb = nmb;
  P.D = D; P.Dp = Dp; P.n_pair = npair; P.n_group = ngroup; P.n_chunk = nchunk; P.n_task = (int)tasks.size();
  P.has_ext = has_ext ? 1 : 0;
  P.lin2 = lin2 ? 1 : 0;
  // ---- chain solver (ba_chain.hpp): the speed/bias blocks must couple only to their neighbours in the order of the reduced
  //      system (an IMU term links consecutive blocks; a marginalisation prior may span two adjacent ones) — what a sliding
  //      window's blocks do, in time order.  Windows solved in HBM (D > MAX_D_LDS) are not concerned.
  bool use_chain = false;
  if (D <= MAX_D_LDS) {
    int min_blocks = 1;
    (void)want_chain(opt, &min_blocks);
    const int Ks = (D - Dp) / 9;
    bool fits = Dp >= 6 && Ks >= min_blocks && Ks <= CH_MAX_KS && LChain::tiles_fit(Dp) &&
                (int)solve_smem_chain(LChain::make(D, Dp).total, ((D + 5) / 6) * 6) <= SOLVE_LDS_LIMIT_CHAIN;
    auto rank = [&](int b) { return sb_off[b] < 0 ? -1 : (sb_off[b] - Dp) / 9; };
    for (int f = 0; f < w.n_imu && fits; ++f) {
      const int r0 = rank(w.imu_sb0[f]), r1 = rank(w.imu_sb1[f]);
      if (r0 >= 0 && r1 >= 0 && r0 != r1 + 1 && r1 != r0 + 1) fits = false;
    }
    int lo = INT_MAX, hi = -1;
    for (int b = 0; b < nmb && fits; ++b)
      if (w.marg_block_type[b] != OKVIS_BA_BLOCK_POSE) {
        const int r = rank(w.marg_block_idx[b]);
        if (r >= 0) lo = std::min(lo, r), hi = std::max(hi, r);
      }
    if (hi - lo > 1) fits = false;
    if (chain && !fits) return BW_CHAIN_UNFIT;
    use_chain = chain;
  }
  H.chain = use_chain;
  P.chain = use_chain ? (D - Dp) / 9 : 0;
  {
    // Diagonal blocks of the dense solver that carry a pose prior or the marginalisation prior: information of a few directions
    // that is orders of magnitude above everything else in the block (the yaw prior of the first pose: 1e16 n n^T across three
    // rotation rows), whose elimination cancels the leading digits of the block.  ba_ldl16.hpp eliminates them with compensated
    // products (profiles/r05_notes.md, "the referee").  The solver numbers the speed/bias part first (L16::perm).
    unsigned m = 0;
    if (D <= MAX_D_LDS && !(opt.tuning.flags & OKVIS_BA_TUNE_NO_LDL_COMP)) {   // (the tiled solver of larger systems: not compensated, ba_chol_tiles.hpp)
      // (chain solver: the blocks of the POSE system, which ldl16_solve factorises on its own; the speed/bias blocks are
      // eliminated 9 x 9, in the order of their rows, without the blocked solver's row / column asymmetry)
      const L16 LY = use_chain ? L16{ldl16_nb(Dp), 0, Dp} : L16{ldl16_nb(D), D - Dp, D};
      auto mark = [&](int off, int n) {
        if (off >= 0 && !(use_chain && off >= Dp))
          for (int k = 0; k < n; ++k) m |= 1u << (LY.perm(off + k) >> 4);
      };
      for (int i = 0; i < w.n_pprior; ++i) mark(pose_off[w.pprior_pose[i]], 6);
      for (int b = 0; b < nmb; ++b) {
        const bool pose = w.marg_block_type[b] == OKVIS_BA_BLOCK_POSE;
        mark(pose ? pose_off[w.marg_block_idx[b]] : sb_off[w.marg_block_idx[b]], pose ? 6 : 9);
      }
      if (opt.tuning.flags & OKVIS_BA_TUNE_LDL_COMP_ALL) m = ~0u;
    }
    P.ldl_comp = m;
  }
  P.gpart_size = gpart_size;
  P.n_tile = ntile;
  P.n_imu_color = n_imu_color;
  P.spart_stride = spart_stride;
  P.spart_buf_stride = (H.group_chunks || H.spec_ok) ? std::max(nchunk, 1) * spart_stride : 0;
  {
    int max_tasks = 0, max_pairs = 0;
    for (const Group& Gq : groups) {
      max_tasks = std::max(max_tasks, Gq.task_end - Gq.task_begin);
      max_pairs = std::max(max_pairs, Gq.pair_end - Gq.pair_begin);
    }
    const int stage = opt.fp32_linearize ? LinCfg<false, float>::STAGE_DOUBLES : LinCfg<false, double>::STAGE_DOUBLES;
    P.fuse_fast = H.group_chunks && !has_ext && max_tasks <= FUSE_MAX_TASKS && (lin2 || 6 * max_pairs <= FUSE_WIT * LIN_THREADS) &&
                  fuse_nlb(Dp, stage) >= 4;
  }
  P.cauchy_b = w.cauchy_b;
  P.imu.sigma_g_c = w.imu_params.sigma_g_c; P.imu.sigma_a_c = w.imu_params.sigma_a_c;
  P.imu.sigma_gw_c = w.imu_params.sigma_gw_c; P.imu.sigma_aw_c = w.imu_params.sigma_aw_c;
  P.imu.g = w.imu_params.g; P.imu.g_max = w.imu_params.g_max; P.imu.a_max = w.imu_params.a_max;

  BW_T("prior + sizes");
  // ---- arena: index lists (the state arrays and the observation records are in place, see above) ----
  OFF(groups, put(A, groups));
  OFF(pair_lm, put_n(A, pair_lm.data(), (size_t)npair));
  OFF(pair_off, put_n(A, pair_off.data(), (size_t)npair));
  OFF(pair_role, put_n(A, pair_role.data(), (size_t)npair));
  OFF(pair_list_begin, put(A, pair_list_begin));
  OFF(pair_list, put(A, pair_list));
  OFF(lm_pair_begin, put(A, lm_pair_begin));
  OFF(lm_obs_begin, put(A, lm_obs_begin));
  OFF(lm_piece_begin, put_n(A, lm_piece_begin.data(), lin2 ? (size_t)nlm + 1 : 0));
  OFF(pair_piece, put_n(A, pair_piece.data(), lin2 ? (size_t)npair : 0));
  OFF(pair_block, put_n(A, pair_block.data(), lin2 ? (size_t)npair : 0));
  OFF(tasks, put(A, tasks));
  OFF(task_list, put(A, task_list));
  OFF(chunks, put(A, chunks));
  OFF(chunk_diag_begin, put(A, chunk_diag_begin));
  OFF(chunk_diag_out, put(A, chunk_diag_out));
  OFF(chunk_cross_begin, put(A, chunk_cross_begin));
  OFF(chunk_cross, put(A, chunk_cross));
  OFF(chunk_desc, put(A, chunk_desc));
  {
    // several Schur tiles per dimension (pose part beyond 96 rows): where the (landmark, block) pairs of every tile start, so that
    // a tile pair's workgroup touches its own pairs only (the pairs of a landmark are sorted by block)
    std::vector<int> lm_tile_begin;
    if (ntile > 1) {
      lm_tile_begin.resize((size_t)nlm * (ntile + 1));
      for (int l = 0; l < nlm; ++l) {
        int p = lm_pair_begin[l];
        for (int t = 0; t <= ntile; ++t) {
          while (p < lm_pair_begin[l + 1] && pair_off[p] < t * SCHUR_TILE_BLOCKS * 6) ++p;
          lm_tile_begin[(size_t)l * (ntile + 1) + t] = p;
        }
      }
    }
    OFF(lm_tile_begin, put(A, lm_tile_begin));
  }
  OFF(imu_order, put(A, imu_order));
  OFF(imu_color_begin, put(A, imu_color_begin));
  OFF(imu_coloff, put(A, imu_coloff));
  BW_T("arena: arrays");
  {
    // destination of every entry of the IMU factors' H (30x30 lower, packed) | g records in the solve kernel's
    // matrix layout (SLayout, ba_solve.hpp), so that the kernel can prefetch value + destination in one round trip
    // (the table only depends on where the terms' blocks sit in the reduced system: a window that slides keeps it from frame to
    // frame, so the last one is kept — 9 us of a 70 us upload)
    struct ImuAsmCache {
      int D = -1, Dp = -1, chain = -1;
      std::vector<int> coloff, color;
      std::vector<int4> table;
      std::vector<int> fastw, pos;
    };
    // (four entries, replaced in turn: the estimator alternates between the window it optimises and the sub-window it marginalises)
    static thread_local ImuAsmCache asm_caches[4];
    static thread_local int asm_next = 0;
    int hit = -1;
    for (int k = 0; k < 4 && hit < 0; ++k)
      if (asm_caches[k].D == D && asm_caches[k].Dp == Dp && asm_caches[k].chain == (int)use_chain && asm_caches[k].coloff == imu_coloff &&
          asm_caches[k].color == imu_color)
        hit = k;
    const bool asm_hit = hit >= 0;
    ImuAsmCache& asm_cache = asm_caches[asm_hit ? hit : asm_next];
    if (!asm_hit) asm_next = (asm_next + 1) & 3;
    std::vector<int4>& imu_asm = asm_cache.table;
    std::vector<int>& imu_fastw = asm_cache.fastw;
    std::vector<int>& imu_pos = asm_cache.pos;
    if (!asm_hit) {
      asm_cache.D = D;
      asm_cache.Dp = Dp;
      asm_cache.chain = (int)use_chain;
      asm_cache.coloff = imu_coloff;
      asm_cache.color = imu_color;
      const int nbk = (D + 5) / 6;   // (the HBM matrix of the large windows has the same block layout)
      auto at = [&](int i, int j) {
        const int bi = i / 6, bj = j / 6;
        return (bj * nbk - (bj * (bj - 1)) / 2 + (bi - bj)) * SBS + (i - 6 * bi) * 6 + (j - 6 * bj);
      };
      // the same entry in the LDS layout of the LDL^T solver (L16::at, ba_ldl16.hpp): i >= j, stored at the mirrored position
      const L16 ly16{ldl16_nb(D), D - Dp, D};   // (the solver's ordering: speed/bias part first)
      const LChain lych = LChain::make(D, Dp);  // (chain solver: ba_chain.hpp)
      auto at16 = [&](int i, int j) { return use_chain ? lych.at(i, j) : ly16.at(i, j); };
      imu_asm.assign(512 * (size_t)w.n_imu, make_int4(-1, -1, 0, 0));
      imu_fastw.assign(512 * (size_t)w.n_imu, -1);
      // (the solve kernel's dynamic LDS: matrix area, then rhs, gradient, diagonal, solution — Dpad doubles each, ba_solve.hpp)
      const bool lds_system = D <= MAX_D_LDS;
      const int goff16 = lds_system ? (use_chain ? lych.total : ldl16_area_doubles(D)) + ((D + 5) / 6) * 6 : 0;
      // Where the entries of a factor's H | g record sit in the record (imu_pos, read by the factor workgroup that writes it): for a
      // system solved in LDS in the order of their places there, so that the lanes of a wave of the solve kernel — consecutive
      // record entries — add to ascending, mostly consecutive LDS addresses.  (In the packed order of the triangle the 64 entries
      // of a wave landed on one bank pair — a row of a 16x16 block lies 128 bytes behind the previous one — and the scatter
      // took 4 us.)  Windows solved in HBM keep the packed order.
      imu_pos.assign(512 * (size_t)w.n_imu, 0);
      std::vector<std::pair<int, int>> keys(495);
      for (int f = 0; f < w.n_imu; ++f) {
        const int* co = imu_coloff.data() + 30 * (size_t)f;
        int e = 0;
        for (int a = 0; a < 30; ++a)
          for (int b = 0; b <= a; ++b, ++e) {
            const int ra = co[a], rb = co[b];
            keys[e] = {(ra < 0 || rb < 0) ? INT_MAX : (lds_system ? at16(ra, rb) : e), e};
          }
        for (int a = 0; a < 30; ++a) keys[465 + a] = {co[a] < 0 ? INT_MAX : (lds_system ? goff16 + co[a] : 465 + a), 465 + a};
        if (lds_system) std::sort(keys.begin(), keys.end());
        int* pos = imu_pos.data() + 512 * (size_t)f;
        for (int rank = 0; rank < 495; ++rank) pos[keys[rank].second] = rank;
        for (int k = 495; k < 512; ++k) pos[k] = k;
        e = 0;
        for (int a = 0; a < 30; ++a)
          for (int b = 0; b <= a; ++b, ++e) {
            const int ra = co[a], rb = co[b];
            if (ra < 0 || rb < 0) continue;
            const size_t at_rec = 512 * (size_t)f + pos[e];
            imu_asm[at_rec] = make_int4((ra >= rb ? at(ra, rb) : at(rb, ra)) | (imu_color[f] << 24), a == b ? ra : -1,
                                        ra >= rb ? (ra << 16 | rb) : (rb << 16 | ra),   // z: the reduced indices
                                        0);
            if (lds_system && imu_color[f] < 16) imu_fastw[at_rec] = at16(ra, rb) | ((a == b ? ra + 1 : 0) << 16) | (imu_color[f] << 24);
          }
        for (int a = 0; a < 30; ++a)
          if (co[a] >= 0) {
            const size_t at_rec = 512 * (size_t)f + pos[465 + a];
            imu_asm[at_rec] = make_int4(co[a] | (1 << 20) | (imu_color[f] << 24), -1, 0, 0);
            if (lds_system && imu_color[f] < 16) imu_fastw[at_rec] = (goff16 + co[a]) | (imu_color[f] << 24);
          }
      }
    }
    if (imu_asm.empty()) OFF(imu_asm, put(A, std::vector<int4>(1, make_int4(-1, -1, 0, 0))));
    else OFF(imu_asm, put(A, imu_asm));
    if (imu_fastw.empty()) OFF(imu_fastw, put(A, std::vector<int>(1, -1)));
    else OFF(imu_fastw, put(A, imu_fastw));
    if (imu_pos.empty()) OFF(imu_pos, put(A, std::vector<int>(1, 0)));
    else OFF(imu_pos, put(A, imu_pos));
    // large windows: the reverse map, so that the tile export (many workgroups) gathers the IMU contributions instead of one
    // workgroup scattering them into HBM.  At most two factors meet in one entry (the chain couples consecutive states).
    std::vector<int2> imu_rev;
    if (D > MAX_D_LDS) {
      const size_t nbk = (D + 5) / 6;
      imu_rev.assign(nbk * (nbk + 1) / 2 * SBS, make_int2(-1, -1));
      for (size_t idx = 0; idx < 512 * (size_t)w.n_imu; ++idx) {
        const int4 d = imu_asm[idx];
        if (d.x < 0 || (d.x & (1 << 20))) continue;   // nothing / an entry of g
        int2& r = imu_rev[d.x & 0xFFFFF];
        if (r.x < 0) r.x = (int)idx;
        else if (r.y < 0) r.y = (int)idx;
        else return OKVIS_BA_ERR_UNSUPPORTED;   // three IMU factors on one block: not a chain
      }
    }
    if (imu_rev.empty()) imu_rev.push_back(make_int2(-1, -1));
    OFF(imu_rev, put(A, imu_rev));
  }
  BW_T("arena: imu tables");
  for (int b = 0; b < 2; ++b) {
    OFF(V[b], put_zero(A, 48 * (size_t)nlm));
    OFF(bl[b], put_zero(A, 24 * (size_t)nlm));
    OFF(Hq[b], put_zero(A, 48 * (size_t)nlm));
    OFF(W[b], put_zero(A, 144 * (size_t)npair));
    OFF(gpart[b], put_zero(A, 8 * (size_t)gpart_size));
    OFF(gscal[b], put_zero(A, 8 * (size_t)GS_COUNT * ngroup));
    OFF(imu_lin[b], put_zero(A, 8 * (size_t)IMU_LIN_STRIDE * w.n_imu));
    OFF(pp_lin[b], put_zero(A, 8 * 42 * (size_t)w.n_pprior));
    OFF(sbp_lin[b], put_zero(A, 8 * 9 * (size_t)w.n_sbprior));
    OFF(rel_lin[b], put_zero(A, 8 * 78 * (size_t)w.n_relpose));
    OFF(marg_lin_e[b], put_zero(A, 8 * 2 * (size_t)Dm));
    OFF(marg_lin_M[b], put_zero(A, 8 * 9 * (size_t)nmb));
    OFF(small_cost[b], put_zero(A, 16));
    if (opt.debug_arrays) OFF(obs_r[b], put_zero(A, 16 * (size_t)nobs));
  }
  OFF(spart, put_zero(A, 8 * (size_t)std::max(nchunk, 1) * (size_t)spart_stride * ((H.group_chunks || H.spec_ok) ? 2 : 1)));
  OFF(spart_sum, put_zero(A, 8 * (size_t)std::max(spart_stride, 1)));
  OFF(sum_sync, put_zero(A, 16));
  OFF(dec, put_zero(A, 8 * (size_t)DEC_COUNT));
  if (opt.debug_arrays) {
    OFF(S, put_zero(A, 8 * (size_t)D * D));
    OFF(rhs, put_zero(A, 8 * (size_t)D));
    OFF(Dp2, put_zero(A, 8 * (size_t)D));
  }
  if (D > MAX_D_LDS) {
    const size_t nbk = (D + 5) / 6;
    OFF(Sg, put_zero(A, 8 * nbk * (nbk + 1) / 2 * 38));
    const size_t nT = (D + CT_TB - 1) / CT_TB, ntile = nT * (nT + 1) / 2;
    P.ct_nT = (int)nT;
    OFF(ct_T, put_zero(A, 8 * ntile * CT_TILE));
    OFF(ct_Linv, put_zero(A, 8 * nT * CT_TILE));
    OFF(ct_rhs, put_zero(A, 8 * nT * CT_TB));
    OFF(ct_y, put_zero(A, 8 * nT * CT_TB));
    OFF(ct_x, put_zero(A, 8 * nT * CT_TB));
    OFF(ct_flag, put_zero(A, sizeof(int) * (ntile + 4 + 2 * nT)));
    OFF(ct_g, put_zero(A, 8 * nT * CT_TB));
    OFF(ct_d2, put_zero(A, 8 * nT * CT_TB));
  }
  OFF(step, put_zero(A, 8 * (size_t)D));
  OFF(scale_p, put_zero(A, 8 * (size_t)D));
  OFF(lm_scale, put_zero(A, 24 * (size_t)nlm));
  OFF(grad, put_zero(A, 8 * (size_t)D));
  OFF(quality, put_zero(A, 8 * (size_t)nlm));
  OFF(results, put_zero(A, results_bytes(npose, nsb, nlm, w.n_imu) + 8));
  if (opt.debug_arrays) OFF(prof, put_zero(A, 8 * (64 + 4 * 160)));   // clock64() phase stamps + tile task timeline: diagnostics only
  OFF(ctrl, put_zero(A, sizeof(Ctrl)));
  OFF(imu_pose0, put_n(A, w.imu_pose0, (size_t)w.n_imu));
  OFF(imu_sb0, put_n(A, w.imu_sb0, (size_t)w.n_imu));
  OFF(imu_pose1, put_n(A, w.imu_pose1, (size_t)w.n_imu));
  OFF(imu_sb1, put_n(A, w.imu_sb1, (size_t)w.n_imu));
  {
    std::vector<long long> t0(w.n_imu), t1(w.n_imu), st(w.n_imu ? w.n_imu_samples : 0);
    for (int f = 0; f < w.n_imu; ++f) {
      t0[f] = w.imu_t0[f];
      t1[f] = w.imu_t1[f];
    }
    for (size_t i = 0; i < st.size(); ++i) st[i] = w.imu_s_t[i];
    OFF(imu_t0, put(A, t0));
    OFF(imu_t1, put(A, t1));
    OFF(imu_s_t, put(A, st));
  }
  OFF(imu_s_begin, put_n(A, w.imu_s_begin, (size_t)w.n_imu));
  OFF(imu_s_count, put_n(A, w.imu_s_count, (size_t)w.n_imu));
  OFF(imu_s_gyr, put_n(A, w.imu_s_gyr, w.n_imu ? 3 * (size_t)w.n_imu_samples : 0));
  OFF(imu_s_acc, put_n(A, w.imu_s_acc, w.n_imu ? 3 * (size_t)w.n_imu_samples : 0));
  {
    std::vector<ImuCacheD> caches((size_t)w.n_imu);
    if (!caches.empty()) std::memset(caches.data(), 0, sizeof(ImuCacheD) * caches.size());
    if (w.imu_sb_ref && w.imu_sb_ref_valid) {
      for (int f = 0; f < w.n_imu; ++f) {
        // (the rule of WindowStore::assign: a window a patchable solver refuses is refused here as well)
        if (w.imu_sb_ref_valid[f] > 2 || (w.imu_sb_ref_valid[f] == 2 && !w.imu_cache)) return OKVIS_BA_ERR_ARG;
        if (w.imu_sb_ref_valid[f] == 2 && w.imu_cache) {
          // the preintegration itself (okvis_ba_fetch_imu_caches): valid as it stands, nothing is rebuilt on first use
          std::memcpy(&caches[f], w.imu_cache + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * f, sizeof(ImuCacheD));
          if (caches[f].valid != 1) return OKVIS_BA_ERR_ARG;   // (a record of a term that was never evaluated)
          for (int k = 0; k < 225; ++k)
            if (!std::isfinite(caches[f].sqrt_info[k])) return OKVIS_BA_ERR_ARG;
          for (int k = 0; k < 4; ++k)
            if (!std::isfinite(caches[f].Delta_q[k])) return OKVIS_BA_ERR_ARG;
          caches[f].redo_count = 0;
        } else if (w.imu_sb_ref_valid[f]) {
          caches[f].valid = 2;
          for (int k = 0; k < 9; ++k) caches[f].sb_ref[k] = w.imu_sb_ref[9 * (size_t)f + k];
        }
      }
    }
    OFF(imu_cache, put(A, caches));
    OFF(imu_cache_prev, put_zero(A, sizeof(ImuCacheD) * (size_t)w.n_imu));
  }
  OFF(pprior_pose, put_n(A, w.pprior_pose, (size_t)w.n_pprior));
  OFF(pprior_meas, put_n(A, w.pprior_meas, 7 * (size_t)w.n_pprior));
  OFF(pprior_sqrtinfo, put_n(A, w.pprior_sqrtinfo, 36 * (size_t)w.n_pprior));
  OFF(sbprior_sb, put_n(A, w.sbprior_sb, (size_t)w.n_sbprior));
  OFF(sbprior_meas, put_n(A, w.sbprior_meas, 9 * (size_t)w.n_sbprior));
  OFF(sbprior_sqrtinfo, put_n(A, w.sbprior_sqrtinfo, 81 * (size_t)w.n_sbprior));
  OFF(rel_pose0, put_n(A, w.rel_pose0, (size_t)w.n_relpose));
  OFF(rel_pose1, put_n(A, w.rel_pose1, (size_t)w.n_relpose));
  OFF(rel_sqrtinfo, put_n(A, w.rel_sqrtinfo, 36 * (size_t)w.n_relpose));
  OFF(marg_block_type, put_n(A, w.marg_block_type, (size_t)nmb));
  OFF(marg_block_idx, put_n(A, w.marg_block_idx, (size_t)nmb));
  OFF(marg_block_off, put_n(A, w.marg_block_off, (size_t)nmb));
  OFF(marg_J, put_n(A, w.marg_J, (size_t)Dm * Dm));
  OFF(marg_H0, put(A, H0));
  OFF(marg_e0, put_n(A, w.marg_e0, (size_t)Dm));
  OFF(marg_lin, put_n(A, w.marg_lin, 9 * (size_t)nmb));
  for (int i = 0; i < w.n_pprior; ++i)
    if (w.pprior_pose[i] < 0 || w.pprior_pose[i] >= npose) return OKVIS_BA_ERR_ARG;
  for (int i = 0; i < w.n_sbprior; ++i)
    if (w.sbprior_sb[i] < 0 || w.sbprior_sb[i] >= nsb) return OKVIS_BA_ERR_ARG;
  for (int i = 0; i < w.n_relpose; ++i)
    if (w.rel_pose0[i] < 0 || w.rel_pose0[i] >= npose || w.rel_pose1[i] < 0 || w.rel_pose1[i] >= npose) return OKVIS_BA_ERR_ARG;
  {
    // where the columns of the pose / speed-bias priors sit in the reduced system (the solve kernel used to look this up
    // through two dependent loads per column)
    std::vector<int> prior_col(6 * (size_t)w.n_pprior + 9 * (size_t)w.n_sbprior + 1, -1);
    for (int i = 0; i < w.n_pprior; ++i) {
      const int off = pose_off[w.pprior_pose[i]];
      for (int k = 0; k < 6; ++k) prior_col[6 * (size_t)i + k] = off < 0 ? -1 : off + k;
    }
    for (int i = 0; i < w.n_sbprior; ++i) {
      const int off = sb_off[w.sbprior_sb[i]];
      for (int k = 0; k < 9; ++k) prior_col[6 * (size_t)w.n_pprior + 9 * (size_t)i + k] = off < 0 ? -1 : off + k;
    }
    OFF(prior_col, put(A, prior_col));
  }

  H.n_pose = npose; H.n_sb = nsb; H.n_lm = nlm; H.n_obs = nobs; H.n_imu = w.n_imu; H.D = D; H.Dp = Dp;
  H.pose_off = pose_off; H.sb_off = sb_off; H.marg_dim = Dm;
  H.n_pair = npair; H.n_group = ngroup; H.n_chunk = nchunk;
  H.pair_lm.assign(pair_lm.data(), pair_lm.data() + npair);
  H.pair_block.assign(pair_block.data(), pair_block.data() + npair);
  H.acc = 0;
  BW_T("arena");
  // ---- algorithmic (compulsory) bytes per iteration, DESIGN.md §4 ----
  const int64_t O = nobs, L = nlm, Pn = npair;
  H.bytes_lin = 32 * O + 56 * (int64_t)npose + 32 * L + 8 * (int64_t)D + 72 * L + 144 * Pn  // reads
                + 32 * L + 120 * L + 144 * Pn + 8 * (int64_t)gpart_size + 8 * GS_COUNT * (int64_t)ngroup;  // writes
  H.bytes_schur = 72 * L + 144 * Pn + 8 * (int64_t)nchunk * spart_stride;
  H.bytes_solve = 8 * (int64_t)nchunk * spart_stride + 8 * (int64_t)gpart_size + 8 * (int64_t)IMU_LIN_STRIDE * w.n_imu +
                  8 * (int64_t)Dm * Dm + 8 * (int64_t)D + 2 * (56 * (int64_t)npose + 72 * (int64_t)nsb);
  H.bytes_small = (int64_t)w.n_imu * (8 * IMU_LIN_STRIDE + (int64_t)sizeof(ImuCacheD) + 2 * 56 + 2 * 72) +
                  8 * (int64_t)Dm * Dm * 2;
  return OKVIS_BA_OK;
}

// convert the arena offsets stored in the pointer fields to device addresses
void relocate(WinPtrs& P, unsigned char* base, unsigned char* zbase, int debug) {
  unsigned char** fields = reinterpret_cast<unsigned char**>(&P.pose[0]);
  // all pointer members are laid out contiguously from pose[0] to marg_lin; relocate by scanning the
  // struct region as an array of pointers (sizes/scalars precede pose[0])
  const size_t first = offsetof(WinPtrs, pose);
  const size_t n = (offsetof(WinPtrs, marg_lin) + sizeof(void*) - first) / sizeof(void*);   // (the record ends with alignment padding)
  for (size_t i = 0; i < n; ++i) {
    const size_t off = reinterpret_cast<size_t>(fields[i]);
    fields[i] = (off & ARENA_ZFLAG) ? zbase + (off & ~ARENA_ZFLAG) : base + off;
  }
  // optional arrays that were never allocated hold offset 0 -> must be null
  if (debug != 1) {   // 2 = phase stamps only: the copies of the linearisation / system would distort the stamps
    P.obs_r[0] = P.obs_r[1] = nullptr;
    P.S = nullptr;
    P.rhs = nullptr;
    P.Dp2 = nullptr;
  }
  if (!debug) P.prof = nullptr;
  P.Hpp = nullptr;
  if (P.D <= MAX_D_LDS) {
    P.Sg = nullptr;
    P.ct_T = P.ct_Linv = P.ct_rhs = P.ct_y = P.ct_x = P.ct_g = P.ct_d2 = nullptr;
    P.ct_flag = nullptr;
  }
}

size_t lin_smem(bool ext, bool f32 = false) {
  const int d = f32 ? (ext ? LinCfg<true, float>::SMEM_DOUBLES : LinCfg<false, float>::SMEM_DOUBLES)
                    : (ext ? LinCfg<true, double>::SMEM_DOUBLES : LinCfg<false, double>::SMEM_DOUBLES);
  return (size_t)d * sizeof(double);
}
size_t solve_smem(int Dpad, bool large) {
  // LDS-resident: the matrix area of the LDL^T solver (ba_ldl16.hpp; Dpad >= D bounds it) + four vectors
  return ((large ? 0 : (size_t)ldl16_area_doubles(Dpad)) + 4 * (size_t)Dpad) * sizeof(double) + 16;
}
// chain solver (ba_chain.hpp): its matrix area (LChain::total of the batch's largest window) + the four vectors
size_t solve_smem_chain(int chain_doubles, int Dpad) { return ((size_t)chain_doubles + 4 * (size_t)Dpad) * sizeof(double) + 16; }
// the solve launch of the LDS-resident windows: the instantiation the batch was laid out for
void launch_solve_small(okvis_ba_solver* s, dim3 grid, hipStream_t st, const WinPtrs* wins, int final_only, CtrlSlot* ctrls) {
  const bool dbuf = s->group_chunks || s->spec_schur;   // (one set of partials per linearisation buffer)
  if (s->chain) {
    const size_t sm = solve_smem_chain(s->max_chain_doubles, s->max_Dpad_small);
    if (dbuf) hipLaunchKernelGGL((solve_kernel<false, true, true>), grid, dim3(SOLVE_THREADS), sm, st, wins, s->d_opt, final_only, ctrls);
    else hipLaunchKernelGGL((solve_kernel<false, false, true>), grid, dim3(SOLVE_THREADS), sm, st, wins, s->d_opt, final_only, ctrls);
  } else {
    const size_t sm = solve_smem(s->max_Dpad_small, false);
    if (dbuf) hipLaunchKernelGGL((solve_kernel<false, true>), grid, dim3(SOLVE_THREADS), sm, st, wins, s->d_opt, final_only, ctrls);
    else hipLaunchKernelGGL((solve_kernel<false, false>), grid, dim3(SOLVE_THREADS), sm, st, wins, s->d_opt, final_only, ctrls);
  }
}
// dynamic LDS of linearize2_kernel: fixed part + the pose part of the step (fused: the aux area of the group reduction)
int lin2_step_doubles(int max_Dp, bool fuse, bool f32) {
  const int aux = fuse ? (f32 ? Lin2Cfg<float, true>::MIN_STEP_DOUBLES : Lin2Cfg<double, true>::MIN_STEP_DOUBLES) : Lin2Cfg<double, false>::MIN_STEP_DOUBLES;
  return ((max_Dp + 1) / 2) * 2 + 8 + aux;   // pose part of the step, then the aux area of the fused reduction
}
size_t lin2_smem(int max_Dp, bool fuse, bool f32, bool two_rounds = false) {
  const int fixed = two_rounds ? (f32 ? Lin2Cfg<float, false, 14>::FIXED_DOUBLES : Lin2Cfg<double, false, 14>::FIXED_DOUBLES)
                    : f32 ? (fuse ? Lin2Cfg<float, true>::FIXED_DOUBLES : Lin2Cfg<float, false>::FIXED_DOUBLES)
                          : (fuse ? Lin2Cfg<double, true>::FIXED_DOUBLES : Lin2Cfg<double, false>::FIXED_DOUBLES);
  return (size_t)(fixed + lin2_step_doubles(max_Dp, fuse, f32)) * sizeof(double);
}
size_t small_smem() { return (size_t)std::max<int>(std::max<int>(ImuLds::TOTAL, EvalLds::TOTAL), 2 * MAX_MARG_DIM) * sizeof(double); }

// okvis_ba_begin for every window in ONE launch (it used to be three device copies and one upload per window: 1 ms of API calls
// for 64 windows): the trial buffers start as copies of the accepted ones, the control record starts a new optimisation.
// `accs[w]` = the accepted buffer as the host knows it (okvis_ba_set_state wrote there).
__global__ void begin_kernel(const WinPtrs* wins, const int* accs, double initial_radius) {
  const WinPtrs& W = wins[blockIdx.x];
  const int acc = accs[blockIdx.x], tr = 1 - acc, tid = threadIdx.x;
  for (int i = tid; i < 7 * W.n_pose; i += blockDim.x) W.pose[tr][i] = W.pose[acc][i];
  for (int i = tid; i < 9 * W.n_sb; i += blockDim.x) W.sb[tr][i] = W.sb[acc][i];
  for (int i = tid; i < 4 * W.n_lm; i += blockDim.x) W.lm[tr][i] = W.lm[acc][i];
  if (tid == 0) {
    Ctrl c;
    for (size_t k = 0; k < sizeof(Ctrl) / 8; ++k) reinterpret_cast<double*>(&c)[k] = 0.0;
    c.acc = acc;
    c.pending = 1;
    c.first = 1;
    c.radius = initial_radius;
    c.decrease_factor = 2.0;
    c.lambda = 1.0 / initial_radius;
    c.mu = DL_MIN_MU;
    *W.ctrl = c;
  }
}
// the control records of all windows into one contiguous array (one device-to-host copy instead of one per window)
__global__ void gather_ctrl_kernel(const WinPtrs* wins, Ctrl* out, int n) {
  const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (w < n && lane < (int)(sizeof(Ctrl) / 8))
    reinterpret_cast<double*>(out + w)[lane] = reinterpret_cast<const double*>(wins[w].ctrl)[lane];
}

// keeps one wave busy for `ticks` of the 100 MHz wall clock (the start stagger of the sub-batch streams)
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

struct Sub {
  hipStream_t st;
  int w0, nw;
};
Sub whole(okvis_ba_solver* s) { return Sub{s->stream, 0, (int)s->wins.size()}; }

// fused mode: the linearise launch reduces every group it has linearised (ba_linearize.hpp); no Schur launch
bool fused(const okvis_ba_solver* s) {
  // (the fused reduction was sized at upload for the observation stage of one precision)
  return s->group_chunks && (s->opt.fp32_linearize != 0) == s->fp32_at_upload &&
         (s->opt.strategy == OKVIS_BA_STRATEGY_DOGLEG || s->opt.gauss_newton);
}
// decision-free Schur launch (okvis_ba_solver::spec_schur): the batch was laid out for it and the options still ask for a mode
// whose damping does not depend on the decision
bool spec_schur_now(const okvis_ba_solver* s) {
  return s->spec_schur && !fused(s) && (s->opt.strategy == OKVIS_BA_STRATEGY_DOGLEG || s->opt.gauss_newton);
}
hipError_t launch_schur(okvis_ba_solver* s, Sub b, int final_call = 0) {
  if (s->max_schur_blocks == 0 || fused(s)) return hipSuccess;
  const int trows = std::min(TILE_DIM, s->max_Dp);
  const bool no_mfma = (s->opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_VALU) != 0;
  // no pose x extrinsics cross blocks: the reduction as a GEMM on the fp64 matrix core (ba_schur2.hpp).  Pose parts beyond 63 rows
  // (several 96-row tile pairs per chunk) keep schur_kernel unless OKVIS_BA_TUNE_SCHUR_MFMA_LARGE is set: every tile pair of a chunk
  // scans all its (landmark, block) rows to fill its tiles, and at configs[2] that makes the matrix-core kernel the slower one
  // (111 against 100 us per launch)
  const bool mfma_large = (s->opt.tuning.flags & OKVIS_BA_TUNE_SCHUR_MFMA_LARGE) != 0;
  if (!s->any_ext && !no_mfma && (trows + 1 <= SCH2_MAXT_SMALL_ROWS || mfma_large)) {
    int nlb = sch2_nlb(trows, 5120);              // 40 KB of tiles: three workgroups per CU
    if (nlb < 12) nlb = sch2_nlb(trows, 9216);    // wide tiles: 72 KB, two per CU
    const size_t sm = (size_t)sch2_tile_doubles(trows, nlb) * sizeof(double);
    if (trows + 1 <= SCH2_MAXT_SMALL_ROWS)
      hipLaunchKernelGGL(schur_mfma_kernel<3>, dim3(s->max_schur_blocks, (unsigned)b.nw), dim3(SCHUR_THREADS), sm, b.st, s->d_wins + b.w0, s->d_opt, trows, final_call, nlb, s->d_ctrl + b.w0, spec_schur_now(s) ? 1 : 0);
    else
      hipLaunchKernelGGL(schur_mfma_kernel<9>, dim3(s->max_schur_blocks, (unsigned)b.nw), dim3(SCHUR_THREADS), sm, b.st, s->d_wins + b.w0, s->d_opt, trows, final_call, nlb, s->d_ctrl + b.w0, 0);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(schur_kernel, dim3(s->max_schur_blocks, (unsigned)b.nw), dim3(SCHUR_THREADS),
                     (size_t)2 * SCHUR_LM_BATCH * trows * 3 * sizeof(double), b.st, s->d_wins + b.w0, s->d_opt, trows,
                     final_call);
  return hipGetLastError();
}
hipError_t launch_solve(okvis_ba_solver* s, Sub b, int final_only) {
  if (s->max_Dpad_small > 0)
    launch_solve_small(s, dim3((unsigned)b.nw, 1 + (b.nw <= SOLVE_HELPED_MAX_WINDOWS ? SOLVE_HELPERS : 0)), b.st, s->d_wins + b.w0, final_only, s->d_ctrl + b.w0);
  if (s->max_Dpad_large > 0) {
    // large windows: assemble + export, tiled multi-workgroup Cholesky (fp64 MFMA), back-substitution + finish
    hipLaunchKernelGGL((solve_kernel<true, false>), dim3((unsigned)b.nw), dim3(SOLVE_THREADS), solve_smem(s->max_Dpad_large, true), b.st,
                       s->d_wins + b.w0, s->d_opt, final_only, s->d_ctrl + b.w0);
    if (!final_only) {
      const int nT = (s->max_Dpad_large + CT_TB - 1) / CT_TB;
      hipLaunchKernelGGL(large_export_kernel, dim3(nT * (nT + 1) / 2, (unsigned)b.nw, CT_TILE / CT_THREADS), dim3(CT_THREADS), 0, b.st,
                         s->d_wins + b.w0);
      hipLaunchKernelGGL(chol_tiles_window_kernel, dim3(nT * (nT + 1) / 2 + nT, (unsigned)b.nw), dim3(CT_THREADS), CT_SMEM_DOUBLES * 8,
                         b.st, s->d_wins + b.w0);
      hipLaunchKernelGGL(solve_large_tail_kernel, dim3((unsigned)b.nw), dim3(SOLVE_THREADS), 0, b.st, s->d_wins + b.w0, s->d_opt);
    }
  }
  return hipGetLastError();
}
// one launch for everything that depends only on the trial state: IMU / prior factors (first max_imu + 1
// workgroups) and the reprojection groups
hipError_t launch_lin(okvis_ba_solver* s, Sub b, int init) {
  const int n_small = s->max_imu + 1;
  const bool f32 = s->opt.fp32_linearize != 0;
  const bool fuse = fused(s);
  if (s->lin2) {   // piece path (ba_linearize2.hpp)
    const int sd = lin2_step_doubles(s->max_Dp, fuse, f32);
    const size_t smem2 = lin2_smem(s->max_Dp, fuse, f32);
    if (s->split_small) {
      hipLaunchKernelGGL(small_kernel, dim3(n_small, (unsigned)b.nw), dim3(LIN_THREADS), small_smem(), b.st, s->d_wins + b.w0, init);
      const dim3 grid2(s->max_group, (unsigned)b.nw);
      const int occ_env = s->opt.tuning.lin2_occupancy > 0 ? s->opt.tuning.lin2_occupancy : 4;
      const bool two_rounds = !fuse && occ_env >= 4;   // block records in two rounds: 37 KB of LDS, four workgroups per CU
      const size_t smem2v = two_rounds ? lin2_smem(s->max_Dp, false, f32, true) : smem2;
      auto go2 = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid2, dim3(LIN_THREADS), smem2v, b.st, s->d_wins + b.w0, s->d_opt, init, 0, sd); };
      const int occ = occ_env;
      if (f32) fuse ? go2(linearize2_kernel<float, true, false>) : (occ >= 4 ? go2(linearize2_kernel<float, false, false, 4, 14>) : go2(linearize2_kernel<float, false, false, 3>));
      else fuse ? go2(linearize2_kernel<double, true, false>) : (occ >= 4 ? go2(linearize2_kernel<double, false, false, 4, 14>) : go2(linearize2_kernel<double, false, false, 3>));
    } else {
      const dim3 grid2(n_small + s->max_group, (unsigned)b.nw);
      const size_t sm = std::max(smem2, small_smem());
      auto go2 = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid2, dim3(LIN_THREADS), sm, b.st, s->d_wins + b.w0, s->d_opt, init, n_small, sd); };
      if (f32) fuse ? go2(linearize2_kernel<float, true, true>) : go2(linearize2_kernel<float, false, true>);
      else fuse ? go2(linearize2_kernel<double, true, true>) : go2(linearize2_kernel<double, false, true>);
    }
    return hipGetLastError();
  }
  const dim3 grid(n_small + s->max_group, (unsigned)b.nw), blk(LIN_THREADS);
  const size_t smem = std::max(lin_smem(s->any_ext, f32), small_smem());
  auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, blk, smem, b.st, s->d_wins + b.w0, s->d_opt, init, n_small); };
  if (s->any_ext) {
    if (f32) fuse ? go(linearize_kernel<true, float, true>) : go(linearize_kernel<true, float, false>);
    else fuse ? go(linearize_kernel<true, double, true>) : go(linearize_kernel<true, double, false>);
  } else {
    if (f32) fuse ? go(linearize_kernel<false, float, true>) : go(linearize_kernel<false, float, false>);
    else fuse ? go(linearize_kernel<false, double, true>) : go(linearize_kernel<false, double, false>);
  }
  return hipGetLastError();
}
// the preintegrations a discarded speculative evaluation left behind, taken back before results leave the device (ba_solve.hpp)
hipError_t launch_imu_take_back(okvis_ba_solver* s, int w0, int nw) {
  if (s->max_imu == 0 || s->opt.strategy != OKVIS_BA_STRATEGY_DOGLEG || s->opt.gauss_newton) return hipSuccess;
  hipLaunchKernelGGL(imu_take_back_kernel, dim3((unsigned)s->max_imu, (unsigned)nw), dim3(64), 0, s->stream, s->d_wins + w0);
  return hipGetLastError();
}
hipError_t launch_iteration(okvis_ba_solver* s, Sub b) {
  hipError_t e;
  if ((e = launch_schur(s, b)) != hipSuccess) return e;
  if ((e = launch_solve(s, b, 0)) != hipSuccess) return e;
  return launch_lin(s, b, 0);
}
// n iterations of every sub-batch: fork from the main stream, one chain per sub-stream, join
hipError_t launch_budget(okvis_ba_solver* s, Sub b, int n) {
  if (s->opt.strategy != OKVIS_BA_STRATEGY_DOGLEG || n <= 0) return hipSuccess;
  hipLaunchKernelGGL(add_budget_kernel, dim3((unsigned)b.nw), dim3(64), 0, b.st, s->d_wins + b.w0, n);
  return hipGetLastError();
}
hipError_t launch_iterations_forked(okvis_ba_solver* s, int n, int budget = -1) {
  const int nsub = (int)s->sub_streams.size();
  if (budget < 0) budget = n;
  if (nsub <= 1) {
    hipError_t e = launch_budget(s, whole(s), budget);
    for (int i = 0; i < n && e == hipSuccess; ++i) e = launch_iteration(s, whole(s));
    return e;
  }
  hipError_t e = hipEventRecord(s->ev_fork, s->stream);
  for (int k = 0; k < nsub && e == hipSuccess; ++k) {
    e = hipStreamWaitEvent(s->sub_streams[k], s->ev_fork, 0);
    const Sub b{s->sub_streams[k], s->sub_begin[k], s->sub_begin[k + 1] - s->sub_begin[k]};
    if (e == hipSuccess) e = launch_budget(s, b, budget);
    for (int i = 0; i < n && e == hipSuccess; ++i) e = launch_iteration(s, b);
    if (e == hipSuccess) e = hipEventRecord(s->sub_events[k], s->sub_streams[k]);
    if (e == hipSuccess) e = hipStreamWaitEvent(s->stream, s->sub_events[k], 0);
  }
  return e;
}

// the accepted-buffer index lives on the device while iterations are in flight: read it back
int refresh_acc(okvis_ba_solver* s, int w) {
  if (s->acc_fresh) return OKVIS_BA_OK;   // read by okvis_ba_finish / set at upload, nothing launched since
  int acc = 0;
  HIP_TRY(hipMemcpyAsync(&acc, &s->wins[w].ptrs.ctrl->acc, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->wins[w].acc = acc & 1;
  return OKVIS_BA_OK;
}

// grow-only staging for begin_kernel's buffer indices and the gathered control records (pinned host + device)
hipError_t reserve_ctrl_stage(okvis_ba_solver* s, size_t n_windows) {
  const size_t bytes = std::max(sizeof(Ctrl), sizeof(int)) * n_windows;
  if (bytes <= s->ctrl_stage_bytes) return hipSuccess;
  if (s->h_ctrl_stage) (void)hipHostFree(s->h_ctrl_stage);
  if (s->d_ctrl_stage) (void)hipFree(s->d_ctrl_stage);
  s->h_ctrl_stage = s->d_ctrl_stage = nullptr;
  s->ctrl_stage_bytes = 0;
  hipError_t e = hipHostMalloc((void**)&s->h_ctrl_stage, bytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_ctrl_stage, bytes);
  if (e == hipSuccess) s->ctrl_stage_bytes = bytes;
  return e;
}

int fetch_ctrl(okvis_ba_solver* s, std::vector<Ctrl>& out) {
  const size_t n = s->wins.size();
  out.resize(n);
  HIP_TRY(reserve_ctrl_stage(s, n));
  hipLaunchKernelGGL(gather_ctrl_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s->stream, s->d_wins, reinterpret_cast<Ctrl*>(s->d_ctrl_stage), (int)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_ctrl_stage, s->d_ctrl_stage, sizeof(Ctrl) * n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::memcpy(out.data(), s->h_ctrl_stage, sizeof(Ctrl) * n);
  return OKVIS_BA_OK;
}

}  // namespace

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
  s->stagger_ticks = 20 * 100;   // start offset of the sub-batch streams (us; default 20: measured 20 / 45 / 70 us all lock the fast interleaving; okvis_ba_tuning::stagger_us)
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
  }
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
  for (auto st : s->sub_streams) (void)hipStreamDestroy(st);
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
  s->stagger_ticks = (long long)(opt->tuning.stagger_us > 0 ? opt->tuning.stagger_us : opt->tuning.stagger_us < 0 ? 0 : 20) * 100;
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
  //      2: 300 k, 3: 325 k, 4: 198 k window-iterations/s — the main stream and the sub-streams together must not
  //      exceed the 4 hardware queues of the default runtime configuration (GPU_MAX_HW_QUEUES) ----
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
      for (auto st : s->sub_streams) (void)hipStreamDestroy(st);
      for (auto ev : s->sub_events) (void)hipEventDestroy(ev);
      s->sub_streams.clear();
      s->sub_events.clear();
    }
    if (nsub > 1 && !same) {
      for (int k = 0; k < nsub; ++k) {
        hipStream_t st;
        hipEvent_t ev;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
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

int okvis_ba_shard(int32_t n_total, int32_t rank, int32_t world, int32_t* ids_out, int32_t* n_out) {
  if (n_total < 0 || world <= 0 || rank < 0 || rank >= world || !ids_out || !n_out) return OKVIS_BA_ERR_ARG;
  int32_t n = 0;
  for (int32_t i = rank; i < n_total; i += world) ids_out[n++] = i;   // window i -> rank i mod world
  *n_out = n;
  return OKVIS_BA_OK;
}

int okvis_ba_batch_run(int device, int32_t rank, int32_t world, int32_t n_total, const okvis_ba_window* all_windows,
                       const okvis_ba_options* opt, int num_iter, okvis_ba_window_record* records_out, int32_t* n_out) {
  if (!all_windows || !records_out || !n_out || num_iter < 0) return OKVIS_BA_ERR_ARG;
  std::vector<int32_t> ids((size_t)std::max(1, (n_total + std::max(world, 1) - 1) / std::max(world, 1)));
  int32_t n = 0;
  int rc = okvis_ba_shard(n_total, rank, world, ids.data(), &n);
  if (rc != OKVIS_BA_OK) return rc;
  *n_out = n;
  if (n == 0) return OKVIS_BA_OK;
  std::vector<okvis_ba_window> mine((size_t)n);
  for (int32_t k = 0; k < n; ++k) mine[(size_t)k] = all_windows[ids[(size_t)k]];
  okvis_ba_solver* s = nullptr;
  rc = okvis_ba_create(&s, device);
  if (rc != OKVIS_BA_OK) return rc;
  if (opt) rc = okvis_ba_set_options(s, opt);
  if (rc == OKVIS_BA_OK) rc = okvis_ba_upload(s, n, mine.data());
  std::vector<okvis_ba_summary> sum((size_t)n);
  double seconds = 0;
  if (rc == OKVIS_BA_OK) {
    const auto t0 = std::chrono::steady_clock::now();
    rc = okvis_ba_optimize(s, num_iter, sum.data());   // ends with a stream synchronisation (summaries are read back)
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (rc == OKVIS_BA_OK)
    for (int32_t k = 0; k < n; ++k) {
      records_out[k].window_id = (uint32_t)ids[(size_t)k];
      records_out[k].iterations = (uint32_t)sum[(size_t)k].iterations;
      records_out[k].final_cost = sum[(size_t)k].final_cost;
      records_out[k].seconds = seconds;
    }
  (void)okvis_ba_destroy(s);
  return rc;
}

int okvis_ba_dense_solve(int device, int32_t n, const double* S, const double* rhs, double* x, int32_t* info) {
  if (n <= 0 || !S || !rhs || !x || !info) return OKVIS_BA_ERR_ARG;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return OKVIS_BA_ERR_NO_DEVICE;
  const int nT = (n + CT_TB - 1) / CT_TB, ntiles = nT * (nT + 1) / 2, np = nT * CT_TB;
  std::vector<double> tiles((size_t)ntiles * CT_TILE, 0.0), r(np, 0.0);
  for (int i = 0; i < nT; ++i)
    for (int j = 0; j <= i; ++j) {
      double* t = &tiles[(size_t)(i * (i + 1) / 2 + j) * CT_TILE];
      for (int a = 0; a < CT_TB; ++a)
        for (int b = 0; b < CT_TB; ++b) {
          const int gi = CT_TB * i + a, gj = CT_TB * j + b;
          t[a * CT_TB + b] = (gi < n && gj < n) ? S[(size_t)gi * n + gj] : ((gi == gj) ? 1.0 : 0.0);
        }
    }
  for (int k = 0; k < n; ++k) r[k] = rhs[k];
  Arena A;
  const size_t oT = A.alloc(8 * tiles.size()), oL = A.alloc(8 * (size_t)nT * CT_TILE), oR = A.alloc(8 * (size_t)np),
               oY = A.alloc(8 * (size_t)np), oX = A.alloc(8 * (size_t)np), oF = A.alloc(sizeof(int) * (ntiles + 1 + 2 * nT));
  unsigned char* d = nullptr;
  if (hipMalloc(&d, A.size) != hipSuccess) return OKVIS_BA_HIP_ERROR_BASE + (int)hipGetLastError();
  struct Free { unsigned char* p; ~Free() { if (p) (void)hipFree(p); } } guard{d};
  auto chk = [](hipError_t err) { return err == hipSuccess ? OKVIS_BA_OK : OKVIS_BA_HIP_ERROR_BASE + (int)err; };
  int rc;
  if ((rc = chk(hipMemcpy(d + oT, tiles.data(), 8 * tiles.size(), hipMemcpyHostToDevice)))) return rc;
  if ((rc = chk(hipMemcpy(d + oR, r.data(), 8 * (size_t)np, hipMemcpyHostToDevice)))) return rc;
  if ((rc = chk(hipMemset(d + oF, 0, sizeof(int) * (ntiles + 1 + 2 * nT))))) return rc;
  {
    std::vector<unsigned long long> sentinel(np, CT_X_SENTINEL);
    if ((rc = chk(hipMemcpy(d + oX, sentinel.data(), 8 * (size_t)np, hipMemcpyHostToDevice)))) return rc;
  }
  CholTiles C;
  C.nT = nT;
  C.T = reinterpret_cast<double*>(d + oT);
  C.Linv = reinterpret_cast<double*>(d + oL);
  C.rhs = reinterpret_cast<double*>(d + oR);
  C.y = reinterpret_cast<double*>(d + oY);
  C.flag = reinterpret_cast<int*>(d + oF);
  C.pflag = C.flag + ntiles + 1;
  C.x = reinterpret_cast<double*>(d + oX);
  if ((rc = chk(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    CT_SMEM_DOUBLES * 8))))
    return rc;
  hipLaunchKernelGGL(chol_tile_kernel, dim3(ntiles + nT), dim3(CT_THREADS), CT_SMEM_DOUBLES * 8, 0, C);
  if ((rc = chk(hipGetLastError()))) return rc;
  if ((rc = chk(hipDeviceSynchronize()))) return rc;
  std::vector<double> xs(np);
  int fail = 0;
  if ((rc = chk(hipMemcpy(xs.data(), d + oX, 8 * (size_t)np, hipMemcpyDeviceToHost)))) return rc;
  if ((rc = chk(hipMemcpy(&fail, d + oF + sizeof(int) * ntiles, sizeof(int), hipMemcpyDeviceToHost)))) return rc;
  for (int k = 0; k < n; ++k) x[k] = xs[k];
  *info = fail;
  return OKVIS_BA_OK;
}

int okvis_ba_reduced_solve(int device, int32_t D, int32_t Dp, int32_t mode, uint32_t comp_mask, const double* S, const double* rhs,
                           double* x, int64_t* ticks, int32_t* info, int32_t repeats, double* lds_dump, int64_t lds_capacity) {
  if (D <= 0 || D > MAX_D_LDS || Dp < 0 || Dp > D || (D - Dp) % 9 != 0 || !S || !rhs || !x || !info) return OKVIS_BA_ERR_ARG;
  const bool chain = mode == OKVIS_BA_SOLVE_CHAIN;
  if (chain && (Dp < 6 || D == Dp || !LChain::tiles_fit(Dp))) return OKVIS_BA_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return OKVIS_BA_ERR_NO_DEVICE;
  if (repeats < 1) repeats = 1;
  auto chk = [](hipError_t err) { return err == hipSuccess ? OKVIS_BA_OK : OKVIS_BA_HIP_ERROR_BASE + (int)err; };
  unsigned char* d = nullptr;
  const size_t n_lds = chain ? (size_t)LChain::make(D, Dp).total : (size_t)ldl16_area_doubles(D);
  const bool dumping = lds_dump && lds_capacity >= (int64_t)n_lds;
  const size_t nS = 8 * (size_t)D * D, total = nS + 16 * (size_t)D + 64 + (dumping ? 8 * n_lds : 0);
  if (hipMalloc(&d, total) != hipSuccess) return OKVIS_BA_HIP_ERROR_BASE + (int)hipGetLastError();
  struct Free { unsigned char* p; ~Free() { if (p) (void)hipFree(p); } } guard{d};
  int rc;
  if ((rc = chk(hipMemcpy(d, S, nS, hipMemcpyHostToDevice)))) return rc;
  if ((rc = chk(hipMemcpy(d + nS, rhs, 8 * (size_t)D, hipMemcpyHostToDevice)))) return rc;
  double* dx = reinterpret_cast<double*>(d + nS + 8 * (size_t)D);
  long long* dt = reinterpret_cast<long long*>(d + nS + 16 * (size_t)D);
  int* di = reinterpret_cast<int*>(dt + 1);
  double* dd = dumping ? reinterpret_cast<double*>(d + nS + 16 * (size_t)D + 64) : nullptr;
  const size_t smem = ((chain ? (size_t)LChain::make(D, Dp).total : (size_t)ldl16_area_doubles(D)) + (size_t)D + 8) * sizeof(double);
  const void* fn = chain ? reinterpret_cast<const void*>(&reduced_solve_kernel<true>) : reinterpret_cast<const void*>(&reduced_solve_kernel<false>);
  if ((rc = chk(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(smem, 65536))))) return rc;
  if (chain)
    hipLaunchKernelGGL(reduced_solve_kernel<true>, dim3(1), dim3(SOLVE_THREADS), smem, 0, reinterpret_cast<const double*>(d),
                       reinterpret_cast<const double*>(d + nS), D, Dp, comp_mask, dx, dt, di, repeats, dd);
  else
    hipLaunchKernelGGL(reduced_solve_kernel<false>, dim3(1), dim3(SOLVE_THREADS), smem, 0, reinterpret_cast<const double*>(d),
                       reinterpret_cast<const double*>(d + nS), D, Dp, comp_mask, dx, dt, di, repeats, dd);
  if ((rc = chk(hipGetLastError()))) return rc;
  if ((rc = chk(hipDeviceSynchronize()))) return rc;
  if ((rc = chk(hipMemcpy(x, dx, 8 * (size_t)D, hipMemcpyDeviceToHost)))) return rc;
  long long t = 0;
  int f = 0;
  if ((rc = chk(hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost)))) return rc;
  if ((rc = chk(hipMemcpy(&f, di, sizeof(f), hipMemcpyDeviceToHost)))) return rc;
  if (ticks) *ticks = t;
  *info = f;
  if (dumping && (rc = chk(hipMemcpy(lds_dump, dd, 8 * n_lds, hipMemcpyDeviceToHost)))) return rc;
  return OKVIS_BA_OK;
}

// MarginalizationError numerics (see include/okvis_amd_ba.h): linearise at the uploaded values, eliminate
// the landmarks (schur_kernel in marg_mode), export the dense system (solve_kernel final_only = 2), then the
// dense elimination + eigen-decomposition (marg_dense_kernel).
int okvis_ba_marginalize(okvis_ba_solver* s, int w, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* res) {
  if (int rc = okvis_ba_marginalize_begin(s, w, spec, res)) return rc;
  return okvis_ba_marginalize_end(s, res);
}

int okvis_ba_marginalize_begin(okvis_ba_solver* s, int w, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* res) {
  if (s) s->acc_fresh = false;
  if (!s || !spec || !res) return OKVIS_BA_ERR_ARG;
  if (!s->uploaded || s->marg_pending.active) return OKVIS_BA_ERR_STATE;
  if (w < 0 || w >= (int)s->wins.size()) return OKVIS_BA_ERR_ARG;
  HostWin& H = s->wins[w];
  if (H.marg_dim != 0) return OKVIS_BA_ERR_ARG;                    // the previous prior comes in through spec
  const bool large_window = H.ptrs.Sg != nullptr;   // reduced system assembled in HBM (D > MAX_D_LDS)
  if ((H.n_pose > 0 && !spec->pose_marg) || (H.n_sb > 0 && !spec->sb_marg)) return OKVIS_BA_ERR_ARG;
  const int pd = spec->prior_dim, pnb = spec->prior_nblocks;
  if (pd < 0 || pnb < 0 || pd > MAX_MARG_DIM) return pd > MAX_MARG_DIM ? OKVIS_BA_ERR_UNSUPPORTED : OKVIS_BA_ERR_ARG;
  if (pd > 0 && (!spec->prior_block_type || !spec->prior_block_idx || !spec->prior_block_off || !spec->prior_H ||
                 !spec->prior_b0 || pnb == 0))
    return OKVIS_BA_ERR_ARG;
  for (int k = 0, expect = 0; k < (pd > 0 ? pnb : 0); ++k) {
    const int t = spec->prior_block_type[k], idx = spec->prior_block_idx[k];
    if (spec->prior_block_off[k] != expect) return OKVIS_BA_ERR_ARG;
    if (t == OKVIS_BA_BLOCK_POSE) {
      if (idx < 0 || idx >= H.n_pose) return OKVIS_BA_ERR_ARG;
      expect += 6;
    } else if (t == OKVIS_BA_BLOCK_SPEEDBIAS) {
      if (idx < 0 || idx >= H.n_sb) return OKVIS_BA_ERR_ARG;
      expect += 9;
    } else {
      return OKVIS_BA_ERR_ARG;
    }
    if (k == pnb - 1 && expect != pd) return OKVIS_BA_ERR_ARG;
  }
  // kept blocks, in reduced order (pose-type blocks first)
  std::vector<int> bt, bi, bo;
  int na = 0;
  for (int i = 0; i < H.n_pose; ++i)
    if (H.pose_off[i] >= 0 && !spec->pose_marg[i]) {
      bt.push_back(OKVIS_BA_BLOCK_POSE); bi.push_back(i); bo.push_back(na);
      na += 6;
    }
  for (int i = 0; i < H.n_sb; ++i)
    if (H.sb_off[i] >= 0 && !spec->sb_marg[i]) {
      bt.push_back(OKVIS_BA_BLOCK_SPEEDBIAS); bi.push_back(i); bo.push_back(na);
      na += 9;
    }
  if (na > res->capacity_dim || (int)bt.size() > res->capacity_blocks) return OKVIS_BA_ERR_ARG;
  if (na > 0 && (!res->H || !res->b0 || !res->J || !res->e0 || !res->block_type || !res->block_idx || !res->block_off))
    return OKVIS_BA_ERR_ARG;
  HIP_TRY(hipSetDevice(s->device));

  // ---- one scratch allocation ----
  const int D = H.D;
  Arena A;
  const size_t o_pm = A.alloc(std::max(1, H.n_pose)), o_sm = A.alloc(std::max(1, H.n_sb));
  const size_t o_pt = A.alloc(sizeof(int) * std::max(1, pnb)), o_pi = A.alloc(sizeof(int) * std::max(1, pnb)),
               o_po = A.alloc(sizeof(int) * std::max(1, pnb));
  const size_t o_pH = A.alloc(8 * std::max<size_t>(1, (size_t)pd * pd)), o_pb = A.alloc(8 * std::max(1, pd));
  const size_t o_win = A.alloc(sizeof(WinPtrs)), o_opt = A.alloc(sizeof(OptD));
  const size_t host_part = A.size;   // everything up to here is written by the host: ONE copy
  const size_t o_work = A.alloc(8 * marg_work_doubles(std::max(1, D)));
  const size_t o_S = A.alloc(8 * std::max<size_t>(1, (size_t)D * D)), o_rhs = A.alloc(8 * std::max(1, D)),
               o_d2 = A.alloc(8 * std::max(1, D));
  // H | J | b0 | e0 | info: contiguous, ONE copy back
  const size_t nn = std::max<size_t>(1, (size_t)na * na), n1 = std::max(1, na);
  const size_t out_bytes = 8 * (2 * nn + 2 * n1);
  const size_t o_out = A.alloc(out_bytes + sizeof(int) * (8 + std::max(1, D)));
  const size_t o_info = o_out + out_bytes;
  // kept blocks beyond the single-workgroup LDS paths: the tail on many workgroups (ba_marg_tiles.hpp)
  const bool no_tiles = (s->opt.tuning.flags & OKVIS_BA_TUNE_NO_MARG_TILES) != 0;   // (A/B switch)
  const bool tiles = na > MARG_PC_NMAX && !no_tiles;
  const int mt_nT = tiles ? (na + CT_TB - 1) / CT_TB : 0, mt_ntiles = mt_nT * (mt_nT + 1) / 2;
  const size_t o_mtT = A.alloc(8 * (size_t)std::max(1, mt_ntiles) * CT_TILE), o_mtZ = A.alloc(8 * (size_t)std::max(1, mt_ntiles) * CT_TILE),
               o_mtL = A.alloc(8 * (size_t)std::max(1, mt_nT) * CT_TILE), o_mtR = A.alloc(8 * (size_t)std::max(1, CT_TB * mt_nT)),
               o_mtY = A.alloc(8 * (size_t)std::max(1, CT_TB * mt_nT)), o_mtP = A.alloc(8 * (size_t)std::max(1, CT_TB * mt_nT)),
               o_mtF = A.alloc(8 * (size_t)std::max(1, mt_ntiles)), o_mtp = A.alloc(8 * (size_t)std::max(1, D)),
               o_mtS = A.alloc(8 * (size_t)std::max(1, CT_TB * mt_nT)),
               o_mtf = A.alloc(sizeof(int) * (size_t)(mt_ntiles + 1 + 2 * mt_nT + 1 + mt_ntiles));
  s->stage_marg.resize(host_part);   // page-locked: the one upload of this call is a true asynchronous copy
  unsigned char* const hb = s->stage_marg.data();
  if (H.n_pose) std::memcpy(&hb[o_pm], spec->pose_marg, H.n_pose);
  if (H.n_sb) std::memcpy(&hb[o_sm], spec->sb_marg, H.n_sb);
  if (pd > 0) {
    std::memcpy(&hb[o_pt], spec->prior_block_type, sizeof(int) * pnb);
    std::memcpy(&hb[o_pi], spec->prior_block_idx, sizeof(int) * pnb);
    std::memcpy(&hb[o_po], spec->prior_block_off, sizeof(int) * pnb);
    std::memcpy(&hb[o_pH], spec->prior_H, 8 * (size_t)pd * pd);
    std::memcpy(&hb[o_pb], spec->prior_b0, 8 * (size_t)pd);
  }
  if (A.size > s->marg_scratch_bytes) {
    if (s->marg_scratch) HIP_TRY(hipFree(s->marg_scratch));
    s->marg_scratch = nullptr;
    s->marg_scratch_bytes = 0;
    HIP_TRY(hipMalloc(&s->marg_scratch, A.size));
    s->marg_scratch_bytes = A.size;
  }
  unsigned char* d = s->marg_scratch;
  WinPtrs P = H.ptrs;   // this window with the export buffers attached
  P.S = (decltype(P.S))(d + o_S);
  P.rhs = (decltype(P.rhs))(d + o_rhs);
  P.Dp2 = (decltype(P.Dp2))(d + o_d2);
  P.grad = nullptr;
  std::memcpy(&hb[o_win], &P, sizeof(P));
  const WinPtrs* d_win = reinterpret_cast<const WinPtrs*>(d + o_win);
  // the marginalisation pass has its own option record in the scratch block (no trust region: one linearisation, no damping);
  // the launches below read it through s->d_opt, which points there for the duration of this call.  The solver's own record is
  // never touched, so nothing has to be restored on the device and captured launch graphs stay valid.
  OptD od = make_optd(s->opt, (int)s->wins.size());
  od.marg_mode = 1;
  od.dogleg = 0;
  std::memcpy(&hb[o_opt], &od, sizeof(od));
  HIP_TRY(hipMemcpyAsync(d, hb, host_part, hipMemcpyHostToDevice, s->stream));
  struct SwapOptions {
    okvis_ba_solver* s;
    OptD* saved;
    ~SwapOptions() { s->d_opt = saved; }
  } swap_options{s, s->d_opt};
  s->d_opt = reinterpret_cast<OptD*>(d + o_opt);

  // ---- linearise + landmark elimination + export ----
  int rc = okvis_ba_begin(s);
  if (rc != OKVIS_BA_OK) return rc;
  s->begun = false;
  const Sub one{s->stream, w, 1};
  HIP_TRY(launch_schur(s, one));
  if (large_window) {
    // assembly of the undamped system, then the kernel that completes it (Schur partials, IMU terms) and, because this copy of
    // the window carries an S pointer, writes it out as one full symmetric D x D matrix
    hipLaunchKernelGGL((solve_kernel<true, false>), dim3(1), dim3(SOLVE_THREADS), solve_smem(s->max_Dpad_large, true), s->stream, d_win, s->d_opt, 2, s->d_ctrl + w);
    const int nT = (((D + 5) / 6) * 6 + CT_TB - 1) / CT_TB;
    hipLaunchKernelGGL(large_export_kernel, dim3(nT * (nT + 1) / 2, 1, CT_TILE / CT_THREADS), dim3(CT_THREADS), 0, s->stream, d_win);
  } else
    launch_solve_small(s, dim3(1, 1 + SOLVE_HELPERS), s->stream, d_win, 2, s->d_ctrl + w);
  HIP_TRY(hipGetLastError());
  MargArgs ma;
  ma.pose_marg = d + o_pm;
  ma.sb_marg = d + o_sm;
  ma.prior_dim = pd;
  ma.prior_nb = pd > 0 ? pnb : 0;
  ma.pb_type = reinterpret_cast<const int*>(d + o_pt);
  ma.pb_idx = reinterpret_cast<const int*>(d + o_pi);
  ma.pb_off = reinterpret_cast<const int*>(d + o_po);
  ma.prior_H = reinterpret_cast<const double*>(d + o_pH);
  ma.prior_b0 = reinterpret_cast<const double*>(d + o_pb);
  ma.work = reinterpret_cast<double*>(d + o_work);
  double* outp = reinterpret_cast<double*>(d + o_out);
  ma.out_H = outp;
  ma.out_J = outp + nn;
  ma.out_b0 = outp + 2 * nn;
  ma.out_e0 = outp + 2 * nn + n1;
  ma.out_info = reinterpret_cast<int*>(d + o_info);
  ma.p_out = reinterpret_cast<double*>(d + o_mtp);
  auto dense = [&](const MargArgs& args, int stage) {
    if (large_window || pd > MARG_SMALL_PRIOR)
      hipLaunchKernelGGL((marg_dense_kernel<MAX_D, MAX_MARG_DIM>), dim3(1), dim3(MARG_THREADS), MARG_LDS_DOUBLES_LARGE * 8, s->stream, d_win, 0,
                         args, MARG_LDS_DOUBLES_LARGE, stage);
    else
      hipLaunchKernelGGL((marg_dense_kernel<MAX_D_LDS, MARG_SMALL_PRIOR>), dim3(1), dim3(MARG_THREADS), MARG_LDS_DOUBLES * 8, s->stream, d_win, 0,
                         args, MARG_LDS_DOUBLES, stage);
  };
  MargTiles mt{};
  if (tiles) {
    // the single workgroup stops after M and b0; Schur complement, scaling, tiled factorisation (matrix core), L^-1 for the proof
    // of full rank, J and e0 on many workgroups
    mt.C.nT = mt_nT;
    mt.C.T = reinterpret_cast<double*>(d + o_mtT);
    mt.C.Linv = reinterpret_cast<double*>(d + o_mtL);
    mt.C.rhs = reinterpret_cast<double*>(d + o_mtR);
    mt.C.y = reinterpret_cast<double*>(d + o_mtY);
    mt.C.flag = reinterpret_cast<int*>(d + o_mtf);
    mt.C.pflag = mt.C.flag + mt_ntiles + 1;
    mt.Z = reinterpret_cast<double*>(d + o_mtZ);
    mt.fro = reinterpret_cast<double*>(d + o_mtF);
    mt.p2 = reinterpret_cast<double*>(d + o_mtP);
    mt.rowsum = reinterpret_cast<double*>(d + o_mtS);
    mt.ok = mt.C.flag + mt_ntiles + 1 + 2 * mt_nT;
    mt.zflag = mt.ok + 1;
    static const bool attrs = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CT_SMEM_DOUBLES * 8);
      return true;
    }();
    (void)attrs;
    if (pd > 0)
      hipLaunchKernelGGL(marg_prior_add_kernel, dim3((unsigned)(((size_t)pd * pd + MARG_TILES_THREADS - 1) / MARG_TILES_THREADS)),
                         dim3(MARG_TILES_THREADS), 0, s->stream, d_win, ma);
    int nm = 0;   // rows of the eliminated block
    for (int i = 0; i < H.n_pose; ++i) nm += (H.pose_off[i] >= 0 && spec->pose_marg[i]) ? 6 : 0;
    for (int i = 0; i < H.n_sb; ++i) nm += (H.sb_off[i] >= 0 && spec->sb_marg[i]) ? 9 : 0;
    if (nm > 0) {
      dense(ma, 1 | 2 | 4);
      hipLaunchKernelGGL(marg_M_kernel, dim3((unsigned)(((size_t)na * nm + MARG_TILES_THREADS - 1) / MARG_TILES_THREADS)), dim3(MARG_TILES_THREADS), 0,
                         s->stream, d_win, ma);
      hipLaunchKernelGGL(marg_b0_kernel, dim3((unsigned)((na + MARG_TILES_THREADS - 1) / MARG_TILES_THREADS)), dim3(MARG_TILES_THREADS), 0, s->stream,
                         d_win, ma);
    } else {
      dense(ma, 1 | 2);   // (nothing to eliminate densely: b0 is a gather)
    }
    const unsigned nb2 = (unsigned)(((size_t)na * na + MARG_TILES_THREADS - 1) / MARG_TILES_THREADS);
    hipLaunchKernelGGL(marg_schur_kernel, dim3(nb2), dim3(MARG_TILES_THREADS), 0, s->stream, d_win, ma);
    hipLaunchKernelGGL(marg_tiles_scale_kernel, dim3((CT_TB * mt_nT + MARG_TILES_THREADS - 1) / MARG_TILES_THREADS), dim3(MARG_TILES_THREADS), 0,
                       s->stream, ma, mt);
    hipLaunchKernelGGL(marg_tiles_fill_kernel, dim3(mt_ntiles), dim3(MARG_TILES_THREADS), 0, s->stream, ma, mt);
    hipLaunchKernelGGL(chol_tile_kernel, dim3(mt_ntiles), dim3(CT_THREADS), CT_SMEM_DOUBLES * 8, s->stream, mt.C);
    hipLaunchKernelGGL(marg_tiles_inverse_kernel, dim3(mt_ntiles), dim3(CT_THREADS), 2 * CT_TB * CT_LD * 8, s->stream, ma, mt);
    hipLaunchKernelGGL(marg_tiles_out_kernel, dim3(mt_ntiles), dim3(MARG_TILES_THREADS), 0, s->stream, ma, mt);
    hipLaunchKernelGGL(marg_tiles_rowsum_kernel, dim3((na + MARG_TILES_THREADS / 64 - 1) / (MARG_TILES_THREADS / 64)), dim3(MARG_TILES_THREADS), 0,
                       s->stream, ma, mt);
    hipLaunchKernelGGL(marg_tiles_decide_kernel, dim3(1), dim3(MARG_THREADS), 0, s->stream, ma, mt);
  } else {
    dense(ma, 0);
  }
  HIP_TRY(hipGetLastError());
  // H | J | b0 | e0 | info are contiguous on the device: one copy into page-locked staging, one synchronisation
  int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  s->stage_dl.resize(out_bytes + sizeof(info) + sizeof(int));
  int* const tiles_ok = reinterpret_cast<int*>(s->stage_dl.data() + out_bytes + sizeof(info));
  // (the tiled route may have to fall back on the single workgroup, which needs this call's arguments: it is waited for here;
  //  the route of the pipeline's sizes only enqueues the copy and leaves the wait to okvis_ba_marginalize_end)
  auto fetch = [&]() -> hipError_t {
    hipError_t e = hipMemcpyAsync(s->stage_dl.data(), outp, out_bytes + sizeof(info), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && tiles) e = hipMemcpyAsync(tiles_ok, mt.ok, sizeof(int), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && tiles) e = hipStreamSynchronize(s->stream);
    return e;
  };
  *tiles_ok = 1;
  HIP_TRY(fetch());
  if (tiles && !*tiles_ok) {
    // no proof of full rank (a rank-deficient kept block, a pivot that is not positive): the single workgroup takes over — the
    // previous prior is part of H already, everything else is done again — and goes on to the eigen-decomposition
    MargArgs again = ma;
    again.prior_dim = 0;
    again.prior_nb = 0;
    dense(again, 0);
    HIP_TRY(hipGetLastError());
    *tiles_ok = 1;
    hipError_t e = hipMemcpyAsync(s->stage_dl.data(), outp, out_bytes + sizeof(info), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    HIP_TRY(e);
    s->marg_tiles_fallbacks++;
  }
  // what is known without the numbers: the blocks the new prior connects
  res->dim = na;
  res->nblocks = (int)bt.size();
  for (size_t k = 0; k < bt.size(); ++k) {
    res->block_type[k] = bt[k];
    res->block_idx[k] = bi[k];
    res->block_off[k] = bo[k];
  }
  okvis_ba_solver::MargPending& mp = s->marg_pending;
  mp.active = true;
  mp.synced = tiles;
  mp.w = w, mp.na = na, mp.nn = nn, mp.n1 = n1, mp.out_bytes = out_bytes;
  mp.bt.swap(bt), mp.bi.swap(bi), mp.bo.swap(bo);
  return OKVIS_BA_OK;
}

int okvis_ba_marginalize_end(okvis_ba_solver* s, okvis_ba_marg_result* res) {
  if (!s || !res) return OKVIS_BA_ERR_ARG;
  okvis_ba_solver::MargPending& mp = s->marg_pending;
  if (!mp.active) return OKVIS_BA_ERR_STATE;
  // the result structure is looked at first: a call with too little room changes nothing and can be repeated with more
  if (mp.na > res->capacity_dim || (int)mp.bt.size() > res->capacity_blocks) return OKVIS_BA_ERR_ARG;
  if (mp.na > 0 && (!res->H || !res->b0 || !res->J || !res->e0 || !res->block_type || !res->block_idx || !res->block_off)) return OKVIS_BA_ERR_ARG;
  mp.active = false;   // (whatever happens below, the call is over)
  HIP_TRY(hipSetDevice(s->device));
  if (!mp.synced) HIP_TRY(hipStreamSynchronize(s->stream));
  const int na = mp.na;
  const size_t nn = mp.nn, n1 = mp.n1, out_bytes = mp.out_bytes;
  const std::vector<int>&bt = mp.bt, &bi = mp.bi, &bo = mp.bo;
  HostWin& H = s->wins[mp.w];
  if (na > res->capacity_dim || (int)bt.size() > res->capacity_blocks) return OKVIS_BA_ERR_ARG;
  if (na > 0 && (!res->H || !res->b0 || !res->J || !res->e0 || !res->block_type || !res->block_idx || !res->block_off)) return OKVIS_BA_ERR_ARG;
  int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::memcpy(info, s->stage_dl.data() + out_bytes, sizeof(info));
  if (na > 0) {
    const double* h = reinterpret_cast<const double*>(s->stage_dl.data());
    std::memcpy(res->H, h, 8 * (size_t)na * na);
    std::memcpy(res->J, h + nn, 8 * (size_t)na * na);
    std::memcpy(res->b0, h + 2 * nn, 8 * (size_t)na);
    std::memcpy(res->e0, h + 2 * nn + n1, 8 * (size_t)na);
  }
  if (info[0] != na) return OKVIS_BA_ERR_NUMERIC;
  res->dim = na;
  res->nblocks = (int)bt.size();
  res->rank = info[2];
  res->sweeps[0] = info[3];
  res->sweeps[1] = info[4];
  if (debug_word().marg)
    std::fprintf(stderr, "marginalize: kept dim %d rank %d sweeps %d %d  pivoted-Cholesky bounds: dropped %.3f tau_hi, kept %d tau_hi\n", info[0],
                 info[2], info[3], info[4], info[6] * 1e-3, info[7]);
  for (size_t k = 0; k < bt.size(); ++k) {
    res->block_type[k] = bt[k];
    res->block_idx[k] = bi[k];
    res->block_off[k] = bo[k];
  }
  H.acc = info[5] & 1;   // read by marg_dense_kernel after the export: no separate copy + synchronisation
  s->acc_fresh = true;
  return OKVIS_BA_OK;
}

}  // extern "C"
