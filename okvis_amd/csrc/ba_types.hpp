// Device-side data model of a batch of sliding windows (shared by the kernels and the host uploader).
//
// One `WinPtrs` per window lives in device memory; kernels index it with blockIdx.y and then use plain
// pointers.  All per-window arrays are carved out of one arena allocation (256-byte aligned), so a batch
// is a single hipMalloc and stays resident in HBM across optimize() calls.
#pragma once
#include <stdint.h>

namespace ba {

constexpr int GROUP_OBS = 256;       // observations handled by one linearise workgroup (= block size)
constexpr int GROUP_LM = 64;         // max landmarks per group
constexpr int GROUP_PAIRS = 512;     // max (landmark, block) pairs per group
constexpr int LIN_THREADS = 256;
constexpr int LIN2_PIECES = 128;     // pieces (<= 2 adjacent observations of one (landmark, pose)) per group on the piece path
constexpr int SCHUR_THREADS = 256;
constexpr int SCHUR_TILE_BLOCKS = 16;  // 16 x 16 blocks of 6x6 = 96 x 96 tile, one 6x6 block per thread
constexpr int SCHUR_LM_BATCH = 16;     // landmarks staged per LDS pass in the Schur kernel
constexpr int SCHUR_CHUNK_LM_MAX = 64;   // landmarks of one Schur workgroup (chunk) = GROUP_LM: a chunk is at least one group;
                                        // more would push the kernel past 80 KB of LDS at 96-row tiles (one workgroup per CU)
constexpr int SCHUR_DESC_INTS = 2 + SCHUR_CHUNK_LM_MAX / 4 + 1 + 1;   // chunk descriptor of the matrix-core Schur kernel (20 ints)
constexpr int SOLVE_THREADS = 1024;
constexpr int MAX_D_LDS = 174;         // reduced systems up to this size are factorised in LDS (block-packed)
constexpr int MAX_D = 900;             // larger ones (up to this) keep the block matrix in HBM/L2 (slower path)
constexpr int MAX_IMU_SAMPLES = 1 << 20; // raw samples of one IMU factor (processed in LDS-sized chunks)
constexpr int IMU_THREADS = 256;
// IMU factor linearisation record: H = J^T J (30x30 lower, packed a(a+1)/2+b) | g = J^T r (30) | r (15) | cost
constexpr int IMU_H = 0, IMU_G = 465, IMU_R = 495, IMU_COST = 510;
constexpr int IMU_LIN_STRIDE = 512;
constexpr int MAX_MARG_DIM = MAX_D;    // rows of a marginalisation prior (any subset of one window's blocks)

// 32-byte observation record (coalesced 2 x 16 B per lane).  idx0 = landmark | cam << 24.
struct ObsRec {
  uint32_t lm_cam;
  uint16_t pose;
  uint16_t ext;
  double u, v, sw;
};
static_assert(sizeof(ObsRec) == 32, "ObsRec must be 32 bytes");

struct Group {  // one linearise workgroup: whole landmarks, <= GROUP_OBS observations
  int lm_begin, lm_end;
  int obs_begin, obs_end;
  int pair_begin, pair_end;
  int task_begin, task_end;
  int plist_begin, plist_end;  // the group's slice of pair_list (= pair_list_begin[pair_begin .. pair_end])
  int tlist_begin, tlist_end;  // the group's slice of task_list
  // piece path (ba_linearize2.hpp): pieces before this group in the window, and before waves 1..3 inside the group
  int piece_begin, pw1, pw2, pw3;
};
static_assert(sizeof(Group) == 64, "Group is fetched as 16 dwords");

// reduction task of a group: accumulate over a list of the group's observations
//   type 0: block Hessian/gradient of a pose-role block   out = 27 doubles (21 upper-tri A + 6 g)
//   type 1: same for an extrinsics-role block
//   type 2: cross block  J_pose^T J_ext                    out = 36 doubles
struct Task {
  int type;
  int off_a;       // reduced offset of the (first) block
  int off_b;       // reduced offset of the second block (type 2)
  int list_begin;  // into task_list (group-local observation indices; piece path: group-local pair indices)
  int list_end;
  int out;         // offset (doubles) into the lin buffer's gpart array
};

struct Chunk {  // one Schur workgroup: a range of groups
  int group_begin, group_end;
};


struct ImuParamsD {
  double sigma_g_c, sigma_a_c, sigma_gw_c, sigma_aw_c, g, g_max, a_max;
};

// IMU preintegration cache of one factor (the `mutable` members of ImuError, ImuError.hpp:248-276)
struct ImuCacheD {
  double Delta_q[4];
  double C_integral[9], C_doubleintegral[9];
  double acc_integral[3], acc_doubleintegral[3];
  double dalpha_db_g[9], dv_db_g[9], dp_db_g[9];
  double sqrt_info[225];   // upper-triangular L^T
  double sb_ref[9];
  int valid;               // 0 until the first preintegration (redo_ = true initially); 2 = sb_ref was handed
                           // over by the host (cache of a previous optimize call), integrals not computed yet;
                           // 3 = integrated ahead of the first evaluation at the uploaded bias (imu_pre_place_kernel)
  int redo_count;
};

static_assert(sizeof(ImuCacheD) == 8 * OKVIS_BA_IMU_CACHE_DOUBLES, "the record okvis_ba_fetch_imu_caches hands out");
// bytes of the packed results record (pack_results_kernel): pose | speed/bias | landmarks | quality | IMU reference biases | IMU caches
__host__ __device__ inline size_t results_bytes(int n_pose, int n_sb, int n_lm, int n_imu) {
  return 56 * (size_t)n_pose + 72 * (size_t)n_sb + 40 * (size_t)n_lm + (72 + sizeof(ImuCacheD)) * (size_t)(n_imu > 0 ? n_imu : 0);
}

// layout of WinPtrs::dec (doubles)
enum { DEC_VALID = 0, DEC_SUMS = 1, DEC_DL = 7, DEC_LM = 19, DEC_COUNT = 32 };

// per-group scalar partials written by the linearise kernel
enum { GS_COST = 0, GS_GD = 1, GS_DDD = 2, GS_STEP2 = 3, GS_X2 = 4, GS_GMAX = 5, GS_COUNT = 8 };

// solver options on the device
struct OptD {
  double initial_radius, max_radius, min_radius, min_lm_diag2, max_lm_diag2;  // diag2 bounds: Ceres clamps the SQUARED column norm
  double min_relative_decrease, function_tolerance, gradient_tolerance, parameter_tolerance;
  int gauss_newton;  // 1 = accept every step, keep the radius fixed
  int marg_mode;     // 1 = marginalisation pass: no damping, landmark blocks eliminated with the preconditioned pseudo-inverse
  int dogleg;        // 1 = Ceres' DOGLEG strategy (the reference's configuration), 0 = Levenberg-Marquardt
  int jacobi_scaling;
  int max_invalid;   // max_num_consecutive_invalid_steps
  int helper_polls;  // how long a solving workgroup waits for its helper workgroups before it sums the partials itself
};

// DoglegStrategy constants (Ceres: kMinMu, kMaxMu, mu_increase_factor_)
constexpr double DL_MIN_MU = 1e-8, DL_MAX_MU = 1.0, DL_MU_INCREASE = 10.0;

// trust-region state of one window; written ONLY by the solve kernel (and the finish kernel)
struct Ctrl {
  int acc;          // index (0/1) of the accepted state + linearisation buffers
  int pending;      // 1 = a trial linearisation sits in buffer 1-acc awaiting the accept/reject decision
  int first;        // 1 = the pending trial is the initial evaluation (iteration 0): accept unconditionally
  int done;         // 0 = running, else termination reason + 1
  int iter;         // iterations started (excluding iteration 0)
  int successful;
  int chol_fail;    // diagnostics: number of non-PD factorisations
  int spec_discard; // 1 = the pending (explicit) trial replaces a speculative Gauss-Newton trial that was evaluated and turned out to lie
                    // outside the trust region: an evaluation the reference never makes.  The IMU terms take back what that
                    // evaluation did to their preintegration (ba_imu.hpp, imu_factor) before they look at the new trial.  (Here, in
                    // the line of the record that every small-factor workgroup reads anyway.)
  double radius, decrease_factor;
  double cost;          // cost at the accepted state
  double lambda;        // 1/radius used for the pending step
  double gd_p, ddd_p, step2_p, x2_p;  // pose/speed-bias part of g.delta, delta^T D^2 delta, |delta|^2, |x|^2
  double initial_cost, abs_grad_tol, grad_max;
  double last_rho, last_model_change;
  // ---- dogleg strategy (OptD::dogleg) ----
  // The trial after a fresh Gauss-Newton solve is launched SPECULATIVELY as the Gauss-Newton point itself (kind 0):
  // whether it lies inside the trust region is only known once the landmark part of its norm has been reduced by
  // the linearise kernel.  The next decision either judges it (inside) or replaces it by an explicit dogleg step
  // delta = -cA xv + beta dGN (kind 1), which is also what follows every rejected step (Ceres' reuse_).
  int tr_kind;        // kind of the pending trial: 0 = Gauss-Newton point / LM step, 1 = explicit coefficients
  int explicit_next;  // the decision just taken asks for an explicit dogleg step (no new Schur reduce / factorisation):
                      // 1 = after a rejected step (a new iteration), 2 = redo of a mis-speculated Gauss-Newton trial (same iteration)
  int invalid_steps;  // consecutive invalid steps
  int max_iter;       // iteration budget of this optimize call (slots spent on mis-speculation do not count)
  double mu;          // regularisation multiplier of the Gauss-Newton solve
  double cA, beta;    // coefficients of the pending explicit trial
  double dl_norm;     // dogleg_step_norm_ of the pending explicit trial
  double pend_model;  // model cost change of the pending explicit trial
  double tot_C, tot_E;  // g.dGN and |gnhat|^2 of the accepted point (pose + landmark parts), valid when have_tot
  double tot_A;       // (diagnostic) |ghat|^2 of the last explicit step
  int have_tot, pad2;
};
static_assert(sizeof(Ctrl) % 8 == 0 && sizeof(Ctrl) / 8 <= 64, "Ctrl is fetched by one wave, one double per lane");
// The control records of a solver's windows sit in ONE array behind its window records (okvis_ba_upload), one 256-byte slot
// each (no two workgroups write the same lines): the solve kernel computes the address of its record from a kernel argument
// and blockIdx.x, so the record — and with it the index of the linearisation buffer every speculative load depends on —
// arrives together with the window record instead of one memory round trip behind it.  WinPtrs::ctrl points to the same slot.
struct alignas(256) CtrlSlot {
  Ctrl c;
};
static_assert(sizeof(CtrlSlot) == 256, "one slot per window");

// On the device every pointer of the record refers to HBM: typed as global-address-space pointers, their accesses are
// global_load / global_store (tracked by the vector-memory counter only) instead of FLAT instructions.
#if defined(__HIP_DEVICE_COMPILE__)
#define BA_G __attribute__((address_space(1)))
#else
#define BA_G
#endif
// (64-byte aligned: a record is exactly 18 lines of the scalar cache, which the solve kernel requests in one go)
struct alignas(64) WinPtrs {
  // ---- sizes ----
  int n_pose, n_sb, n_lm, n_cam, n_obs, n_imu, n_pprior, n_sbprior, n_rel;
  int marg_dim, marg_nb;
  int D, Dp;              // reduced dimension; leading part that belongs to pose blocks
  int n_pair, n_group, n_chunk, n_task;
  int has_ext;            // any non-fixed extrinsics-role block
  int gpart_size;         // doubles in gpart
  int n_tile;             // Schur tiles per dimension
  int n_imu_color;
  int ct_nT;              // tile rows of the tiled dense solver (0 = the LDS solver handles this window)
  int spart_stride;       // doubles per chunk partial: (Dp/6)(Dp/6+1)/2*36 + 3*Dp  (S | Y b | g | diag U)
  int fuse_fast;          // fused mode: the groups of this window qualify for the matrix-core reduction (ba_linearize.hpp)
  int spart_buf_stride;   // doubles between the partials of linearisation buffer 0 and 1 (fused mode: one set per buffer); 0 = one set
  int lin2;               // the index lists are those of the piece path (ba_linearize2.hpp)
  unsigned ldl_comp;      // bit b: diagonal block b of the solver's ordering (ba_ldl16.hpp) holds columns of a pose prior or of the
                          // marginalisation prior and is eliminated with compensated products (D <= MAX_D_LDS; 0 otherwise)
  int chain;              // > 0: the reduced system is laid out for and solved by ba_chain.hpp (speed/bias blocks eliminated along the
                          // IMU chain first); the value is the number of speed/bias blocks.  0: the dense blocked LDL^T (ba_ldl16.hpp)
  double cauchy_b;
  ImuParamsD imu;

  // ---- state (index = buffer 0/1) ----
  BA_G double* pose[2];
  BA_G double* sb[2];
  BA_G double* lm[2];
  const BA_G int* pose_off;    // reduced offset or -1 (fixed)
  const BA_G int* sb_off;

  // ---- structure ----
  const BA_G double* cam_intr;
  const BA_G int* cam_model;
  const BA_G ObsRec* obs;
  const BA_G Group* groups;
  const BA_G int* pair_lm;     // [n_pair]
  const BA_G int* pair_off;    // [n_pair] reduced offset of the pair's block
  const BA_G int* pair_role;   // [n_pair] 0 = pose role, 1 = extrinsics role
  const BA_G int* pair_list_begin;  // [n_pair+1] into pair_list
  const BA_G uint16_t* pair_list;   // group-local observation indices
  const BA_G int* lm_pair_begin;    // [n_lm+1]
  const BA_G int* lm_obs_begin;     // [n_lm+1] observation range of each landmark (sorted order)
  // piece path (ba_linearize2.hpp)
  const BA_G int* lm_piece_begin;   // [n_lm+1] window-wide piece index of each landmark's first piece
  const BA_G int* pair_piece;       // [n_pair] group-local first piece | piece count << 16
  const BA_G int* pair_block;       // [n_pair] pose index of the pair's block
  const BA_G Task* tasks;
  const BA_G uint16_t* task_list;
  const BA_G Chunk* chunks;
  // per-chunk lists of the per-group partials (task 'out' offsets) that sum into each pose block / cross block
  const BA_G int* chunk_diag_begin;   // [n_chunk * (Dp/6) + 1] into chunk_diag_out
  const BA_G int* chunk_diag_out;
  const BA_G int* chunk_cross_begin;  // [n_chunk + 1] into chunk_cross (triples off_a, off_b, out)
  const BA_G int* chunk_cross;
  const BA_G int* lm_tile_begin;      // [n_lm][n_tile + 1] (only with n_tile > 1): first pair of landmark l whose block lies in Schur tile t or beyond
  const BA_G int* chunk_desc;         // [n_chunk][SCHUR_DESC_INTS]: lm_begin, lm_end, then lm_pair_begin at every 4th landmark of the chunk (ba_schur2.hpp)
  const BA_G int* imu_order;          // [n_imu] factor indices sorted by colour (one colour shares no parameter block)
  const BA_G int* imu_color_begin;    // [n_imu_color+1]
  const BA_G int* imu_coloff;         // [n_imu][30] reduced index of each local column (or -1)
  const BA_G int2* imu_rev;           // large windows (matrix in HBM): per entry of the block-packed matrix the (up to two) IMU record
                                 // entries f * 512 + e that land there, or -1: gathered by large_export_kernel
  const BA_G int4* imu_asm;           // [n_imu][512] where entry e of factor f's H|g record lands in the solve kernel's LDS
                                 // system: x = offset | is_g << 20 | colour << 24 (or -1), y = d2 index or -1 (D <= MAX_D_LDS)
  const BA_G int* imu_fastw;          // [n_imu][512] the same for the solve kernel's prefetch (D <= MAX_D_LDS), one word per entry:
                                 // offset in the kernel's dynamic LDS (doubles: the matrix in the layout of ba_ldl16.hpp, the gradient
                                 // behind it) | (d2 index + 1) << 16 | colour << 24, or -1
  const BA_G int* imu_pos;            // [n_imu][512] where logical entry e of a factor's H | g part (H 30x30 lower packed a(a+1)/2+b, then g) sits in
                                 // its record: the order of the entries' places in the solve kernel's LDS system (identity for D > MAX_D_LDS)
  const BA_G int* prior_col;          // [6 n_pprior | 9 n_sbprior] reduced index of every column of the pose priors, then of the
                                 // speed/bias priors, or -1 (fixed block)

  // ---- linearisation (index = buffer 0/1) ----
  BA_G double* V[2];           // [n_lm][6]
  BA_G double* bl[2];          // [n_lm][3]
  BA_G double* Hq[2];          // [n_lm][6]
  BA_G double* W[2];           // [n_pair][18]
  BA_G double* gpart[2];       // task outputs
  BA_G double* gscal[2];       // [n_group][GS_COUNT]
  BA_G double* imu_lin[2];     // [n_imu][IMU_LIN_STRIDE]
  BA_G double* pp_lin[2];      // [n_pprior][36 J + 6 r]
  BA_G double* sbp_lin[2];     // [n_sbprior][9 r]
  BA_G double* rel_lin[2];     // [n_rel][36 J0 + 36 J1 + 6 r]
  BA_G double* marg_lin_e[2];  // [marg_dim] e = e0 + J dchi
  BA_G double* marg_lin_M[2];  // [marg_nb][9] rotation blocks oplus(q (x) q_lin^-1)[0:3,0:3]
  BA_G double* small_cost[2];  // [2]: cost of priors+marg (one workgroup), spare
  BA_G double* obs_r[2];       // [n_obs][2] (debug/parity)

  // ---- Schur / solve ----
  BA_G double* spart;          // [1 or 2][n_chunk][spart_stride]: block-packed lower triangle | Y b
  BA_G double* spart_sum;      // [spart_stride] the sum of the chunk partials, made by the helper workgroups of the solve launch
  BA_G int* sum_sync;          // [0] number of helper workgroups that have delivered (all launches), [1] solve launches so far
  BA_G double* dec;            // [DEC_COUNT] the accept / reject decision the Schur kernel took on the pending trial (same function, same
                               // inputs as the solve kernel would use): [0] valid, [1..6] trial sums, then the DecisionDL / Decision fields
  BA_G double* S;              // [D][D] debug copy of the damped reduced matrix
  BA_G double* Sg;             // block-packed reduced matrix workspace in HBM when D > MAX_D_LDS, else null
  // tiled multi-workgroup solver of the large windows (ba_chol_tiles.hpp), null when D <= MAX_D_LDS
  BA_G double* ct_T;           // lower 48x48 tiles
  BA_G double* ct_Linv;        // inverses of the diagonal tiles
  BA_G double* ct_rhs;         // [48 nT]
  BA_G double* ct_y;           // [48 nT]
  BA_G double* ct_x;           // [48 nT] solution of the tiled solver (sentinel until a value is final, ba_chol_tiles.hpp)
  BA_G int* ct_flag;           // [ntiles] done flags, [ntiles] failure, [ntiles + 1] 1 = a system was exported this iteration,
                          // [ntiles + 2] 1 = the HBM matrix Sg is all zero (cleared by the export)
  BA_G double* ct_g;           // [48 nT] gradient of the accepted linearisation (for the step scalars)
  BA_G double* ct_d2;          // [48 nT] damping diagonal
  BA_G double* rhs;            // [D]
  BA_G double* step;           // [D] reduced step of the last solve (dogleg: the Gauss-Newton point dGN)
  BA_G double* scale_p;        // [D] Jacobi scale of the pose/speed-bias columns (first linearisation of the call)
  BA_G double* lm_scale;       // [n_lm][3] Jacobi scale of the landmark columns
  BA_G double* grad;           // [D]
  BA_G double* Dp2;            // [D]
  BA_G double* Hpp;            // [D][D] undamped U (debug/parity), optional
  BA_G double* quality;        // [n_lm]
  BA_G double* results;        // [7 n_pose + 9 n_sb + 4 n_lm + n_lm + 9 n_imu] packed by pack_results_kernel for okvis_ba_fetch_results
  BA_G double* prof;           // [64] clock64() phase stamps of workgroup 0 (diagnostics)
  BA_G Ctrl* ctrl;

  // ---- IMU ----
  const BA_G int* imu_pose0; const BA_G int* imu_sb0; const BA_G int* imu_pose1; const BA_G int* imu_sb1;
  const long long* imu_t0; const long long* imu_t1;
  const BA_G int* imu_s_begin; const BA_G int* imu_s_count;
  const long long* imu_s_t; const BA_G double* imu_s_gyr; const BA_G double* imu_s_acc;
  BA_G ImuCacheD* imu_cache;
  BA_G ImuCacheD* imu_cache_prev;   // [n_imu] the record as it was before the last evaluation re-preintegrated it (valid = 0: it did not)

  // ---- priors ----
  const BA_G int* pprior_pose; const BA_G double* pprior_meas; const BA_G double* pprior_sqrtinfo;
  const BA_G int* sbprior_sb; const BA_G double* sbprior_meas; const BA_G double* sbprior_sqrtinfo;
  const BA_G int* rel_pose0; const BA_G int* rel_pose1; const BA_G double* rel_sqrtinfo;
  const BA_G int* marg_block_type; const BA_G int* marg_block_idx; const BA_G int* marg_block_off;
  const BA_G double* marg_J; const BA_G double* marg_H0; const BA_G double* marg_e0; const BA_G double* marg_lin;
};

}  // namespace ba
