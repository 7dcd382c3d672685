/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C API of the CPU restatement of the okvis_ceres optimisation path (okvis::Estimator::optimize,
 * reference okvis_ceres/src/Estimator.cpp:843-906).  Used ONLY by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker; the product (okvis_amd/) never includes or links it.
 *
 * PARITY STATUS: pinned at the factor / linearisation / marginalisation level against the reference's own code:
 * oracle/_ref/libokvis_ref.so is built from the reference's unmodified sources (oracle/ref/Makefile; Eigen, Ceres,
 * glog and OpenCV are replaced by the stand-in headers of oracle/shim because none is installed) and
 * tests/test_oracle_vs_ref.py holds every restated factor, the whole-window normal equations, the landmark quality
 * and the MarginalizationError numerics to <= 1e-12 of it (b0 / J^T e0 of the marginalisation: 1e-10 / 1e-9, they
 * cancel digits).  NOT pinned: ::ceres::Solve itself (Ceres 1.9 is not in the tree) - the trust-region policy is a
 * restatement from Ceres' documentation, see DESIGN.md section 2.
 *
 * The window data format (okvis_ba_window, options, summary, array ids) is shared with the product's
 * public header include/okvis_amd_ba.h — data layout only, no code.
 */
#ifndef OKVIS_ORACLE_H_
#define OKVIS_ORACLE_H_

#include "../include/okvis_amd_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_window orc_window;

/* ---- factor-level entry points (each cites the reference in orc_factors.hpp) ---- */
void orc_pose_plus(const double x[7], const double delta[6], double out[7]);
void orc_pose_minus(const double x[7], const double xpd[7], double delta[6]);
void orc_pose_lift_jacobian(const double x[7], double J_6x7[42]);
void orc_pose_plus_jacobian(const double x[7], double J_7x6[42]);
/* intr[12] = fu fv cu cv d0..d7; returns status bits: 1 = valid (not "behind"), 2 = defined */
int orc_reprojection(const double pose[7], const double point[4], const double extr[7],
                     const double intr[12], int model, const double uv[2], const double sqrtInfo[4],
                     double r[2], double* Jp_2x6, double* Jl_2x3, double* Je_2x6);
int orc_project(const double intr[12], int model, const double point[3], double kp[2], double* J_2x3);
/* IMU: stateless helper that preintegrates then evaluates once (fresh cache, like the first
 * Evaluate call of a new ImuError).  Returns number of integration steps. */
int orc_imu_evaluate_fresh(int n, const int64_t* t, const double* gyr, const double* acc,
                           const okvis_ba_imu_params* p, int64_t t0, int64_t t1, const double pose0[7],
                           const double sb0[9], const double pose1[7], const double sb1[9], double r[15],
                           double* J0_15x6, double* J1_15x9, double* J2_15x6, double* J3_15x9,
                           double* sqrtInfo_15x15);
/* same but the cache is linearised at sb_ref (redo at sb_ref first), then evaluated at sb0 WITHOUT
 * allowing a redo — exposes the first-order bias correction path (ImuError.cpp:564-601). */
int orc_imu_evaluate_at_ref(int n, const int64_t* t, const double* gyr, const double* acc,
                            const okvis_ba_imu_params* p, int64_t t0, int64_t t1, const double sb_ref[9],
                            const double pose0[7], const double sb0[9], const double pose1[7],
                            const double sb1[9], double r[15], double* J0, double* J1, double* J2,
                            double* J3);
int orc_imu_propagation(int n, const int64_t* t, const double* gyr, const double* acc,
                        const okvis_ba_imu_params* p, double T_WS[7], double sb[9], int64_t t_start,
                        int64_t t_end, double* cov_15x15, double* jac_15x15);
void orc_pose_error(const double pose[7], const double meas[7], const double sqrtInfo[36], double r[6],
                    double* J_6x6);
void orc_speedbias_error(const double sb[9], const double meas[9], const double sqrtInfo[81], double r[9],
                         double* J_9x9);
void orc_relative_pose_error(const double p0[7], const double p1[7], const double sqrtInfo[36],
                             double r[6], double* J0_6x6, double* J1_6x6);
/* squareRootInformation_ = LLT(information).matrixL().transpose() with Eigen's early-exit behaviour */
void orc_sqrt_information(const double* info, int n, double* out);

/* ---- window-level ---- */
orc_window* orc_window_create(const okvis_ba_window* w);
void orc_window_destroy(orc_window* h);
/* CPU-baseline mode (bench.py): OpenMP threads for the observation sweep and the landmark Schur reduction; 1 = the
 * serial reference order used by every parity test */
int orc_set_threads(int n);
int orc_max_threads(void);
/* marginalisation-prior Jacobian convention: 1 (default) = what Ceres effectively uses,
 * J_min * lift(x_lin) * plusJacobian(x); 0 = constant J_min columns (MarginalizationError.cpp:904-938) */
void orc_window_set_marg_exact(orc_window* h, int exact);
int orc_window_reduced_dim(orc_window* h);
int orc_window_pair_count(orc_window* h);
void orc_window_pairs(orc_window* h, int32_t* pair_lm, int32_t* pair_block);
/* evaluate + linearise at the current state; returns cost */
double orc_window_linearize(orc_window* h);
/* cost only at the current state */
double orc_window_cost(orc_window* h);
/* Schur-reduce + solve with damping 1/radius on the current linearisation; returns 0 ok, 1 not PD */
int orc_window_solve(orc_window* h, double radius, const okvis_ba_options* opt);
/* full LM loop == product okvis_ba_optimize */
void orc_window_optimize(orc_window* h, const okvis_ba_options* opt, int num_iter,
                         okvis_ba_summary* summary);
/* run exactly n LM iterations with all tolerances disabled; returns wall seconds (CPU baseline) */
double orc_window_time_iterations(orc_window* h, const okvis_ba_options* opt, int n);
void orc_window_get_state(orc_window* h, double* pose, double* sb, double* lm);
void orc_window_set_state(orc_window* h, const double* pose, const double* sb, const double* lm);
int64_t orc_window_array_size(orc_window* h, int which);
int orc_window_download(orc_window* h, int which, double* out, int64_t n);
/* dense numeric check helper: full (un-reduced) gradient entry count = D + 3*n_lm */
void orc_window_full_gradient(orc_window* h, double* g_full);
/* MarginalizationError numerics, re-stated literally on the FULL (dense + landmark) matrix
 * (okvis_ceres/src/MarginalizationError.cpp:127-435 addResidualBlock, :507-802 marginalizeOut, :806-846
 * updateErrorComputation): linearise every residual of the window at its current values, add the previous
 * prior (H_, b0_), eliminate all landmarks and the flagged blocks, compute J and e0.  Same argument structs as
 * okvis_ba_marginalize. */
int orc_window_marginalize(orc_window* h, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* res);
/* symmetric eigen-decomposition used above (cyclic Jacobi): A [n*n] row-major in, eigenvalues out[n],
 * eigenvectors as columns of Q [n*n] */
void orc_sym_eig(const double* A, int n, double* eigenvalues, double* Q);

#ifdef __cplusplus
}
#endif
#endif
