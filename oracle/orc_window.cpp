// ORACLE — TEST INFRASTRUCTURE ONLY (see okvis_oracle.h header).
//
// Window-level CPU restatement: evaluates every error term of one sliding window exactly as Ceres would
// call them from okvis::Estimator::optimize (Estimator.cpp:843-877), applies the Cauchy corrector
// (semantics mirrored at MarginalizationError.cpp:325-365), eliminates the landmark blocks by Schur
// complement (SPARSE_SCHUR, Estimator.cpp:854; same algebra in-tree at MarginalizationError.cpp:617-689)
// and runs the documented trust-region policy (DESIGN.md "solver policy": Ceres-1.9
// LevenbergMarquardtStrategy semantics).  Plain loops, one thread.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>
#include <map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif
#include "okvis_oracle.h"
#include "orc_factors.hpp"

using namespace orc;

static int g_threads = 1;  // > 1: OpenMP CPU-baseline mode (bench.py only), see orc_set_threads

struct orc_window {
  // ---- copied structure ----
  int n_pose, n_sb, n_lm, n_cam, n_obs, n_imu, n_pprior, n_sbprior, n_relpose;
  std::vector<real> pose, sb, lm;
  std::vector<uint8_t> pose_fixed, sb_fixed;
  std::vector<Camera> cams;
  std::vector<int> obs_lm, obs_pose, obs_ext, obs_cam;
  std::vector<real> obs_uv, obs_sqrtw;
  real cauchy_b;
  std::vector<int> imu_pose0, imu_sb0, imu_pose1, imu_sb1, imu_s_begin, imu_s_count;
  std::vector<int64_t> imu_t0, imu_t1, imu_s_t;
  std::vector<real> imu_s_gyr, imu_s_acc;
  ImuParams imu_params;
  std::vector<ImuCache> imu_cache;
  std::vector<int> pprior_pose, sbprior_sb, rel_pose0, rel_pose1;
  std::vector<real> pprior_meas, pprior_sqrtinfo, sbprior_meas, sbprior_sqrtinfo, rel_sqrtinfo;
  int marg_dim, marg_nblocks;
  std::vector<int> marg_block_type, marg_block_idx, marg_block_off;
  std::vector<real> marg_J, marg_e0, marg_lin;
  bool marg_exact = true;

  // ---- derived ordering ----
  int D = 0;
  std::vector<int> pose_off, sb_off;
  std::vector<int> pair_lm, pair_block, lm_pair_begin, obs_pair_p, obs_pair_e, lm_obs_begin;
  int n_pair = 0;

  // ---- linearisation at the accepted state ----
  std::vector<real> V, b, Hq, W, U, g, obs_r, imu_r, quality;
  real cost = 0;
  // ---- last solve ----
  std::vector<real> S, rhs, step_p, step_l, Dp2, Dl2;
  real lambda = 0;

  ImuSamples samples(int f) const {
    ImuSamples s;
    s.n = imu_s_count[f];
    s.t = imu_s_t.data() + imu_s_begin[f];
    s.gyr = imu_s_gyr.data() + 3 * imu_s_begin[f];
    s.acc = imu_s_acc.data() + 3 * imu_s_begin[f];
    return s;
  }
};

namespace {

template <class T>
std::vector<T> copyv(const T* p, size_t n) {
  return p ? std::vector<T>(p, p + n) : std::vector<T>(n);
}
std::vector<real> copyr(const double* p, size_t n) { return p ? std::vector<real>(p, p + n) : std::vector<real>(n); }

void build_ordering(orc_window* h) {
  h->pose_off.assign(h->n_pose, -1);
  h->sb_off.assign(h->n_sb, -1);
  int off = 0;
  for (int i = 0; i < h->n_pose; ++i)
    if (!h->pose_fixed[i]) {
      h->pose_off[i] = off;
      off += 6;
    }
  for (int i = 0; i < h->n_sb; ++i)
    if (!h->sb_fixed[i]) {
      h->sb_off[i] = off;
      off += 9;
    }
  h->D = off;
  // (landmark, free block) pairs, sorted by landmark then block index
  h->lm_pair_begin.assign(h->n_lm + 1, 0);
  std::vector<std::vector<int>> blocks(h->n_lm);
  for (int o = 0; o < h->n_obs; ++o) {
    int l = h->obs_lm[o];
    int cand[2] = {h->obs_pose[o], h->obs_ext[o]};
    for (int c = 0; c < 2; ++c)
      if (!h->pose_fixed[cand[c]] &&
          std::find(blocks[l].begin(), blocks[l].end(), cand[c]) == blocks[l].end())
        blocks[l].push_back(cand[c]);
  }
  h->pair_lm.clear();
  h->pair_block.clear();
  for (int l = 0; l < h->n_lm; ++l) {
    std::sort(blocks[l].begin(), blocks[l].end());
    h->lm_pair_begin[l] = (int)h->pair_lm.size();
    for (int bk : blocks[l]) {
      h->pair_lm.push_back(l);
      h->pair_block.push_back(bk);
    }
  }
  h->lm_pair_begin[h->n_lm] = (int)h->pair_lm.size();
  h->n_pair = (int)h->pair_lm.size();
  h->lm_obs_begin.assign(h->n_lm + 1, 0);
  for (int o = 0; o < h->n_obs; ++o) h->lm_obs_begin[h->obs_lm[o] + 1]++;
  for (int l = 0; l < h->n_lm; ++l) h->lm_obs_begin[l + 1] += h->lm_obs_begin[l];
  h->obs_pair_p.assign(h->n_obs, -1);
  h->obs_pair_e.assign(h->n_obs, -1);
  for (int o = 0; o < h->n_obs; ++o) {
    int l = h->obs_lm[o];
    for (int p = h->lm_pair_begin[l]; p < h->lm_pair_begin[l + 1]; ++p) {
      if (h->pair_block[p] == h->obs_pose[o]) h->obs_pair_p[o] = p;
      if (h->pair_block[p] == h->obs_ext[o]) h->obs_pair_e[o] = p;
    }
  }
}

// accumulate a factor's J^T J and J^T r into the dense pose-side system
void add_factor_to(real* U, real* g, int D, int nres, const real* r, int nb, const int* off, const int* dim,
                   const real* const* J);
void add_factor(orc_window* h, int nres, const real* r, int nb, const int* off, const int* dim,
                const real* const* J) {
  add_factor_to(h->U.data(), h->g.data(), h->D, nres, r, nb, off, dim, J);
}
void add_factor_to(real* U, real* g, int D, int nres, const real* r, int nb, const int* off, const int* dim,
                   const real* const* J) {
  for (int a = 0; a < nb; ++a) {
    if (off[a] < 0) continue;
    for (int i = 0; i < dim[a]; ++i) {
      real s = 0;
      for (int k = 0; k < nres; ++k) s += J[a][k * dim[a] + i] * r[k];
      g[off[a] + i] += s;
    }
    for (int bb = 0; bb < nb; ++bb) {
      if (off[bb] < 0) continue;
      for (int i = 0; i < dim[a]; ++i)
        for (int j = 0; j < dim[bb]; ++j) {
          real s = 0;
          for (int k = 0; k < nres; ++k) s += J[a][k * dim[a] + i] * J[bb][k * dim[bb] + j];
          U[(size_t)(off[a] + i) * D + off[bb] + j] += s;
        }
    }
  }
}

// reprojection residuals o0 <= o < o1 (ReprojectionError + CauchyLoss, implementation/Estimator.hpp:68-82); the
// landmark-side sums (V, b, Hq, W) go to the window (observations are sorted by landmark: ranges cut at landmark
// boundaries never share an entry), the pose-side sums to U / g of the caller (per thread in the OpenMP baseline)
static void reprojection_range(orc_window* h, bool lin, int o0, int o1, real* U, real* g, real* cost) {
  static const int ut[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  for (int o = o0; o < o1; ++o) {
    const int l = h->obs_lm[o], ip = h->obs_pose[o], ie = h->obs_ext[o];
    const real w = h->obs_sqrtw[o];
    const real sqrtInfo[4] = {w, 0, 0, w};
    ReprojOut out;
    reprojection_error(&h->pose[7 * ip], &h->lm[4 * l], &h->pose[7 * ie], h->cams[h->obs_cam[o]],
                       &h->obs_uv[2 * o], sqrtInfo, lin, &out);
    const real s = out.r[0] * out.r[0] + out.r[1] * out.r[1];
    real sr = 1.0;
    if (h->cauchy_b > 0) {
      real rho[3];
      cauchy_loss(h->cauchy_b, s, rho);
      *cost += 0.5 * rho[0];
      sr = std::sqrt(rho[1]);  // Corrector: rho'' <= 0 -> residual_scaling = sqrt(rho'), alpha = 0
    } else {
      *cost += 0.5 * s;
    }
    if (!lin) continue;
    h->obs_r[2 * o] = out.r[0];
    h->obs_r[2 * o + 1] = out.r[1];
    // un-robustified landmark Hessian (Map::getLhs, Map.cpp:101-156)
    for (int e = 0; e < 6; ++e)
      h->Hq[6 * l + e] += out.Jl(0, ut[e][0]) * out.Jl(0, ut[e][1]) + out.Jl(1, ut[e][0]) * out.Jl(1, ut[e][1]);
    const real rt[2] = {sr * out.r[0], sr * out.r[1]};
    Mat<2, 6> Jp = sr * out.Jp;
    Mat<2, 3> Jl = sr * out.Jl;
    Mat<2, 6> Je = sr * out.Je;
    for (int e = 0; e < 6; ++e)
      h->V[6 * l + e] += Jl(0, ut[e][0]) * Jl(0, ut[e][1]) + Jl(1, ut[e][0]) * Jl(1, ut[e][1]);
    for (int i = 0; i < 3; ++i) h->b[3 * l + i] += Jl(0, i) * rt[0] + Jl(1, i) * rt[1];
    const int pp = h->obs_pair_p[o], pe = h->obs_pair_e[o];
    if (pp >= 0)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) h->W[18 * pp + 3 * i + j] += Jp(0, i) * Jl(0, j) + Jp(1, i) * Jl(1, j);
    if (pe >= 0)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) h->W[18 * pe + 3 * i + j] += Je(0, i) * Jl(0, j) + Je(1, i) * Jl(1, j);
    const int off[2] = {h->pose_off[ip], h->pose_off[ie]};
    const int dim[2] = {6, 6};
    const real* J[2] = {Jp.a, Je.a};
    add_factor_to(U, g, h->D, 2, rt, 2, off, dim, J);
  }
}

// Evaluate all error terms at the current state.  lin=true additionally fills the linearisation.
real evaluate(orc_window* h, bool lin) {
  const int D = h->D;
  if (lin) {
    h->V.assign(6 * (size_t)h->n_lm, 0.0);
    h->b.assign(3 * (size_t)h->n_lm, 0.0);
    h->Hq.assign(6 * (size_t)h->n_lm, 0.0);
    h->W.assign(18 * (size_t)h->n_pair, 0.0);
    h->U.assign((size_t)D * D, 0.0);
    h->g.assign(D, 0.0);
    h->obs_r.assign(2 * (size_t)h->n_obs, 0.0);
    h->imu_r.assign(15 * (size_t)h->n_imu, 0.0);
  }
  real cost = 0;
  // ---- reprojection ----
  if (g_threads <= 1 || h->n_lm < 2 * g_threads) {
    reprojection_range(h, lin, 0, h->n_obs, h->U.data(), h->g.data(), &cost);
  } else {
#ifdef _OPENMP
    // CPU-baseline mode (orc_set_threads): landmark ranges per thread, private pose-side accumulators summed in thread
    // order.  Same arithmetic per observation; the order of the pose-side sums differs from the serial path.
    const int T = g_threads;
    std::vector<std::vector<real>> Ut(T), gt(T);
    std::vector<real> ct(T, 0.0);
#pragma omp parallel num_threads(T)
    {
      const int t = omp_get_thread_num();
      const int l0 = (int)((long long)h->n_lm * t / T), l1 = (int)((long long)h->n_lm * (t + 1) / T);
      if (lin) {
        Ut[t].assign((size_t)D * D, 0.0);
        gt[t].assign(D, 0.0);
      }
      reprojection_range(h, lin, h->lm_obs_begin[l0], h->lm_obs_begin[l1], Ut[t].data(), gt[t].data(), &ct[t]);
    }
    for (int t = 0; t < T; ++t) {
      cost += ct[t];
      if (lin) {
        for (size_t i = 0; i < (size_t)D * D; ++i) h->U[i] += Ut[t][i];
        for (int i = 0; i < D; ++i) h->g[i] += gt[t][i];
      }
    }
#endif
  }
  // ---- IMU ----
  for (int f = 0; f < h->n_imu; ++f) {
    real r[15], J0[90], J1[135], J2[90], J3[135];
    imu_evaluate(h->samples(f), h->imu_params, h->imu_t0[f], h->imu_t1[f], &h->imu_cache[f],
                 &h->pose[7 * h->imu_pose0[f]], &h->sb[9 * h->imu_sb0[f]], &h->pose[7 * h->imu_pose1[f]],
                 &h->sb[9 * h->imu_sb1[f]], r, lin ? J0 : nullptr, lin ? J1 : nullptr,
                 lin ? J2 : nullptr, lin ? J3 : nullptr);
    real s = 0;
    for (int k = 0; k < 15; ++k) s += r[k] * r[k];
    cost += 0.5 * s;
    if (!lin) continue;
    for (int k = 0; k < 15; ++k) h->imu_r[15 * f + k] = r[k];
    const int off[4] = {h->pose_off[h->imu_pose0[f]], h->sb_off[h->imu_sb0[f]],
                        h->pose_off[h->imu_pose1[f]], h->sb_off[h->imu_sb1[f]]};
    const int dim[4] = {6, 9, 6, 9};
    const real* J[4] = {J0, J1, J2, J3};
    add_factor(h, 15, r, 4, off, dim, J);
  }
  // ---- pose priors ----
  for (int f = 0; f < h->n_pprior; ++f) {
    real r[6], J[36];
    const int ip = h->pprior_pose[f];
    pose_error(&h->pose[7 * ip], &h->pprior_meas[7 * f], &h->pprior_sqrtinfo[36 * f], r, lin ? J : nullptr);
    real s = 0;
    for (int k = 0; k < 6; ++k) s += r[k] * r[k];
    cost += 0.5 * s;
    if (!lin) continue;
    const int off[1] = {h->pose_off[ip]};
    const int dim[1] = {6};
    const real* Jp[1] = {J};
    add_factor(h, 6, r, 1, off, dim, Jp);
  }
  // ---- speed/bias priors ----
  for (int f = 0; f < h->n_sbprior; ++f) {
    real r[9], J[81];
    const int is = h->sbprior_sb[f];
    speedbias_error(&h->sb[9 * is], &h->sbprior_meas[9 * f], &h->sbprior_sqrtinfo[81 * f], r, lin ? J : nullptr);
    real s = 0;
    for (int k = 0; k < 9; ++k) s += r[k] * r[k];
    cost += 0.5 * s;
    if (!lin) continue;
    const int off[1] = {h->sb_off[is]};
    const int dim[1] = {9};
    const real* Jp[1] = {J};
    add_factor(h, 9, r, 1, off, dim, Jp);
  }
  // ---- relative pose ----
  for (int f = 0; f < h->n_relpose; ++f) {
    real r[6], J0[36], J1[36];
    const int i0 = h->rel_pose0[f], i1 = h->rel_pose1[f];
    relative_pose_error(&h->pose[7 * i0], &h->pose[7 * i1], &h->rel_sqrtinfo[36 * f], r, lin ? J0 : nullptr,
                        lin ? J1 : nullptr);
    real s = 0;
    for (int k = 0; k < 6; ++k) s += r[k] * r[k];
    cost += 0.5 * s;
    if (!lin) continue;
    const int off[2] = {h->pose_off[i0], h->pose_off[i1]};
    const int dim[2] = {6, 6};
    const real* Jp[2] = {J0, J1};
    add_factor(h, 6, r, 2, off, dim, Jp);
  }
  // ---- marginalisation prior (MarginalizationError.cpp:867-946) ----
  if (h->marg_dim > 0) {
    const int Dm = h->marg_dim, nb = h->marg_nblocks;
    std::vector<real> dchi(Dm, 0.0), e(h->marg_e0);
    std::vector<std::vector<real>> Jb(nb);
    std::vector<int> off(nb), dim(nb);
    std::vector<const real*> Jp(nb);
    for (int i = 0; i < nb; ++i) {
      const int idx = h->marg_block_idx[i], o = h->marg_block_off[i];
      M3 Mrot = M3::Identity();
      if (h->marg_block_type[i] == OKVIS_BA_BLOCK_POSE) {
        dim[i] = 6;
        off[i] = h->pose_off[idx];
        if (off[i] >= 0) {  // fixed blocks are skipped by computeDeltaChi (:873,:888)
          pose_minus(&h->marg_lin[9 * i], &h->pose[7 * idx], &dchi[o]);
          if (h->marg_exact) {
            // what Ceres multiplies together: J_min * lift(x_lin) (:931-938) * plusJacobian(x)
            // = J_min * blkdiag(I, oplus(q (x) q_lin^-1)[0:3,0:3])
            Quat q{h->pose[7 * idx + 3], h->pose[7 * idx + 4], h->pose[7 * idx + 5], h->pose[7 * idx + 6]};
            const real* xl = &h->marg_lin[9 * i];
            Quat ql_inv{-xl[3], -xl[4], -xl[5], xl[6]};
            Mrot = qoplusMat(qmul(q, ql_inv)).block<3, 3>(0, 0);
          }
        }
      } else {
        dim[i] = 9;
        off[i] = h->sb_off[idx];
        if (off[i] >= 0)
          for (int k = 0; k < 9; ++k) dchi[o + k] = h->sb[9 * idx + k] - h->marg_lin[9 * i + k];
      }
      if (lin) {
        Jb[i].assign((size_t)Dm * dim[i], 0.0);
        for (int r = 0; r < Dm; ++r) {
          const real* Jr = &h->marg_J[(size_t)r * Dm + o];
          if (dim[i] == 6) {
            for (int k = 0; k < 3; ++k) Jb[i][r * 6 + k] = Jr[k];
            for (int k = 0; k < 3; ++k)
              Jb[i][r * 6 + 3 + k] = Jr[3] * Mrot(0, k) + Jr[4] * Mrot(1, k) + Jr[5] * Mrot(2, k);
          } else {
            for (int k = 0; k < 9; ++k) Jb[i][r * 9 + k] = Jr[k];
          }
        }
        Jp[i] = Jb[i].data();
      }
    }
    real s = 0;
    for (int r = 0; r < Dm; ++r) {
      real acc = 0;
      for (int c = 0; c < Dm; ++c) acc += h->marg_J[(size_t)r * Dm + c] * dchi[c];
      e[r] += acc;
      s += e[r] * e[r];
    }
    cost += 0.5 * s;
    if (lin) add_factor(h, Dm, e.data(), nb, off.data(), dim.data(), Jp.data());
  }
  if (lin) h->cost = cost;
  return cost;
}

inline real clampd(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

// inverse of a symmetric 3x3 given upper-tri (00,01,02,11,12,22) via cofactors
void inv3sym(const real v[6], real out[6]) {
  const real a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5];
  const real c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const real det = a * c00 + b * c01 + c * c02;
  const real id = 1.0 / det;
  out[0] = c00 * id;
  out[1] = c01 * id;
  out[2] = c02 * id;
  out[3] = (a * f - c * c) * id;
  out[4] = (b * c - a * e) * id;
  out[5] = (a * d - b * b) * id;
}

// dense Cholesky solve A x = b (A row-major n x n, SPD). returns false if not PD.
bool chol_solve(std::vector<real> A, int n, const std::vector<real>& b, std::vector<real>* x) {
  for (int k = 0; k < n; ++k) {
    real d = A[(size_t)k * n + k];
    for (int j = 0; j < k; ++j) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)k * n + k] = d;
    for (int i = k + 1; i < n; ++i) {
      real s = A[(size_t)i * n + k];
      for (int j = 0; j < k; ++j) s -= A[(size_t)i * n + j] * A[(size_t)k * n + j];
      A[(size_t)i * n + k] = s / d;
    }
  }
  std::vector<real> y(n);
  for (int i = 0; i < n; ++i) {
    real s = b[i];
    for (int j = 0; j < i; ++j) s -= A[(size_t)i * n + j] * y[j];
    y[i] = s / A[(size_t)i * n + i];
  }
  x->assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    real s = y[i];
    for (int j = i + 1; j < n; ++j) s -= A[(size_t)j * n + i] * (*x)[j];
    (*x)[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

// (H + lambda Dd) delta = -g via landmark Schur complement, Dd = diag(h->Dp2, h->Dl2) given by the caller.
// Returns false if S is not PD.
bool solve_damped(orc_window* h, real lambda) {
  const int D = h->D;
  h->lambda = lambda;
  h->S = h->U;
  h->rhs.assign(D, 0.0);
  for (int i = 0; i < D; ++i) {
    h->S[(size_t)i * D + i] += lambda * h->Dp2[i];
    h->rhs[i] = -h->g[i];
  }
  std::vector<real> Vinv(6 * (size_t)h->n_lm);
  auto reduce_range = [&](int l0, int l1, real* S_, real* rhs_) {
    for (int l = l0; l < l1; ++l) {
    real v[6];
    for (int e = 0; e < 6; ++e) v[e] = h->V[6 * l + e];
    v[0] += lambda * h->Dl2[3 * l + 0];
    v[3] += lambda * h->Dl2[3 * l + 1];
    v[5] += lambda * h->Dl2[3 * l + 2];
    real* vi = &Vinv[6 * l];
    inv3sym(v, vi);
    const real Vi[3][3] = {{vi[0], vi[1], vi[2]}, {vi[1], vi[3], vi[4]}, {vi[2], vi[4], vi[5]}};
    const int p0 = h->lm_pair_begin[l], p1 = h->lm_pair_begin[l + 1];
    for (int pa = p0; pa < p1; ++pa) {
      real Y[18];
      const real* Wa = &h->W[18 * pa];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j)
          Y[3 * i + j] = Wa[3 * i + 0] * Vi[0][j] + Wa[3 * i + 1] * Vi[1][j] + Wa[3 * i + 2] * Vi[2][j];
      const int oa = h->pose_off[h->pair_block[pa]];
      for (int i = 0; i < 6; ++i)
        rhs_[oa + i] += Y[3 * i] * h->b[3 * l] + Y[3 * i + 1] * h->b[3 * l + 1] + Y[3 * i + 2] * h->b[3 * l + 2];
      for (int pb = p0; pb < p1; ++pb) {
        const real* Wb = &h->W[18 * pb];
        const int ob = h->pose_off[h->pair_block[pb]];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j)
            S_[(size_t)(oa + i) * D + ob + j] -=
                Y[3 * i] * Wb[3 * j] + Y[3 * i + 1] * Wb[3 * j + 1] + Y[3 * i + 2] * Wb[3 * j + 2];
      }
    }
    }
  };
  if (g_threads <= 1 || h->n_lm < 2 * g_threads) {
    reduce_range(0, h->n_lm, h->S.data(), h->rhs.data());
  } else {
#ifdef _OPENMP
    const int T = g_threads;   // CPU-baseline mode: private partial reduced systems, summed in thread order
    std::vector<std::vector<real>> St(T), rt(T);
#pragma omp parallel num_threads(T)
    {
      const int t = omp_get_thread_num();
      St[t].assign((size_t)D * D, 0.0);
      rt[t].assign(D, 0.0);
      reduce_range((int)((long long)h->n_lm * t / T), (int)((long long)h->n_lm * (t + 1) / T), St[t].data(), rt[t].data());
    }
    for (int t = 0; t < T; ++t) {
      for (size_t i = 0; i < (size_t)D * D; ++i) h->S[i] += St[t][i];
      for (int i = 0; i < D; ++i) h->rhs[i] += rt[t][i];
    }
#endif
  }
  if (!chol_solve(h->S, D, h->rhs, &h->step_p)) return false;
  // back-substitution: delta_l = -Vinv (g_l + W^T delta_p)
  h->step_l.assign(3 * (size_t)h->n_lm, 0.0);
  for (int l = 0; l < h->n_lm; ++l) {
    real t[3] = {h->b[3 * l], h->b[3 * l + 1], h->b[3 * l + 2]};
    for (int p = h->lm_pair_begin[l]; p < h->lm_pair_begin[l + 1]; ++p) {
      const real* Wp = &h->W[18 * p];
      const real* dp = &h->step_p[h->pose_off[h->pair_block[p]]];
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 6; ++i) t[j] += Wp[3 * i + j] * dp[i];
    }
    const real* vi = &Vinv[6 * l];
    h->step_l[3 * l + 0] = -(vi[0] * t[0] + vi[1] * t[1] + vi[2] * t[2]);
    h->step_l[3 * l + 1] = -(vi[1] * t[0] + vi[3] * t[1] + vi[4] * t[2]);
    h->step_l[3 * l + 2] = -(vi[2] * t[0] + vi[4] * t[1] + vi[5] * t[2]);
  }
  return true;
}

// LevenbergMarquardtStrategy: D^2 = clamp(diag J^T J, min_lm_diagonal, max_lm_diagonal), damping D^2 / radius
bool solve(orc_window* h, real radius, const okvis_ba_options& opt) {
  h->Dp2.assign(h->D, 0.0);
  h->Dl2.assign(3 * (size_t)h->n_lm, 0.0);
  for (int i = 0; i < h->D; ++i) h->Dp2[i] = clampd(h->U[(size_t)i * h->D + i], opt.min_lm_diagonal, opt.max_lm_diagonal);
  static const int dg[3] = {0, 3, 5};
  for (int l = 0; l < h->n_lm; ++l)
    for (int k = 0; k < 3; ++k) h->Dl2[3 * l + k] = clampd(h->V[6 * l + dg[k]], opt.min_lm_diagonal, opt.max_lm_diagonal);
  return solve_damped(h, 1.0 / radius);
}

void apply_step(orc_window* h, std::vector<real>* pose, std::vector<real>* sb, std::vector<real>* lm) {
  *pose = h->pose;
  *sb = h->sb;
  *lm = h->lm;
  for (int i = 0; i < h->n_pose; ++i)
    if (h->pose_off[i] >= 0) pose_plus(&h->pose[7 * i], &h->step_p[h->pose_off[i]], &(*pose)[7 * i]);
  for (int i = 0; i < h->n_sb; ++i)
    if (h->sb_off[i] >= 0)
      for (int k = 0; k < 9; ++k) (*sb)[9 * i + k] = h->sb[9 * i + k] + h->step_p[h->sb_off[i] + k];
  for (int l = 0; l < h->n_lm; ++l)  // HomogeneousPointLocalParameterization::plus (:59-71)
    for (int k = 0; k < 3; ++k) (*lm)[4 * l + k] = h->lm[4 * l + k] + h->step_l[3 * l + k];
}

real gradient_max_norm(const orc_window* h) {
  real m = 0;
  for (real v : h->g) m = std::max(m, std::fabs(v));
  for (real v : h->b) m = std::max(m, std::fabs(v));
  return m;
}

// 3x3 symmetric eigenvalues by cyclic Jacobi (replaces Eigen::SelfAdjointEigenSolver<Matrix3d>)
void eig3sym(const real v[6], real ev[3]) {
  real A[3][3] = {{v[0], v[1], v[2]}, {v[1], v[3], v[4]}, {v[2], v[4], v[5]}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    real off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        real theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        real t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        real c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          real akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          real apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = A[0][0];
  ev[1] = A[1][1];
  ev[2] = A[2][2];
  std::sort(ev, ev + 3);
}

void landmark_quality(orc_window* h) {
  // Estimator.cpp:880-896
  h->quality.assign(h->n_lm, 0.0);
  for (int l = 0; l < h->n_lm; ++l) {
    real ev[3];
    eig3sym(&h->Hq[6 * l], ev);
    if (ev[0] < 1.0e-12)
      h->quality[l] = 0.0;
    else
      h->quality[l] = std::sqrt(ev[0]) / std::sqrt(ev[2]);
  }
}

void lm_loop(orc_window* h, const okvis_ba_options& opt, int num_iter, okvis_ba_summary* sum) {
  real radius = opt.initial_radius, decrease_factor = 2.0;
  evaluate(h, true);
  okvis_ba_summary s;
  std::memset(&s, 0, sizeof(s));
  s.initial_cost = h->cost;
  const real g0 = gradient_max_norm(h);
  s.gradient_max_norm = g0;
  const real abs_grad_tol = opt.gradient_tolerance * std::max(g0, real(2.220446049250313e-16));
  s.termination = 0;
  bool done = false;
  if (opt.gradient_tolerance > 0 && g0 <= abs_grad_tol) {
    s.termination = 2;
    done = true;
  }
  for (int it = 1; it <= num_iter && !done; ++it) {
    s.iterations = it;
    bool ok = solve(h, radius, opt);
    real model_change = 0;
    if (ok) {
      real gd = 0, dDd = 0;
      for (int i = 0; i < h->D; ++i) {
        gd += h->g[i] * h->step_p[i];
        dDd += h->Dp2[i] * h->step_p[i] * h->step_p[i];
      }
      for (size_t i = 0; i < h->step_l.size(); ++i) {
        gd += h->b[i] * h->step_l[i];
        dDd += h->Dl2[i] * h->step_l[i] * h->step_l[i];
      }
      model_change = -0.5 * gd + 0.5 * h->lambda * dDd;
      if (!(model_change > 0) && !opt.gauss_newton) ok = false;
    }
    if (!ok) {  // invalid step: treated like a rejected step
      radius /= decrease_factor;
      decrease_factor *= 2;
      if (radius < opt.min_radius) {
        s.termination = ok ? 4 : 5;
        done = true;
      }
      continue;
    }
    // parameter tolerance (checked before the trial evaluation)
    real step2 = 0, x2 = 0;
    for (real v : h->step_p) step2 += v * v;
    for (real v : h->step_l) step2 += v * v;
    for (int i = 0; i < h->n_pose; ++i)
      if (h->pose_off[i] >= 0)
        for (int k = 0; k < 7; ++k) x2 += h->pose[7 * i + k] * h->pose[7 * i + k];
    for (int i = 0; i < h->n_sb; ++i)
      if (h->sb_off[i] >= 0)
        for (int k = 0; k < 9; ++k) x2 += h->sb[9 * i + k] * h->sb[9 * i + k];
    for (real v : h->lm) x2 += v * v;
    if (opt.parameter_tolerance > 0 &&
        std::sqrt(step2) <= opt.parameter_tolerance * (std::sqrt(x2) + opt.parameter_tolerance)) {
      s.termination = 3;
      done = true;
      continue;
    }
    std::vector<real> pose_t, sb_t, lm_t, pose_s = h->pose, sb_s = h->sb, lm_s = h->lm;
    apply_step(h, &pose_t, &sb_t, &lm_t);
    h->pose = pose_t;
    h->sb = sb_t;
    h->lm = lm_t;
    const real old_cost = h->cost;
    const real new_cost = evaluate(h, false);
    const real rho = (old_cost - new_cost) / model_change;
    if (opt.gauss_newton || rho > opt.min_relative_decrease) {
      evaluate(h, true);  // Ceres re-evaluates residuals + Jacobians at the accepted point
      s.successful_steps++;
      if (!opt.gauss_newton) {
        const real t = 2.0 * rho - 1.0;
        radius = radius / std::max(real(1.0 / 3.0), 1.0 - t * t * t);
        radius = std::min(real(opt.max_radius), radius);
        decrease_factor = 2.0;
      }
      const real gm = gradient_max_norm(h);
      s.gradient_max_norm = gm;
      if (opt.gradient_tolerance > 0 && gm <= abs_grad_tol) {
        s.termination = 2;
        done = true;
      } else if (opt.function_tolerance > 0 && std::fabs(old_cost - new_cost) < opt.function_tolerance * old_cost) {
        s.termination = 1;
        done = true;
      }
    } else {
      h->pose = pose_s;
      h->sb = sb_s;
      h->lm = lm_s;
      radius /= decrease_factor;
      decrease_factor *= 2;
      if (radius < opt.min_radius) {
        s.termination = 4;
        done = true;
      }
    }
  }
  s.final_cost = h->cost;
  s.final_radius = radius;
  landmark_quality(h);
  if (sum) *sum = s;
}


// ---------------------------------------------------------------------------------------------------
// The reference's configured policy (Estimator.cpp:854-873): Ceres 1.9 TrustRegionMinimizer with
// trust_region_strategy_type = DOGLEG (dogleg_type = TRADITIONAL_DOGLEG by default), jacobi_scaling = true,
// linear solver SPARSE_SCHUR (exact: here the landmark Schur complement + dense Cholesky).  Ceres is not in the
// reference tree; this follows its documented algorithm (trust_region_minimizer.cc / dogleg_strategy.cc of 1.9,
// restated, not copied).  Written in UNSCALED variables; with s = Jacobi scale (from the first linearisation of
// this call), h_i = diag(J^T J)_i and d_i = sqrt(clamp(s_i^2 h_i, min_lm_diagonal, max_lm_diagonal)) (Ceres'
// `diagonal_` of the column-scaled Jacobian):
//   gradient_ (D-normalised, scaled)      ghat_i  = s_i g_i / d_i
//   Cauchy direction in unscaled space    xv_i    = s_i^2 g_i / d_i^2,  alpha = |ghat|^2 / |J xv|^2
//   Gauss-Newton point                    (H + mu diag(d^2/s^2)) dGN = -g,   gnhat_i = d_i dGN_i / s_i
//   step                                  delta = -cA xv + beta dGN  (dogleg interpolation in the hat space)
// ---------------------------------------------------------------------------------------------------
struct Full {  // a vector over [reduced pose/speed-bias part | 3 per landmark]
  std::vector<real> p, l;
};
static real dotf(const Full& a, const Full& b) {
  real s = 0;
  for (size_t i = 0; i < a.p.size(); ++i) s += a.p[i] * b.p[i];
  for (size_t i = 0; i < a.l.size(); ++i) s += a.l[i] * b.l[i];
  return s;
}
// y = H x with H = [U W; W^T V] of the current linearisation
static void hessian_times(const orc_window* h, const Full& x, Full* y) {
  const int D = h->D;
  y->p.assign(D, 0.0);
  y->l.assign(3 * (size_t)h->n_lm, 0.0);
  for (int i = 0; i < D; ++i) {
    real s = 0;
    for (int j = 0; j < D; ++j) s += h->U[(size_t)i * D + j] * x.p[j];
    y->p[i] = s;
  }
  for (int l = 0; l < h->n_lm; ++l) {
    const real* v = &h->V[6 * l];
    const real* xl = &x.l[3 * l];
    y->l[3 * l + 0] = v[0] * xl[0] + v[1] * xl[1] + v[2] * xl[2];
    y->l[3 * l + 1] = v[1] * xl[0] + v[3] * xl[1] + v[4] * xl[2];
    y->l[3 * l + 2] = v[2] * xl[0] + v[4] * xl[1] + v[5] * xl[2];
    for (int pr = h->lm_pair_begin[l]; pr < h->lm_pair_begin[l + 1]; ++pr) {
      const real* Wp = &h->W[18 * pr];  // 6x3 row-major block (pose block rows, landmark columns)
      const int o = h->pose_off[h->pair_block[pr]];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) {
          y->p[o + i] += Wp[3 * i + j] * xl[j];
          y->l[3 * l + j] += Wp[3 * i + j] * x.p[o + i];
        }
    }
  }
}
// Ceres 1.9: gradient_max_norm = || x - Plus(x, -g) ||_inf over the ambient coordinates
static real projected_gradient_max_norm(const orc_window* h) {
  real m = 0;
  for (int i = 0; i < h->n_pose; ++i)
    if (h->pose_off[i] >= 0) {
      real d[6], xp[7];
      for (int k = 0; k < 6; ++k) d[k] = -h->g[h->pose_off[i] + k];
      pose_plus(&h->pose[7 * i], d, xp);
      for (int k = 0; k < 7; ++k) m = std::max(m, std::fabs(h->pose[7 * i + k] - xp[k]));
    }
  for (int i = 0; i < h->n_sb; ++i)
    if (h->sb_off[i] >= 0)
      for (int k = 0; k < 9; ++k) m = std::max(m, std::fabs(h->g[h->sb_off[i] + k]));
  for (real v : h->b) m = std::max(m, std::fabs(v));
  return m;
}
static real free_x_norm(const orc_window* h) {
  real x2 = 0;
  for (int i = 0; i < h->n_pose; ++i)
    if (h->pose_off[i] >= 0)
      for (int k = 0; k < 7; ++k) x2 += h->pose[7 * i + k] * h->pose[7 * i + k];
  for (int i = 0; i < h->n_sb; ++i)
    if (h->sb_off[i] >= 0)
      for (int k = 0; k < 9; ++k) x2 += h->sb[9 * i + k] * h->sb[9 * i + k];
  for (real v : h->lm) x2 += v * v;
  return std::sqrt(x2);
}

void dogleg_loop(orc_window* h, const okvis_ba_options& opt, int num_iter, okvis_ba_summary* sum) {
  const real kMinMu = 1e-8, kMaxMu = 1.0, kMuIncrease = 10.0;  // DoglegStrategy constants
  const int D = h->D, NL = 3 * h->n_lm;
  static const int dg[3] = {0, 3, 5};
  real radius = opt.initial_radius, mu = kMinMu;
  bool reuse = false, gn_ok = false;
  real alpha = 0, dogleg_step_norm = 0;
  int invalid_steps = 0;
  evaluate(h, true);
  okvis_ba_summary s;
  std::memset(&s, 0, sizeof(s));
  s.initial_cost = h->cost;
  // Jacobi scaling, estimated once (TrustRegionMinimizer: EstimateScale at the initial point only)
  Full scale, dd, ghat, xv, gn, gnhat, step, g;
  scale.p.assign(D, 1.0);
  scale.l.assign(NL, 1.0);
  if (opt.jacobi_scaling) {
    for (int i = 0; i < D; ++i) scale.p[i] = 1.0 / (1.0 + std::sqrt(h->U[(size_t)i * D + i]));
    for (int l = 0; l < h->n_lm; ++l)
      for (int k = 0; k < 3; ++k) scale.l[3 * l + k] = 1.0 / (1.0 + std::sqrt(h->V[6 * l + dg[k]]));
  }
  s.gradient_max_norm = projected_gradient_max_norm(h);
  bool done = false;
  if (opt.gradient_tolerance > 0 && s.gradient_max_norm <= opt.gradient_tolerance) {
    s.termination = 2;
    done = true;
  }
  real x_norm = free_x_norm(h);
  for (int it = 1; it <= num_iter && !done; ++it) {
    s.iterations = it;
    // ---------------- DoglegStrategy::ComputeStep ----------------
    if (!reuse) {
      reuse = true;
      dd.p.assign(D, 0.0), dd.l.assign(NL, 0.0);
      g.p = h->g, g.l = h->b;
      for (int i = 0; i < D; ++i)
        dd.p[i] = std::sqrt(clampd(scale.p[i] * scale.p[i] * h->U[(size_t)i * D + i], opt.min_lm_diagonal, opt.max_lm_diagonal));
      for (int l = 0; l < h->n_lm; ++l)
        for (int k = 0; k < 3; ++k)
          dd.l[3 * l + k] = std::sqrt(clampd(scale.l[3 * l + k] * scale.l[3 * l + k] * h->V[6 * l + dg[k]],
                                             opt.min_lm_diagonal, opt.max_lm_diagonal));
      ghat = g, xv = g;
      for (int i = 0; i < D; ++i) {
        ghat.p[i] = scale.p[i] * g.p[i] / dd.p[i];
        xv.p[i] = scale.p[i] * (ghat.p[i] / dd.p[i]);
      }
      for (int i = 0; i < NL; ++i) {
        ghat.l[i] = scale.l[i] * g.l[i] / dd.l[i];
        xv.l[i] = scale.l[i] * (ghat.l[i] / dd.l[i]);
      }
      // Cauchy point: alpha * -gradient_
      Full Hxv;
      hessian_times(h, xv, &Hxv);
      alpha = dotf(ghat, ghat) / dotf(xv, Hxv);
      // Gauss-Newton step, regularised by mu * diagonal_^2 (in scaled variables) until the factorisation succeeds
      h->Dp2.assign(D, 0.0), h->Dl2.assign(NL, 0.0);
      for (int i = 0; i < D; ++i) h->Dp2[i] = (dd.p[i] / scale.p[i]) * (dd.p[i] / scale.p[i]);
      for (int i = 0; i < NL; ++i) h->Dl2[i] = (dd.l[i] / scale.l[i]) * (dd.l[i] / scale.l[i]);
      gn_ok = false;
      while (mu < kMaxMu) {
        bool ok = solve_damped(h, mu);
        if (ok) {
          for (real v : h->step_p) ok = ok && std::isfinite(v);
          for (real v : h->step_l) ok = ok && std::isfinite(v);
        }
        if (!ok) {
          mu *= kMuIncrease;
          continue;
        }
        gn_ok = true;
        break;
      }
      if (gn_ok) {
        gn.p = h->step_p, gn.l = h->step_l;
        gnhat = gn;
        for (int i = 0; i < D; ++i) gnhat.p[i] = dd.p[i] * gn.p[i] / scale.p[i];
        for (int i = 0; i < NL; ++i) gnhat.l[i] = dd.l[i] * gn.l[i] / scale.l[i];
      }
    }
    bool valid = gn_ok;
    real model_change = 0;
    if (gn_ok) {
      // ---------------- ComputeTraditionalDoglegStep ----------------
      real cA = 0, beta = 1;
      const real gradient_norm = std::sqrt(dotf(ghat, ghat)), gn_norm = std::sqrt(dotf(gnhat, gnhat));
      if (opt.gauss_newton || gn_norm <= radius) {  // case 1: the Gauss-Newton point lies inside the trust region
        cA = 0, beta = 1;
        dogleg_step_norm = gn_norm;
      } else if (gradient_norm * alpha >= radius) {  // case 2: even the Cauchy point lies outside
        cA = radius / gradient_norm, beta = 0;
        dogleg_step_norm = radius;
      } else {  // case 3: on the segment Cauchy point -> Gauss-Newton point
        const real b_dot_a = -alpha * dotf(ghat, gnhat);
        const real a2 = (alpha * gradient_norm) * (alpha * gradient_norm);
        const real bma2 = a2 - 2 * b_dot_a + gn_norm * gn_norm;
        const real c = b_dot_a - a2;
        const real dsc = std::sqrt(c * c + bma2 * (radius * radius - a2));
        beta = (c <= 0) ? (dsc - c) / bma2 : (radius * radius - a2) / (dsc + c);
        cA = alpha * (1.0 - beta);
        Full t = gnhat;
        for (int i = 0; i < D; ++i) t.p[i] = -cA * ghat.p[i] + beta * gnhat.p[i];
        for (int i = 0; i < NL; ++i) t.l[i] = -cA * ghat.l[i] + beta * gnhat.l[i];
        dogleg_step_norm = std::sqrt(dotf(t, t));
      }
      step = gn;
      for (int i = 0; i < D; ++i) step.p[i] = -cA * xv.p[i] + beta * gn.p[i];
      for (int i = 0; i < NL; ++i) step.l[i] = -cA * xv.l[i] + beta * gn.l[i];
      // model_cost_change = -(J step)^T (r + J step / 2) = -g^T step - step^T H step / 2
      Full Hs;
      hessian_times(h, step, &Hs);
      model_change = -dotf(g, step) - 0.5 * dotf(step, Hs);
      if (model_change < 0.0 && !opt.gauss_newton) valid = false;
      if (std::getenv("ORC_TRACE"))
        std::fprintf(stderr, "orc it %d radius %.17g mu %g A %.17g B %.17g C %.17g E %.17g alpha %.17g cA %.17g beta %.17g dl %.17g model %.17g\n",
                     it, radius, mu, dotf(ghat, ghat), dotf(ghat, ghat) / alpha, dotf(ghat, gnhat), dotf(gnhat, gnhat), alpha, cA,
                     beta, dogleg_step_norm, model_change);
    }
    if (!valid) {  // StepIsInvalid: counted as an unsuccessful iteration of zero length
      if (++invalid_steps >= std::max(1, opt.max_consecutive_invalid_steps)) {
        s.termination = 5;
        done = true;
        continue;
      }
      mu *= kMuIncrease;
      reuse = false;
      continue;
    }
    invalid_steps = 0;
    h->step_p = step.p, h->step_l = step.l;
    std::vector<real> pose_t, sb_t, lm_t, pose_s = h->pose, sb_s = h->sb, lm_s = h->lm;
    apply_step(h, &pose_t, &sb_t, &lm_t);
    real dx2 = 0;
    for (size_t i = 0; i < pose_t.size(); ++i) dx2 += (pose_t[i] - pose_s[i]) * (pose_t[i] - pose_s[i]);
    for (size_t i = 0; i < sb_t.size(); ++i) dx2 += (sb_t[i] - sb_s[i]) * (sb_t[i] - sb_s[i]);
    for (size_t i = 0; i < lm_t.size(); ++i) dx2 += (lm_t[i] - lm_s[i]) * (lm_t[i] - lm_s[i]);
    h->pose = pose_t, h->sb = sb_t, h->lm = lm_t;
    const real old_cost = h->cost;
    const real new_cost = evaluate(h, false);
    h->pose = pose_s, h->sb = sb_s, h->lm = lm_s;
    if (!opt.gauss_newton) {
      if (opt.parameter_tolerance > 0 && std::sqrt(dx2) <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
        s.termination = 3;  // returns WITHOUT taking the step
        done = true;
        continue;
      }
      if (opt.function_tolerance > 0 && std::fabs(old_cost - new_cost) < opt.function_tolerance * old_cost) {
        s.termination = 1;  // Ceres <= 1.10 returns here WITHOUT taking the step
        done = true;
        continue;
      }
    }
    const real rho = (old_cost - new_cost) / model_change;
    if (opt.gauss_newton || rho > opt.min_relative_decrease) {
      s.successful_steps++;
      if (!opt.gauss_newton) {  // StepAccepted
        if (rho < 0.25) radius *= 0.5;
        if (rho > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
        mu = std::max(kMinMu, 2.0 * mu / kMuIncrease);
      }
      reuse = false;
      h->pose = pose_t, h->sb = sb_t, h->lm = lm_t;
      x_norm = free_x_norm(h);
      evaluate(h, true);
      s.gradient_max_norm = projected_gradient_max_norm(h);
      if (opt.gradient_tolerance > 0 && s.gradient_max_norm <= opt.gradient_tolerance) {
        s.termination = 2;
        done = true;
      }
    } else {  // StepRejected: only the interpolation is redone with the smaller radius
      radius *= 0.5;
      reuse = true;
    }
    if (!done && radius < opt.min_radius) {
      s.termination = 4;
      done = true;
    }
  }
  s.final_cost = h->cost;
  s.final_radius = radius;
  landmark_quality(h);
  if (sum) *sum = s;
}

}  // namespace

// ===================================================================================================
// C API
// ===================================================================================================
// The entry points take and return double whatever `real` is: arguments are copied into `real` arrays, results copied back
// (exact in the fp64 build; the long double build rounds its results to double on the way out).
namespace {
struct In {
  std::vector<real> v;
  bool has;
  In(const double* p, size_t n) : has(p != nullptr) {
    if (p) v.assign(p, p + n);
  }
  operator const real*() const { return has ? v.data() : nullptr; }
};
struct Out {  // written back when the object goes out of scope; `init` for arguments that are read and written
  double* dst;
  std::vector<real> v;
  Out(double* p, size_t n, bool init = false) : dst(p), v(p ? n : 0) {
    if (p && init) v.assign(p, p + n);
  }
  Out(const Out&) = delete;
  operator real*() { return dst ? v.data() : nullptr; }
  ~Out() {
    if (dst)
      for (size_t i = 0; i < v.size(); ++i) dst[i] = (double)v[i];
  }
};
template <class M>
void put(double* dst, const M& m) {
  if (dst)
    for (size_t i = 0; i < sizeof(m.a) / sizeof(m.a[0]); ++i) dst[i] = (double)m.a[i];
}
void put(double* dst, const std::vector<real>& v) {
  if (dst)
    for (size_t i = 0; i < v.size(); ++i) dst[i] = (double)v[i];
}
}  // namespace

extern "C" {

void orc_pose_plus(const double x[7], const double d[6], double out[7]) { pose_plus(In(x, 7), In(d, 6), Out(out, 7)); }
void orc_pose_minus(const double x[7], const double xpd[7], double d[6]) { pose_minus(In(x, 7), In(xpd, 7), Out(d, 6)); }
void orc_pose_lift_jacobian(const double x[7], double J[42]) { pose_lift_jacobian(In(x, 7), Out(J, 42)); }
void orc_pose_plus_jacobian(const double x[7], double J[42]) { pose_plus_jacobian(In(x, 7), Out(J, 42)); }

static Camera make_cam(const double intr[12], int model) {
  Camera c;
  c.fu = intr[0];
  c.fv = intr[1];
  c.cu = intr[2];
  c.cv = intr[3];
  c.model = model;
  for (int i = 0; i < 8; ++i) c.d[i] = intr[4 + i];
  return c;
}

int orc_reprojection(const double pose[7], const double point[4], const double extr[7],
                     const double intr[12], int model, const double uv[2], const double sqrtInfo[4],
                     double r[2], double* Jp, double* Jl, double* Je) {
  ReprojOut out;
  reprojection_error(In(pose, 7), In(point, 4), In(extr, 7), make_cam(intr, model), In(uv, 2), In(sqrtInfo, 4), Jp || Jl || Je,
                     &out);
  r[0] = (double)out.r[0];
  r[1] = (double)out.r[1];
  put(Jp, out.Jp);
  put(Jl, out.Jl);
  put(Je, out.Je);
  return (out.valid ? 1 : 0) | (out.defined ? 2 : 0);
}

int orc_project(const double intr[12], int model, const double point[3], double kp[2], double* J) {
  Mat<2, 3> Jm;
  bool ok = project(make_cam(intr, model), vec3(point[0], point[1], point[2]), Out(kp, 2), J ? &Jm : nullptr);
  if (ok) put(J, Jm);
  return ok ? 1 : 0;
}

static ImuParams make_params(const okvis_ba_imu_params* p) {
  ImuParams q;
  q.sigma_g_c = p->sigma_g_c;
  q.sigma_a_c = p->sigma_a_c;
  q.sigma_gw_c = p->sigma_gw_c;
  q.sigma_aw_c = p->sigma_aw_c;
  q.g = p->g;
  q.g_max = p->g_max;
  q.a_max = p->a_max;
  return q;
}

int orc_imu_evaluate_fresh(int n, const int64_t* t, const double* gyr, const double* acc,
                           const okvis_ba_imu_params* p, int64_t t0, int64_t t1, const double pose0[7],
                           const double sb0[9], const double pose1[7], const double sb1[9], double r[15],
                           double* J0, double* J1, double* J2, double* J3, double* sqrtInfo) {
  In g(gyr, 3 * (size_t)n), a(acc, 3 * (size_t)n);
  ImuSamples s{n, t, g, a};
  ImuCache c;
  imu_evaluate(s, make_params(p), t0, t1, &c, In(pose0, 7), In(sb0, 9), In(pose1, 7), In(sb1, 9), Out(r, 15), Out(J0, 90),
               Out(J1, 135), Out(J2, 90), Out(J3, 135));
  put(sqrtInfo, c.sqrtInfo);
  return c.redoCounter;
}

int orc_imu_evaluate_at_ref(int n, const int64_t* t, const double* gyr, const double* acc,
                            const okvis_ba_imu_params* p, int64_t t0, int64_t t1, const double sb_ref[9],
                            const double pose0[7], const double sb0[9], const double pose1[7],
                            const double sb1[9], double r[15], double* J0, double* J1, double* J2,
                            double* J3) {
  In g(gyr, 3 * (size_t)n), a(acc, 3 * (size_t)n);
  ImuSamples s{n, t, g, a};
  ImuCache c;
  ImuParams prm = make_params(p);
  imu_redo_preintegration(s, prm, t0, t1, In(sb_ref, 9), &c);
  c.redo = false;
  // evaluate with the redo threshold effectively disabled: temporarily fake a huge threshold by
  // evaluating through a copy whose reference equals sb_ref; the reference's own threshold
  // (|dbg|*dt > 1e-4, ImuError.cpp:549) applies — callers keep |dbg| below it.
  imu_evaluate(s, prm, t0, t1, &c, In(pose0, 7), In(sb0, 9), In(pose1, 7), In(sb1, 9), Out(r, 15), Out(J0, 90), Out(J1, 135),
               Out(J2, 90), Out(J3, 135));
  return c.redoCounter;
}

int orc_imu_propagation(int n, const int64_t* t, const double* gyr, const double* acc,
                        const okvis_ba_imu_params* p, double T_WS[7], double sb[9], int64_t t_start,
                        int64_t t_end, double* cov, double* jac) {
  In g(gyr, 3 * (size_t)n), a(acc, 3 * (size_t)n);
  ImuSamples s{n, t, g, a};
  return imu_propagation(s, make_params(p), Out(T_WS, 7, true), Out(sb, 9, true), t_start, t_end, Out(cov, 225), Out(jac, 225));
}

void orc_pose_error(const double pose[7], const double meas[7], const double si[36], double r[6], double* J) {
  pose_error(In(pose, 7), In(meas, 7), In(si, 36), Out(r, 6), Out(J, 36));
}
void orc_speedbias_error(const double sb[9], const double meas[9], const double si[81], double r[9], double* J) {
  speedbias_error(In(sb, 9), In(meas, 9), In(si, 81), Out(r, 9), Out(J, 81));
}
void orc_relative_pose_error(const double p0[7], const double p1[7], const double si[36], double r[6],
                             double* J0, double* J1) {
  relative_pose_error(In(p0, 7), In(p1, 7), In(si, 36), Out(r, 6), Out(J0, 36), Out(J1, 36));
}
void orc_sqrt_information(const double* info, int n, double* out) {
  sqrt_information_upper(In(info, (size_t)n * n), n, Out(out, (size_t)n * n));
}

orc_window* orc_window_create(const okvis_ba_window* w) {
  orc_window* h = new orc_window();
  h->n_pose = w->n_pose;
  h->n_sb = w->n_sb;
  h->n_lm = w->n_lm;
  h->n_cam = w->n_cam;
  h->n_obs = w->n_obs;
  h->n_imu = w->n_imu;
  h->n_pprior = w->n_pprior;
  h->n_sbprior = w->n_sbprior;
  h->n_relpose = w->n_relpose;
  h->pose = copyr(w->pose, 7 * (size_t)w->n_pose);
  h->sb = copyr(w->sb, 9 * (size_t)w->n_sb);
  h->lm = copyr(w->lm, 4 * (size_t)w->n_lm);
  h->pose_fixed = copyv(w->pose_fixed, (size_t)w->n_pose);
  h->sb_fixed = copyv(w->sb_fixed, (size_t)w->n_sb);
  for (int i = 0; i < w->n_cam; ++i) h->cams.push_back(make_cam(w->cam_intr + 12 * i, w->cam_model[i]));
  h->obs_lm = copyv(w->obs_lm, (size_t)w->n_obs);
  h->obs_pose = copyv(w->obs_pose, (size_t)w->n_obs);
  h->obs_ext = copyv(w->obs_ext, (size_t)w->n_obs);
  h->obs_cam = copyv(w->obs_cam, (size_t)w->n_obs);
  h->obs_uv = copyr(w->obs_uv, 2 * (size_t)w->n_obs);
  h->obs_sqrtw = copyr(w->obs_sqrtw, (size_t)w->n_obs);
  h->cauchy_b = w->cauchy_b;
  h->imu_pose0 = copyv(w->imu_pose0, (size_t)w->n_imu);
  h->imu_sb0 = copyv(w->imu_sb0, (size_t)w->n_imu);
  h->imu_pose1 = copyv(w->imu_pose1, (size_t)w->n_imu);
  h->imu_sb1 = copyv(w->imu_sb1, (size_t)w->n_imu);
  h->imu_t0 = copyv(w->imu_t0, (size_t)w->n_imu);
  h->imu_t1 = copyv(w->imu_t1, (size_t)w->n_imu);
  h->imu_s_begin = copyv(w->imu_s_begin, (size_t)w->n_imu);
  h->imu_s_count = copyv(w->imu_s_count, (size_t)w->n_imu);
  h->imu_s_t = copyv(w->imu_s_t, (size_t)w->n_imu_samples);
  h->imu_s_gyr = copyr(w->imu_s_gyr, 3 * (size_t)w->n_imu_samples);
  h->imu_s_acc = copyr(w->imu_s_acc, 3 * (size_t)w->n_imu_samples);
  h->imu_params = make_params(&w->imu_params);
  h->imu_cache.assign(w->n_imu, ImuCache());
  // an ImuError object that lived through earlier optimize() calls: its cache was built at sb_ref
  if (w->imu_sb_ref && w->imu_sb_ref_valid)
    for (int f = 0; f < w->n_imu; ++f)
      if (w->imu_sb_ref_valid[f]) {
        imu_redo_preintegration(h->samples(f), h->imu_params, w->imu_t0[f], w->imu_t1[f], In(w->imu_sb_ref + 9 * (size_t)f, 9),
                                &h->imu_cache[f]);
        h->imu_cache[f].redo = false;
      }
  h->pprior_pose = copyv(w->pprior_pose, (size_t)w->n_pprior);
  h->pprior_meas = copyr(w->pprior_meas, 7 * (size_t)w->n_pprior);
  h->pprior_sqrtinfo = copyr(w->pprior_sqrtinfo, 36 * (size_t)w->n_pprior);
  h->sbprior_sb = copyv(w->sbprior_sb, (size_t)w->n_sbprior);
  h->sbprior_meas = copyr(w->sbprior_meas, 9 * (size_t)w->n_sbprior);
  h->sbprior_sqrtinfo = copyr(w->sbprior_sqrtinfo, 81 * (size_t)w->n_sbprior);
  h->rel_pose0 = copyv(w->rel_pose0, (size_t)w->n_relpose);
  h->rel_pose1 = copyv(w->rel_pose1, (size_t)w->n_relpose);
  h->rel_sqrtinfo = copyr(w->rel_sqrtinfo, 36 * (size_t)w->n_relpose);
  h->marg_dim = w->marg_dim;
  h->marg_nblocks = w->marg_dim > 0 ? w->marg_nblocks : 0;
  h->marg_block_type = copyv(w->marg_block_type, (size_t)h->marg_nblocks);
  h->marg_block_idx = copyv(w->marg_block_idx, (size_t)h->marg_nblocks);
  h->marg_block_off = copyv(w->marg_block_off, (size_t)h->marg_nblocks);
  h->marg_J = copyr(w->marg_J, (size_t)w->marg_dim * w->marg_dim);
  h->marg_e0 = copyr(w->marg_e0, (size_t)w->marg_dim);
  h->marg_lin = copyr(w->marg_lin, 9 * (size_t)h->marg_nblocks);
  build_ordering(h);
  return h;
}

void orc_window_destroy(orc_window* h) { delete h; }
// CPU-baseline mode of bench.py: n > 1 evaluates the reprojection residuals and the landmark Schur reduction with n
// OpenMP threads (private accumulators; the order of the sums differs from the serial path, so parity tests use 1).
// Returns the number of threads that will actually be used.
int orc_set_threads(int n) {
#ifdef _OPENMP
  g_threads = n < 1 ? 1 : n;
#else
  (void)n;
  g_threads = 1;
#endif
  return g_threads;
}
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_window_set_marg_exact(orc_window* h, int exact) { h->marg_exact = exact != 0; }
int orc_window_reduced_dim(orc_window* h) { return h->D; }
int orc_window_pair_count(orc_window* h) { return h->n_pair; }
void orc_window_pairs(orc_window* h, int32_t* pair_lm, int32_t* pair_block) {
  for (int i = 0; i < h->n_pair; ++i) {
    pair_lm[i] = h->pair_lm[i];
    pair_block[i] = h->pair_block[i];
  }
}
double orc_window_linearize(orc_window* h) {
  double c = evaluate(h, true);
  landmark_quality(h);
  return c;
}
double orc_window_cost(orc_window* h) { return evaluate(h, false); }
int orc_window_solve(orc_window* h, double radius, const okvis_ba_options* opt) {
  if (opt->strategy == OKVIS_BA_STRATEGY_DOGLEG) {
    // the Gauss-Newton solve of the FIRST dogleg iteration: Jacobi scale from the current linearisation, mu = min_mu
    static const int dg[3] = {0, 3, 5};
    const int D = h->D;
    h->Dp2.assign(D, 0.0);
    h->Dl2.assign(3 * (size_t)h->n_lm, 0.0);
    for (int i = 0; i < D; ++i) {
      const real hh = h->U[(size_t)i * D + i];
      const real sc = opt->jacobi_scaling ? 1.0 / (1.0 + std::sqrt(hh)) : 1.0;
      h->Dp2[i] = clampd(sc * sc * hh, opt->min_lm_diagonal, opt->max_lm_diagonal) / (sc * sc);
    }
    for (int l = 0; l < h->n_lm; ++l)
      for (int k = 0; k < 3; ++k) {
        const real hh = h->V[6 * l + dg[k]];
        const real sc = opt->jacobi_scaling ? 1.0 / (1.0 + std::sqrt(hh)) : 1.0;
        h->Dl2[3 * l + k] = clampd(sc * sc * hh, opt->min_lm_diagonal, opt->max_lm_diagonal) / (sc * sc);
      }
    return solve_damped(h, 1e-8) ? 0 : 1;
  }
  return solve(h, radius, *opt) ? 0 : 1;
}
void orc_window_optimize(orc_window* h, const okvis_ba_options* opt, int num_iter, okvis_ba_summary* summary) {
  if (opt->strategy == OKVIS_BA_STRATEGY_DOGLEG) {
    dogleg_loop(h, *opt, num_iter, summary);
    return;
  }
  lm_loop(h, *opt, num_iter, summary);
}
double orc_window_time_iterations(orc_window* h, const okvis_ba_options* opt, int n) {
  okvis_ba_options o = *opt;
  o.function_tolerance = 0;
  o.gradient_tolerance = 0;
  o.parameter_tolerance = 0;
  auto t0 = std::chrono::steady_clock::now();
  if (o.strategy == OKVIS_BA_STRATEGY_DOGLEG) dogleg_loop(h, o, n, nullptr);
  else lm_loop(h, o, n, nullptr);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}
void orc_window_get_state(orc_window* h, double* pose, double* sb, double* lm) {
  put(pose, h->pose);
  put(sb, h->sb);
  put(lm, h->lm);
}
void orc_window_set_state(orc_window* h, const double* pose, const double* sb, const double* lm) {
  if (pose) h->pose.assign(pose, pose + h->pose.size());
  if (sb) h->sb.assign(sb, sb + h->sb.size());
  if (lm) h->lm.assign(lm, lm + h->lm.size());
}

static const std::vector<real>* pick(orc_window* h, int which) {
  switch (which) {
    case OKVIS_BA_ARR_POSE: return &h->pose;
    case OKVIS_BA_ARR_SB: return &h->sb;
    case OKVIS_BA_ARR_LM: return &h->lm;
    case OKVIS_BA_ARR_OBS_RESIDUAL: return &h->obs_r;
    case OKVIS_BA_ARR_LM_V: return &h->V;
    case OKVIS_BA_ARR_LM_B: return &h->b;
    case OKVIS_BA_ARR_LM_HQ: return &h->Hq;
    case OKVIS_BA_ARR_PAIR_W: return &h->W;
    case OKVIS_BA_ARR_REDUCED_S: return &h->S;
    case OKVIS_BA_ARR_REDUCED_RHS: return &h->rhs;
    case OKVIS_BA_ARR_STEP: return &h->step_p;
    case OKVIS_BA_ARR_LM_QUALITY: return &h->quality;
    case OKVIS_BA_ARR_GRADIENT: return &h->g;
    case OKVIS_BA_ARR_IMU_RESIDUAL: return &h->imu_r;
    case OKVIS_BA_ARR_HPP: return &h->U;
  }
  return nullptr;
}
int64_t orc_window_array_size(orc_window* h, int which) {
  if (which == OKVIS_BA_ARR_IMU_SB_REF) return 9 * (int64_t)h->n_imu;
  const std::vector<real>* v = pick(h, which);
  return v ? (int64_t)v->size() : -1;
}
int orc_window_download(orc_window* h, int which, double* out, int64_t n) {
  if (which == OKVIS_BA_ARR_IMU_SB_REF) {
    if (n != 9 * (int64_t)h->n_imu) return -1;
    for (int f = 0; f < h->n_imu; ++f)
      for (int k = 0; k < 9; ++k) out[9 * f + k] = (double)h->imu_cache[f].sb_ref[k];
    return 0;
  }
  const std::vector<real>* v = pick(h, which);
  if (!v || (int64_t)v->size() != n) return -1;
  put(out, *v);
  return 0;
}
void orc_window_full_gradient(orc_window* h, double* g_full) {
  for (int i = 0; i < h->D; ++i) g_full[i] = (double)h->g[i];
  for (size_t i = 0; i < h->b.size(); ++i) g_full[h->D + i] = (double)h->b[i];
}

}  // extern "C"

// ===================================================================================================
// Marginalisation (MarginalizationError.cpp), literal restatement on dense matrices
// ===================================================================================================
namespace {
typedef std::vector<real> Vec;

// Eigen::SelfAdjointEigenSolver stand-in: classical cyclic-by-row Jacobi; ascending eigenvalues like Eigen.
void sym_eig(const Vec& Ain, int n, Vec* ev, Vec* Qout) {
  Vec A(Ain), Q((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
  const real eps = std::numeric_limits<double>::epsilon();
  real dmax = 0.0;
  for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(A[(size_t)i * n + i]));
  const real thr = std::max(eps * dmax, real(1e-300));  // absolute accuracy of a backward-stable solver
  for (int sweep = 0; sweep < 100; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const real apq = A[(size_t)p * n + q], app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        if (std::fabs(apq) <= thr || std::fabs(apq) <= eps * std::sqrt(std::fabs(app * aqq))) continue;
        rotated = true;
        const real theta = (aqq - app) / (2.0 * apq);
        const real t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const real c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const real akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
          const real qkp = Q[(size_t)k * n + p], qkq = Q[(size_t)k * n + q];
          Q[(size_t)k * n + p] = c * qkp - s * qkq;
          Q[(size_t)k * n + q] = s * qkp + c * qkq;
        }
        for (int k = 0; k < n; ++k) {
          const real apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
      }
    if (!rotated) break;
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * n + a] < A[(size_t)b * n + b]; });
  ev->assign(n, 0.0);
  Qout->assign((size_t)n * n, 0.0);
  for (int k = 0; k < n; ++k) {
    (*ev)[k] = A[(size_t)order[k] * n + order[k]];
    for (int i = 0; i < n; ++i) (*Qout)[(size_t)i * n + k] = Q[(size_t)i * n + order[k]];
  }
}

// pseudoInverseSymmSqrt (implementation/MarginalizationError.hpp:215-243): result = Q diag(sqrt(1/l) | 0)
void pinv_symm_sqrt(const Vec& a, int n, Vec* result) {
  Vec ev, Q;
  sym_eig(a, n, &ev, &Q);
  real mx = ev[0];
  for (int i = 1; i < n; ++i) mx = std::max(mx, ev[i]);
  const real tol = std::numeric_limits<double>::epsilon() * n * mx;
  result->assign((size_t)n * n, 0.0);
  for (int k = 0; k < n; ++k) {
    const real s = ev[k] > tol ? std::sqrt(1.0 / ev[k]) : 0.0;
    for (int i = 0; i < n; ++i) (*result)[(size_t)i * n + k] = Q[(size_t)i * n + k] * s;
  }
}

// one Schur step of marginalizeOut on (H, b0) of size n: eliminate the index set `mb` (ascending), keeping the
// others in order.  landmark = true: block-diagonal V in 3x3 blocks (:617-684); false: dense V (:686-739).
void marginalize_out(Vec* Hp, Vec* bp, int n, const std::vector<int>& mb, bool landmark) {
  Vec& H = *Hp;
  Vec& b0 = *bp;
  std::vector<char> is_m(n, 0);
  for (int i : mb) is_m[i] = 1;
  std::vector<int> ka;
  for (int i = 0; i < n; ++i)
    if (!is_m[i]) ka.push_back(i);
  const int na = (int)ka.size(), nm = (int)mb.size();
  // preconditioner (:619-625 / :689-695)
  Vec p(n), Hs((size_t)n * n), bs(n);
  for (int i = 0; i < n; ++i) p[i] = H[(size_t)i * n + i] > 1.0e-9 ? std::sqrt(H[(size_t)i * n + i]) : 1.0e-3;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) Hs[(size_t)i * n + j] = (1.0 / p[i]) * H[(size_t)i * n + j] * (1.0 / p[j]);
    bs[i] = (1.0 / p[i]) * b0[i];
  }
  // split (:627-647)
  Vec U((size_t)na * na), W((size_t)na * nm), V((size_t)nm * nm), b_a(na), b_b(nm), p_a(na);
  for (int i = 0; i < na; ++i) {
    for (int j = 0; j < na; ++j) U[(size_t)i * na + j] = Hs[(size_t)ka[i] * n + ka[j]];
    for (int j = 0; j < nm; ++j) W[(size_t)i * nm + j] = Hs[(size_t)ka[i] * n + mb[j]];
    b_a[i] = bs[ka[i]];
    p_a[i] = p[ka[i]];
  }
  for (int i = 0; i < nm; ++i) {
    for (int j = 0; j < nm; ++j) V[(size_t)i * nm + j] = Hs[(size_t)mb[i] * n + mb[j]];
    b_b[i] = bs[mb[i]];
  }
  Vec dH((size_t)na * na, 0.0), db(na, 0.0);
  if (landmark) {
    for (int i = 0; i < nm; i += 3) {  // :657-677
      Vec V1(9), Vis;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) V1[3 * r + c] = V[(size_t)(i + r) * nm + i + c];
      pinv_symm_sqrt(V1, 3, &Vis);
      Vec M((size_t)na * 3), M1((size_t)na * 3);
      for (int r = 0; r < na; ++r)
        for (int c = 0; c < 3; ++c) {
          real s = 0;
          for (int k = 0; k < 3; ++k) s += W[(size_t)r * nm + i + k] * Vis[3 * k + c];
          M[3 * r + c] = s;
        }
      for (int r = 0; r < na; ++r)
        for (int c = 0; c < 3; ++c) {
          real s = 0;
          for (int k = 0; k < 3; ++k) s += M[3 * r + k] * Vis[3 * c + k];  // M * Vis^T
          M1[3 * r + c] = s;
        }
      for (int r = 0; r < na; ++r) {
        for (int c = 0; c < na; ++c) {
          real s = 0;
          for (int k = 0; k < 3; ++k) s += M[3 * r + k] * M[3 * c + k];
          dH[(size_t)r * na + c] += s;
        }
        real s = 0;
        for (int k = 0; k < 3; ++k) s += M1[3 * r + k] * b_b[i + k];
        db[r] += s;
      }
    }
  } else {
    Vec V1((size_t)nm * nm), Vis;  // :724-736
    for (int r = 0; r < nm; ++r)
      for (int c = 0; c < nm; ++c) V1[(size_t)r * nm + c] = 0.5 * (V[(size_t)r * nm + c] + V[(size_t)c * nm + r]);
    pinv_symm_sqrt(V1, nm, &Vis);
    Vec M((size_t)na * nm), t(nm);
    for (int r = 0; r < na; ++r)
      for (int c = 0; c < nm; ++c) {
        real s = 0;
        for (int k = 0; k < nm; ++k) s += W[(size_t)r * nm + k] * Vis[(size_t)k * nm + c];
        M[(size_t)r * nm + c] = s;
      }
    for (int c = 0; c < nm; ++c) {
      real s = 0;
      for (int k = 0; k < nm; ++k) s += Vis[(size_t)k * nm + c] * b_b[k];
      t[c] = s;
    }
    for (int r = 0; r < na; ++r) {
      real s = 0;
      for (int k = 0; k < nm; ++k) s += M[(size_t)r * nm + k] * t[k];
      db[r] = s;
      for (int c = 0; c < na; ++c) {
        real q = 0;
        for (int k = 0; k < nm; ++k) q += M[(size_t)r * nm + k] * M[(size_t)c * nm + k];
        dH[(size_t)r * na + c] = q;
      }
    }
  }
  // Schur + unscale (:679-684 / :731-739)
  Vec Hn((size_t)na * na), bn(na);
  for (int r = 0; r < na; ++r) {
    bn[r] = p_a[r] * (b_a[r] - db[r]);
    for (int c = 0; c < na; ++c) Hn[(size_t)r * na + c] = p_a[r] * (U[(size_t)r * na + c] - dH[(size_t)r * na + c]) * p_a[c];
  }
  H.swap(Hn);
  b0.swap(bn);
}
}  // namespace

void orc_sym_eig(const double* A, int n, double* eigenvalues, double* Q) {
  Vec a(A, A + (size_t)n * n), ev, q;
  sym_eig(a, n, &ev, &q);
  std::copy(ev.begin(), ev.end(), eigenvalues);
  std::copy(q.begin(), q.end(), Q);
}

int orc_window_marginalize(orc_window* h, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* res) {
  if (!h || !spec || !res || h->marg_dim != 0) return -1;
  // ---- addResidualBlock for every residual of the window: H += J~^T J~, b0 -= J~^T r~ (:367-420), evaluated
  // at the current (= linearisation point) values with the loss-function corrector (:325-365)
  evaluate(h, true);
  const int D = h->D, L = h->n_lm, n = D + 3 * L;
  Vec H((size_t)n * n, 0.0), b0(n, 0.0);
  for (int i = 0; i < D; ++i) {
    for (int j = 0; j < D; ++j) H[(size_t)i * n + j] = h->U[(size_t)i * D + j];
    b0[i] = -h->g[i];
  }
  static const int ut[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  for (int l = 0; l < L; ++l) {
    for (int e = 0; e < 6; ++e) {
      H[(size_t)(D + 3 * l + ut[e][0]) * n + D + 3 * l + ut[e][1]] = h->V[6 * l + e];
      H[(size_t)(D + 3 * l + ut[e][1]) * n + D + 3 * l + ut[e][0]] = h->V[6 * l + e];
    }
    for (int i = 0; i < 3; ++i) b0[D + 3 * l + i] = -h->b[3 * l + i];
  }
  for (int pr = 0; pr < h->n_pair; ++pr) {
    const int l = h->pair_lm[pr], off = h->pose_off[h->pair_block[pr]];
    if (off < 0) continue;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 3; ++j) {
        H[(size_t)(off + i) * n + D + 3 * l + j] = h->W[18 * pr + 3 * i + j];
        H[(size_t)(D + 3 * l + j) * n + off + i] = h->W[18 * pr + 3 * i + j];
      }
  }
  // ---- the previous prior's H_ and b0_ (they persist inside the MarginalizationError object) ----
  if (spec->prior_dim > 0) {
    const int pd = spec->prior_dim;
    std::vector<int> ridx(pd, -1);
    for (int k = 0; k < spec->prior_nblocks; ++k) {
      const bool pose = spec->prior_block_type[k] == OKVIS_BA_BLOCK_POSE;
      const int base = pose ? h->pose_off[spec->prior_block_idx[k]] : h->sb_off[spec->prior_block_idx[k]];
      for (int i = 0; i < (pose ? 6 : 9); ++i) ridx[spec->prior_block_off[k] + i] = base < 0 ? -1 : base + i;
    }
    for (int r = 0; r < pd; ++r) {
      if (ridx[r] < 0) continue;
      b0[ridx[r]] += spec->prior_b0[r];
      for (int c = 0; c < pd; ++c)
        if (ridx[c] >= 0) H[(size_t)ridx[r] * n + ridx[c]] += spec->prior_H[(size_t)r * pd + c];
    }
  }
  // ---- marginalizeOut: landmark part, then dense part ----
  if (L > 0) {
    std::vector<int> mb;
    for (int i = D; i < n; ++i) mb.push_back(i);
    marginalize_out(&H, &b0, n, mb, true);
  }
  std::vector<int> mb, bt, bi, bo;
  int na = 0;
  for (int i = 0; i < h->n_pose; ++i) {
    if (h->pose_off[i] < 0) continue;
    if (spec->pose_marg[i]) {
      for (int k = 0; k < 6; ++k) mb.push_back(h->pose_off[i] + k);
    } else {
      bt.push_back(OKVIS_BA_BLOCK_POSE); bi.push_back(i); bo.push_back(na);
      na += 6;
    }
  }
  for (int i = 0; i < h->n_sb; ++i) {
    if (h->sb_off[i] < 0) continue;
    if (spec->sb_marg[i]) {
      for (int k = 0; k < 9; ++k) mb.push_back(h->sb_off[i] + k);
    } else {
      bt.push_back(OKVIS_BA_BLOCK_SPEEDBIAS); bi.push_back(i); bo.push_back(na);
      na += 9;
    }
  }
  std::sort(mb.begin(), mb.end());
  if (!mb.empty()) marginalize_out(&H, &b0, D, mb, false);
  if (na > res->capacity_dim || (int)bt.size() > res->capacity_blocks) return -2;
  res->dim = na;
  res->nblocks = (int)bt.size();
  for (size_t k = 0; k < bt.size(); ++k) {
    res->block_type[k] = bt[k];
    res->block_idx[k] = bi[k];
    res->block_off[k] = bo[k];
  }
  res->rank = 0;
  res->sweeps[0] = res->sweeps[1] = 0;
  if (na == 0) return 0;
  // ---- updateErrorComputation (:806-846) ----
  Vec p(na), A((size_t)na * na), ev, Q;
  for (int i = 0; i < na; ++i) p[i] = H[(size_t)i * na + i] > 1.0e-9 ? std::sqrt(H[(size_t)i * na + i]) : 1.0e-3;
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < na; ++j)
      A[(size_t)i * na + j] = 0.5 * (1.0 / p[i]) * (H[(size_t)i * na + j] + H[(size_t)j * na + i]) * (1.0 / p[j]);
  sym_eig(A, na, &ev, &Q);
  real mx = ev[0];
  for (int i = 1; i < na; ++i) mx = std::max(mx, ev[i]);
  const real tol = std::numeric_limits<double>::epsilon() * na * mx;
  for (int r = 0; r < na; ++r) {
    const real S = ev[r] > tol ? ev[r] : 0.0, Sp = ev[r] > tol ? 1.0 / ev[r] : 0.0;
    if (ev[r] > tol) res->rank++;
    real s = 0;
    for (int c = 0; c < na; ++c) {
      res->J[(size_t)r * na + c] = p[c] * Q[(size_t)c * na + r] * std::sqrt(S);
      s += std::sqrt(Sp) * Q[(size_t)c * na + r] * (1.0 / p[c]) * b0[c];
    }
    res->e0[r] = -s;
  }
  std::copy(H.begin(), H.end(), res->H);
  std::copy(b0.begin(), b0.end(), res->b0);
  return 0;
}
