// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// ::ceres::Solve for the stand-in Ceres interface of oracle/shim/ceres/ceres.h.  THIS IS NOT CERES: Ceres-Solver 1.9
// is not in the reference tree and cannot be fetched.  It is this repository's second, deliberately different
// statement of the policy the reference configures (Estimator.cpp:854-873: TRUST_REGION, DOGLEG [traditional],
// jacobi_scaling, exact linear solver), written against the generic Problem / CostFunction / LocalParameterization /
// LossFunction interface so that the okvis reference's OWN error terms (compiled unmodified into oracle/_ref) drive
// it:
//   * residuals and ambient Jacobians come from CostFunction::Evaluate, local Jacobians = J * ComputeJacobian(x)
//     (how Ceres chains the parameterisation), robustification by Ceres' Corrector formulas;
//   * the column-scaled Jacobian is kept block-wise and used literally: gradient = J^T r, model_cost_change =
//     -(J step).(r + J step / 2), Cauchy point from |J v|^2 — no Schur complement, the FULL normal equations are
//     factorised by a dense Cholesky (the oracle and the GPU eliminate the landmarks first);
//   * the iteration logic follows the published structure of Ceres 1.9's TrustRegionMinimizer, DoglegStrategy
//     (TRADITIONAL_DOGLEG) and LevenbergMarquardtStrategy.
// tests/test_dogleg_policy.py compares iterates / radius / bookkeeping of oracle/orc_window.cpp (Schur-based, unscaled
// variables) with this one.
#include <ceres/ceres.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <vector>

namespace ceres {

std::string Solver::Summary::BriefReport() const { return message; }
std::string Solver::Summary::FullReport() const {
  char buf[256];
  std::snprintf(buf, sizeof(buf), "%s: cost %.9e -> %.9e, %d successful / %d unsuccessful steps", message.c_str(),
                initial_cost, final_cost, num_successful_steps, num_unsuccessful_steps);
  return buf;
}

namespace {

typedef std::vector<double> Vec;

struct BlockRef {
  internal::ParameterBlock* b;
  int global_off, local_off;  // offsets into the ambient / tangent vectors; local_off = -1 for constant blocks
};
struct ResidualEval {
  int row0, nrows;
  std::vector<int> col0, ncols;  // per free block of this residual: tangent offset, width
  std::vector<Vec> J;            // per free block: nrows x ncols row-major, robustified, column-scaled
};

struct Program {
  std::vector<internal::ResidualBlock*> residuals;  // evaluation order = insertion order
  std::map<internal::ParameterBlock*, int> index;
  std::vector<BlockRef> blocks;  // free blocks only
  int n_global = 0, n_local = 0, n_rows = 0;

  explicit Program(Problem* p) {
    for (auto* r : p->residual_blocks()) residuals.push_back(r);
    std::sort(residuals.begin(), residuals.end(),
              [](internal::ResidualBlock* a, internal::ResidualBlock* b) { return a->serial < b->serial; });
    // free blocks in order of first use (blocks used by no residual do not enter the program, like in Ceres)
    for (auto* r : residuals) {
      n_rows += r->cost->num_residuals();
      for (auto* b : r->blocks) {
        if (b->constant || index.count(b)) continue;
        index[b] = (int)blocks.size();
        BlockRef br = {b, n_global, n_local};
        blocks.push_back(br);
        n_global += b->size;
        n_local += b->LocalSize();
      }
    }
  }
  void get_state(Vec* x) const {
    x->assign(n_global, 0.0);
    for (const BlockRef& br : blocks) std::copy(br.b->values, br.b->values + br.b->size, x->begin() + br.global_off);
  }
  void set_state(const Vec& x) const {
    for (const BlockRef& br : blocks) std::copy(x.begin() + br.global_off, x.begin() + br.global_off + br.b->size, br.b->values);
  }
  void plus(const Vec& x, const Vec& delta, Vec* out) const {
    out->assign(n_global, 0.0);
    for (const BlockRef& br : blocks) {
      if (br.b->parameterization)
        br.b->parameterization->Plus(&x[br.global_off], &delta[br.local_off], &(*out)[br.global_off]);
      else
        for (int k = 0; k < br.b->size; ++k) (*out)[br.global_off + k] = x[br.global_off + k] + delta[br.local_off + k];
    }
  }
  // cost (and optionally residuals + block Jacobians, both robustified) at the state currently in the user's blocks
  double evaluate(Vec* residuals_out, std::vector<ResidualEval>* jac) const {
    double cost = 0;
    if (residuals_out) residuals_out->assign(n_rows, 0.0);
    if (jac) jac->clear();
    int row = 0;
    for (auto* rb : residuals) {
      const int nr = rb->cost->num_residuals();
      const size_t nb = rb->blocks.size();
      std::vector<const double*> params(nb);
      std::vector<Vec> Jg(nb);
      std::vector<double*> Jp(nb, (double*)0);
      for (size_t i = 0; i < nb; ++i) {
        params[i] = rb->blocks[i]->values;
        if (jac && !rb->blocks[i]->constant) {
          Jg[i].assign((size_t)nr * rb->blocks[i]->size, 0.0);
          Jp[i] = Jg[i].data();
        }
      }
      Vec r(nr, 0.0);
      rb->cost->Evaluate(params.data(), r.data(), jac ? Jp.data() : 0);
      double sq = 0;
      for (int k = 0; k < nr; ++k) sq += r[k] * r[k];
      double rho[3] = {sq, 1.0, 0.0};
      if (rb->loss) rb->loss->Evaluate(sq, rho);
      cost += 0.5 * rho[0];
      // Corrector (Triggs): scaling of the residual and rank-one correction of the Jacobian
      const double sqrt_rho1 = std::sqrt(rho[1]);
      double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
      if (rb->loss && !(sq == 0.0 || rho[2] <= 0.0)) {
        const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(Dd);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
      }
      if (jac) {
        ResidualEval ev;
        ev.row0 = row, ev.nrows = nr;
        for (size_t i = 0; i < nb; ++i) {
          internal::ParameterBlock* b = rb->blocks[i];
          if (b->constant) continue;
          const int gs = b->size, ls = b->LocalSize();
          Vec Jl((size_t)nr * ls, 0.0);
          if (b->parameterization) {
            Vec P((size_t)gs * ls, 0.0);  // global x local, row-major
            b->parameterization->ComputeJacobian(b->values, P.data());
            for (int a = 0; a < nr; ++a)
              for (int c = 0; c < ls; ++c) {
                double s = 0;
                for (int k = 0; k < gs; ++k) s += Jg[i][(size_t)a * gs + k] * P[(size_t)k * ls + c];
                Jl[(size_t)a * ls + c] = s;
              }
          } else {
            Jl = Jg[i];
          }
          if (rb->loss) {  // CorrectJacobian (uses the un-corrected residuals)
            if (alpha_sq_norm == 0.0) {
              for (double& v : Jl) v *= sqrt_rho1;
            } else {
              for (int c = 0; c < ls; ++c) {
                double rtj = 0;
                for (int a = 0; a < nr; ++a) rtj += Jl[(size_t)a * ls + c] * r[a];
                for (int a = 0; a < nr; ++a)
                  Jl[(size_t)a * ls + c] = sqrt_rho1 * (Jl[(size_t)a * ls + c] - alpha_sq_norm * r[a] * rtj);
              }
            }
          }
          ev.col0.push_back(blocks[index.at(b)].local_off);
          ev.ncols.push_back(ls);
          ev.J.push_back(Jl);
        }
        jac->push_back(ev);
      }
      if (residuals_out)
        for (int k = 0; k < nr; ++k) (*residuals_out)[row + k] = r[k] * (rb->loss ? residual_scaling : 1.0);
      row += nr;
    }
    return cost;
  }
};

// J x (x in tangent space) and J^T y for the block Jacobian
void right_multiply(const std::vector<ResidualEval>& J, const Vec& x, Vec* y, int n_rows) {
  y->assign(n_rows, 0.0);
  for (const ResidualEval& e : J)
    for (size_t b = 0; b < e.J.size(); ++b)
      for (int a = 0; a < e.nrows; ++a) {
        double s = 0;
        for (int c = 0; c < e.ncols[b]; ++c) s += e.J[b][(size_t)a * e.ncols[b] + c] * x[e.col0[b] + c];
        (*y)[e.row0 + a] += s;
      }
}
void left_multiply(const std::vector<ResidualEval>& J, const Vec& y, Vec* x, int n) {
  x->assign(n, 0.0);
  for (const ResidualEval& e : J)
    for (size_t b = 0; b < e.J.size(); ++b)
      for (int c = 0; c < e.ncols[b]; ++c) {
        double s = 0;
        for (int a = 0; a < e.nrows; ++a) s += e.J[b][(size_t)a * e.ncols[b] + c] * y[e.row0 + a];
        (*x)[e.col0[b] + c] += s;
      }
}
void squared_column_norm(const std::vector<ResidualEval>& J, Vec* d, int n) {
  d->assign(n, 0.0);
  for (const ResidualEval& e : J)
    for (size_t b = 0; b < e.J.size(); ++b)
      for (int a = 0; a < e.nrows; ++a)
        for (int c = 0; c < e.ncols[b]; ++c) {
          const double v = e.J[b][(size_t)a * e.ncols[b] + c];
          (*d)[e.col0[b] + c] += v * v;
        }
}
void scale_columns(std::vector<ResidualEval>* J, const Vec& s) {
  for (ResidualEval& e : *J)
    for (size_t b = 0; b < e.J.size(); ++b)
      for (int a = 0; a < e.nrows; ++a)
        for (int c = 0; c < e.ncols[b]; ++c) e.J[b][(size_t)a * e.ncols[b] + c] *= s[e.col0[b] + c];
}
// solve min |J y - r|^2 + |diag(D) y|^2 by the normal equations (J^T J + D^2) y = J^T r, dense Cholesky
bool normal_solve(const std::vector<ResidualEval>& J, const Vec& r, const Vec& D, int n, Vec* y) {
  std::vector<double> A((size_t)n * n, 0.0);
  for (const ResidualEval& e : J)
    for (size_t b = 0; b < e.J.size(); ++b)
      for (size_t c2 = 0; c2 < e.J.size(); ++c2)
        for (int i = 0; i < e.ncols[b]; ++i)
          for (int j = 0; j < e.ncols[c2]; ++j) {
            double s = 0;
            for (int a = 0; a < e.nrows; ++a)
              s += e.J[b][(size_t)a * e.ncols[b] + i] * e.J[c2][(size_t)a * e.ncols[c2] + j];
            A[(size_t)(e.col0[b] + i) * n + e.col0[c2] + j] += s;
          }
  for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += D[i] * D[i];
  Vec rhs;
  left_multiply(J, r, &rhs, n);
  // Cholesky, lower triangle in place
  for (int k = 0; k < n; ++k) {
    double d = A[(size_t)k * n + k];
    for (int j = 0; j < k; ++j) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)k * n + k] = d;
    for (int i = k + 1; i < n; ++i) {
      double s = A[(size_t)i * n + k];
      const double* ai = &A[(size_t)i * n];
      const double* ak = &A[(size_t)k * n];
      for (int j = 0; j < k; ++j) s -= ai[j] * ak[j];
      A[(size_t)i * n + k] = s / d;
    }
  }
  y->assign(n, 0.0);
  for (int i = 0; i < n; ++i) {
    double s = rhs[i];
    for (int j = 0; j < i; ++j) s -= A[(size_t)i * n + j] * (*y)[j];
    (*y)[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = (*y)[i];
    for (int j = i + 1; j < n; ++j) s -= A[(size_t)j * n + i] * (*y)[j];
    (*y)[i] = s / A[(size_t)i * n + i];
  }
  for (double v : *y)
    if (!std::isfinite(v)) return false;
  return true;
}
double dot(const Vec& a, const Vec& b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}
double norm(const Vec& a) { return std::sqrt(dot(a, a)); }

// ---- trust-region strategies (in the variables of the column-scaled Jacobian) ----
struct Strategy {
  virtual ~Strategy() {}
  virtual bool ComputeStep(const std::vector<ResidualEval>& J, const Vec& r, int n, int n_rows, Vec* step) = 0;
  virtual void StepAccepted(double q) = 0;
  virtual void StepRejected(double q) = 0;
  virtual void StepIsInvalid() = 0;
  virtual double Radius() const = 0;
};
struct LMStrategy : Strategy {
  double radius, max_radius, min_d, max_d, decrease_factor;
  bool reuse;
  Vec diag;
  explicit LMStrategy(const Solver::Options& o)
      : radius(o.initial_trust_region_radius), max_radius(o.max_trust_region_radius), min_d(o.min_lm_diagonal),
        max_d(o.max_lm_diagonal), decrease_factor(2.0), reuse(false) {}
  bool ComputeStep(const std::vector<ResidualEval>& J, const Vec& r, int n, int, Vec* step) {
    if (!reuse) {
      squared_column_norm(J, &diag, n);
      for (double& v : diag) v = std::min(std::max(v, min_d), max_d);
    }
    Vec lm(n);
    for (int i = 0; i < n; ++i) lm[i] = std::sqrt(diag[i] / radius);
    reuse = true;
    if (!normal_solve(J, r, lm, n, step)) return false;
    for (double& v : *step) v = -v;
    return true;
  }
  void StepAccepted(double q) {
    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * q - 1.0, 3));
    radius = std::min(max_radius, radius);
    decrease_factor = 2.0;
    reuse = false;
  }
  void StepRejected(double) {
    radius = radius / decrease_factor;
    decrease_factor *= 2.0;
    reuse = true;
  }
  void StepIsInvalid() { StepRejected(0.0); }
  double Radius() const { return radius; }
};
struct DoglegStrategy : Strategy {
  double radius, min_d, max_d, mu, min_mu, max_mu, mu_increase, alpha, dogleg_step_norm;
  bool reuse, gn_ok;
  Vec diag, gradient, gn;
  explicit DoglegStrategy(const Solver::Options& o)
      : radius(o.initial_trust_region_radius), min_d(o.min_lm_diagonal), max_d(o.max_lm_diagonal), mu(1e-8), min_mu(1e-8),
        max_mu(1.0), mu_increase(10.0), alpha(0), dogleg_step_norm(0), reuse(false), gn_ok(false) {}
  void Interpolate(Vec* step) {
    const int n = (int)gradient.size();
    step->assign(n, 0.0);
    const double gradient_norm = norm(gradient), gn_norm = norm(gn);
    if (gn_norm <= radius) {
      *step = gn;
      dogleg_step_norm = gn_norm;
    } else if (gradient_norm * alpha >= radius) {
      for (int i = 0; i < n; ++i) (*step)[i] = -(radius / gradient_norm) * gradient[i];
      dogleg_step_norm = radius;
    } else {
      const double b_dot_a = -alpha * dot(gradient, gn);
      const double a2 = std::pow(alpha * gradient_norm, 2.0);
      const double bma2 = a2 - 2 * b_dot_a + std::pow(gn_norm, 2);
      const double c = b_dot_a - a2;
      const double d = std::sqrt(c * c + bma2 * (std::pow(radius, 2.0) - a2));
      const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
      for (int i = 0; i < n; ++i) (*step)[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i];
      dogleg_step_norm = norm(*step);
    }
    for (int i = 0; i < n; ++i) (*step)[i] /= diag[i];
  }
  bool ComputeStep(const std::vector<ResidualEval>& J, const Vec& r, int n, int n_rows, Vec* step) {
    if (reuse) {  // only the interpolation changes with the radius
      if (gn_ok) Interpolate(step);
      return gn_ok;
    }
    reuse = true;
    squared_column_norm(J, &diag, n);
    for (double& v : diag) v = std::sqrt(std::min(std::max(v, min_d), max_d));
    left_multiply(J, r, &gradient, n);
    for (int i = 0; i < n; ++i) gradient[i] /= diag[i];
    Vec sg(n), Jg;
    for (int i = 0; i < n; ++i) sg[i] = gradient[i] / diag[i];
    right_multiply(J, sg, &Jg, n_rows);
    alpha = dot(gradient, gradient) / dot(Jg, Jg);
    gn_ok = false;
    while (mu < max_mu) {
      Vec lm(n);
      for (int i = 0; i < n; ++i) lm[i] = diag[i] * std::sqrt(mu);
      if (!normal_solve(J, r, lm, n, &gn)) {
        mu *= mu_increase;
        continue;
      }
      gn_ok = true;
      break;
    }
    if (!gn_ok) return false;
    for (int i = 0; i < n; ++i) gn[i] *= -diag[i];
    Interpolate(step);
    return true;
  }
  void StepAccepted(double q) {
    if (q < 0.25) radius *= 0.5;
    if (q > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
    mu = std::max(min_mu, 2.0 * mu / mu_increase);
    reuse = false;
  }
  void StepRejected(double) {
    radius *= 0.5;
    reuse = true;
  }
  void StepIsInvalid() {
    mu *= mu_increase;
    reuse = false;
  }
  double Radius() const { return radius; }
};

}  // namespace

void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  const auto t_start = std::chrono::steady_clock::now();
  auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  *summary = Solver::Summary();
  summary->num_successful_steps = summary->num_unsuccessful_steps = 0;
  Program prog(problem);
  const int n = prog.n_local;
  Vec x, x_plus_delta, residuals, gradient, scale(n, 1.0), step, delta(n), model_residuals;
  std::vector<ResidualEval> J;
  prog.get_state(&x);
  double cost = prog.evaluate(&residuals, &J);
  left_multiply(J, residuals, &gradient, n);
  summary->initial_cost = summary->final_cost = cost;
  auto gradient_norms = [&](IterationSummary* it) {
    Vec neg(n), proj;
    for (int i = 0; i < n; ++i) neg[i] = -gradient[i];
    prog.plus(x, neg, &proj);
    double m = 0, s2 = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      m = std::max(m, std::fabs(x[i] - proj[i]));
      s2 += (x[i] - proj[i]) * (x[i] - proj[i]);
    }
    it->gradient_max_norm = m;
    it->gradient_norm = std::sqrt(s2);
  };
  IterationSummary it;
  it.iteration = 0;
  it.cost = cost;
  gradient_norms(&it);
  std::unique_ptr<Strategy> strategy;
  if (options.trust_region_strategy_type == DOGLEG) strategy.reset(new DoglegStrategy(options));
  else strategy.reset(new LMStrategy(options));
  it.trust_region_radius = strategy->Radius();
  it.cumulative_time_in_seconds = now();
  summary->iterations.push_back(it);
  if (it.gradient_max_norm <= options.gradient_tolerance) {
    summary->message = "Gradient tolerance reached.";
    summary->termination_type = CONVERGENCE;
    return;
  }
  if (options.jacobi_scaling) {
    squared_column_norm(J, &scale, n);
    for (double& v : scale) v = 1.0 / (1.0 + std::sqrt(v));
    scale_columns(&J, scale);
  }
  double x_norm = norm(x);
  int invalid = 0;
  while (true) {
    for (IterationCallback* cb : options.callbacks) {
      const CallbackReturnType rc = (*cb)(summary->iterations.back());
      if (rc == SOLVER_TERMINATE_SUCCESSFULLY) {
        summary->message = "User callback returned SOLVER_TERMINATE_SUCCESSFULLY.";
        summary->termination_type = USER_SUCCESS;
        goto finish;
      }
      if (rc == SOLVER_ABORT) {
        summary->message = "User callback returned SOLVER_ABORT.";
        summary->termination_type = USER_FAILURE;
        goto finish;
      }
    }
    {
      const double t_iter = now();
      if (summary->iterations.back().iteration >= options.max_num_iterations) {
        summary->message = "Maximum number of iterations reached.";
        summary->termination_type = NO_CONVERGENCE;
        break;
      }
      if (t_iter >= options.max_solver_time_in_seconds) {
        summary->message = "Maximum solver time reached.";
        summary->termination_type = NO_CONVERGENCE;
        break;
      }
      const bool solved = strategy->ComputeStep(J, residuals, n, prog.n_rows, &step);
      it = IterationSummary();
      it.iteration = summary->iterations.back().iteration + 1;
      double model_cost_change = 0;
      if (solved) {
        right_multiply(J, step, &model_residuals, prog.n_rows);
        double s = 0;
        for (int k = 0; k < prog.n_rows; ++k) s += model_residuals[k] * (residuals[k] + model_residuals[k] / 2.0);
        model_cost_change = -s;
        it.step_is_valid = !(model_cost_change < 0.0);
      }
      if (!it.step_is_valid) {
        if (++invalid >= options.max_num_consecutive_invalid_steps) {
          summary->message = "Number of successive invalid steps more than max_num_consecutive_invalid_steps.";
          summary->termination_type = FAILURE;
          break;
        }
        it.cost = cost;
        it.gradient_max_norm = summary->iterations.back().gradient_max_norm;
        it.gradient_norm = summary->iterations.back().gradient_norm;
      } else {
        invalid = 0;
        for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];  // undo the column scaling
        prog.plus(x, delta, &x_plus_delta);
        prog.set_state(x_plus_delta);
        const double new_cost = prog.evaluate(0, 0);
        prog.set_state(x);
        double s2 = 0;
        for (size_t i = 0; i < x.size(); ++i) s2 += (x[i] - x_plus_delta[i]) * (x[i] - x_plus_delta[i]);
        it.step_norm = std::sqrt(s2);
        if (it.step_norm <= options.parameter_tolerance * (x_norm + options.parameter_tolerance)) {
          summary->message = "Parameter tolerance reached.";
          summary->termination_type = CONVERGENCE;
          break;
        }
        it.cost_change = cost - new_cost;
        if (std::fabs(it.cost_change) < options.function_tolerance * cost) {
          summary->message = "Function tolerance reached.";
          summary->termination_type = CONVERGENCE;
          break;
        }
        it.relative_decrease = it.cost_change / model_cost_change;
        it.step_is_successful = it.relative_decrease > options.min_relative_decrease;
      }
      if (it.step_is_successful) {
        ++summary->num_successful_steps;
        strategy->StepAccepted(it.relative_decrease);
        x = x_plus_delta;
        x_norm = norm(x);
        prog.set_state(x);
        cost = prog.evaluate(&residuals, &J);
        left_multiply(J, residuals, &gradient, n);
        gradient_norms(&it);
        if (it.gradient_max_norm <= options.gradient_tolerance) {
          summary->message = "Gradient tolerance reached.";
          summary->termination_type = CONVERGENCE;
          it.cost = cost;
          it.trust_region_radius = strategy->Radius();
          summary->iterations.push_back(it);
          break;
        }
        if (options.jacobi_scaling) scale_columns(&J, scale);
      } else {
        ++summary->num_unsuccessful_steps;
        if (it.step_is_valid) strategy->StepRejected(it.relative_decrease);
        else strategy->StepIsInvalid();
      }
      it.cost = cost;
      it.trust_region_radius = strategy->Radius();
      it.iteration_time_in_seconds = now() - t_iter;
      it.cumulative_time_in_seconds = now();
      if (it.trust_region_radius < options.min_trust_region_radius) {
        summary->message = "Termination. Minimum trust region radius reached.";
        summary->termination_type = CONVERGENCE;
        break;
      }
      summary->iterations.push_back(it);
    }
  }
finish:
  prog.set_state(x);
  summary->final_cost = cost;
  summary->total_time_in_seconds = now();
}

}  // namespace ceres
