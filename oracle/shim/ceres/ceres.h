// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the PUBLIC interface of Ceres-Solver 1.9 that the okvis reference sources compile against
// (CostFunction / SizedCostFunction / LocalParameterization / LossFunction / Problem / Solver), written
// from scratch for this repository: Ceres is not installed here and there is no network.  It lets
// oracle/ref/Makefile compile the reference's own sources unmodified.  Problem only keeps the book;
// ceres::Solve is provided by oracle/ref/ceres_shim_solve.cpp (this repository's restatement of the
// TRUST_REGION / DOGLEG policy — it is NOT Ceres).
#pragma once
#include <Eigen/Core>  // the real ceres.h pulls Eigen in (jet.h); reference sources rely on that
#include <glog/logging.h>  // as the real ceres.h does
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace ceres {

typedef int int32;
typedef short int16;
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR,
                        ITERATIVE_SCHUR, CGNR };
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
enum MinimizerType { LINE_SEARCH, TRUST_REGION };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32> parameter_block_sizes_;
  int num_residuals_;
};

template <int kNumResiduals, int N0 = 0, int N1 = 0, int N2 = 0, int N3 = 0, int N4 = 0, int N5 = 0, int N6 = 0,
          int N7 = 0, int N8 = 0, int N9 = 0>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    const int n[10] = {N0, N1, N2, N3, N4, N5, N6, N7, N8, N9};
    for (int i = 0; i < 10 && n[i] > 0; ++i) mutable_parameter_block_sizes()->push_back(n[i]);
  }
  virtual ~SizedCostFunction() {}
};

class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};

class LossFunction {
 public:
  virtual ~LossFunction() {}
  // rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s), s = squared residual norm
  virtual void Evaluate(double s, double rho[3]) const = 0;
};
class TrivialLoss : public LossFunction {
 public:
  virtual void Evaluate(double s, double rho[3]) const { rho[0] = s, rho[1] = 1.0, rho[2] = 0.0; }
};
class CauchyLoss : public LossFunction {  // rho(s) = b log(1 + s/b), b = a^2
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / b_) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }

 private:
  const double b_, c_;
};
class HuberLoss : public LossFunction {  // rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 above
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  virtual void Evaluate(double s, double rho[3]) const {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s, rho[1] = 1.0, rho[2] = 0.0;
    }
  }

 private:
  const double a_, b_;
};

struct IterationSummary {
  int32 iteration;
  bool step_is_valid, step_is_nonmonotonic, step_is_successful;
  double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, trust_region_radius, eta,
      step_size;
  int line_search_function_evaluations, linear_solver_iterations;
  double iteration_time_in_seconds, step_solver_time_in_seconds, cumulative_time_in_seconds;
  IterationSummary()
      : iteration(0), step_is_valid(false), step_is_nonmonotonic(false), step_is_successful(false), cost(0),
        cost_change(0), gradient_max_norm(0), gradient_norm(0), step_norm(0), relative_decrease(0),
        trust_region_radius(0), eta(0), step_size(0), line_search_function_evaluations(0),
        linear_solver_iterations(0), iteration_time_in_seconds(0), step_solver_time_in_seconds(0),
        cumulative_time_in_seconds(0) {}
};
class IterationCallback {
 public:
  virtual ~IterationCallback() {}
  virtual CallbackReturnType operator()(const IterationSummary& summary) = 0;
};

namespace internal {
struct ParameterBlock {
  double* values;
  int size;
  LocalParameterization* parameterization;
  bool constant;
  std::set<struct ResidualBlock*> residuals;
  int LocalSize() const { return parameterization ? parameterization->LocalSize() : size; }
};
struct ResidualBlock {
  CostFunction* cost;
  LossFunction* loss;
  std::vector<ParameterBlock*> blocks;
  uint64_t serial;  // insertion order (evaluation order of the stand-in solver)
};
}  // namespace internal
typedef internal::ResidualBlock* ResidualBlockId;

template <typename T>
class OrderedGroups {
 public:
  bool AddElementToGroup(const T e, const int g) {
    group_[e] = g;
    return true;
  }
  int GroupId(const T e) const {
    typename std::map<T, int>::const_iterator it = group_.find(e);
    return it == group_.end() ? -1 : it->second;
  }
  void Clear() { group_.clear(); }

 private:
  std::map<T, int> group_;
};
typedef OrderedGroups<double*> ParameterBlockOrdering;

class Problem {
 public:
  struct Options {
    Options()
        : cost_function_ownership(TAKE_OWNERSHIP), loss_function_ownership(TAKE_OWNERSHIP),
          local_parameterization_ownership(TAKE_OWNERSHIP), enable_fast_parameter_block_removal(false),
          enable_fast_removal(false), disable_all_safety_checks(false) {}
    Ownership cost_function_ownership, loss_function_ownership, local_parameterization_ownership;
    bool enable_fast_parameter_block_removal, enable_fast_removal, disable_all_safety_checks;
  };
  Problem() : serial_(0) {}
  explicit Problem(const Options& o) : options_(o), serial_(0) {}
  ~Problem() {
    for (auto& kv : blocks_) delete kv.second;
    for (auto* r : residuals_) delete r;
  }
  void AddParameterBlock(double* values, int size) { AddParameterBlock(values, size, NULL); }
  void AddParameterBlock(double* values, int size, LocalParameterization* lp) {
    internal::ParameterBlock*& b = blocks_[values];
    if (!b) {
      b = new internal::ParameterBlock();
      b->values = values, b->size = size, b->constant = false;
    }
    b->parameterization = lp;
  }
  void RemoveParameterBlock(double* values) {
    auto it = blocks_.find(values);
    if (it == blocks_.end()) return;
    std::set<internal::ResidualBlock*> rs = it->second->residuals;
    for (auto* r : rs) RemoveResidualBlock(r);
    delete it->second;
    blocks_.erase(it);
  }
  ResidualBlockId AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& params) {
    internal::ResidualBlock* r = new internal::ResidualBlock();
    r->cost = cost, r->loss = loss, r->serial = serial_++;
    for (size_t i = 0; i < params.size(); ++i) {
      if (!blocks_.count(params[i])) AddParameterBlock(params[i], cost->parameter_block_sizes()[i]);
      r->blocks.push_back(blocks_[params[i]]);
      blocks_[params[i]]->residuals.insert(r);
    }
    residuals_.insert(r);
    return r;
  }
  void RemoveResidualBlock(ResidualBlockId r) {
    if (!residuals_.count(r)) return;
    for (auto* b : r->blocks) b->residuals.erase(r);
    residuals_.erase(r);
    delete r;
  }
  void SetParameterBlockConstant(double* v) { blocks_.at(v)->constant = true; }
  void SetParameterBlockVariable(double* v) { blocks_.at(v)->constant = false; }
  void SetParameterization(double* v, LocalParameterization* lp) { blocks_.at(v)->parameterization = lp; }
  int NumParameterBlocks() const { return (int)blocks_.size(); }
  int NumResidualBlocks() const { return (int)residuals_.size(); }
  // stand-in internals, used by ceres_shim_solve.cpp
  const std::map<double*, internal::ParameterBlock*>& parameter_blocks() const { return blocks_; }
  const std::set<internal::ResidualBlock*>& residual_blocks() const { return residuals_; }

 private:
  Options options_;
  std::map<double*, internal::ParameterBlock*> blocks_;
  std::set<internal::ResidualBlock*> residuals_;
  uint64_t serial_;
};

class Solver {
 public:
  struct Options {
    Options()
        : minimizer_type(TRUST_REGION), trust_region_strategy_type(LEVENBERG_MARQUARDT),
          dogleg_type(TRADITIONAL_DOGLEG), use_nonmonotonic_steps(false), max_consecutive_nonmonotonic_steps(5),
          max_num_iterations(50), max_solver_time_in_seconds(1e9), num_threads(1),
          initial_trust_region_radius(1e4), max_trust_region_radius(1e16), min_trust_region_radius(1e-32),
          min_relative_decrease(1e-3), min_lm_diagonal(1e-6), max_lm_diagonal(1e32),
          max_num_consecutive_invalid_steps(5), function_tolerance(1e-6), gradient_tolerance(1e-10),
          parameter_tolerance(1e-8), linear_solver_type(SPARSE_NORMAL_CHOLESKY), preconditioner_type(JACOBI),
          num_linear_solver_threads(1), linear_solver_ordering(NULL), use_inner_iterations(false),
          jacobi_scaling(true), minimizer_progress_to_stdout(false), update_state_every_iteration(false) {}
    MinimizerType minimizer_type;
    TrustRegionStrategyType trust_region_strategy_type;
    DoglegType dogleg_type;
    bool use_nonmonotonic_steps;
    int max_consecutive_nonmonotonic_steps, max_num_iterations;
    double max_solver_time_in_seconds;
    int num_threads;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius, min_relative_decrease,
        min_lm_diagonal, max_lm_diagonal;
    int max_num_consecutive_invalid_steps;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    LinearSolverType linear_solver_type;
    PreconditionerType preconditioner_type;
    int num_linear_solver_threads;
    ParameterBlockOrdering* linear_solver_ordering;
    bool use_inner_iterations, jacobi_scaling, minimizer_progress_to_stdout, update_state_every_iteration;
    std::vector<IterationCallback*> callbacks;
  };
  struct Summary {
    Summary()
        : termination_type(FAILURE), initial_cost(-1), final_cost(-1), num_successful_steps(-1),
          num_unsuccessful_steps(-1), total_time_in_seconds(-1) {}
    std::string BriefReport() const;
    std::string FullReport() const;
    TerminationType termination_type;
    std::string message;
    double initial_cost, final_cost;
    std::vector<IterationSummary> iterations;
    int num_successful_steps, num_unsuccessful_steps;
    double total_time_in_seconds;
  };
};

// implemented in oracle/ref/ceres_shim_solve.cpp
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);

}  // namespace ceres
