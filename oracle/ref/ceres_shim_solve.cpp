// ORACLE — TEST INFRASTRUCTURE ONLY.  Placeholder: filled in below.
#include <ceres/ceres.h>
namespace ceres {
std::string Solver::Summary::BriefReport() const { return message; }
std::string Solver::Summary::FullReport() const { return message; }
void Solve(const Solver::Options&, Problem*, Solver::Summary* summary) { summary->message = "not implemented"; }
}  // namespace ceres
