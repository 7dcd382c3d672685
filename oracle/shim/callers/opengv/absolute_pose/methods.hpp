// stand-in (declarations only) for OpenGV's absolute-pose solvers
#pragma once
#include <vector>
#include <opengv/types.hpp>
#include <opengv/absolute_pose/AbsoluteAdapterBase.hpp>
namespace opengv { namespace absolute_pose {
translation_t p2p(const AbsoluteAdapterBase& adapter);
translation_t p2p(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformations_t p3p_kneip(const AbsoluteAdapterBase& adapter);
transformations_t p3p_kneip(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformations_t p3p_gao(const AbsoluteAdapterBase& adapter);
transformations_t p3p_gao(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformations_t gp3p(const AbsoluteAdapterBase& adapter);
transformations_t gp3p(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformation_t epnp(const AbsoluteAdapterBase& adapter);
transformation_t epnp(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformation_t gpnp(const AbsoluteAdapterBase& adapter);
transformation_t gpnp(const AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
transformation_t optimize_nonlinear(AbsoluteAdapterBase& adapter);
transformation_t optimize_nonlinear(AbsoluteAdapterBase& adapter, const std::vector<int>& indices);
}}
