// stand-in (declarations only) for OpenGV's relative-pose solvers
#pragma once
#include <vector>
#include <opengv/types.hpp>
#include <opengv/relative_pose/RelativeAdapterBase.hpp>
namespace opengv { namespace relative_pose {
translation_t twopt(const RelativeAdapterBase& adapter, bool unrotate, const std::vector<int>& indices);
translation_t twopt(const RelativeAdapterBase& adapter, bool unrotate, size_t index0, size_t index1);
rotation_t twopt_rotationOnly(const RelativeAdapterBase& adapter);
rotation_t twopt_rotationOnly(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
rotation_t twopt_rotationOnly(const RelativeAdapterBase& adapter, size_t index0, size_t index1);
rotation_t rotationOnly(const RelativeAdapterBase& adapter);
rotation_t rotationOnly(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
essentials_t fivept_stewenius(const RelativeAdapterBase& adapter);
essentials_t fivept_stewenius(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
essentials_t fivept_nister(const RelativeAdapterBase& adapter);
essentials_t fivept_nister(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
essentials_t sevenpt(const RelativeAdapterBase& adapter);
essentials_t sevenpt(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
essential_t eightpt(const RelativeAdapterBase& adapter);
essential_t eightpt(const RelativeAdapterBase& adapter, const std::vector<int>& indices);
transformation_t optimize_nonlinear(RelativeAdapterBase& adapter);
transformation_t optimize_nonlinear(RelativeAdapterBase& adapter, const std::vector<int>& indices);
}}
