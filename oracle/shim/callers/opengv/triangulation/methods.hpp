// stand-in (declarations only) for OpenGV's triangulation methods
#pragma once
#include <opengv/types.hpp>
#include <opengv/relative_pose/RelativeAdapterBase.hpp>
namespace opengv { namespace triangulation {
point_t triangulate(const relative_pose::RelativeAdapterBase& adapter, size_t index);
point_t triangulate2(const relative_pose::RelativeAdapterBase& adapter, size_t index);
}}
