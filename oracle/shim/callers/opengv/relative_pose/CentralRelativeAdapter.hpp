#pragma once
#include <opengv/relative_pose/RelativeAdapterBase.hpp>
