#pragma once
#include <opengv/absolute_pose/AbsoluteAdapterBase.hpp>
