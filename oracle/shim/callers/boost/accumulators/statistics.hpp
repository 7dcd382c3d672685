#pragma once
#include <boost/accumulators/accumulators.hpp>
