// ORACLE — TEST INFRASTRUCTURE ONLY: stand-in, see core/core.hpp
#pragma once
#include "../core/core.hpp"
