// ORACLE — TEST INFRASTRUCTURE ONLY: stand-in, see ceres.h
#pragma once
#include "ceres.h"
