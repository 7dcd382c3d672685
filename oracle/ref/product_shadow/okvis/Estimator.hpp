// ORACLE — TEST INFRASTRUCTURE ONLY.
// Shadows the reference's <okvis/Estimator.hpp> on the include path of oracle/_ref/reference_test_estimator: the reference's
// own test source then gets `okvis::Estimator` = the MI355X backend behind the reference's method set (what a maintainer's
// header swap does).
#pragma once
#include "okvis_estimator_adapter.hpp"
#ifndef OKVIS_AMD_HAVE_OKVIS
#error "the okvis headers were not found"
#endif
