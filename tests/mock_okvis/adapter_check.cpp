// Compile check of okvis_estimator_adapter.hpp against the stand-in interface (see README.md): instantiating the class
// proves that every pure virtual of okvis::VioBackendInterface is overridden with the exact signature.
#include "okvis_estimator_adapter.hpp"
#ifndef OKVIS_AMD_HAVE_OKVIS
#error "the stand-in headers were not found: the adapter body was not compiled"
#endif
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/cameras/RadialTangentialDistortion8.hpp>

void touch(okvis::Estimator& e, okvis::MapPointVector& removed) {
  okvis::VioBackendInterface& backend = e;   // what ThreadedKFVio / Frontend see
  backend.optimize(10, 2, false);
  e.applyMarginalizationStrategy(5, 3, removed);
  e.addObservation<okvis::cameras::PinholeCamera<okvis::cameras::RadialTangentialDistortion> >(1, 2, 0, 3);
  e.addObservation<okvis::cameras::PinholeCamera<okvis::cameras::EquidistantDistortion> >(1, 2, 0, 3);
  e.addObservation<okvis::cameras::PinholeCamera<okvis::cameras::RadialTangentialDistortion8> >(1, 2, 0, 3);
  okvis::kinematics::Transformation T;
  okvis::ImuMeasurementDeque imu;
  okvis::Estimator::initPoseFromImu(imu, T);
}

okvis::Estimator* make() { return new okvis::Estimator(); }   // not abstract
