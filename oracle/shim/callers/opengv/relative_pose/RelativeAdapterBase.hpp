// stand-in (declarations only) for OpenGV's relative-pose adapter interface
#pragma once
#include <stdlib.h>
#include <opengv/types.hpp>
namespace opengv { namespace relative_pose {
class RelativeAdapterBase {
 public:
  RelativeAdapterBase() : _t12(translation_t()), _R12(rotation_t()) {}
  RelativeAdapterBase(const rotation_t& R12) : _t12(translation_t()), _R12(R12) {}
  RelativeAdapterBase(const translation_t& t12, const rotation_t& R12) : _t12(t12), _R12(R12) {}
  virtual ~RelativeAdapterBase() {}
  virtual opengv::bearingVector_t getBearingVector1(size_t index) const = 0;
  virtual opengv::bearingVector_t getBearingVector2(size_t index) const = 0;
  virtual double getWeight(size_t index) const = 0;
  virtual opengv::translation_t getCamOffset1(size_t index) const = 0;
  virtual opengv::rotation_t getCamRotation1(size_t index) const = 0;
  virtual opengv::translation_t getCamOffset2(size_t index) const = 0;
  virtual opengv::rotation_t getCamRotation2(size_t index) const = 0;
  virtual size_t getNumberCorrespondences() const = 0;
  opengv::translation_t gett12() const { return _t12; }
  void sett12(const translation_t& t12) { _t12 = t12; }
  opengv::rotation_t getR12() const { return _R12; }
  void setR12(const rotation_t& R12) { _R12 = R12; }
 protected:
  opengv::translation_t _t12;
  opengv::rotation_t _R12;
};
}}
