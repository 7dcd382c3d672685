// stand-in (declarations only) for OpenGV's absolute-pose adapter interface
#pragma once
#include <stdlib.h>
#include <opengv/types.hpp>
namespace opengv { namespace absolute_pose {
class AbsoluteAdapterBase {
 public:
  AbsoluteAdapterBase() : _t(translation_t()), _R(rotation_t()) {}
  AbsoluteAdapterBase(const rotation_t& R) : _t(translation_t()), _R(R) {}
  AbsoluteAdapterBase(const translation_t& t, const rotation_t& R) : _t(t), _R(R) {}
  virtual ~AbsoluteAdapterBase() {}
  virtual opengv::bearingVector_t getBearingVector(size_t index) const = 0;
  virtual double getWeight(size_t index) const = 0;
  virtual opengv::translation_t getCamOffset(size_t index) const = 0;
  virtual opengv::rotation_t getCamRotation(size_t index) const = 0;
  virtual opengv::point_t getPoint(size_t index) const = 0;
  virtual size_t getNumberCorrespondences() const = 0;
  opengv::translation_t gett() const { return _t; }
  void sett(const translation_t& t) { _t = t; }
  opengv::rotation_t getR() const { return _R; }
  void setR(const rotation_t& R) { _R = R; }
 protected:
  opengv::translation_t _t;
  opengv::rotation_t _R;
};
}}
