// stand-in for okvis_common/include/okvis/FrameTypedefs.hpp:58-70 (KeypointIdentifier), :140-174 (MapPoint, MapPointVector,
// PointMap), :234 (SpeedAndBias)
#pragma once
#include <Eigen/Core>
#include <cstdint>
#include <map>
#include <vector>
namespace okvis {
struct KeypointIdentifier {
  KeypointIdentifier(uint64_t fi = 0, size_t ci = 0, size_t ki = 0) : frameId(fi), cameraIndex(ci), keypointIndex(ki) {}
  bool operator<(const KeypointIdentifier& o) const {
    return frameId != o.frameId ? frameId < o.frameId : cameraIndex != o.cameraIndex ? cameraIndex < o.cameraIndex : keypointIndex < o.keypointIndex;
  }
  uint64_t frameId;
  size_t cameraIndex, keypointIndex;
};
struct MapPoint {
  MapPoint() : id(0), quality(0.0), distance(0.0) {}
  MapPoint(uint64_t id, const Eigen::Vector4d& point, double quality, double distance)
      : id(id), point(point), quality(quality), distance(distance) {}
  uint64_t id;
  Eigen::Vector4d point;
  double quality, distance;
  std::map<okvis::KeypointIdentifier, uint64_t> observations;
};
typedef std::vector<MapPoint, Eigen::aligned_allocator<MapPoint> > MapPointVector;
typedef std::map<uint64_t, MapPoint, std::less<uint64_t>, Eigen::aligned_allocator<std::pair<const uint64_t, MapPoint> > > PointMap;
typedef Eigen::Matrix<double, 9, 1> SpeedAndBias;
}  // namespace okvis
